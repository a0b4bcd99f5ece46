#!/bin/bash
# rebuild the native parts in place (from any working directory); prints compiler errors only
cd "$(dirname "$0")/.." && python -c "from myfm_amd import _build; _build.build_all()" 2>&1 | grep -iE "error|Error" ; ls -la --time-style=+%T myfm_amd/libmyfm_hip.so | awk '{print $6, $7}'
