#!/bin/bash
# phase profile of the resident latent sweep (MFM_RES_PROF): usage res_prof.sh [ENV=VAL ...]
env "$@" MFM_RES_PROF=5 python bench.py --steps 10 --warmup 3 --no-other-configs --cpu-seconds 0 --fit-iters 0 2>&1 | grep -A30 "resident profile" | cut -c1-330 | sed 's/"unit".*//'
