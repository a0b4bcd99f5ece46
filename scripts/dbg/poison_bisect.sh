#!/bin/bash
# debug: which launch sites read LDS they have not written (MFM_DEBUG_POISON_LDS, mfm_common.hpp). $1 = TASK
export MFM_RNG_FUSED_JUMP=1 TASK=${1:-reg} NIT=2
MFM_DEBUG_POISON_LDS=1000 MFM_DEBUG_POISON_LIST=1 python scripts/dbg/ordered_nan.py 2>&1 | grep "launch site" | sort -u > /tmp/sites.txt
wc -l /tmp/sites.txt
while read -r _ _ _ cls _ line; do
  out=$(MFM_DEBUG_POISON_LDS=$cls MFM_DEBUG_POISON_LINE=$line python scripts/dbg/ordered_nan.py 2>&1 | grep -c " nan\|finite False\|EXC")
  echo "class $cls line $line -> bad lines $out"
done < /tmp/sites.txt
