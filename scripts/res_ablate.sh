#!/bin/bash
# timing experiments of the resident latent sweep (mfm_res.hpp, MFM_RES_DBG): each switch cuts one part off (results wrong)
for dbg in ${@:-0 32 64 128 192 4 256 512 1024 1536 1792} ; do
  v=$(MFM_RES_DBG=$dbg python bench.py --steps 40 --warmup 3 --no-other-configs --cpu-seconds 0 --fit-iters 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], r['by_kernel_ms_per_step'].get('sweep_V_resident'))")
  echo "dbg=$dbg it/s,ms_resident: $v"
done
