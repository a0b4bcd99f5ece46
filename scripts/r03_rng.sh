# RNG scheduling experiments: generator granularity x stream priority
for e in "X=1" "MFM_RNG_PAR_BLOCKS=256" "MFM_RNG_PAR_BLOCKS=128" "MFM_RNG_PAR_BLOCKS=96" "MFM_RNG_PAR_BLOCKS=128 MFM_RNG_NO_PRIORITY=1" "MFM_RES_NO_LINEAR=1"; do
  v=$(env $e python bench.py --steps 60 --warmup 3 --no-other-configs --cpu-seconds 0 --fit-iters 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "$e: $v"
done
