# the parallel generator's batch (iterations of outputs per producing launch) under the two-part random set
cd $GRAFT_REPO_ROOT
Q="--cpu-seconds 0 --fit-iters 0 --no-other-configs --long-seconds 4"
for rep in 1 2; do
for b in 6 3 12 16; do
  MFM_RNG_GEN_BATCH=$b python bench.py --steps 40 --warmup 5 $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch=$b value', d['value'], 'long', d.get('value_long'))"
done
done
