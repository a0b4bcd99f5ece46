# a random set produced in two parts (default) against the whole set behind one gate (MFM_RNG_ONE_PART=1): tests, bench A/B, timelines
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_rng.py tests/test_gpu_device_hypers.py tests/test_golden_gpu.py -x -q 2>&1 | tail -3
Q="--cpu-seconds 0 --fit-iters 0 --no-other-configs --long-seconds 4"
for rep in 1 2 3; do
for g in 1 0; do
  if [ $g = 1 ]; then export MFM_RNG_ONE_PART=1; else unset MFM_RNG_ONE_PART; fi
  python bench.py --steps 40 --warmup 5 $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one_part=$g value', d['value'], 'long', d.get('value_long'))"
done
done
unset MFM_RNG_ONE_PART
TL_FLAGS=-v bash scripts/prof_timeline.sh two_parts k_mf_resident --steps 12 --warmup 3 --no-other-configs --long-seconds 0 > /dev/null
grep -A60 "gap_before_us" gpurun_out/timeline_two_parts.txt | grep -v "^kernel" | tail -42 | cut -c1-120
