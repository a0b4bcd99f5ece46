B="python bench.py --steps 400 --warmup 20 --cpu-iters 0 --fit-iters 0 --no-other-configs --no-kernel-timing --long-seconds 0"
mkdir -p gpurun_out/ab
for i in 1 2 3; do
  $B 2>/dev/null | tail -1 > gpurun_out/ab/nostore_$i.json
  MYFM_AMD_KEEP_RESIDUAL=1 $B 2>/dev/null | tail -1 > gpurun_out/ab/store_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab/*store_*.json')):
    print(f, json.loads(open(f).read())['value'])
PY
timeout 600 python -m pytest tests/test_gpu_capi.py -q -x -k "resident" 2>&1 | tail -3
