# round-3 first measurement: resident sweep parity + A/B against the per-factor pass
set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_capi.py -x -q -m gpu 2>&1 | tail -5
B="--steps 40 --warmup 3 --no-other-configs --cpu-seconds 0 --fit-iters 0"
MFM_SETUP_TIMING=1 timeout 600 python bench.py $B > gpurun_out/r03_a_res.json 2> gpurun_out/r03_a_res.err
MFM_NO_RESIDENT=1 timeout 600 python bench.py $B > gpurun_out/r03_a_nores.json 2> gpurun_out/r03_a_nores.err
grep resident gpurun_out/r03_a_res.err
python - <<'PY'
import json
for n in ("res","nores"):
    try:
        d=json.loads(open("gpurun_out/r03_a_%s.json"%n).read().strip().splitlines()[-1]); r=d["roofline"]
        print(n, d["value"], d["ms_per_step"], r.get("kernel"), r.get("frac"), r.get("by_kernel_ms_per_step"))
    except Exception as e: print(n, "failed", e)
PY
bash scripts/res_ablate.sh
