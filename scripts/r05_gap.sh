cd $GRAFT_REPO_ROOT
Q="--steps 300 --warmup 10 --cpu-seconds 0 --fit-iters 0 --no-other-configs --long-seconds 0"
for v in "" "MYFM_AMD_DEVICE_HYPERS=1" "MFM_RNG_DBG_REUSE=1" "MFM_RNG_DBG_REUSE=1 MYFM_AMD_DEVICE_HYPERS=1"; do
  echo "[$v]: $(env $v python bench.py $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'it/s', d['ms_per_step'], 'ms', (d.get('roofline') or {}).get('avg_launch_us'))")"
done
