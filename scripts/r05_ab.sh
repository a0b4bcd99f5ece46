# A/B of MFM_RES_DBG values on config 3 in one box: bash scripts/r05_ab.sh 0 1048576 2097152
cd $GRAFT_REPO_ROOT
Q="--steps 300 --warmup 5 --cpu-seconds 0 --fit-iters 0 --no-other-configs --long-seconds 0"
for i in 1 2; do for d in "$@"; do
  echo "MFM_RES_DBG=$d: $(MFM_RES_DBG=$d python bench.py $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'it/s', d['ms_per_step'], 'ms', d['roofline']['avg_launch_us'])")"
done; done
