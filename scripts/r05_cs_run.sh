# usage: bash scripts/r05_cs_run.sh <tag> "name ENV=.. ENV=.." ...   -- config 5 at full size under the given environments, phase stamps
cd $GRAFT_REPO_ROOT
T=$1; shift
C5="--config 5 --scale 1.0 --steps 3 --warmup 1 --cpu-iters 0 --fit-iters 0 --long-seconds 0"
for spec in "$@"; do
  set -- $spec; name=$1; shift
  env "$@" MFM_CB_PROF=130 MFM_SETUP_TIMING=1 python bench.py $C5 > gpurun_out/${T}_$name.json 2> gpurun_out/${T}_$name.err
  echo "== $name: $(python -c "import json; d=json.loads(open('gpurun_out/${T}_$name.json').read().strip().splitlines()[-1]); print(d['value'], 'it/s', d['ms_per_step'], 'ms', 'setup', d['config'].get('setup_s'))" 2>&1 | tail -1)"
  grep "k_cs_stream\|k_cb_persist" gpurun_out/${T}_$name.err | tail -1 | cut -c1-330
done
