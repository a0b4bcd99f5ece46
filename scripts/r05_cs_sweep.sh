# round 5: config 5 at full size with the streamed chain (k_cs_stream) under several plan parameters, against the round-4 form
cd $GRAFT_REPO_ROOT
T=${1:-r05_b}
C5="--config 5 --scale 1.0 --steps 3 --warmup 1 --cpu-iters 0 --fit-iters 0 --long-seconds 0"
run() {  # name, env...
  name=$1; shift
  env "$@" MFM_CB_PROF=130 MFM_SETUP_TIMING=1 python bench.py $C5 > gpurun_out/${T}_$name.json 2> gpurun_out/${T}_$name.err
  echo "== $name: $(python -c "import json,sys; d=json.loads(open('gpurun_out/${T}_$name.json').read().strip().splitlines()[-1]); print(d['value'], 'it/s', d['ms_per_step'], 'ms', 'setup', d['config'].get('setup_s'), {k: round(v['ms_per_step'],1) for k,v in (d.get('kernel_classes') or {}).items()} if isinstance(d.get('kernel_classes'), dict) else '')" 2>&1 | tail -1)"
  grep "streamed chain\|k_cs_stream\|k_cb_persist" gpurun_out/${T}_$name.err | tail -4
}
run persist MFM_NO_CB_STREAM=1
run default MFM_CS_X=0
run lw3 MFM_CS_LW=3
run lw2 MFM_CS_LW=2
run cg2lw6 MFM_CS_CG=2 MFM_CS_LW=6
run cg2lw4 MFM_CS_CG=2 MFM_CS_LW=4
run nb32 MFM_CS_NB=32
run nb32lw3 MFM_CS_NB=32 MFM_CS_LW=3
