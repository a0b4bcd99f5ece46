# the first attempt's window half-width: miss rate and rate at k = 3.5 against 4.0
cd $GRAFT_REPO_ROOT
soak() {
  MFM_LATENT_TIMING=1 python bench.py --gpus 1 --fit-iters 0 --no-other-configs --no-kernel-timing --cpu-seconds 0 --long-seconds 0 "$@" 2>/tmp/soak.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   value', d['value'])"
  grep "^\[latent\]" /tmp/soak.err > /tmp/soak.txt
  echo "k=$MFM_LAT_KSIGMA $* : draws $(wc -l < /tmp/soak.txt), second attempts $(grep -c '2 attempt' /tmp/soak.txt), status != 0: $(grep -vc 'status 0' /tmp/soak.txt)"
}
for k in 3.2 3.0; do
  export MFM_LAT_KSIGMA=$k
  soak --config 3 --task classification --steps 400 --warmup 2
  soak --config 2 --task classification --steps 2000 --warmup 2
  soak --config 2 --task ordered --steps 1000 --warmup 2
  soak --config 5 --scale 1.0 --steps 12 --warmup 1
done
