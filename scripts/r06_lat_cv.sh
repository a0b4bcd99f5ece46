# the windows' control variate: parity tests, then second attempts and rates against the plain windows (MFM_LAT_NO_CV=1) for two safety factors
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_exact_latent.py tests/test_gpu_host_rng_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
soak() {
  MFM_LATENT_TIMING=1 python bench.py --gpus 1 --fit-iters 0 --no-other-configs --no-kernel-timing --cpu-seconds 0 --long-seconds 0 "$@" 2>/tmp/soak.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   value', d['value'])"
  grep "^\[latent\]" /tmp/soak.err > /tmp/soak.txt
  echo "$TAG $* : draws $(wc -l < /tmp/soak.txt), second attempts $(grep -c '2 attempt' /tmp/soak.txt), status != 0: $(grep -vc 'status 0' /tmp/soak.txt) | $(tail -1 /tmp/soak.txt | sed 's/.*walkers/walkers/' | cut -c1-200)"
}
for mode in nocv 1.15; do
  unset MFM_LAT_NO_CV MFM_LAT_CV_SAFETY
  if [ $mode = nocv ]; then export MFM_LAT_NO_CV=1; else export MFM_LAT_CV_SAFETY=$mode; fi
  TAG="[$mode]"
  soak --config 3 --task classification --steps 400 --warmup 2
  soak --config 3 --task ordered --steps 200 --warmup 2
  soak --config 5 --scale 0.2 --steps 40 --warmup 2
  soak --config 5 --scale 1.0 --steps 12 --warmup 1
done
