# many draws: how often the first attempt's windows (+- 4 sigma) miss the path, and that no draw ever falls back (MFM_LATENT_TIMING prints
# one line per draw: "N attempt(s), status S")
cd $GRAFT_REPO_ROOT
soak() {
  MFM_LATENT_TIMING=1 python bench.py --gpus 1 --fit-iters 0 --no-other-configs --no-kernel-timing --cpu-seconds 0 --long-seconds 0 "$@" 2>&1 >/dev/null | grep "^\[latent\]" > /tmp/soak.txt
  echo "$* : draws $(wc -l < /tmp/soak.txt), second attempts $(grep -c '2 attempt' /tmp/soak.txt), status != 0: $(grep -vc 'status 0' /tmp/soak.txt)"
}
soak --config 3 --task classification --steps 400 --warmup 2
soak --config 3 --task ordered --steps 200 --warmup 2
soak --config 2 --task classification --steps 2000 --warmup 2
soak --config 2 --task ordered --steps 1000 --warmup 2
soak --config 5 --scale 0.2 --steps 40 --warmup 2
