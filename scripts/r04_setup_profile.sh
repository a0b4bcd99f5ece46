# round 4, f4: mfm_finalize and the session setup phase by phase (MFM_SETUP_TIMING=1) for configs 3, 4, 5 at full size, and the
# bench lines that carry setup_s. Run on the GPU box through gpurun; outputs under gpurun_out/${T}_*.
T=${1:-r04_s}
Q="--cpu-iters 0 --fit-iters 0 --no-other-configs --no-kernel-timing --long-seconds 0"
for cfg in "3:--steps 200 --warmup 20" "4:--config 4 --steps 100 --warmup 5" "5:--config 5 --scale 1.0 --steps 4 --warmup 1"; do
  n=${cfg%%:*}; args=${cfg#*:}
  MFM_SETUP_TIMING=1 python bench.py $args $Q > gpurun_out/${T}_bench_config${n}.json 2> gpurun_out/${T}_setup_config${n}.err
  { echo "## bench.py $args $Q  (MFM_SETUP_TIMING=1)"; grep "^\[" gpurun_out/${T}_setup_config${n}.err; echo; } >> gpurun_out/${T}_setup_timing.txt
  rm -f gpurun_out/${T}_setup_config${n}.err
done
python - <<'PY' >> gpurun_out/${T:-r04_s}_setup_timing.txt
import json, glob, os
T = os.environ.get("T", "r04_s")
for f in sorted(glob.glob("gpurun_out/%s_bench_config*.json" % T)):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print("%-44s value %9.3f  setup_s %s  datagen_s %s  plan_flags %s" % (os.path.basename(f), d["value"], d["config"].get("setup_s"), d["config"].get("datagen_s"), d["config"].get("plan_flags")))
PY
