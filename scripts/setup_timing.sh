# mfm_finalize phase by phase (MFM_SETUP_TIMING=1) for configs 3, 4 and 5 at full size
Q="--cpu-iters 0 --fit-iters 0 --no-other-configs --no-kernel-timing --long-seconds 0"
T=${1:-setup}
MFM_SETUP_TIMING=1 python bench.py --steps 3 --warmup 1 $Q > gpurun_out/${T}_cfg3.json 2> gpurun_out/${T}_cfg3.err
MFM_SETUP_TIMING=1 python bench.py --config 4 --steps 3 --warmup 1 $Q > gpurun_out/${T}_cfg4.json 2> gpurun_out/${T}_cfg4.err
MFM_SETUP_TIMING=1 python bench.py --config 5 --scale 1.0 --steps 2 --warmup 1 $Q > gpurun_out/${T}_cfg5.json 2> gpurun_out/${T}_cfg5.err
grep -h "mfm_finalize\|setup" gpurun_out/${T}_cfg3.err | tail -40
echo ==== ; grep -h "mfm_finalize\|setup" gpurun_out/${T}_cfg4.err | tail -40
echo ==== ; grep -h "mfm_finalize\|setup" gpurun_out/${T}_cfg5.err | tail -60
