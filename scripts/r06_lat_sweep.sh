#!/bin/bash
# env sweep of the exact latent draws' geometry at config 5 (full size unless SCALE is set): flows time per draw
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export MFM_LATENT_TIMING=1
S=${SCALE:-1.0}
run() { echo "### $*"; env "$@" python bench.py --gpus 1 --fit-iters 0 --no-other-configs --no-kernel-timing --cpu-seconds 0 --config 5 --scale $S --latent exact --steps 2 --warmup 1 2>&1 | grep -E "^\[latent\]|\"value\"" | tail -2 | cut -c1-60,100-180,215-420; }
run MFM_LAT_CHUNKS=512 MFM_LAT_PLAIN_ROUNDS=1
run MFM_LAT_CHUNKS=512
run MFM_LAT_CHUNKS=1024 MFM_LAT_RES_NT=256
run MFM_LAT_CHUNKS=768 MFM_LAT_RES_NT=256
run MFM_LAT_CHUNKS=1536 MFM_LAT_RES_NT=256
run MFM_LAT_CHUNKS=512 MFM_LAT_RES_NT=256
