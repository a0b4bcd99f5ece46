# round 4: `bench.py --gpus 2` with both ranks on ONE GPU (124 CUs each, collectives over gloo): the row-sharded persistent sweep
# with the in-launch exchange against the per-factor passes with one all-reduce per factor. Not a scaling measurement (one GPU,
# CPU collectives) -- it shows the collectives per step and that the exchange path runs config 3 at full size.
T=${1:-r04_u}
export MYFM_BENCH_BACKEND=gloo MYFM_BENCH_DEVICE=0 MFM_RES_NO_PROCESS_LOCK=1 MFM_RES_CUS=124
Q="--gpus 2 --steps 30 --warmup 3 --weak-steps 0 --cpu-seconds 0 --fit-iters 0"
python bench.py $Q 2>gpurun_out/${T}_peer.err | tail -1 > gpurun_out/${T}_bench_2ranks_peer_exchange.json
MYFM_BENCH_NO_PEER_EXCHANGE=1 python bench.py $Q 2>gpurun_out/${T}_nopeer.err | tail -1 > gpurun_out/${T}_bench_2ranks_per_factor.json
python - <<PY
import json
for n in ("peer_exchange", "per_factor"):
    d = json.loads(open("gpurun_out/${T}_bench_2ranks_%s.json" % n).read())
    c = d["config"]
    print("%-14s value %8.2f it/s  ms/step %8.3f  allreduce_calls_per_step %5.1f  peer_exchange %s  plan_flags %s" % (n, d["value"], d["ms_per_step"], c["allreduce_calls_per_step"], c.get("peer_exchange"), c["plan_flags"]))
PY
tail -3 gpurun_out/${T}_peer.err
