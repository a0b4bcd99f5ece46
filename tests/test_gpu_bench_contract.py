"""bench.py's output contract: ONE JSON line on stdout with the driver's keys, the roofline and cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, capture_output=True, text=True, timeout=900,
                       env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines  # native libraries' banners must not reach stdout
    return json.loads(lines[0])


def test_bench_line_small_config3():
    d = _run("--rows", "300000", "--steps", "4", "--warmup", "2", "--cpu-seconds", "2", "--fit-iters", "3", "--no-other-configs")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["cores"] == 1 and c["kind"] == "port" and c["value"] > 0
    assert d["fit"]["fit_it_per_s"] > 0


def test_bench_line_relation_blocks():
    d = _run("--config", "4", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--fit-iters", "0")
    assert d["config"]["relation_blocks"] and d["value"] > 0 and d["cpu_baseline"] is None


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher and no WORLD_SIZE (the driver's scaling command): bench.py spawns the two
    ranks itself and prints rank 0's single line. On a 1-GPU box both ranks share device 0 and the library's all-reduces go
    through torch.distributed / gloo (RCCL refuses two ranks on one device)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MYFM_BENCH_BACKEND="gloo", MYFM_BENCH_DEVICE="0")
    d = _run("--gpus", "2", "--rows", "300000", "--steps", "3", "--warmup", "1", "--weak-steps", "2", env=env)
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["scaling"] == "strong"
    c = d["config"]
    assert c["torch_world_size"] == 2 and c["allreduce_calls_per_step"] > 0 and c["rows_this_rank"] < c["rows"]
    assert c["rccl_ranks"] == 0 and c["rccl_path"] == ""  # (gloo carried the collectives here; on N GPUs: rccl_ranks == N)
    assert d["weak_scaling"]["it_per_s"] > 0
    # the in-launch exchange is opt-in (--peer-exchange); every rank's own collective count and volume are in the line
    assert c["peer_exchange"] is False
    assert [r["rank"] for r in c["per_rank"]] == [0, 1] and all(r["allreduce_calls_per_step"] > 0 and r["allreduce_bytes_per_step"] > 0 for r in c["per_rank"])
    assert sum(r["rows"] for r in c["per_rank"]) == c["rows"]


def test_bench_two_ranks_persistent_sweep_with_in_launch_exchange():
    """`python bench.py --gpus 2` with the row-sharded persistent sweep: both ranks on device 0 with a share of the CUs each
    (MFM_RES_CUS, no cross-process CU lock), exchange buffers through IPC handles; the line says so (`peer_exchange`) and the
    collectives per step are the per-iteration ones only."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MYFM_BENCH_BACKEND="gloo", MYFM_BENCH_DEVICE="0", MFM_RES_NO_PROCESS_LOCK="1", MFM_RES_CUS="100", MFM_RES_MIN_ROWS="0")
    d = _run("--gpus", "2", "--peer-exchange", "--rows", "300000", "--users", "3000", "--items", "2000", "--steps", "3", "--warmup", "1",
             "--weak-steps", "2", env=env)
    c = d["config"]
    assert d["n_gpus"] == 2 and d["value"] > 0 and c["peer_exchange"] is True, c
    assert c["plan_flags"] & 256 and c["plan_flags"] & 8
    assert c["allreduce_calls_per_step"] <= 6, c["allreduce_calls_per_step"]
    assert d["weak_scaling"]["peer_exchange"] is True and d["weak_scaling"]["it_per_s"] > 0


def test_bench_two_ranks_fall_back_when_the_exchange_fails_its_trial():
    """The same run with rank 1's flags held down (MFM_RES_XCH_BREAK): the persistent launches time out inside the trial iterations,
    every rank agrees to drop the path, and the line comes from fresh sessions on the per-factor passes."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MYFM_BENCH_BACKEND="gloo", MYFM_BENCH_DEVICE="0", MFM_RES_NO_PROCESS_LOCK="1", MFM_RES_CUS="100", MFM_RES_MIN_ROWS="0",
               MFM_RES_XCH_BREAK="1")
    d = _run("--gpus", "2", "--peer-exchange", "--rows", "300000", "--users", "3000", "--items", "2000", "--steps", "3", "--warmup", "1",
             "--weak-steps", "0", env=env)
    c = d["config"]
    assert d["n_gpus"] == 2 and d["value"] > 0 and c["peer_exchange"] is False, c
    assert not (c["plan_flags"] & 256) and c["allreduce_calls_per_step"] > 6


def test_bench_sharded_world1_reports_the_rccl_communicator():
    """world = 1 through the library's own RCCL communicator: the line proves which librccl carried the all-reduces."""
    env = dict(os.environ, MYFM_BENCH_FORCE_SHARDED="1")
    d = _run("--rows", "300000", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--fit-iters", "0", "--weak-steps", "0", env=env)
    c = d["config"]
    assert c["rccl_ranks"] == 1 and "librccl" in c["rccl_path"], c
