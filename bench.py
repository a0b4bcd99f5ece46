#!/usr/bin/env python
"""bench.py -- Gibbs iterations/sec of the MI355X sampler, one JSON line.

Default workload = BASELINE configs[2]: MovieLens-10M-shaped synthetic CSR (N = 10 M rows, 69 878 + 10 677 one-hot
features, nnz = 20 M), MyFMRegressor rank 32, fp64, the full update_all per step (BaseFMTrainer.hpp:135-152).

    python bench.py --gpus N --steps K --warmup W
    (N > 1 without a launcher: bench.py starts its own N ranks -- python -m torch.distributed.run --nnodes=1
     --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py ... -- and passes rank 0's line through;
     under a launcher (WORLD_SIZE set) it is one of the ranks.)

N > 1: ONE chain over the SAME table, rows sharded over the N GPUs at user boundaries, the all-reduces issued by
libmyfm_hip.so through RCCL (strong scaling: `value` = iterations/s of that chain).

Other BASELINE workloads (not the judged line; same code path): --config 2 | 4 | 5 [--scale S].

Besides the contract's fields the line carries
  "roofline":     HBM roofline of the dominant kernel class (its own algorithmic bytes / HIP-event time in the timed region)
  "cpu_baseline": the CPU oracle (Eigen-free restatement of the reference, built here with -O3 -march=native), ONE thread
                  pinned to core 0, >= 30 s of full iterations of the same design
  "fit":          the rate of the path users call: MyFMRegressor(rank).fit(X, y, n_iter) with the default number of kept
                  samples and the default callback (create_train_fm: retention + callback inside the loop)
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def b_iter_bytes(N, nnz, D, K):
    """algorithmic bytes per Gibbs iteration, main table, regression (the UNFUSED algorithm)."""
    return nnz * (40 + 56 * K) + N * (40 + 8 * K) + D * (16 * K + 8)


def workload(cfg, a):
    """-> dict(X, blocks [(map, csr)], y, shapes, rank, task, name)"""
    from myfm_amd.utils import synthetic as ds

    if cfg == 3:
        X, y, shapes = ds.movielens_like(a.rows, a.users, a.items, rank_true=32, seed=1)
        return dict(X=X, blocks=[], y=y, shapes=shapes, rank=a.rank or 32, task="regression",
                    name="BASELINE configs[2]: MovieLens-10M-shaped synthetic CSR, MyFMRegressor rank=%d fp64, full update_all" % (a.rank or 32))
    if cfg == 2:
        X, y, shapes = ds.movielens_like(80000, 943, 1682, rank_true=8, seed=0, user_offset=30.0, item_offset=20.0)
        return dict(X=X, blocks=[], y=y, shapes=shapes, rank=a.rank or 8, task="regression",
                    name="BASELINE configs[1]: MovieLens-100k-shaped one-hot CSR (943 + 1682 features, N = 80 000), rank=%d" % (a.rank or 8))
    if cfg == 4:
        main, blocks, y, shapes = ds.ml100k_extended_like()
        return dict(X=main, blocks=blocks, y=y, shapes=shapes, rank=a.rank or 16, task="regression",
                    name="BASELINE configs[3]: ML-100k-extended-shaped RelationBlock design (user / movie side information), rank=%d" % (a.rank or 16))
    if cfg == 5:
        main, blocks, y, shapes = ds.config5_like(a.scale, ordered=True)
        return dict(X=main, blocks=blocks, y=y, shapes=shapes, rank=a.rank or 64, task="ordered",
                    name="BASELINE configs[4] at scale %g: N = %d rows, nnz = %d + 4 relation blocks, MyFMOrderedProbit rank=%d"
                         % (a.scale, main.shape[0], main.nnz, a.rank or 64))
    raise SystemExit("--config must be 2, 3, 4 or 5")


def make_config(_myfm, gi, n_iter, n_kept, task, n_rows, latent=None):
    b = _myfm.ConfigBuilder()
    b.set_alpha_0(1.0).set_beta_0(1.0).set_gamma_0(1.0).set_mu_0(0.0).set_reg_0(1.0)
    b.set_group_index([int(g) for g in gi]).set_n_iter(n_iter).set_n_kept_samples(n_kept)
    if task == "ordered":
        b.set_task_type(_myfm.TaskType.ORDERED)
        b.set_cutpoint_groups([(5, np.arange(n_rows))])
    elif task == "classification":
        b.set_task_type(_myfm.TaskType.CLASSIFICATION)
    else:
        b.set_task_type(_myfm.TaskType.REGRESSION)
    if latent is not None:  # the latent-draw mode of classification / ordered probit: "exact" | "host" | "philox"
        b.set_latent_mode(latent)
    return b.build()


def retarget(W, task):
    """--task: the same design with the target of another task type (BASELINE.md 1 quotes classification and ordered-probit
    rates of the reference next to regression): probit labels y > 3.5, ordered classes round(y) - 1 in 0..4."""
    if task is None or task == W["task"]:
        return W
    y = np.asarray(W["y"], dtype=np.float64)
    if W["task"] != "regression":
        raise SystemExit("--task: only the regression workloads (configs 2, 3, 4) can be retargeted")
    W = dict(W)
    if task == "classification":
        W["y"] = np.where(y > 3.5, 1.0, -1.0)
    elif task == "ordered":
        W["y"] = (np.clip(np.round(y), 1, 5) - 1).astype(np.float64)
    else:
        raise SystemExit("--task must be regression, classification or ordered")
    W["task"] = task
    W["name"] = W["name"] + " [target of task %s]" % task
    return W


def cpu_baseline(W, gi, min_seconds, max_iters):
    """the oracle, 1 thread pinned to core 0, full iterations of the same design, in a subprocess (affinity, fresh build)"""
    import pickle
    import tempfile

    if min_seconds <= 0:
        return None
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "w.pkl")
        with open(f, "wb") as fh:
            pickle.dump(dict(X=W["X"], blocks=W["blocks"], y=W["y"], gi=gi, rank=W["rank"], task=W["task"]), fh, protocol=4)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "cpu_baseline.py"), f, str(min_seconds), str(max_iters)],
                           capture_output=True, text=True, cwd=ROOT)
    if r.returncode != 0:
        return {"error": (r.stderr or r.stdout)[-400:]}
    return json.loads(r.stdout.strip().splitlines()[-1])


def replica_digest(sess):
    """64-bit digest of everything a rank's replica of the chain holds (w0, w, V, every hyper-parameter): replicas are compared
    bit for bit, not through a sum that could hide a swapped pair of coefficients."""
    import hashlib

    fm, hy = sess.fm, sess.hyper
    h = hashlib.blake2b(digest_size=8)
    for arr in (np.float64(fm.w0), np.asarray(fm.w), np.asarray(fm.V), np.float64(hy.alpha), np.asarray(hy.mu_w), np.asarray(hy.lambda_w),
                np.asarray(hy.mu_V), np.asarray(hy.lambda_V)):
        h.update(np.ascontiguousarray(arr, dtype=np.float64).tobytes())
    return int.from_bytes(h.digest(), "little")


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher: run the N ranks under torch.distributed.run (one process per GPU,
    rendezvous on 127.0.0.1) and pass rank 0's single JSON line through. Returns the exit code."""
    import socket

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n_gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: RCCL across processes needs it on this driver)
    env.setdefault("OMP_NUM_THREADS", "8")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, cwd=ROOT)
    lines = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith('{"metric"')]
    for l in r.stdout.decode(errors="replace").splitlines():
        if not l.startswith('{"metric"') and l.strip():
            print(l, file=sys.stderr)
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    return r.returncode if (r.returncode != 0 or lines) else 1


def other_configs(budget_s):
    """BASELINE's other workloads through the same code path, each as `bench.py --config C` in a subprocess (a fresh context,
    bounded samples): iterations/s on the GPU next to the pinned CPU oracle. Reported inside the default line."""
    out = {}
    runs = [("configs[1]", ["--config", "2", "--steps", "300", "--warmup", "20", "--cpu-seconds", "3"]),
            ("configs[3]", ["--config", "4", "--steps", "40", "--warmup", "5", "--cpu-seconds", "3"]),
            ("configs[4] at scale 0.1", ["--config", "5", "--scale", "0.1", "--steps", "6", "--warmup", "2", "--cpu-seconds", "1"]),
            # ... and at its own size, N = 50 M rows (one CPU-oracle iteration there is ~130 s: profiles/r03_e_bench_config5_full_cpu.json)
            # -- the default latent mode ("exact": the reference's own stream, draw for draw) and the per-row Philox streams
            ("configs[4] at full size", ["--config", "5", "--scale", "1.0", "--steps", "5", "--warmup", "2", "--cpu-seconds", "0", "--kernel-timing"]),
            ("configs[4] at full size, latent=philox", ["--config", "5", "--scale", "1.0", "--steps", "5", "--warmup", "2", "--cpu-seconds", "0",
                                                        "--latent", "philox"]),
            # the task types BASELINE.md 1 quotes next to regression (examples/ml-100k.ipynb: classification 48.85 it/s, ordered probit
            # 20.40 it/s on the reference's CPU): configs[1]'s and configs[2]'s designs with those targets, both latent modes
            ("tasks: classification, configs[1] shape", ["--config", "2", "--task", "classification", "--steps", "200", "--warmup", "10", "--cpu-seconds", "3"]),
            ("tasks: classification, configs[1] shape, latent=philox", ["--config", "2", "--task", "classification", "--latent", "philox", "--steps", "300",
                                                                        "--warmup", "20", "--cpu-seconds", "0"]),
            ("tasks: ordered probit, configs[1] shape", ["--config", "2", "--task", "ordered", "--steps", "200", "--warmup", "10", "--cpu-seconds", "3"]),
            ("tasks: ordered probit, configs[1] shape, latent=philox", ["--config", "2", "--task", "ordered", "--latent", "philox", "--steps", "300",
                                                                        "--warmup", "20", "--cpu-seconds", "0"]),
            ("tasks: classification, configs[2] shape", ["--config", "3", "--task", "classification", "--steps", "30", "--warmup", "3", "--cpu-seconds", "0"]),
            ("tasks: classification, configs[2] shape, latent=philox", ["--config", "3", "--task", "classification", "--latent", "philox", "--steps", "60",
                                                                        "--warmup", "5", "--cpu-seconds", "0"])]
    t_all = time.time()
    for name, args in runs:
        if time.time() - t_all > budget_s:
            out[name] = {"skipped": "time budget"}
            continue
        t0 = time.time()
        timing_args = [] if "--kernel-timing" in args else ["--no-kernel-timing"]
        args = [x for x in args if x != "--kernel-timing"]
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", "1", "--fit-iters", "0", "--no-other-configs"] + timing_args + args,
                           capture_output=True, text=True, cwd=ROOT)
        line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
        if r.returncode != 0 or not line:
            out[name] = {"error": (r.stderr or r.stdout)[-300:]}
            continue
        d = json.loads(line[-1])
        cpu = d.get("cpu_baseline") or {}
        out[name] = {"workload": d["config"]["workload"], "task": d["config"].get("task"), "latent": d["config"].get("latent"),
                     "it_per_s": d["value"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                     "cpu_it_per_s": cpu.get("value"), "cpu_iterations": cpu.get("iterations"), "cpu_pinned": cpu.get("pinned_to_one_core"),
                     "setup_s": d["config"]["setup_s"], "wall_s": round(time.time() - t0, 1)}
        if d.get("roofline"):  # (the legs run with kernel timing: dominant class, time per chain column, the passes' HBM fractions)
            rf = d["roofline"]
            out[name]["roofline"] = {k: rf.get(k) for k in ("kernel", "avg_launch_us", "achieved", "frac", "unit", "kernel_ms_per_step",
                                                            "chain_us_per_column", "chain_columns_per_step", "by_kernel_hbm")}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed iterations (0: enough for >= 10 s of GPU work at config 3, so that "
                    "a 5-second utilisation sampler sees the timed region; 100 for the other configs)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--scale", type=float, default=1.0, help="config 5: fraction of the full N = 50 M shape")
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--users", type=int, default=69878)
    ap.add_argument("--items", type=int, default=10677)
    ap.add_argument("--rank", type=int, default=0, help="0: the config's own rank")
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="CPU-oracle time budget (0 disables)")
    ap.add_argument("--cpu-iters", type=int, default=-1, help="(compat) 0 disables the CPU baseline")
    ap.add_argument("--fit-iters", type=int, default=100, help="iterations of the MyFM*.fit() leg (0 disables)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--weak-steps", type=int, default=20, help="N > 1: timed iterations of the weak-scaling leg (0 disables)")
    ap.add_argument("--long-seconds", type=float, default=2.0, help="when the timed region is shorter than 1 s: also time the same loop "
                    "for about this long and report it as value_long (0 disables)")
    ap.add_argument("--peer-exchange", action="store_true", help="N > 1, config 3: the persistent sweep row-sharded with the ranks' item sums "
                    "exchanged inside the launch (never run on two physical GPUs yet: opt-in; a trial of three iterations decides)")
    ap.add_argument("--no-other-configs", action="store_true", help="default run (config 3, 1 GPU): skip the other_configs legs")
    ap.add_argument("--task", default=None, help="regression | classification | ordered: the config's design with another task's target")
    ap.add_argument("--latent", default=None, help="latent draws of classification / ordered probit: exact (the reference's own stream, "
                    "draw for draw, evaluated on the device) | host (the same on the host) | philox (per-row streams)")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a.gpus))
    if a.cpu_iters == 0:
        a.cpu_seconds = 0.0
    if a.steps <= 0:
        a.steps = 2000 if a.config == 3 else (100 if a.config != 5 else 10)

    # stdout carries exactly ONE line (the JSON): whatever native libraries print there (RCCL's version banner is flushed
    # at exit) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and not (a.gpus == 1 and world == 1):
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (a.gpus, world))

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the sampler has no CPU fallback")
    # MYFM_BENCH_BACKEND=gloo MYFM_BENCH_DEVICE=0: several ranks on ONE GPU (the multi-process flow on a 1-GPU box: RCCL
    # refuses two ranks on a device, so the all-reduces then go through torch.distributed / gloo from the library's callback)
    backend = os.environ.get("MYFM_BENCH_BACKEND", "nccl")
    dev = int(os.environ.get("MYFM_BENCH_DEVICE", local_rank))
    if backend != "nccl":
        os.environ["MYFM_BENCH_TORCH_ALLREDUCE"] = "1"
    torch.cuda.set_device(dev)
    os.environ["MYFM_AMD_DEVICE"] = str(dev)  # one process per GPU
    dist = None
    force_sharded = bool(os.environ.get("MYFM_BENCH_FORCE_SHARDED"))  # exercise the N > 1 code path at world = 1
    if world > 1 or force_sharded:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if not any(f.startswith("_myfm.") and f.endswith(".so") for f in os.listdir(os.path.join(ROOT, "myfm_amd"))):
        import __graft_entry__ as _g  # (a tree without the built extension: build it in place, hipcc cross-compiles)

        _g.build()
    from myfm_amd import _capi, _myfm
    from myfm_amd.utils import synthetic as ds

    t0 = time.time()
    W = retarget(workload(a.config, a), a.task)
    X, y, blocks, K = W["X"], W["y"], W["blocks"], W["rank"]
    gi = ds.group_index_from_shapes(W["shapes"])
    N, D0, nnz = X.shape[0], X.shape[1], X.nnz
    D = len(gi)
    t_data = time.time() - t0

    t0 = time.time()
    sharded = world > 1 or force_sharded
    lo, hi = 0, N
    parallelism = "1 GPU"
    if not sharded:
        rels = [_myfm.RelationBlock(np.asarray(m, dtype=np.int64), B) for m, B in blocks]
        t1 = time.time()
        cfg1 = make_config(_myfm, gi, a.steps + a.warmup + 8, 0, W["task"], N, a.latent)
        t2 = time.time()
        sess = _myfm.GibbsSession(K, 0.1, X, rels, y, 42, cfg1)
        if os.environ.get("MFM_SETUP_TIMING"):
            print("[bench] RelationBlock %.3f s, config %.3f s, GibbsSession %.3f s" % (t1 - t0, t2 - t1, time.time() - t2),
                  file=sys.stderr)
    else:
        from myfm_amd import distributed as mdist

        # strong scaling: the SAME table, rows [lo, hi) on this rank, cut between two users; the model state and the random
        # variates are replicated (same seed); libmyfm_hip.so issues the all-reduces itself (ncclAllReduce on its stream)
        levels, _ = _capi.column_levels(X)
        cuts = mdist.shard_cuts(X.indices[X.indptr[:-1]], world)
        lo, hi = cuts[rank], cuts[rank + 1]
        rels = [_myfm.RelationBlock(np.asarray(m, dtype=np.int64)[lo:hi], B) for m, B in blocks]
        cfg = make_config(_myfm, gi, a.steps + a.warmup + 12, 0, W["task"], hi - lo, a.latent)

        def make_session():
            sess, err = None, ""
            if not os.environ.get("MYFM_BENCH_TORCH_ALLREDUCE"):
                try:
                    cid = mdist.native_comm_id()
                    sess = _myfm.GibbsSession(K, 0.1, X[lo:hi], rels, y[lo:hi], 42, cfg, n_total_rows=N, row_offset=lo,
                                              main_levels=levels, comm_id=cid, shard_rank=rank, shard_world=world)
                except Exception as ex:  # noqa: BLE001 -- reported below; every rank must take the same branch
                    err = "%s: %s" % (type(ex).__name__, ex)
            ok = torch.tensor([1 if sess is not None else 0], device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            how = "RCCL all-reduce called by libmyfm_hip.so on its stream"
            if int(ok.item()) == 0:
                # the library could not open its own communicator on some rank: the same all-reduces through torch.distributed
                # (the same librccl, called back from the library on a torch stream)
                print("bench.py rank %d: native RCCL communicator unavailable (%s); using the torch.distributed callback" % (rank, err),
                      file=sys.stderr)
                sess = None
                ar = mdist.TorchAllReduce()
                sess = _myfm.GibbsSession(K, 0.1, X[lo:hi], rels, y[lo:hi], 42, cfg, allreduce=ar, n_total_rows=N, row_offset=lo,
                                          stream=ar.stream_ptr, main_levels=levels, shard_rank=rank, shard_world=world)
                how = "all-reduce through torch.distributed (%s) called back from libmyfm_hip.so on a shared stream" % backend
            return sess, how

        sess, how = make_session()
        # The persistent sweep row-sharded: every rank keeps the residual of its rows on chip, the ranks' item sums
        # meet INSIDE the launch through IPC-mapped exchange buffers (no collective between the sweeps of an iteration). It has
        # only ever run with the ranks side by side on ONE GPU (one L2, no xGMI), so it is OPT-IN here as in distributed.enable():
        # --peer-exchange / MYFM_BENCH_PEER_EXCHANGE=1. Even then a trial decides: three iterations, after each of which every rank's
        # 64-bit digest of (w0, w, V, every hyper-parameter) must be the same; a rank that times out, or replicas that differ, send
        # every rank back to the per-factor passes (RCCL all-reduce per level) on a fresh session.
        peer_live = False
        want_peer = a.peer_exchange or os.environ.get("MYFM_BENCH_PEER_EXCHANGE", "") not in ("", "0")
        if not blocks and want_peer and not os.environ.get("MYFM_BENCH_NO_PEER_EXCHANGE"):
            peer_live = mdist.connect_peers(sess)
            if peer_live:
                good = 1.0
                for _trial in range(3):
                    dig = 0
                    try:
                        # (after every iteration the stream is checked: a rank whose launch timed out raises here, after it has
                        #  issued the same collectives as everybody else)
                        sess.step()
                        sess.synchronize()
                        dig = replica_digest(sess)
                    except Exception as ex:  # noqa: BLE001 -- every rank takes part in the agreement below
                        print("bench.py rank %d: row-sharded persistent sweep failed its trial (%s: %s)" % (rank, type(ex).__name__, ex),
                              file=sys.stderr)
                        good = 0.0
                    # (the digest as two 32-bit halves: exact in float64)
                    t = torch.tensor([good, float(dig >> 32), -float(dig >> 32), float(dig & 0xFFFFFFFF), -float(dig & 0xFFFFFFFF)],
                                     dtype=torch.float64, device="cuda")
                    dist.all_reduce(t, op=dist.ReduceOp.MIN)
                    if float(t[0]) < 1.0 or float(t[1]) != -float(t[2]) or float(t[3]) != -float(t[4]):
                        good = 0.0
                        break
                if good < 1.0:
                    peer_live = False
                    sess = None
                    import gc

                    gc.collect()
                    os.environ["MFM_NO_SHARDED_RESIDENT"] = "1"
                    sess, how = make_session()
        if peer_live:
            parallelism = ("one chain over the same %d rows, sharded over %d GPUs at user boundaries (%d rows on rank 0); persistent sweep "
                           "on every rank, the ranks' item sums exchanged inside the launch through IPC-mapped buffers over xGMI; %s: per "
                           "iteration the residual sums and one model synchronisation" % (N, world, hi - lo, how))
        else:
            parallelism = ("one chain over the same %d rows, sharded over %d GPUs at user boundaries (%d rows on rank 0); %s: per factor "
                           "the item level's statistics, per sweep one model synchronisation" % (N, world, hi - lo, how))
    t_setup = time.time() - t0

    def sync():
        sess.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # Kernel timing: bracketing EVERY launch with HIP events costs ~10 % of an iteration at config 3, so the per-class
    # breakdown comes from 3 extra diagnostic steps before the warm-up (not part of the timed region), and inside the timed
    # region only the dominant class is bracketed.
    breakdown, dom_name = {}, None
    if not a.no_kernel_timing:
        sess.step()
        sess.timing_enable(True)
        sess.timing_reset()
        for _ in range(3):
            sess.step()
        sess.synchronize()
        breakdown = {k: (v[0] / 3.0, v[1] / 3.0, v[2] / 3.0) for k, v in dict(sess.timing()).items()}
        sess.timing_enable(False)
        if breakdown:
            dom_name = max(breakdown.items(), key=lambda kv: kv[1][0])[0]
    for _ in range(a.warmup):
        sess.step()
    # (several ranks: no events inside the timed region -- the dominant class then contains the collective, and its figures come
    # from the diagnostic steps above)
    live_timing = bool(dom_name) and world == 1
    if live_timing:
        sess.timing_select(dom_name)
        sess.timing_enable(True)
        sess.timing_reset()
    calls0, doubles0 = (sess.comm_stats()[0], sess.comm_stats()[1]) if sharded else (0, 0)
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        sess.step()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    timing = dict(sess.timing()) if live_timing else ({dom_name: breakdown[dom_name]} if dom_name else {})
    if live_timing:
        sess.timing_enable(False)
        sess.timing_select("")
    calls = (sess.comm_stats()[0] - calls0) if sharded else 0
    doubles = (sess.comm_stats()[1] - doubles0) if sharded else 0
    per_rank = None
    if dist is not None and world > 1:  # every rank's own collective count and volume (a first multi-GPU run must be readable)
        mine = torch.tensor([float(rank), float(hi - lo), calls / a.steps, 8.0 * doubles / a.steps], dtype=torch.float64, device="cuda")
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": int(v[0]), "rows": int(v[1]), "allreduce_calls_per_step": round(float(v[2]), 1),
                     "allreduce_bytes_per_step": round(float(v[3]))} for v in allr]
    # a short timed region (the driver's 20 steps at config 3 are 65 ms) gets an in-line cross-check: the same loop for >= 2 s
    long_run = None
    if world == 1 and elapsed < 1.0 and a.long_seconds > 0:
        n_long = max(a.steps, int(np.ceil(a.long_seconds / max(elapsed / a.steps, 1e-6))))
        sync()
        t0 = time.perf_counter()
        for _ in range(n_long):
            sess.step()
        sync()
        t_long_run = time.perf_counter() - t0
        long_run = (n_long, t_long_run)

    # sanity: the chain is alive and, when sharded, the replicated model is identical on every rank
    alpha = sess.hyper.alpha
    assert np.isfinite(alpha) and alpha > 0, alpha
    if dist is not None and world > 1:
        dig = replica_digest(sess)
        mine = torch.tensor([float(dig >> 32), float(dig & 0xFFFFFFFF)], dtype=torch.float64, device="cuda")
        allv = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        for v in allv[1:]:
            assert torch.equal(v, allv[0]), ("replicas diverged (64-bit digests of w0, w, V and the hyper-parameters)", [x.tolist() for x in allv])

    plan_flags = int(sess.plan_flags())
    latent_info = dict(sess.latent_info()) if hasattr(sess, "latent_info") else {}
    rccl_ranks, rccl_path = sess.comm_info() if sharded else (0, "")
    # Weak-scaling leg (N > 1, config 3): the table grows with the GPUs -- rank r holds ~a.rows rows of ONE user-sorted table of
    # world * a.rows rows over the same users / items (myfm_amd/utils/synthetic.py::movielens_like_shard). Reported as an extra field;
    # `value` stays the strong-scaling rate of config 3 itself.
    weak = None
    if sharded and a.config == 3 and a.weak_steps > 0 and not blocks:
        del sess
        import gc

        gc.collect()  # (the context gives its CUs back before the weak-scaling session asks for them)
        Xw, yw, shw, lo_w, total_w = ds.movielens_like_shard(a.rows, rank, world, a.users, a.items)
        levels_w = np.concatenate([np.zeros(a.users, np.int32), np.ones(a.items, np.int32)])
        gi_w = ds.group_index_from_shapes(shw)
        cfg_w = make_config(_myfm, gi_w, a.weak_steps + 8, 0, "regression", Xw.shape[0])
        if os.environ.get("MYFM_BENCH_TORCH_ALLREDUCE") or "torch.distributed" in how:
            ar_w = mdist.TorchAllReduce()
            sw = _myfm.GibbsSession(K, 0.1, Xw, [], yw, 42, cfg_w, allreduce=ar_w, n_total_rows=total_w, row_offset=lo_w,
                                    stream=ar_w.stream_ptr, main_levels=levels_w, shard_rank=rank, shard_world=world)
        else:
            sw = _myfm.GibbsSession(K, 0.1, Xw, [], yw, 42, cfg_w, n_total_rows=total_w, row_offset=lo_w, main_levels=levels_w,
                                    comm_id=mdist.native_comm_id(), shard_rank=rank, shard_world=world)
        # (the strong leg's trial has shown that the in-launch exchange works between these ranks: the weak leg takes it too)
        peer_w = bool(peer_live) and mdist.connect_peers(sw)
        for _ in range(3):
            sw.step()
        sw.synchronize()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(a.weak_steps):
            sw.step()
        sw.synchronize()
        torch.cuda.synchronize()
        dist.barrier()
        tw = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        tw = float(tw.item())
        if world > 1:  # the replicated model is identical on every rank here too
            fw = sw.fm
            mine = torch.tensor([float(fw.w0), float(sw.hyper.alpha), float(np.abs(np.asarray(fw.V)).sum()), float(np.asarray(fw.w).sum())],
                                dtype=torch.float64, device="cuda")
            allv = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(allv, mine)
            for v in allv[1:]:
                assert torch.allclose(v, allv[0], rtol=1e-12, atol=0), ("weak-scaling replicas diverged", [x.tolist() for x in allv])
        weak = {"rows_total": int(total_w), "rows_per_gpu": int(a.rows), "steps": a.weak_steps, "it_per_s": round(a.weak_steps / tw, 3),
                "allreduce_calls_per_step": round(sw.comm_stats()[0] / (a.weak_steps + 3), 1),
                "ms_per_step": round(tw / a.weak_steps * 1e3, 3), "row_iterations_per_s": round(total_w * a.weak_steps / tw),
                "peer_exchange": bool(peer_w),
                "note": "ONE chain over a user-sorted table of world x rows_per_gpu rows (same users / items), every rank holds a "
                        "contiguous range of users; timing as for `value` (barrier + synchronize, max over ranks)"}
        del sw

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    it_per_s = a.steps / elapsed  # iterations of THE chain per second (all N GPUs work on it)

    roofline = None
    if timing:
        name, (ms, launches, alg_bytes) = max(timing.items(), key=lambda kv: kv[1][0])
        achieved = alg_bytes / (ms * 1e-3) / 1e9
        # HBM bytes per launch of that class from the committed PMC passes of THIS round (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE of the same command, corrected as MI355X_MICROARCH.md prescribes; profiles/r03_*_pmc_traffic.*)
        traffic, src = None, None
        default_workload = a.config == 3 and (a.rows, a.users, a.items, K) == (10_000_000, 69878, 10677, 32) and world == 1
        pdir = os.path.join(ROOT, "profiles")
        import re as _re
        pmc = sorted(f for f in os.listdir(pdir) if _re.fullmatch(r"r\d\d_[a-z]_pmc_traffic\.json", f)) if os.path.isdir(pdir) else []
        if default_workload and pmc:
            tr = json.load(open(os.path.join(pdir, pmc[-1])))
            if name in tr:
                traffic, src = round(tr[name]["bytes_per_level_launch"]), pmc[-1]
        us = ms / launches * 1e3
        roofline = {
            "bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic, "traffic_source": src, "traffic_measured": False,
            "traffic_gbs": round(traffic / (us * 1e-6) / 1e9, 1) if traffic else None,
            "traffic_over_algorithmic": round(traffic / (alg_bytes / launches), 3) if traffic else None,
            "avg_launch_us": round(us, 2), "launches": int(launches), "alg_bytes_per_launch": round(alg_bytes / launches),
            "note": "achieved = the kernel's OWN algorithmic bytes / HIP-event time of its launches in the timed region. "
                    "sweep_V_resident = update_w0's shift + update_w + update_V of a two-field table as ONE persistent launch, "
                    "the residual on chip for all K + 1 sweeps: e read once (8 B / row, in slot order) and NOT written back -- "
                    "update_e, which follows, recomputes it (FMTrainer.hpp:494), so the launch's copy would be a dead store "
                    "(mfm_set_residual_policy; MYFM_AMD_KEEP_RESIDUAL=1 keeps it: + 8 B / row) --, per sweep one 16-byte "
                    "statistics partial per (workgroup, item) run written + read, "
                    "its 8-byte list entry and two 4-byte item reads. The launch is bound by instruction issue, LDS atomics "
                    "and two grid barriers per sweep, not by HBM: its fraction of the HBM roofline is low BECAUSE the bytes are "
                    "gone (r02's per-factor pass moved 289 MB per factor, this one 98 MB); see DESIGN.md 4.3. "
                    "sweep_V_fused_next (other shapes) = one per-factor pass of the two-field latent sweep.",
            "kernel_ms_per_step": round(sum(v[0] for v in breakdown.values()), 3),
            "by_kernel_ms_per_step": {k: round(v[0], 3) for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1][0])},
            "by_kernel_note": "per-class times from 3 diagnostic steps with every launch bracketed (outside the timed region)",
            # every class on its own algorithmic bytes (the same accounting as `achieved`), from the diagnostic steps
            "by_kernel_hbm": {k: {"ms_per_step": round(v[0], 3), "launches_per_step": round(v[1], 1),
                                  "alg_gbs": round(v[2] / (v[0] * 1e-3) / 1e9, 1) if v[0] > 0 and v[2] > 0 else None,
                                  "frac": round(v[2] / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if v[0] > 0 and v[2] > 0 else None}
                              for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1][0])[:8]},
        }
        if blocks and "block_sweep" in breakdown:
            # the block-feature chains are sequential in the column order (FMTrainer.hpp:276-302, :419-470): their figure of merit is time
            # per column, not bandwidth
            n_cols_iter = sum(int(B.shape[1]) for _, B in blocks) * (K + 1)
            roofline["chain_us_per_column"] = round(breakdown["block_sweep"][0] * 1e3 / max(n_cols_iter, 1), 3)
            roofline["chain_columns_per_step"] = n_cols_iter
        # second bound of the same kernel: VALU instruction issue. One wave-instruction per CU and clock is the machine's rate
        # (4 SIMDs x 16 lanes, wave64); the instruction count per launch comes from this round's committed SQ-counter pass of the
        # same workload (rocprofv3 --pmc SQ_INSTS_VALU ..., profiles/r04_sq_counters_config3.*), the launch time is live
        sq = sorted(f for f in os.listdir(pdir) if _re.fullmatch(r"r\d\d_sq_counters_config3\.json", f)) if os.path.isdir(pdir) else []
        if default_workload and sq:
            cnt = json.load(open(os.path.join(pdir, sq[-1]))).get(name)
            if cnt:
                n_cu, clock_hz = 256, 2.4e9
                insts = cnt["insts_valu"]
                ach = insts / (us * 1e-6)
                roofline["second_bound"] = {
                    "bound": "valu_issue", "achieved": round(ach / 1e9, 1), "peak": round(n_cu * clock_hz / 1e9, 1), "unit": "G wave-instructions/s",
                    "frac": round(ach / (n_cu * clock_hz), 4), "valu_wave_instructions_per_launch": insts,
                    "all_wave_instructions_per_launch": insts + cnt["insts_salu"] + cnt["insts_lds"] + cnt["insts_vmem_rd"] + cnt["insts_vmem_wr"],
                    "wave_cycles_waiting_frac": cnt["wave_wait_any_frac"], "lds_active_frac": cnt["lds_active_frac"],
                    "lds_bank_conflict_frac_of_active": cnt["lds_bank_conflict_frac_of_active"], "counters_source": sq[-1], "counters_measured_live": False,
                    "note": "no throughput unit of the launch is near its peak (HBM, VALU issue, LDS each <= 0.3): it is bound by dependent "
                            "latency inside the sweeps and by its grid barriers (62 % of the wave-cycles wait), DESIGN.md 4.3"}
        if a.config in (2, 3) and not blocks:
            B_iter = b_iter_bytes(N, nnz, D, K)
            unfused = 56.0 * nnz + 8.0 * N + 8.0 * D
            roofline["survey_8d"] = {
                "alg_bytes_per_iteration_unfused": B_iter,
                "iteration_gbs": round(B_iter * it_per_s / 1e9, 1),
                "iteration_frac": round(B_iter * it_per_s / 1e9 / HBM_PEAK_GBS, 4),
                "updateV_gbs": round(unfused * K / (sum(v[0] for k2, v in breakdown.items() if k2.startswith("sweep_V")) * 1e-3) / 1e9, 1)
                if breakdown else None,
                "note": "SURVEY 8d's byte model of the UNFUSED algorithm ((56 nnz + 8 N + 8 D) K per update_V, B_iter per "
                        "iteration) divided by measured time: 'bytes the fusion avoids', not achieved bandwidth",
            }

    # ---- the path users call: MyFM*.fit() (create_train_fm: retention of the last n_kept samples + callback per iteration)
    fit = None
    if a.fit_iters > 0 and world == 1 and a.config in (2, 3, 4):
        import myfm_amd

        del sess
        rbs = [myfm_amd.RelationBlock(np.asarray(m, dtype=np.int64), B) for m, B in blocks]
        def timed_fit(n_iter):
            est = myfm_amd.MyFMRegressor(K)
            f0 = time.perf_counter()
            est.fit(X, y, rbs, group_shapes=W["shapes"], n_iter=n_iter)
            return time.perf_counter() - f0, len(est.predictor_.samples)

        # two fits of different length (after a short one that pays the first-call costs): the difference is free of the
        # one-off part (design upload, plan, first iteration)
        timed_fit(5)
        for _attempt in range(3):  # (short fits on a cold box: the one-off part can jitter by more than the difference)
            t_short, _ = timed_fit(a.fit_iters)
            t_long, kept = timed_fit(3 * a.fit_iters)
            per_it = (t_long - t_short) / (2 * a.fit_iters)
            if per_it > 0:
                break
        if per_it <= 0:
            per_it = t_long / (3 * a.fit_iters)  # (upper bound on the time per iteration: includes the setup)
        fit = {"fit_it_per_s": round(1.0 / per_it, 3), "n_iter": [a.fit_iters, 3 * a.fit_iters], "n_kept_samples": kept,
               "fit_seconds": [round(t_short, 3), round(t_long, 3)], "ratio_to_value": round(1.0 / per_it / it_per_s, 3),
               "note": "MyFMRegressor(rank).fit(X, y, n_iter): default n_kept_samples (n_iter - 5), default callback, row-order check "
                       "included; per-iteration rate = (fit(3 n) - fit(n)) / 2 n, a kept sample snapshotted (device to device) every iteration"}

    cpu = None
    if a.cpu_seconds > 0 and world == 1:
        cpu = cpu_baseline(W, gi, a.cpu_seconds, max_iters=2000)

    out = {
        "metric": "Gibbs iterations/sec (rank=%d)" % K,
        "value": round(it_per_s, 3),
        "unit": "Gibbs iterations/sec",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": W["name"], "task": W["task"],
            # how the latent z of classification / ordered probit are drawn: "exact" = from the reference's own random
            # stream, draw for draw, in parallel on the device; "philox" = per-row counter-based streams; "host" = the host loop
            "latent": latent_info.get("mode") if W["task"] != "regression" else None,
            "latent_draw": latent_info if W["task"] != "regression" else None,
            "rows": N, "nnz": nnz, "features": D, "main_features": D0, "rank": K, "groups": int(max(gi)) + 1,
            "relation_blocks": [{"rows": int(B.shape[0]), "features": int(B.shape[1]), "nnz": int(B.nnz)} for _, B in blocks],
            "parallelism": parallelism,
            "row_iterations_per_s": round(N * it_per_s),
            "setup_s": round(t_setup, 2), "datagen_s": round(t_data, 2),
            "plan_flags": plan_flags,
            # regression: update_e recomputes e = score - y after every update_V (FMTrainer.hpp:494), so the persistent sweep does not
            # write its on-chip residual back (a dead store); MYFM_AMD_KEEP_RESIDUAL=1 restores the write-back (- 2.5 % at config 3)
            "residual_write_back": bool(os.environ.get("MYFM_AMD_KEEP_RESIDUAL")) or W["task"] != "regression",
        },
        "roofline": roofline,
        "cpu_baseline": cpu,
        "fit": fit,
    }
    if long_run:
        out["value_long"] = round(long_run[0] / long_run[1], 3)
        out["value_long_steps"] = long_run[0]
        out["value_long_seconds"] = round(long_run[1], 3)
    if cpu and cpu.get("value"):
        out["speedup_vs_cpu_baseline"] = round(it_per_s / cpu["value"], 1)
    if weak:
        out["weak_scaling"] = weak
        # (top level, so that a first N > 1 curve reads without digging: `value` is the STRONG-scaling rate of one chain over the same
        #  10 M rows; the weak-scaling leg is one chain over N x 10 M rows)
        out["weak_scaling_it_per_s"] = weak["it_per_s"]
        out["weak_scaling_row_iterations_per_s"] = weak["row_iterations_per_s"]
    if a.config == 3 and world == 1 and not a.no_other_configs and not force_sharded:
        out["other_configs"] = other_configs(budget_s=300.0)
    if sharded:
        out["config"]["allreduce_calls_per_step"] = round(calls / a.steps, 1)
        out["config"]["allreduce_bytes_per_step"] = round(8.0 * doubles / a.steps)
        out["allreduce_calls_per_step"] = out["config"]["allreduce_calls_per_step"]
        out["allreduce_bytes_per_step"] = out["config"]["allreduce_bytes_per_step"]
        # what DESIGN.md 7 budgets for this run (NO multi-GPU curve had been measured when it was written: times from 1-GPU phase stamps,
        # xGMI hop assumed 3-5 us) -- printed beside the measurement so that a first curve explains itself
        if a.config == 3 and not blocks:
            budget = {1: (2.95, 2.95), 2: (2.4, 2.6), 4: (2.0, 2.15), 8: (1.7, 1.9)} if peer_live else \
                     {1: (2.95, 2.95), 2: (2.2, 2.4), 4: (2.0, 2.2), 8: (1.9, 2.1)}
            lo_ms, hi_ms = budget.get(world, (None, None))
            out["design_budget"] = {
                "source": "DESIGN.md section 7, config 3 strong scaling (%s)" % ("in-launch exchange" if peer_live else "per-factor passes, one all-reduce per factor"),
                "expected_ms_per_step": [lo_ms, hi_ms], "expected_it_per_s": [round(1e3 / hi_ms, 1), round(1e3 / lo_ms, 1)] if lo_ms else None,
                "measured_ms_per_step": round(elapsed / a.steps * 1e3, 3),
                "note": "strong scaling of ONE latency-bound chain is poor by design (1.3-1.8x at 8 GPUs): N GPUs are better spent on N "
                        "chains; sharding is what makes tables beyond one GPU possible -- see weak_scaling_it_per_s"}
        out["config"]["per_rank"] = per_rank  # [{rank, rows, allreduce_calls_per_step, allreduce_bytes_per_step}]: every rank's own count
        out["config"]["rows_this_rank"] = hi - lo
        # evidence that the collective spans the ranks: ncclCommCount of the communicator libmyfm_hip.so opened itself, and the
        # librccl it bound (0 / "": the torch.distributed callback carried the all-reduces instead, see `parallelism`)
        out["config"]["peer_exchange"] = bool(peer_live)
        out["config"]["rccl_ranks"] = int(rccl_ranks)
        out["config"]["rccl_path"] = rccl_path
        out["config"]["torch_world_size"] = world
    os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
