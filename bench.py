#!/usr/bin/env python
"""bench.py -- Gibbs iterations/sec of the MI355X sampler on BASELINE config 3:
MovieLens-10M-shaped synthetic CSR (N = 10 M rows, 69 878 + 10 677 one-hot features, nnz = 20 M),
MyFMRegressor rank 32, fp64, full update_all per step (BaseFMTrainer.hpp:135-152).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Prints ONE JSON line (rank 0). Besides the contract's fields it carries
  "roofline":     HBM roofline of the dominant kernel class (algorithmic bytes / HIP-event time)
  "cpu_baseline": the CPU oracle (Eigen-free restatement, 1 thread) timed on this box's host cores
                  on a bounded sample of the same workload (N = 1: full iterations of the same design)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def b_iter_bytes(N, nnz, D, K):
    """algorithmic bytes per Gibbs iteration, main table, regression."""
    return nnz * (40 + 56 * K) + N * (40 + 8 * K) + D * (16 * K + 8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--users", type=int, default=69878)
    ap.add_argument("--items", type=int, default=10677)
    ap.add_argument("--rank", type=int, default=32)
    ap.add_argument("--cpu-iters", type=int, default=2, help="CPU-oracle iterations timed (0 disables)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if a.gpus != 1 or world != 1:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (a.gpus, world))

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the sampler has no CPU fallback")
    if os.environ.get("MYFM_BENCH_DEVICE"):  # debugging: several ranks on one GPU (with MYFM_BENCH_BACKEND=gloo)
        local_rank = int(os.environ["MYFM_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    os.environ["HIP_VISIBLE_DEVICES"] = os.environ.get("HIP_VISIBLE_DEVICES", "")
    dist = None
    force_sharded = bool(os.environ.get("MYFM_BENCH_FORCE_SHARDED"))  # exercise the N > 1 code path at world = 1
    if world > 1 or force_sharded:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        backend = os.environ.get("MYFM_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from myfm_amd import _myfm
    from tests import datasets as ds

    # ---- workload: every rank holds one ML-10M-shaped shard of `rows` rows (weak scaling) --------
    t0 = time.time()
    row_lo, rows_total = 0, a.rows
    if world == 1:
        X, y, shapes = ds.movielens_like(a.rows, a.users, a.items, rank_true=32, seed=1)
    else:
        # rank r holds rows [r * rows, (r + 1) * rows) of ONE user-sorted table of world * rows rows (same users / items)
        X, y, shapes, row_lo, rows_total = ds.movielens_like_shard(a.rows, rank, world, a.users, a.items, rank_true=32, seed=1)
    gi = ds.group_index_from_shapes(shapes)
    N, D, nnz, K = X.shape[0], X.shape[1], X.nnz, a.rank
    t_data = time.time() - t0

    b = _myfm.ConfigBuilder()
    b.set_alpha_0(1.0).set_beta_0(1.0).set_gamma_0(1.0).set_mu_0(0.0).set_reg_0(1.0)
    b.set_group_index([int(g) for g in gi]).set_n_iter(a.steps + a.warmup + 4).set_n_kept_samples(0)
    b.set_task_type(_myfm.TaskType.REGRESSION)
    t0 = time.time()
    os.environ["MYFM_AMD_DEVICE"] = str(local_rank)  # one process per GPU
    parallelism = "1 GPU"
    if world == 1 and not force_sharded:
        sess = _myfm.GibbsSession(K, 0.1, X, [], y, 42, b.build())
    else:
        # Row-sharded: rank r holds rows [r*rows, (r+1)*rows) of ONE chain over world*rows rows;
        # model state / variates replicated (same seed), one RCCL all-reduce per level of every sweep.
        from myfm_amd.distributed import TorchAllReduce

        ar = TorchAllReduce()
        levels = np.concatenate([np.zeros(a.users, np.int32), np.ones(a.items, np.int32)])  # two one-hot fields
        sess = _myfm.GibbsSession(K, 0.1, X, [], y, 42, b.build(), allreduce=ar, n_total_rows=rows_total, row_offset=row_lo,
                                  stream=ar.stream_ptr, main_levels=levels)
        parallelism = ("one chain over %d rows, user-sorted, rows sharded over %d GPUs at user boundaries (weak: ~%d rows/GPU); "
                       "per factor one RCCL all-reduce of the item level's statistics (+ one for users split between "
                       "two ranks, if any), one model all-reduce per sweep" % (rows_total, world, a.rows))
    t_setup = time.time() - t0

    def sync():
        sess.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # Kernel timing: bracketing EVERY launch with HIP events costs ~10 % of an iteration at this shape, so the
    # per-class breakdown comes from 3 extra diagnostic steps before the warm-up (not part of the timed region),
    # and inside the timed region only the dominant class is bracketed (31 launches per step: free).
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
    if dom_name:
        sess.timing_select(dom_name)
        sess.timing_enable(True)
        sess.timing_reset()
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
    timing = dict(sess.timing()) if dom_name else {}
    if dom_name:
        sess.timing_enable(False)
        sess.timing_select("")

    # sanity: the chain is alive (finite state, plausible noise precision) and, when sharded, the replicated
    # model is the same on every rank (w0, alpha and a checksum of V)
    alpha = sess.hyper.alpha
    assert np.isfinite(alpha) and alpha > 0, alpha
    if dist is not None and world > 1:
        fm = sess.fm
        mine = torch.tensor([float(fm.w0), float(alpha), float(np.abs(np.asarray(fm.V)).sum()), float(np.asarray(fm.w).sum())],
                            dtype=torch.float64, device="cuda")
        allv = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        for v in allv[1:]:
            assert torch.allclose(v, allv[0], rtol=1e-12, atol=0), ("replicas diverged", [x.tolist() for x in allv])

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # Weak scaling: every rank sweeps its own `rows`-row shard per step (N > 1: the shards form ONE chain
    # over world*rows rows, synchronised by the per-level all-reduces). Whole-job value = shard-iterations/s
    # summed over the ranks; the chain itself advances at value / world iterations/s.
    it_per_s = world * a.steps / elapsed
    B_iter = b_iter_bytes(N, nnz, D, K)

    roofline = None
    if timing:
        dom = max(timing.items(), key=lambda kv: kv[1][0])
        name, (ms, launches, alg_bytes) = dom
        achieved = alg_bytes / (ms * 1e-3) / 1e9
        unfused_bytes = 56.0 * nnz + 8.0 * N + 8.0 * D  # SURVEY 8d, one factor of update_V
        # HBM bytes per launch of that class from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE of this same command, corrected as MI355X_MICROARCH.md prescribes; profiles/*_pmc_traffic.*)
        traffic = None
        default_workload = (a.rows, a.users, a.items, a.rank) == (10_000_000, 69878, 10677, 32) and world == 1
        pmc = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json")) \
            if os.path.isdir(os.path.join(ROOT, "profiles")) else []
        if default_workload and pmc:
            tr = json.load(open(os.path.join(ROOT, "profiles", pmc[-1])))
            if name in tr:
                traffic = round(tr[name]["bytes_per_level_launch"])
        roofline = {
            "bound": "hbm",
            "kernel": name,
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "traffic_source": pmc[-1] if traffic else None,
            "traffic_gbs": round(traffic / (ms / launches * 1e-3) / 1e9, 1) if traffic else None,
            "traffic_frac": round(traffic / (ms / launches * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
            "unfused_equivalent_gbs": round(unfused_bytes / (ms / launches * 1e-3) / 1e9, 1) if name == "sweep_V_fused_next" else None,
            "note": "achieved = algorithmic bytes of the FUSED pass (e, q read + written once, 4-byte entries, one 16-byte "
                    "slot per run) / event time; traffic = PMC-measured HBM bytes of the same launch (above the algorithmic "
                    "figure: the scattered 16-byte slot stores cost whole lines); unfused_equivalent_gbs = SURVEY 8d bytes of "
                    "the unfused algorithm for the same work ((56 nnz + 8 N + 8 D) per factor) / time: what the fusion saves",
            # SURVEY 8d's figure for the "q-cache / e-update sweep": (56 nnz + 8 N + 8 D) K / t_updateV, t_updateV = all
            # update_V kernel classes of one step (diagnostic steps); can exceed what any unfused implementation could
            # reach because the fused pass moves fewer bytes than that formula assumes
            "updateV_survey_gbs": round(unfused_bytes * K / (sum(v[0] for k2, v in breakdown.items() if k2.startswith("sweep_V")) * 1e-3) / 1e9, 1)
            if breakdown else None,
            "avg_launch_us": round(ms / launches * 1e3, 2),
            "launches": int(launches),
            "alg_bytes_per_launch": round(alg_bytes / launches),
            "kernel_ms_per_step": round(sum(v[0] for v in breakdown.values()), 3),
            "iteration_alg_gbs": round(B_iter * (a.steps / elapsed) / 1e9, 1),
            "iteration_frac": round(B_iter * (a.steps / elapsed) / 1e9 / HBM_PEAK_GBS, 4),
            "by_kernel_ms_per_step": {k: round(v[0], 3) for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1][0])},
            "by_kernel_note": "per-class times from 3 diagnostic steps with every launch bracketed (outside the timed region); "
                              "the dominant class's figures above are from events inside the timed region",
        }

    cpu = None
    if a.cpu_iters > 0 and world == 1:  # (rank 0, N = 1 only)
        from oracle import oracle as O

        O.build()
        ot = O.OracleTrainer(X, y, rank=K, group_index=gi, seed=42)
        c0 = time.perf_counter()
        for _ in range(a.cpu_iters):
            ot.step()
        c_el = time.perf_counter() - c0
        cpu = {
            "value": round(a.cpu_iters / c_el, 5),
            "unit": "Gibbs iterations/sec",
            "cores": 1,
            "kind": "port",
            "sample": "%d full update_all iterations of the same design (N=%d, nnz=%d, rank %d) by oracle/libmyfm_oracle.so, "
                      "1 thread; the reference core itself needs Eigen and cannot be built here" % (a.cpu_iters, N, nnz, K),
            "seconds": round(c_el, 2),
            "host_cpus": os.cpu_count(),
        }

    out = {
        "metric": "Gibbs iterations/sec (rank=32)",
        "value": round(it_per_s, 3),
        "unit": "Gibbs iterations/sec",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[2]: MovieLens-10M-shaped synthetic CSR, MyFMRegressor rank=%d fp64, full update_all" % K,
            "rows": N, "nnz": nnz, "features": D, "users": a.users, "items": a.items, "rank": K, "groups": 2,
            "parallelism": parallelism,
            "row_iterations_per_s": round((rows_total if world > 1 else N) * a.steps / elapsed),
            "chain_iterations_per_s": round(a.steps / elapsed, 3),
            "alg_bytes_per_iteration": B_iter,
            "setup_s": round(t_setup, 2), "datagen_s": round(t_data, 2),
        },
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    if cpu:
        out["speedup_vs_cpu_baseline"] = round((a.steps / elapsed) / cpu["value"], 1)
    if world > 1 or force_sharded:
        # sanity: the replicated model state is identical on every rank after the timed steps
        out["config"]["allreduce_calls_per_step"] = round(ar.calls / (a.steps + a.warmup), 1)
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
