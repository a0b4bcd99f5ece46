"""Worker of tests/test_gpu_sharded.py::test_two_processes_one_gpu_persistent_sweep (two ranks under torch.distributed.run, both
on device 0, process group over gloo): the row-sharded persistent sweep with the ranks in DIFFERENT processes -- the exchange
buffers travel as IPC handles (mfm_peer_export / mfm_peer_import through myfm_amd.distributed.connect_peers), the two persistent
launches run side by side and read each other's item sums inside the launch. Every rank must reproduce the oracle's unsharded
chain; the replicas must agree bit for bit."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ["MYFM_AMD_DEVICE"] = "0"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from myfm_amd import _capi, _myfm
    from myfm_amd import distributed as D
    from oracle import oracle as O
    from tests import datasets as ds

    n, K = 120001, 3
    X, y, shapes = ds.onehot_mf(n, 300, 170, seed=13, sort_by_user=True)
    gi = ds.group_index_from_shapes(shapes)
    cuts = D.shard_cuts(X.indices[X.indptr[:-1]], world)
    lo, hi = cuts[rank], cuts[rank + 1]
    levels = _capi.column_levels(X)[0]
    b = _myfm.ConfigBuilder()
    b.set_alpha_0(1.0).set_beta_0(1.0).set_gamma_0(1.0).set_mu_0(0.0).set_reg_0(1.0)
    b.set_group_index([int(g) for g in gi]).set_n_iter(10).set_n_kept_samples(0).set_task_type(_myfm.TaskType.REGRESSION)
    ar = D.TorchAllReduce()
    s = _myfm.GibbsSession(K, 0.1, X[lo:hi], [], y[lo:hi], 42, b.build(), allreduce=ar, n_total_rows=n, row_offset=lo,
                           stream=ar.stream_ptr, main_levels=levels, shard_rank=rank, shard_world=world)
    live = D.connect_peers(s)
    assert live and (s.plan_flags() & 256), (live, s.plan_flags())
    for it in range(3):
        s.step()
    s.synchronize()
    t = O.OracleTrainer(X, y, rank=K, group_index=gi)
    for it in range(3):
        t.step()
    w0, w, V = t.fm()
    np.testing.assert_allclose(np.asarray(s.fm.V), V, rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(np.asarray(s.fm.w), w, rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(s.residual(), t.e(n)[lo:hi], rtol=1e-7, atol=1e-7)
    mine = torch.tensor([float(np.abs(np.asarray(s.fm.V)).sum())], dtype=torch.float64)
    allv = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    assert all(float(v) == float(allv[0]) for v in allv)
    # giving it up walks the same chain on the per-factor passes
    s.peer_drop()
    assert not (s.plan_flags() & 256)
    s.step()
    t.step()
    np.testing.assert_allclose(np.asarray(s.fm.V), t.fm()[2], rtol=1e-7, atol=1e-8)
    del s
    # MyFMRegressor.fit() row-sharded with the same hand-over (distributed.enable(peer_exchange=True))
    import myfm_amd

    D.enable(native=False, peer_exchange=True)
    fm = myfm_amd.MyFMRegressor(K).fit(X, y, group_shapes=shapes, n_iter=4, n_kept_samples=2)
    samples, _, _ = O.fit(X, y, rank=K, group_index=gi, n_iter=4, n_kept_samples=2)
    for got, (w0_, w_, V_) in zip(fm.predictor_.samples, samples):
        np.testing.assert_allclose(got.V, V_, rtol=1e-7, atol=1e-7)
        np.testing.assert_allclose(got.w, w_, rtol=1e-7, atol=1e-7)
    assert fm.history_ is not None and D._STATE.get("last_connect") is True  # (the fit did take the in-launch exchange)
    D.disable()
    dist.barrier()
    if rank == 0:
        print("mp_peer_worker ok: world", world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
