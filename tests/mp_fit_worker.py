"""Worker of tests/test_gpu_sharded.py::test_multi_gpu_sharded_fit (one process per GPU, launched by
torch.distributed.run): MyFMRegressor.fit() row-sharded over the ranks through the library's RCCL all-reduce must
reproduce the oracle's unsharded chain on every rank."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist

    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    # MP_FIT_ONE_GPU=1: every rank on device 0, process group over gloo, the library's all-reduces through the
    # torch.distributed callback (RCCL refuses two ranks on one device): the multi-PROCESS flow on a 1-GPU box
    one_gpu = bool(os.environ.get("MP_FIT_ONE_GPU"))
    if one_gpu:
        local = 0
        os.environ["MYFM_AMD_DEVICE"] = "0"
    torch.cuda.set_device(local)
    if one_gpu:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    import myfm_amd
    from myfm_amd import distributed as D
    from oracle import oracle as O
    from tests import datasets as ds

    D.enable(native=not one_gpu)
    X, y, shapes = ds.onehot_mf(40000, 300, 80, seed=5)
    gi = ds.group_index_from_shapes(shapes)
    fm = myfm_amd.MyFMRegressor(6).fit(X, y, group_shapes=shapes, n_iter=6, n_kept_samples=6)
    samples, hypers, _ = O.fit(X, y, rank=6, group_index=gi, n_iter=6, n_kept_samples=6)
    for s, (w0, w, V) in zip(fm.predictor_.samples, samples):
        assert abs(s.w0 - w0) < 1e-7
        np.testing.assert_allclose(s.w, w, rtol=1e-7, atol=1e-7)
        np.testing.assert_allclose(s.V, V, rtol=1e-7, atol=1e-7)
    # replicas agree bit for bit
    mine = torch.tensor([float(np.abs(fm.predictor_.samples[-1].V).sum())], dtype=torch.float64, device="cuda")
    allv = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    assert all(float(v) == float(allv[0]) for v in allv)
    # relation blocks + ordered probit run through the same entry point
    main_X, X_flat, blocks, yb, bshapes = ds.multihot_block_design(n_train=900)
    rbs = [myfm_amd.RelationBlock([int(v) for v in m], b) for m, b in blocks]
    fb = myfm_amd.MyFMRegressor(3).fit(main_X, yb, rbs, group_shapes=bshapes, n_iter=4, n_kept_samples=4)
    sb, _, _ = O.fit(main_X, yb, blocks, rank=3, group_index=ds.group_index_from_shapes(bshapes), n_iter=4, n_kept_samples=4)
    np.testing.assert_allclose(fb.predictor_.samples[-1].V, sb[-1][2], rtol=1e-7, atol=1e-7)
    dist.barrier()
    if rank == 0:
        print("mp_fit_worker ok: world", world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
