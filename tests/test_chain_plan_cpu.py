"""The plan of the streamed block-feature sweep (csrc/mfm_chain_plan.hpp -> k_cs_stream) without a GPU: `mfm_cs_plan_selftest` builds
the plan of a random multi-hot block and emulates the launch's data flow on the host -- the walker, its X / Y helpers, the ranges' S / U
wavefronts each on their own copy of what they can see (records in global memory, LDS slots, ring positions), run in the EARLIEST order
their flags allow -- against the plain sequential sweep of FMTrainer.hpp:419-470's dependence structure. A plan that lets an actor read
a record before its last update is in, or reuse an LDS slot / a ring position too early, gives different numbers."""
import ctypes

import numpy as np
import pytest
import scipy.sparse as sps


def _selftest(B, cg, lw, nb, rd, cap):
    from myfm_amd import _capi

    L = _capi.lib()
    csc = sps.csc_matrix(B)
    csc.sort_indices()
    ptr = np.ascontiguousarray(csc.indptr, dtype=np.int64)
    idx = np.ascontiguousarray(csc.indices, dtype=np.int32)
    val = np.ascontiguousarray(csc.data, dtype=np.float64)
    diff = ctypes.c_double(-1.0)
    info = (ctypes.c_int64 * 8)()
    f = L.mfm_cs_plan_selftest
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                  ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]
    rc = f(B.shape[0], B.shape[1], ptr.ctypes.data, idx.ctypes.data, val.ctypes.data, cg, lw, nb, rd, cap, ctypes.byref(diff), info)
    return rc, diff.value, list(info)


def _block(n_rows, n_cols, per_row, seed):
    rng = np.random.default_rng(seed)
    cols = rng.integers(0, n_cols, size=(n_rows, per_row))
    rows = np.repeat(np.arange(n_rows), per_row)
    B = sps.csr_matrix((rng.normal(size=rows.size) * 0.3, (rows, cols.ravel())), shape=(n_rows, n_cols))
    B.sum_duplicates()
    return B


@pytest.mark.parametrize("cg,lw,nb,rd", [(4, 3, 16, 2), (4, 4, 16, 2), (2, 5, 8, 1), (1, 6, 3, 2), (4, 2, 32, 3), (3, 1, 5, 2), (4, 7, 16, 3)])
def test_plan_data_flow_matches_the_sequential_sweep(cg, lw, nb, rd):
    B = _block(20000, 300, 6, seed=cg * 100 + lw)
    rc, diff, info = _selftest(B, cg, lw, nb, rd, 1 << 20)
    assert rc == 0 and info[0] == 1, (rc, info)
    assert diff < 1e-9, diff
    assert info[4] + info[5] == B.nnz and info[7] == -(-300 // cg)
    assert info[4] > 0 and info[5] > 0  # both kinds of entries occur


def test_small_block_is_all_hot_and_dense_columns_work():
    # a block so small that every row is touched inside every window: no cold entry at all; and two dense columns
    rng = np.random.default_rng(5)
    B = sps.hstack([_block(60, 40, 5, seed=1), sps.csr_matrix(rng.normal(size=(60, 2)))]).tocsr()
    rc, diff, info = _selftest(B, 4, 3, 16, 2, 1 << 20)
    assert rc == 0 and info[0] == 1 and diff < 1e-9, (rc, diff, info)
    assert info[1] <= 60  # never more slots than rows


def test_window_that_does_not_fit_is_refused():
    B = _block(20000, 300, 6, seed=9)
    rc, diff, info = _selftest(B, 4, 4, 16, 2, 50)
    assert rc == 0 and info[0] == 0 and info[1] > 50


def test_config5_user_block_shape_fits_the_lds():
    # BASELINE configs[4]'s user-side block at 1/10 of its rows and columns' density kept: 50 000 rows x 2000 columns, 10 entries per row
    # (p = 0.005 per (row, column) as at full size): the window the launch takes by default must fit 1900 slots
    B = _block(50000, 2000, 10, seed=2)
    rc, diff, info = _selftest(B, 4, 3, 16, 2, 1900)
    assert rc == 0 and info[0] == 1 and diff < 1e-9, (rc, diff, info)
