"""CPU-only tests of the drop-in boundary: everything that must work without a GPU -- the C ABI
exports, host-side validation / pickling of the _myfm types, the level schedule, and the rule that
the product never touches the oracle."""
import os
import pickle
import re
import subprocess

import numpy as np
import pytest
import scipy.sparse as sps

from . import datasets as ds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g

    g.build()
    import myfm_amd

    return myfm_amd


def test_capi_exports_every_declared_symbol(built):
    from myfm_amd import _capi

    header = open(os.path.join(ROOT, "include", "myfm_hip.h")).read()
    declared = set(re.findall(r"\b(mfm_[A-Za-z0-9_]+)\s*\(", header))
    declared -= {"mfm_ctx", "mfm_design"}
    assert declared == set(_capi.SYMBOLS), declared ^ set(_capi.SYMBOLS)
    L = _capi.lib()
    for s in declared:
        assert hasattr(L, s), s
    out = subprocess.check_output(["nm", "-D", "--defined-only", _capi.LIB_PATH]).decode()
    exported = set(re.findall(r" T (mfm_[A-Za-z0-9_]+)", out))
    assert declared <= exported


def test_no_gpu_fails_loudly(built):
    from myfm_amd import _capi, _myfm

    if _myfm.device_count() > 0:
        pytest.skip("a GPU is visible")
    X, y = ds.toy()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        built.MyFMRegressor(2).fit(X, y, n_iter=3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _capi.Context(X, y, rank=2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _capi.Design(X)


def test_product_never_touches_the_oracle(built):
    pat = re.compile(r"oracle", re.I)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "myfm_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                for line in txt.splitlines():
                    if pat.search(line):
                        assert "import" not in line and "#include" not in line and "CDLL" not in line, (f, line)
    out = subprocess.check_output(["ldd", os.path.join(ROOT, "myfm_amd", "libmyfm_hip.so")]).decode()
    assert "oracle" not in out


def test_config_builder_validation(built):
    from myfm_amd import _myfm

    b = _myfm.ConfigBuilder()
    with pytest.raises(ValueError, match="No matching index for group index 1"):
        b.set_group_index([0, 0, 2]).build()
    b = _myfm.ConfigBuilder().set_identical_groups(3)
    with pytest.raises(ValueError, match="n_iter must be positive"):
        b.set_n_iter(0).build()
    with pytest.raises(ValueError, match="n_kept_samples must not exceed n_iter"):
        b.set_n_iter(5).set_n_kept_samples(6).build()
    with pytest.raises(ValueError, match="non-negative"):
        b.set_n_kept_samples(-1).build()
    cfg = b.set_n_iter(10).set_n_kept_samples(3).set_task_type(_myfm.TaskType.CLASSIFICATION).build()
    assert isinstance(cfg, _myfm.FMLearningConfig)
    assert int(_myfm.TaskType.ORDERED) == 2


def test_relation_block(built):
    rb = built.RelationBlock([0, 1, 1, 0], sps.csr_matrix(np.array([[1.0, 0, 2.0], [0, 3.0, 0]])))
    assert (rb.mapper_size, rb.block_size, rb.feature_size) == (4, 2, 3)
    assert rb.original_to_block == [0, 1, 1, 0]
    assert (rb.data.toarray() == np.array([[1.0, 0, 2.0], [0, 3.0, 0]])).all()
    assert repr(rb) == "<RelationBlock with mapper size = 4, block data size = 2, feature size = 3>"
    rb2 = pickle.loads(pickle.dumps(rb))
    assert rb2.original_to_block == rb.original_to_block and (rb2.data != rb.data).nnz == 0
    with pytest.raises(RuntimeError, match="index mapping points to non-existing row"):
        built.RelationBlock([0, 2], sps.csr_matrix(np.eye(2)))
    # non-CSR inputs are coerced (declare_module.hpp:95-137 via the scipy caster)
    assert built.RelationBlock([0], np.eye(2)).feature_size == 2


def test_pickle_of_boundary_types(built):
    from myfm_amd import _myfm

    rng = np.random.default_rng(0)
    w, V = rng.normal(size=5), rng.normal(size=(5, 3))
    # FM state tuples: 4-tuple and the 3-tuple of earlier versions (declare_module.hpp:172-192)
    fm = _myfm.FM.__new__(_myfm.FM)
    fm.__setstate__((0.5, w, V, [np.array([0.0, 1.0])]))
    assert fm.w0 == 0.5 and np.allclose(fm.w, w) and np.allclose(fm.V, V) and np.allclose(fm.cutpoints[0], [0, 1])
    assert repr(fm) == "<Factorization Machine sample with feature size = 5, rank = 3>"
    fm3 = _myfm.FM.__new__(_myfm.FM)
    fm3.__setstate__((0.5, w, V))
    assert fm3.cutpoints == []
    back = pickle.loads(pickle.dumps(fm))
    assert np.allclose(back.V, V) and back.w0 == 0.5
    hy = _myfm.FMHyperParameters.__new__(_myfm.FMHyperParameters)
    hy.__setstate__((2.0, np.ones(2), np.ones(2) * 3, np.arange(6.0).reshape(2, 3), np.ones((2, 3))))
    hy2 = pickle.loads(pickle.dumps(hy))
    assert hy2.alpha == 2.0 and np.allclose(hy2.mu_V, np.arange(6.0).reshape(2, 3)) and hy2.lambda_w[1] == 3
    pr = _myfm.Predictor.__new__(_myfm.Predictor)
    pr.__setstate__((3, 5, int(_myfm.TaskType.REGRESSION), [fm, back]))
    pr2 = pickle.loads(pickle.dumps(pr))
    assert len(pr2.samples) == 2 and np.allclose(pr2.samples[1].V, V)
    hist = _myfm.LearningHistory.__new__(_myfm.LearningHistory)
    hist.__setstate__(([hy], [], [4]))
    h2 = pickle.loads(pickle.dumps(hist))
    assert h2.n_mh_accept == [4] and h2.hypers[0].alpha == 2.0


def test_predictor_input_checks_need_no_gpu(built):
    from myfm_amd import _myfm

    pr = _myfm.Predictor.__new__(_myfm.Predictor)
    pr.__setstate__((3, 5, int(_myfm.TaskType.REGRESSION), []))
    with pytest.raises(ValueError, match="Told to predict for 4 but this->feature_size is 5"):
        pr.predict(sps.csr_matrix(np.eye(4)), [])
    with pytest.raises(RuntimeError, match="Empty samples!"):
        pr.predict(sps.csr_matrix(np.ones((2, 5))), [])
    with pytest.raises(RuntimeError, match="main table has size 2 but the relation\\[0\\] has size 3"):
        pr.predict(sps.csr_matrix(np.ones((2, 3))), [built.RelationBlock([0, 1, 0], np.eye(2))])


def test_estimator_argument_checks(built):
    X, y = ds.toy()
    with pytest.raises(ValueError, match="At least X or X_rel"):
        built.MyFMRegressor(2).fit(None, y)
    with pytest.raises(RuntimeError, match="Must specify both"):
        built.MyFMRegressor(2).fit(X, y, X_test=X)
    with pytest.raises(RuntimeError, match="Predictor called before fit"):
        built.MyFMRegressor(2).predict(X)
    with pytest.raises(RuntimeError, match="Sampler not run yet"):
        built.MyFMRegressor(2).get_hyper_trace()
    rb1 = built.RelationBlock([0, 1], np.eye(2))
    rb2 = built.RelationBlock([0, 1, 1], np.eye(2))
    with pytest.raises(ValueError, match="Inconsistent case size"):
        built.MyFMRegressor(2).fit(None, y, [rb1, rb2])


def _brute_levels(X):
    X = sps.csc_matrix(X)
    D = X.shape[1]
    supp = [set(X.indices[X.indptr[j]:X.indptr[j + 1]]) for j in range(D)]
    level = np.zeros(D, dtype=np.int32)
    for j in range(D):
        lv = -1
        for k in range(j):
            if supp[j] & supp[k]:
                lv = max(lv, level[k])
        level[j] = lv + 1
    return level


@pytest.mark.parametrize("seed", range(4))
def test_level_schedule_matches_definition(built, seed):
    # level(j) = 1 + max level of earlier conflicting columns
    from myfm_amd import _capi

    rng = np.random.RandomState(seed)
    X = sps.random(60, 25, density=0.08, random_state=rng, format="csr")
    level, n = _capi.column_levels(X)
    np.testing.assert_array_equal(level, _brute_levels(X))
    assert n == level.max() + 1
    X, _, _ = ds.onehot_mf(500, 40, 12, seed=seed)
    level, n = _capi.column_levels(X)
    assert n == 2 and (level[:40] == 0).all() and (level[40:] == 1).all()
    Xt, _ = ds.toy()
    assert list(_capi.column_levels(Xt)[0]) == [0, 1, 1, 1, 1, 2, 2, 2, 2]


def test_mt19937_jump_ahead_polynomials():
    """csrc/mfm_mtjump.hpp (Berlekamp-Massey characteristic polynomial, x^J mod phi) against std::mt19937 itself:
    the parallel device generator starts its workgroups from these polynomials"""
    from myfm_amd import _myfm

    assert _myfm.mt_jump_selftest(64, 3, 12345) == 0
    assert _myfm.mt_jump_selftest(512, 2, 7) == 0


def test_native_row_sort_helpers(built):
    # fit()'s row preparation (estimators._device_row_order): stable counting sort by the first stored column and the
    # threaded CSR row gather must agree with numpy's stable argsort / scipy's fancy indexing, for int32 and int64 index
    # arrays, ragged rows and explicit values
    from myfm_amd import _myfm

    rng = np.random.default_rng(3)
    n, d = 5000, 37
    lens = rng.integers(1, 5, size=n)
    indptr = np.concatenate([[0], np.cumsum(lens)])
    indices = np.concatenate([np.sort(rng.choice(d, size=k, replace=False)) for k in lens]).astype(np.int32)
    data = rng.normal(size=indices.size)
    X = sps.csr_matrix((data, indices, indptr), shape=(n, d))
    first = X.indices[X.indptr[:-1]]
    for ptr_t in (np.int32, np.int64):
        order = _myfm.row_order_by_first_column(X.indptr.astype(ptr_t), X.indices, d)
        assert order.dtype == np.int64 and np.array_equal(order, np.argsort(first, kind="stable"))
        ptr, idx, val = _myfm.permute_csr_rows(X.indptr.astype(ptr_t), X.indices, X.data, order)
        Y = sps.csr_matrix((val, idx, ptr), shape=X.shape)
        Z = X[order]
        assert np.array_equal(Y.indptr, Z.indptr) and np.array_equal(Y.indices, Z.indices) and np.array_equal(Y.data, Z.data)
    with pytest.raises(ValueError, match="no stored entry"):
        _myfm.row_order_by_first_column(np.array([0, 1, 1]), np.array([0], dtype=np.int32), 3)
    with pytest.raises(ValueError, match="out of range"):
        _myfm.permute_csr_rows(X.indptr, X.indices, X.data, np.array([n], dtype=np.int64))
    from myfm_amd.estimators import _device_row_order

    assert _device_row_order(X[np.argsort(first, kind="stable")]) is None  # already sorted: nothing to do
    assert np.array_equal(_device_row_order(X), np.argsort(first, kind="stable"))


def test_cutpoint_groups_accept_index_arrays(built):
    # set_cutpoint_groups takes the reference's list-of-lists (declare_module.hpp:139-156) and numpy index arrays
    from myfm_amd import _myfm

    b = _myfm.ConfigBuilder().set_identical_groups(2).set_n_iter(3).set_n_kept_samples(1).set_task_type(_myfm.TaskType.ORDERED)
    assert isinstance(b.set_cutpoint_groups([(3, [0, 1, 2]), (4, [3, 4])]).build(), _myfm.FMLearningConfig)
    assert isinstance(b.set_cutpoint_groups([(3, np.arange(3)), (4, np.array([3, 4], dtype=np.int32))]).build(), _myfm.FMLearningConfig)
    with pytest.raises(ValueError, match="negative row index"):
        b.set_cutpoint_groups([(3, np.array([0, -1]))])
    with pytest.raises(ValueError, match="expected"):
        b.set_cutpoint_groups([(3,)])
