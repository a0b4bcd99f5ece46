"""ctypes binding of libmyfm_hip.so (include/myfm_hip.h) -- used by the parity tests and bench.py
to reach the device path through the C ABI itself. The estimator layer goes through the pybind11
module ``myfm_amd._myfm`` (host C++), which links the same library.

There is no fallback: a missing library is an ImportError, a missing GPU is a RuntimeError.
"""
import ctypes as C
import os

import numpy as np
import scipy.sparse as sps

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmyfm_hip.so")

MFM_OK, MFM_ERR_INVALID, MFM_ERR_RUNTIME, MFM_ERR_DEVICE = 0, 1, 2, 3
TASK_REGRESSION, TASK_CLASSIFICATION, TASK_ORDERED = 0, 1, 2

# every symbol declared in include/myfm_hip.h (tests check the library exports all of them)
SYMBOLS = [
    "mfm_version", "mfm_device_count", "mfm_global_error", "mfm_create", "mfm_destroy", "mfm_last_error",
    "mfm_set_stream", "mfm_synchronize", "mfm_set_main", "mfm_add_block", "mfm_set_groups", "mfm_finalize",
    "mfm_peer_info", "mfm_peer_set", "mfm_peer_model_info", "mfm_peer_set_model", "mfm_peer_export", "mfm_peer_import", "mfm_peer_drop", "mfm_set_residual_policy", "mfm_dim_all", "mfm_plan_info", "mfm_plan_flags", "mfm_set_state", "mfm_get_state", "mfm_set_w0", "mfm_zero_w", "mfm_get_e",
    "mfm_get_q", "mfm_set_e", "mfm_reduce_e", "mfm_shift_e", "mfm_group_stats_w", "mfm_group_stats_V",
    "mfm_sweep_w", "mfm_sweep_V", "mfm_sweep_wV", "mfm_update_e_regression", "mfm_update_e_classification", "mfm_score_train",
    "mfm_oprobit_add_group", "mfm_oprobit_eval", "mfm_oprobit_sample_z", "mfm_hyper_stats", "mfm_timing_enable", "mfm_timing_select", "mfm_timing_reset",
    "mfm_timing_n_classes", "mfm_timing_class_name", "mfm_timing_get", "mfm_design_create", "mfm_design_add_block",
    "mfm_design_destroy", "mfm_design_last_error", "mfm_design_dim_all", "mfm_design_predict",
    "mfm_host_column_levels", "mfm_rng_prepare", "mfm_rng_seed_mt19937", "mfm_rng_set_program", "mfm_rng_prefetch", "mfm_rng_acquire",
    "mfm_rng_get_z", "mfm_design_score_ctx", "mfm_design_n_rows", "mfm_set_allreduce", "mfm_set_row_offset", "mfm_set_main_levels",
    "mfm_test_erfcx", "mfm_test_truncated_normal", "mfm_get_device", "mfm_set_shard", "mfm_comm_unique_id", "mfm_comm_init",
    "mfm_comm_stats", "mfm_comm_info", "mfm_store_create", "mfm_store_destroy", "mfm_store_last_error", "mfm_store_size", "mfm_store_push_ctx",
    "mfm_store_push_host", "mfm_store_get", "mfm_design_predict_store", "mfm_store_reserve", "mfm_cs_plan_selftest",
    "mfm_regression_iteration_ready", "mfm_regression_iteration",
    "mfm_update_e_classification_exact", "mfm_oprobit_sample_z_exact", "mfm_latent_stats", "mfm_rng_host_read", "mfm_rng_host_advance", "mfm_set_latent_order",
]

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH
        )
    from ._build import preload_hip_runtime

    preload_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, i64, i32, dbl, P = C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_void_p
    u64 = C.c_uint64
    L.mfm_version.restype = C.c_char_p
    L.mfm_global_error.restype = C.c_char_p
    L.mfm_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.mfm_destroy.argtypes = [vp]
    L.mfm_destroy.restype = None
    L.mfm_last_error.restype = C.c_char_p
    L.mfm_last_error.argtypes = [vp]
    L.mfm_set_stream.argtypes = [vp, vp]
    L.mfm_synchronize.argtypes = [vp]
    L.mfm_set_main.argtypes = [vp, i64, i64, P, P, P, P]
    L.mfm_add_block.argtypes = [vp, i64, i64, P, P, P, P]
    L.mfm_set_groups.argtypes = [vp, P, i64, i32]
    L.mfm_finalize.argtypes = [vp, i32]
    L.mfm_peer_info.argtypes = [vp, P, P, P, P, P]
    L.mfm_peer_set.argtypes = [vp, i32, i32, P, P]
    L.mfm_peer_model_info.argtypes = [vp, P, P]
    L.mfm_peer_set_model.argtypes = [vp, i32, i32, P, P]
    L.mfm_peer_export.argtypes = [vp, P]
    L.mfm_peer_import.argtypes = [vp, i32, i32, P]
    L.mfm_peer_drop.argtypes = [vp]
    L.mfm_regression_iteration_ready.argtypes = [vp]
    L.mfm_regression_iteration.argtypes = [vp, P, P, P, P, P, P, P, P]
    L.mfm_set_residual_policy.argtypes = [vp, i32]
    L.mfm_dim_all.restype = i64
    L.mfm_dim_all.argtypes = [vp]
    L.mfm_plan_info.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    L.mfm_plan_flags.argtypes = [vp]
    L.mfm_set_state.argtypes = [vp, dbl, P, P]
    L.mfm_get_state.argtypes = [vp, C.POINTER(dbl), P, P]
    L.mfm_set_w0.argtypes = [vp, dbl]
    L.mfm_zero_w.argtypes = [vp]
    L.mfm_get_e.argtypes = [vp, P]
    L.mfm_get_q.argtypes = [vp, P]
    L.mfm_set_e.argtypes = [vp, P]
    L.mfm_reduce_e.argtypes = [vp, C.POINTER(dbl), C.POINTER(dbl)]
    L.mfm_shift_e.argtypes = [vp, dbl]
    L.mfm_group_stats_w.argtypes = [vp, P, P, P]
    L.mfm_group_stats_V.argtypes = [vp, P, P, P]
    L.mfm_hyper_stats.argtypes = [vp, C.c_int32, P, P, P, P, P, P, P, P]
    L.mfm_sweep_w.argtypes = [vp, dbl, P, P, P]
    L.mfm_sweep_V.argtypes = [vp, i32, i32, dbl, P, P, P]
    L.mfm_sweep_wV.argtypes = [vp, dbl, dbl, P, P, P, i32, i32, P, P, P]
    L.mfm_update_e_regression.argtypes = [vp]
    L.mfm_update_e_classification.argtypes = [vp, u64, u64]
    L.mfm_score_train.argtypes = [vp]
    L.mfm_oprobit_add_group.argtypes = [vp, i32, P, i64, C.POINTER(i32)]
    L.mfm_oprobit_eval.argtypes = [vp, i32, P, C.POINTER(dbl), P, P]
    L.mfm_oprobit_sample_z.argtypes = [vp, i32, P, u64, u64]
    L.mfm_timing_enable.argtypes = [vp, C.c_int]
    L.mfm_timing_select.argtypes = [vp, C.c_int32]
    L.mfm_timing_reset.argtypes = [vp]
    L.mfm_timing_class_name.restype = C.c_char_p
    L.mfm_timing_class_name.argtypes = [C.c_int]
    L.mfm_timing_get.argtypes = [vp, C.c_int, C.POINTER(dbl), C.POINTER(i64), C.POINTER(dbl)]
    L.mfm_design_create.argtypes = [C.c_int, i64, i64, P, P, P, C.POINTER(vp)]
    L.mfm_design_add_block.argtypes = [vp, i64, i64, P, P, P, P]
    L.mfm_design_destroy.argtypes = [vp]
    L.mfm_design_destroy.restype = None
    L.mfm_design_last_error.restype = C.c_char_p
    L.mfm_design_last_error.argtypes = [vp]
    L.mfm_design_dim_all.restype = i64
    L.mfm_design_dim_all.argtypes = [vp]
    L.mfm_design_predict.argtypes = [vp, i32, i32, P, P, P, i32, i32, P, P]
    L.mfm_host_column_levels.argtypes = [i64, i64, P, P, P, C.POINTER(i32)]
    L.mfm_rng_prepare.argtypes = [i64, i32, i32]
    L.mfm_rng_seed_mt19937.argtypes = [vp, P, i32]
    L.mfm_rng_set_program.argtypes = [vp, P, i32]
    L.mfm_rng_prefetch.argtypes = [vp]
    L.mfm_rng_acquire.argtypes = [vp, P, i64]
    L.mfm_rng_get_z.argtypes = [vp, P, P]
    L.mfm_update_e_classification_exact.argtypes = [vp, C.POINTER(i32)]
    L.mfm_oprobit_sample_z_exact.argtypes = [vp, i32, P, C.POINTER(i32)]
    L.mfm_latent_stats.argtypes = [vp, P]
    L.mfm_set_latent_order.argtypes = [vp, P, i64]
    L.mfm_rng_host_read.argtypes = [vp, u64, i64, P]
    L.mfm_rng_host_advance.argtypes = [vp, u64]
    L.mfm_design_score_ctx.argtypes = [vp, vp, P]
    L.mfm_design_n_rows.restype = i64
    L.mfm_design_n_rows.argtypes = [vp]
    L.mfm_set_allreduce.argtypes = [vp, vp, vp]
    L.mfm_set_row_offset.argtypes = [vp, i64]
    L.mfm_set_main_levels.argtypes = [vp, P, i64]
    L.mfm_get_device.argtypes = [vp]
    L.mfm_set_shard.argtypes = [vp, i32, i32]
    L.mfm_comm_unique_id.argtypes = [P]
    L.mfm_comm_init.argtypes = [vp, P, i32, i32]
    L.mfm_comm_stats.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    L.mfm_comm_info.argtypes = [vp, C.POINTER(i32), C.c_char_p, i64]
    L.mfm_store_create.argtypes = [C.c_int, i64, i32, C.POINTER(vp)]
    L.mfm_store_destroy.argtypes = [vp]
    L.mfm_store_destroy.restype = None
    L.mfm_store_last_error.restype = C.c_char_p
    L.mfm_store_last_error.argtypes = [vp]
    L.mfm_store_size.argtypes = [vp]
    L.mfm_store_reserve.argtypes = [vp, i32]
    L.mfm_store_push_ctx.argtypes = [vp, vp]
    L.mfm_store_push_host.argtypes = [vp, dbl, P, P]
    L.mfm_store_get.argtypes = [vp, i32, C.POINTER(dbl), P, P]
    L.mfm_design_predict_store.argtypes = [vp, vp, i32, i32, i32, i32, P, P]
    L.mfm_test_erfcx.argtypes = [C.c_int, P, i64, P]
    L.mfm_test_truncated_normal.argtypes = [C.c_int, i32, dbl, dbl, u64, u64, i64, P]
    _lib = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def csr_parts(X):
    X = sps.csr_matrix(X, dtype=np.float64)
    return (
        X,
        np.ascontiguousarray(X.indptr, dtype=np.int64),
        np.ascontiguousarray(X.indices, dtype=np.int32),
        np.ascontiguousarray(X.data, dtype=np.float64),
    )


def _raise(code, msg):
    msg = msg.decode() if isinstance(msg, bytes) else msg
    if code == MFM_ERR_INVALID:
        raise ValueError(msg)
    raise RuntimeError(msg)


def column_levels(X):
    """Host-only: level schedule of the columns of X (no GPU needed)."""
    X, ip, ix, _ = csr_parts(X)
    level = np.zeros(X.shape[1], dtype=np.int32)
    n = C.c_int32()
    rc = lib().mfm_host_column_levels(X.shape[0], X.shape[1], _p(ip), _p(ix), _p(level), C.byref(n))
    if rc:
        _raise(rc, lib().mfm_global_error())
    return level, n.value


def device_erfcx(x, device=0):
    """the device erfcx of mfm_oprobit_eval on x (test hook)"""
    x = _f64(x)
    out = np.empty_like(x)
    rc = lib().mfm_test_erfcx(device, _p(x), x.size, _p(out))
    if rc:
        _raise(rc, lib().mfm_global_error())
    return out


def device_truncated_normal(kind, lo, hi, n, seed=1, draw_index=0, device=0):
    """n draws of the device truncated-normal samplers (test hook): kind 'left' (z > lo), 'right' (z < hi), 'twoside'"""
    k = {"left": 0, "right": 1, "twoside": 2}[kind]
    out = np.empty(n)
    rc = lib().mfm_test_truncated_normal(device, k, float(lo), float(hi), seed, draw_index, n, _p(out))
    if rc:
        _raise(rc, lib().mfm_global_error())
    return out


class Context:
    """One training problem on one GPU: thin, explicit wrapper over the mfm_* entry points."""

    def __init__(self, X, y, blocks=(), rank=4, group_index=None, device=0):
        L = lib()
        h = C.c_void_p()
        rc = L.mfm_create(device, C.byref(h))
        if rc:
            _raise(rc, L.mfm_global_error())
        self.h = h
        X, ip, ix, dv = csr_parts(X)
        y = _f64(y)
        self.N, self.D0 = X.shape
        self._ck(L.mfm_set_main(h, X.shape[0], X.shape[1], _p(ip), _p(ix), _p(dv), _p(y)))
        D = X.shape[1]
        for mp, B in blocks:
            B, bp, bx, bv = csr_parts(B)
            mp = np.ascontiguousarray(mp, dtype=np.int64)
            self._ck(L.mfm_add_block(h, B.shape[0], B.shape[1], _p(bp), _p(bx), _p(bv), _p(mp)))
            D += B.shape[1]
        if group_index is None:
            group_index = np.zeros(D, dtype=np.int32)
        group_index = np.ascontiguousarray(group_index, dtype=np.int32)
        self.G = int(group_index.max()) + 1 if D else 1
        self._ck(L.mfm_set_groups(h, _p(group_index), group_index.shape[0], self.G))
        self._ck(L.mfm_finalize(h, rank))
        self.D, self.K = D, rank

    def _ck(self, rc):
        if rc:
            _raise(rc, lib().mfm_last_error(self.h))

    def close(self):
        if getattr(self, "h", None):
            lib().mfm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- state; V crosses this wrapper in the (D, K) shape of the boundary
    def set_state(self, w0, w, V):
        w = _f64(w)
        Vt = _f64(np.asarray(V, dtype=np.float64).T)
        self._ck(lib().mfm_set_state(self.h, float(w0), _p(w), _p(Vt)))

    def get_state(self):
        w0 = C.c_double()
        w = np.empty(self.D)
        V = np.empty((self.K, self.D))
        self._ck(lib().mfm_get_state(self.h, C.byref(w0), _p(w), _p(V)))
        return w0.value, w, np.ascontiguousarray(V.T)

    def set_w0(self, w0):
        self._ck(lib().mfm_set_w0(self.h, float(w0)))

    def zero_w(self):
        self._ck(lib().mfm_zero_w(self.h))

    def get_e(self):
        e = np.empty(self.N)
        self._ck(lib().mfm_get_e(self.h, _p(e)))
        return e

    def get_q(self):
        q = np.empty(self.N)
        self._ck(lib().mfm_get_q(self.h, _p(q)))
        return q

    def set_e(self, e):
        e = _f64(e)
        self._ck(lib().mfm_set_e(self.h, _p(e)))

    # --- iteration pieces
    def reduce_e(self):
        a, b = C.c_double(), C.c_double()
        self._ck(lib().mfm_reduce_e(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def shift_e(self, delta):
        self._ck(lib().mfm_shift_e(self.h, float(delta)))

    def group_stats_w(self, mu_w):
        mu_w = _f64(mu_w)
        s, ss = np.empty(self.G), np.empty(self.G)
        self._ck(lib().mfm_group_stats_w(self.h, _p(mu_w), _p(s), _p(ss)))
        return s, ss

    def group_stats_V(self, mu_V):
        """mu_V: (G, K) -> (sum (G, K), ssd (G, K))"""
        mu = _f64(np.asarray(mu_V).T)
        s, ss = np.empty((self.K, self.G)), np.empty((self.K, self.G))
        self._ck(lib().mfm_group_stats_V(self.h, _p(mu), _p(s), _p(ss)))
        return s.T.copy(), ss.T.copy()

    def sweep_w(self, alpha, lambda_w, mu_w, z):
        lambda_w, mu_w = _f64(lambda_w), _f64(mu_w)
        if z is not None:
            z = _f64(z)
            assert z.shape[0] == self.D
        self._ck(lib().mfm_sweep_w(self.h, float(alpha), _p(lambda_w), _p(mu_w), _p(z)))

    def sweep_V(self, f_begin, f_end, alpha, lambda_V, mu_V, z):
        """lambda_V, mu_V: (G, K); z: ((f_end - f_begin), D)"""
        lam, mu = _f64(np.asarray(lambda_V).T), _f64(np.asarray(mu_V).T)
        if z is not None:
            z = _f64(z)
            assert z.size == (f_end - f_begin) * self.D
        self._ck(lib().mfm_sweep_V(self.h, f_begin, f_end, float(alpha), _p(lam), _p(mu), _p(z)))

    def sweep_wV(self, alpha, e_shift, lambda_w, mu_w, zw, f_begin, f_end, lambda_V, mu_V, zv):
        """update_w0's residual shift, update_w and update_V of factors [f_begin, f_end) as one call"""
        lambda_w, mu_w = _f64(lambda_w), _f64(mu_w)
        lam, mu = _f64(np.asarray(lambda_V).T), _f64(np.asarray(mu_V).T)
        if zw is not None:
            zw, zv = _f64(zw), _f64(zv)
            assert zw.shape[0] == self.D and zv.size == (f_end - f_begin) * self.D
        self._ck(lib().mfm_sweep_wV(self.h, float(alpha), float(e_shift), _p(lambda_w), _p(mu_w), _p(zw), f_begin, f_end,
                                    _p(lam), _p(mu), _p(zv)))

    def update_e_regression(self):
        self._ck(lib().mfm_update_e_regression(self.h))

    def update_e_classification(self, seed, draw_index):
        self._ck(lib().mfm_update_e_classification(self.h, seed, draw_index))

    def score_train(self):
        self._ck(lib().mfm_score_train(self.h))

    def oprobit_add_group(self, n_class, rows=None):
        g = C.c_int32()
        if rows is None:
            self._ck(lib().mfm_oprobit_add_group(self.h, n_class, None, 0, C.byref(g)))
        else:
            rows = np.ascontiguousarray(rows, dtype=np.int64)
            self._ck(lib().mfm_oprobit_add_group(self.h, n_class, _p(rows), rows.shape[0], C.byref(g)))
        return g.value

    def oprobit_eval(self, group, gamma, want_h=True):
        gamma = _f64(gamma)
        m = gamma.shape[0]
        ll = C.c_double()
        dg = np.empty(m)
        H = np.empty((m, m)) if want_h else None
        self._ck(lib().mfm_oprobit_eval(self.h, group, _p(gamma), C.byref(ll), _p(dg), _p(H)))
        return ll.value, dg, H

    def oprobit_sample_z(self, group, gamma, seed, draw_index):
        gamma = _f64(gamma)
        self._ck(lib().mfm_oprobit_sample_z(self.h, group, _p(gamma), seed, draw_index))

    def synchronize(self):
        self._ck(lib().mfm_synchronize(self.h))

    # --- device random stream
    RNG_OP = np.dtype([("kind", np.int32), ("dest", np.int32), ("count", np.int64), ("offset", np.int64), ("shape", np.float64)])

    def rng_seed_mt19937(self, state624, position):
        st = np.ascontiguousarray(state624, dtype=np.uint32)
        assert st.shape[0] == 624
        self._ck(lib().mfm_rng_seed_mt19937(self.h, _p(st), int(position)))

    def rng_set_program(self, ops):
        """ops: list of (kind, dest, count, offset, shape)"""
        arr = np.array([tuple(o) for o in ops], dtype=self.RNG_OP)
        self._n_hv = int(sum(o[2] for o in ops if o[1] == 0))
        self._ck(lib().mfm_rng_set_program(self.h, _p(arr), len(ops)))

    def rng_prefetch(self):
        self._ck(lib().mfm_rng_prefetch(self.h))

    def rng_acquire(self):
        hv = np.empty(max(self._n_hv, 1))
        self._ck(lib().mfm_rng_acquire(self.h, _p(hv), self._n_hv))
        return hv[: self._n_hv]

    def rng_get_z(self):
        zw, zv = np.empty(self.D), np.empty((max(self.K, 1), self.D))
        self._ck(lib().mfm_rng_get_z(self.h, _p(zw), _p(zv)))
        return zw, zv[: self.K]

    # --- the state-dependent draws on the same stream (latent mode "exact")
    def update_e_classification_exact(self):
        st = C.c_int32()
        self._ck(lib().mfm_update_e_classification_exact(self.h, C.byref(st)))
        return st.value

    def oprobit_sample_z_exact(self, group, gamma):
        gamma = _f64(gamma)
        st = C.c_int32()
        self._ck(lib().mfm_oprobit_sample_z_exact(self.h, group, _p(gamma), C.byref(st)))
        return st.value

    def latent_stats(self):
        out = np.zeros(8, dtype=np.int64)
        self._ck(lib().mfm_latent_stats(self.h, _p(out)))
        return dict(zip(("status", "chunks", "subs", "lq", "quads", "walkers", "attempts"), (int(v) for v in out)))

    def rng_host_read(self, offset, n):
        out = np.empty(max(int(n), 1), dtype=np.uint32)
        self._ck(lib().mfm_rng_host_read(self.h, int(offset), int(n), _p(out)))
        return out[: int(n)]

    def rng_host_advance(self, words):
        self._ck(lib().mfm_rng_host_advance(self.h, int(words)))

    def set_residual_policy(self, recomputable):
        self._ck(lib().mfm_set_residual_policy(self.h, 1 if recomputable else 0))

    def plan_flags(self):
        f = lib().mfm_plan_flags(self.h)
        return {"qfree": bool(f & 1), "unit": bool(f & 2), "ell": bool(f & 4), "sharded": bool(f & 8), "soa": bool(f & 16),
                "fused_next": bool(f & 32), "sharded_fused": bool(f & 64), "mf": bool(f & 128), "resident": bool(f & 256), "cell": bool(f & 512), "streamed_chain": bool(f & 1024), "resident_overflow": bool(f & 2048)}

    def plan_info(self):
        a, b = C.c_int64(), C.c_int64()
        lib().mfm_plan_info(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    # --- timing
    def timing_enable(self, on=True):
        self._ck(lib().mfm_timing_enable(self.h, int(on)))

    def timing_reset(self):
        self._ck(lib().mfm_timing_reset(self.h))

    def timing(self):
        """{class name: (ms_total, launches, algorithmic_bytes_total)} for classes that ran."""
        L = lib()
        out = {}
        for c in range(L.mfm_timing_n_classes()):
            ms, n, by = C.c_double(), C.c_int64(), C.c_double()
            self._ck(L.mfm_timing_get(self.h, c, C.byref(ms), C.byref(n), C.byref(by)))
            if n.value:
                out[L.mfm_timing_class_name(c).decode()] = (ms.value, n.value, by.value)
        return out


class Design:
    """A prediction design (main CSR + relation blocks) resident on the GPU."""

    def __init__(self, X, blocks=(), device=0):
        L = lib()
        h = C.c_void_p()
        X, ip, ix, dv = csr_parts(X)
        rc = L.mfm_design_create(device, X.shape[0], X.shape[1], _p(ip), _p(ix), _p(dv), C.byref(h))
        if rc:
            _raise(rc, L.mfm_global_error())
        self.h = h
        self.N = X.shape[0]
        for mp, B in blocks:
            B, bp, bx, bv = csr_parts(B)
            mp = np.ascontiguousarray(mp, dtype=np.int64)
            if mp.shape[0] != self.N:
                self.close()
                raise ValueError("Relation blocks have inconsistent mapper size with case_size")
            rc = L.mfm_design_add_block(h, B.shape[0], B.shape[1], _p(bp), _p(bx), _p(bv), _p(mp))
            if rc:
                msg = L.mfm_design_last_error(h)
                self.close()
                _raise(rc, msg)
        self.D = L.mfm_design_dim_all(h)

    def close(self):
        if getattr(self, "h", None):
            lib().mfm_design_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def score_ctx(self, ctx):
        """FM::predict_score with the model state resident in training context `ctx`."""
        out = np.empty(self.N)
        rc = lib().mfm_design_score_ctx(self.h, ctx.h, _p(out))
        if rc:
            _raise(rc, lib().mfm_design_last_error(self.h))
        return out

    def predict(self, samples, mode=0, cutpoints=None):
        """samples: list of (w0, w[D], V[D, K]); mode 0 mean score, 1 mean Phi(score), 2 ordered probit."""
        S = len(samples)
        K = np.asarray(samples[0][2]).shape[1] if S else 0
        w0s = _f64([s[0] for s in samples])
        ws = _f64(np.stack([np.asarray(s[1], dtype=np.float64) for s in samples])) if S else np.empty(0)
        Vs = _f64(np.stack([np.asarray(s[2], dtype=np.float64).T for s in samples])) if S else np.empty(0)
        n_cut = 0
        cp = None
        if mode == 2:
            cp = _f64(np.stack([np.asarray(c, dtype=np.float64) for c in cutpoints]))
            n_cut = cp.shape[1]
        out = np.empty((self.N, n_cut + 1) if mode == 2 else self.N)
        rc = lib().mfm_design_predict(self.h, K, S, _p(w0s), _p(ws), _p(Vs), mode, n_cut, _p(cp), _p(out))
        if rc:
            _raise(rc, lib().mfm_design_last_error(self.h))
        return out


class Store:
    """posterior samples resident on the GPU (mfm_store_*)"""

    def __init__(self, D, K, device=0):
        h = C.c_void_p()
        rc = lib().mfm_store_create(device, D, K, C.byref(h))
        if rc:
            _raise(rc, lib().mfm_global_error())
        self.h, self.D, self.K = h, D, K

    def close(self):
        if getattr(self, "h", None):
            lib().mfm_store_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc:
            _raise(rc, lib().mfm_store_last_error(self.h))

    def __len__(self):
        return lib().mfm_store_size(self.h)

    def push_ctx(self, ctx):
        self._ck(lib().mfm_store_push_ctx(self.h, ctx.h))

    def push(self, w0, w, V):
        w, Vt = _f64(w), _f64(np.asarray(V, dtype=np.float64).T)
        self._ck(lib().mfm_store_push_host(self.h, float(w0), _p(w), _p(Vt)))

    def get(self, idx):
        w0, w, V = C.c_double(), np.empty(self.D), np.empty((self.K, self.D))
        self._ck(lib().mfm_store_get(self.h, idx, C.byref(w0), _p(w), _p(V)))
        return w0.value, w, np.ascontiguousarray(V.T)

    def predict(self, design, mode=0, cutpoints=None, first=0, count=None):
        count = len(self) - first if count is None else count
        n_cut, cp = 0, None
        if mode == 2:
            cp = _f64(np.stack([np.asarray(c, dtype=np.float64) for c in cutpoints]))
            n_cut = cp.shape[1]
        out = np.empty((design.N, n_cut + 1) if mode == 2 else design.N)
        rc = lib().mfm_design_predict_store(design.h, self.h, first, count, mode, n_cut, _p(cp), _p(out))
        if rc:
            _raise(rc, lib().mfm_design_last_error(design.h))
        return out
