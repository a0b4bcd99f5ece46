"""ctypes front-end of oracle/libmyfm_oracle.so (see myfm_oracle.cpp header).

TEST INFRASTRUCTURE ONLY: the checker for tests/, smoke() and the cpu_baseline leg of bench.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sps

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmyfm_oracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libfaddeeva_ref.so")

REGRESSION, CLASSIFICATION, ORDERED = 0, 1, 2


def build(force: bool = False) -> None:
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    src = os.path.join(_HERE, "myfm_oracle.cpp")
    stale = (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    if force or stale or not os.path.exists(_REF_PATH):
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("MYFM_ORACLE_LIB") or _LIB_PATH  # (bench.py's CPU leg: a -march=native build of the same source)
        if path == _LIB_PATH and not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(path)
        vp, i64, i32, dbl = C.c_void_p, C.c_int64, C.c_int32, C.c_double
        P = C.c_void_p  # raw data pointers
        L.orc_last_error.restype = C.c_char_p
        L.orc_new.restype = vp
        L.orc_free.argtypes = [vp]
        L.orc_clone.restype = vp
        L.orc_clone.argtypes = [vp]
        L.orc_set_main.argtypes = [vp, i64, i64, P, P, P, P, i64]
        L.orc_add_block.argtypes = [vp, i64, i64, P, P, P, P, i64]
        L.orc_set_config.argtypes = [vp, C.c_int, dbl, dbl, dbl, dbl, dbl, C.c_int, C.c_int, dbl, P, i64]
        L.orc_add_cutpoint_group.argtypes = [vp, C.c_int, P, i64]
        L.orc_start.argtypes = [vp, C.c_int, dbl, C.c_int]
        L.orc_step.argtypes = [vp]
        L.orc_substep.argtypes = [vp, C.c_int]
        L.orc_update_V_factor.argtypes = [vp, C.c_int]
        L.orc_dim_all.restype = i64
        L.orc_dim_all.argtypes = [vp]
        L.orc_n_groups.argtypes = [vp]
        L.orc_get_fm.argtypes = [vp, P, P, P]
        L.orc_set_fm.argtypes = [vp, dbl, P, P]
        L.orc_get_hyper.argtypes = [vp, P, P, P, P, P]
        L.orc_set_hyper.argtypes = [vp, dbl, P, P, P, P]
        L.orc_get_e.argtypes = [vp, P]
        L.orc_get_q.argtypes = [vp, P]
        L.orc_set_e.argtypes = [vp, P]
        L.orc_n_cutpoint_groups.argtypes = [vp]
        L.orc_cutpoint_size.argtypes = [vp, C.c_int]
        L.orc_get_cutpoints.argtypes = [vp, C.c_int, P]
        L.orc_mh_accept.restype = i64
        L.orc_mh_accept.argtypes = [vp, C.c_int]
        L.orc_rng_sample_normals.argtypes = [vp, i64, P]
        L.orc_rng_gamma.restype = dbl
        L.orc_rng_gamma.argtypes = [vp, dbl, dbl]
        L.orc_rng_raw.restype = C.c_uint32
        L.orc_rng_raw.argtypes = [vp]
        L.orc_rng_state.argtypes = [vp, P]
        L.orc_rng_set_state.argtypes = [vp, P]
        L.orc_trace_enable.argtypes = [vp, C.c_int]
        L.orc_trace_size.restype = i64
        L.orc_trace_size.argtypes = [vp]
        L.orc_trace_get.argtypes = [vp, P]
        L.orc_predict_score.argtypes = [vp, dbl, P, P, C.c_int, P]
        L.orc_tn_left.restype = dbl
        L.orc_tn_left.argtypes = [C.c_uint32, dbl]
        L.orc_tn_twoside.restype = dbl
        L.orc_tn_twoside.argtypes = [C.c_uint32, dbl, dbl]
        L.orc_tn_left_many.argtypes = [C.c_uint32, dbl, i64, P]
        L.orc_tn_twoside_many.argtypes = [C.c_uint32, dbl, dbl, i64, P]
        L.orc_erfcx.restype = dbl
        L.orc_erfcx.argtypes = [dbl]
        L.orc_oprobit_eval.argtypes = [C.c_int, P, dbl, P, P, i64, P, i64, C.POINTER(dbl), P, P, C.POINTER(dbl), P, P]
        _lib = L
    return _lib


def ref_faddeeva():
    """The reference's own Faddeeva.cc, compiled in place into oracle/_ref (None if absent)."""
    if not os.path.exists(_REF_PATH):
        return None
    R = C.CDLL(_REF_PATH)
    for n in ("ref_erfcx", "ref_erf", "ref_erfc"):
        getattr(R, n).restype = C.c_double
        getattr(R, n).argtypes = [C.c_double]
    return R


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _csr_parts(X):
    X = sps.csr_matrix(X, dtype=np.float64)
    X.sort_indices()
    return (
        X,
        np.ascontiguousarray(X.indptr, dtype=np.int64),
        np.ascontiguousarray(X.indices, dtype=np.int32),
        np.ascontiguousarray(X.data, dtype=np.float64),
    )


def _check(rc):
    if rc != 0:
        msg = lib().orc_last_error().decode()
        raise (ValueError if rc == -2 else RuntimeError)(msg)


class OracleTrainer:
    """One GibbsFMTrainer + its FM + hyper, steppable (create_train_fm unrolled)."""

    def __init__(
        self,
        X,
        y,
        blocks=(),  # sequence of (original_to_block, csr)
        rank=4,
        init_std=0.1,
        seed=42,
        task=REGRESSION,
        alpha_0=1.0,
        beta_0=1.0,
        gamma_0=1.0,
        mu_0=0.0,
        reg_0=1.0,
        fit_w0=True,
        fit_linear=True,
        group_index=None,
        nu_oprobit=5.0,
        cutpoint_groups=None,
        _handle=None,
    ):
        L = lib()
        self.rank = rank
        if _handle is not None:
            self.h = _handle
            return
        self.h = L.orc_new()
        X, ip, ix, dv = _csr_parts(X)
        y = np.ascontiguousarray(y, dtype=np.float64)
        _check(L.orc_set_main(self.h, X.shape[0], X.shape[1], _p(ip), _p(ix), _p(dv), _p(y), y.shape[0]))
        D = X.shape[1]
        for mp, B in blocks:
            B, bp, bx, bv = _csr_parts(B)
            mp = np.ascontiguousarray(mp, dtype=np.int64)
            _check(L.orc_add_block(self.h, B.shape[0], B.shape[1], _p(bp), _p(bx), _p(bv), _p(mp), mp.shape[0]))
            D += B.shape[1]
        if group_index is None:
            group_index = np.zeros(D, dtype=np.int32)
        group_index = np.ascontiguousarray(group_index, dtype=np.int32)
        _check(
            L.orc_set_config(
                self.h, task, alpha_0, beta_0, gamma_0, mu_0, reg_0, int(fit_w0), int(fit_linear), nu_oprobit,
                _p(group_index), group_index.shape[0],
            )
        )
        if task == ORDERED:
            if cutpoint_groups is None:
                cutpoint_groups = [(int(y.max()) + 1, np.arange(y.shape[0]))]
            for n_class, rows in cutpoint_groups:
                rows = np.ascontiguousarray(rows, dtype=np.int64)
                _check(L.orc_add_cutpoint_group(self.h, int(n_class), _p(rows), rows.shape[0]))
        _check(L.orc_start(self.h, rank, init_std, seed))

    def __del__(self):
        try:
            lib().orc_free(self.h)
        except Exception:
            pass

    def clone(self):
        o = OracleTrainer(None, None, rank=self.rank, _handle=lib().orc_clone(self.h))
        return o

    @property
    def D(self):
        return lib().orc_dim_all(self.h)

    @property
    def G(self):
        return lib().orc_n_groups(self.h)

    def step(self):
        _check(lib().orc_step(self.h))

    def substep(self, which):
        _check(lib().orc_substep(self.h, which))

    def update_V_factor(self, f):
        _check(lib().orc_update_V_factor(self.h, f))

    def fm(self):
        """(w0, w[D], V[D, K]) -- V returned in the (D, K) shape the boundary exposes."""
        D, K = self.D, self.rank
        w0 = C.c_double()
        w = np.empty(D)
        V = np.empty((K, D))
        lib().orc_get_fm(self.h, C.byref(w0), _p(w), _p(V))
        return w0.value, w, np.ascontiguousarray(V.T)

    def set_fm(self, w0, w, V):
        w = np.ascontiguousarray(w, dtype=np.float64)
        Vt = np.ascontiguousarray(np.asarray(V, dtype=np.float64).T)
        lib().orc_set_fm(self.h, float(w0), _p(w), _p(Vt))

    def hyper(self):
        """dict(alpha, mu_w[G], lambda_w[G], mu_V[G,K], lambda_V[G,K])"""
        G, K = self.G, self.rank
        a = C.c_double()
        mu_w, lam_w = np.empty(G), np.empty(G)
        mu_V, lam_V = np.empty((K, G)), np.empty((K, G))
        lib().orc_get_hyper(self.h, C.byref(a), _p(mu_w), _p(lam_w), _p(mu_V), _p(lam_V))
        return dict(alpha=a.value, mu_w=mu_w, lambda_w=lam_w, mu_V=mu_V.T.copy(), lambda_V=lam_V.T.copy())

    def set_hyper(self, alpha, mu_w, lambda_w, mu_V, lambda_V):
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        mu_w, lambda_w = f(mu_w), f(lambda_w)
        mu_Vt, lam_Vt = f(np.asarray(mu_V).T), f(np.asarray(lambda_V).T)
        lib().orc_set_hyper(self.h, float(alpha), _p(mu_w), _p(lambda_w), _p(mu_Vt), _p(lam_Vt))

    def e(self, n):
        out = np.empty(n)
        lib().orc_get_e(self.h, _p(out))
        return out

    def q(self, n):
        out = np.empty(n)
        lib().orc_get_q(self.h, _p(out))
        return out

    def set_e(self, e):
        e = np.ascontiguousarray(e, dtype=np.float64)
        lib().orc_set_e(self.h, _p(e))

    def cutpoints(self, g=0):
        n = lib().orc_cutpoint_size(self.h, g)
        out = np.empty(n)
        lib().orc_get_cutpoints(self.h, g, _p(out))
        return out

    def mh_accept(self, g=0):
        return lib().orc_mh_accept(self.h, g)

    def rng_sample_normals(self, n):
        out = np.empty(n)
        lib().orc_rng_sample_normals(self.h, n, _p(out))
        return out

    def rng_gamma(self, shape, scale):
        return lib().orc_rng_gamma(self.h, shape, scale)

    def rng_state(self):
        """(state[624] uint32, position) of the trainer's std::mt19937"""
        out = np.empty(625, dtype=np.uint32)
        lib().orc_rng_state(self.h, _p(out))
        return out[:624].copy(), int(out[624])

    def set_rng_state(self, state624, position):
        buf = np.empty(625, dtype=np.uint32)
        buf[:624] = state624
        buf[624] = position
        lib().orc_rng_set_state(self.h, _p(buf))

    def trace_enable(self, on=True):
        lib().orc_trace_enable(self.h, int(on))

    def trace(self):
        n = lib().orc_trace_size(self.h)
        out = np.empty(n)
        if n:
            lib().orc_trace_get(self.h, _p(out))
        return out.reshape(-1, 5)


def oprobit_eval(n_class, alpha, scores, y, rows=None, reg=1.0):
    """OprobitSampler::operator() (OProbitSampler.hpp:389-463) on given scores / labels for the rows of one cutpoint
    group: dict(ll, dgamma, Hg) = the row loop in gamma space (:402-413), (neg_ll, dalpha, Ha) = the full result."""
    alpha = np.ascontiguousarray(alpha, dtype=np.float64)
    m = n_class - 1
    assert alpha.shape[0] == m
    x = np.ascontiguousarray(scores, dtype=np.float64)
    yv = np.ascontiguousarray(y, dtype=np.float64)
    rows = np.arange(x.shape[0], dtype=np.int64) if rows is None else np.ascontiguousarray(rows, dtype=np.int64)
    ll, nll = C.c_double(), C.c_double()
    dg, Hg, da, Ha = np.empty(m), np.empty((m, m)), np.empty(m), np.empty((m, m))
    _check(lib().orc_oprobit_eval(n_class, _p(alpha), float(reg), _p(x), _p(yv), x.shape[0], _p(rows), rows.shape[0],
                                  C.byref(ll), _p(dg), _p(Hg), C.byref(nll), _p(da), _p(Ha)))
    return dict(ll=ll.value, dgamma=dg, Hg=Hg, neg_ll=nll.value, dalpha=da, Ha=Ha)


def tn_left_many(seed, mu_minus, n):
    """util.hpp:15-37 on a std::mt19937(seed): n draws of z ~ N(0,1) | z > mu_minus"""
    out = np.empty(n)
    lib().orc_tn_left_many(seed, float(mu_minus), n, _p(out))
    return out


def tn_twoside_many(seed, lo, hi, n):
    """util.hpp:39-60: n draws of z ~ N(0,1) | lo < z < hi"""
    out = np.empty(n)
    lib().orc_tn_twoside_many(seed, float(lo), float(hi), n, _p(out))
    return out


class OracleDesign:
    """A (main CSR, blocks) design to score FM samples on: FM::predict_score (FM.hpp:47-52)."""

    def __init__(self, X, blocks=()):
        L = lib()
        self.h = L.orc_new()
        X, ip, ix, dv = _csr_parts(X)
        self.n = X.shape[0]
        y = np.zeros(self.n)
        _check(L.orc_set_main(self.h, X.shape[0], X.shape[1], _p(ip), _p(ix), _p(dv), _p(y), self.n))
        for mp, B in blocks:
            B, bp, bx, bv = _csr_parts(B)
            mp = np.ascontiguousarray(mp, dtype=np.int64)
            _check(L.orc_add_block(self.h, B.shape[0], B.shape[1], _p(bp), _p(bx), _p(bv), _p(mp), mp.shape[0]))

    def __del__(self):
        try:
            lib().orc_free(self.h)
        except Exception:
            pass

    def predict_score(self, w0, w, V):
        w = np.ascontiguousarray(w, dtype=np.float64)
        V = np.asarray(V, dtype=np.float64)
        Vt = np.ascontiguousarray(V.T)
        out = np.empty(self.n)
        _check(lib().orc_predict_score(self.h, float(w0), _p(w), _p(Vt), V.shape[1], _p(out)))
        return out


def fit(X, y, blocks=(), n_iter=100, n_kept_samples=None, **kw):
    """create_train_fm (declare_module.hpp:30-45) + learn_with_callback (FMTrainer.hpp:56-87):
    returns (samples=[(w0, w, V[, cutpoints])...], hypers=[dict...], trainer)."""
    if n_kept_samples is None:
        n_kept_samples = min(max(n_iter - 5, 5), n_iter)
    t = OracleTrainer(X, y, blocks, **kw)
    samples, hypers = [], []
    for it in range(n_iter):
        t.step()
        if n_iter <= it + n_kept_samples:
            s = t.fm()
            if kw.get("task", REGRESSION) == ORDERED:
                s = s + (t.cutpoints(0),)
            samples.append(s)
        hypers.append(t.hyper())
    return samples, hypers, t
