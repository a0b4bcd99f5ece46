"""sklearn-style estimators over the MI355X backend: the MyFMRegressor / MyFMClassifier /
MyFMOrderedProbit surface of the reference (src/myfm/base.py:70-399, src/myfm/gibbs.py:32-543),
written against ``myfm_amd._myfm``. Same constructor / fit / predict arguments, defaults and
error behaviour; numpy-2 safe.
"""
import os
from collections import OrderedDict
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
from scipy import sparse as sps
from scipy import special

from . import _myfm
from ._myfm import ConfigBuilder, RelationBlock, TaskType

REAL = np.float64


def std_cdf(x):
    """Standard normal CDF (base.py:41-43)."""
    return (1 + special.erf(np.asarray(x) * np.sqrt(0.5))) / 2


def check_data_consistency(X, X_rel) -> int:
    """Number of cases shared by X and the relation blocks (base.py:46-61)."""
    if X_rel:
        sizes = {rel.mapper_size for rel in X_rel}
        if len(sizes) > 1:
            raise ValueError("Inconsistent case size for X_rel.")
        n = sizes.pop()
        if X is not None and X.shape[0] != n:
            raise ValueError("X and X_rel have different shape.")
        return n
    if X is None:
        raise ValueError("At least X or X_rel must be provided.")
    return int(X.shape[0])


def _as_csr(X, n_rows) -> sps.csr_matrix:
    if X is None:
        return sps.csr_matrix((n_rows, 0), dtype=REAL)
    X = sps.csr_matrix(X)
    if X.dtype != REAL:
        X = X.astype(REAL)
    # canonical CSR is a precondition of the sampler (duplicates would change sum x^2)
    if not X.has_canonical_format:
        X = X.copy()
        X.sum_duplicates()
    return X


class _ProgressBar:
    """tqdm when available (base.py:303-312), silent otherwise."""

    def __init__(self, total):
        try:
            from tqdm import tqdm

            self.bar = tqdm(total=total)
        except Exception:  # pragma: no cover
            self.bar = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        if self.bar is not None:
            self.bar.close()

    def update(self, message, n=1):
        if self.bar is not None:
            if message is not None:
                self.bar.set_description(message)
            self.bar.update(n)


class MyFMGibbsBase:
    """Common part of the Gibbs estimators (base.py:70-323 + gibbs.py:32-142)."""

    _task_type = TaskType.REGRESSION

    def __init__(
        self,
        rank: int,
        init_stdev: float = 0.1,
        random_seed: int = 42,
        alpha_0: float = 1.0,
        beta_0: float = 1.0,
        gamma_0: float = 1.0,
        mu_0: float = 0.0,
        reg_0: float = 1.0,
        fit_w0: bool = True,
        fit_linear: bool = True,
        exact_latent_draws: bool = True,
    ):
        # exact_latent_draws (not in the reference, which has one generator and nothing to choose). True (default): under the same
        # seed EVERY draw is the reference's own -- also the latent draws of MyFMClassifier / MyFMOrderedProbit, which the reference
        # makes row after row inside rejection loops on its one std::mt19937: they are evaluated in parallel on the device from that
        # same stream (csrc/mfm_latent.hip), in the caller's row order whatever order the device paths keep the rows in. False: the
        # latent draws come from per-row Philox streams instead (same law, other numbers: posterior means agree with the reference's
        # chain only statistically) -- 1.15x faster at 5 10^7 rows, 5x at 10^7 rows of a two-field table. Regression chains are the
        # reference's draw for draw either way. Row-sharded fits always use the per-row streams.
        self.exact_latent_draws = bool(exact_latent_draws)
        self.rank = rank
        self.init_stdev = init_stdev
        self.random_seed = random_seed
        self.alpha_0 = alpha_0
        self.beta_0 = beta_0
        self.gamma_0 = gamma_0
        self.mu_0 = mu_0
        self.reg_0 = reg_0
        self.fit_w0 = fit_w0
        self.fit_linear = fit_linear
        self.predictor_ = None
        self.history_ = None
        self.n_groups_: Optional[int] = None

    def __str__(self) -> str:
        return (
            "{}(init_stdev={}, alpha_0={}, beta_0={}, gamma_0={}, mu_0={}, reg_0={})".format(
                self.__class__.__name__, self.init_stdev, self.alpha_0, self.beta_0, self.gamma_0, self.mu_0, self.reg_0
            )
        )

    # ---- hooks specialised per task ------------------------------------------------------------
    def _process_y(self, y):
        return np.asarray(y).astype(REAL)

    def _status_report(self, fm, hyper) -> str:
        raise NotImplementedError

    def _prepare_prediction_for_test(self, fm, X, X_rel):
        raise NotImplementedError

    def _measure_score(self, prediction, y) -> Dict[str, float]:
        raise NotImplementedError

    # ---- fitting -----------------------------------------------------------------------------------
    def _default_callback(self, freq, do_test, X_test, X_rel_test, y_test):
        def callback(i, fm, hyper, history):
            if i % freq:
                return False, None
            msg = self._status_report(fm, hyper)
            if do_test:
                pred = self._prepare_prediction_for_test(fm, X_test, X_rel_test)
                for key, metric in self._measure_score(pred, y_test).items():
                    msg += " {}_this: {:.2f}".format(key, metric)
            return False, msg

        return callback

    def _fit(
        self,
        X,
        y,
        X_rel=[],
        X_test=None,
        y_test=None,
        X_rel_test=[],
        n_iter: int = 100,
        n_kept_samples: Optional[int] = None,
        grouping: Optional[List[int]] = None,
        group_shapes: Optional[List[int]] = None,
        callback=None,
        config_builder: Optional[ConfigBuilder] = None,
        callback_default_freq: int = 10,
    ) -> None:
        if config_builder is None:
            config_builder = ConfigBuilder()
        y = np.asarray(y)
        train_size = check_data_consistency(X, X_rel)
        X = _as_csr(X, train_size)
        assert X.shape[0] == y.shape[0]
        dim_all = X.shape[1] + sum(rel.feature_size for rel in X_rel)

        if n_kept_samples is None:
            n_kept_samples = min(max(n_iter - 5, 5), n_iter)  # base.py:238-239
        else:
            assert n_iter >= n_kept_samples

        for key in ("alpha_0", "beta_0", "gamma_0", "mu_0", "reg_0", "fit_w0", "fit_linear"):
            getattr(config_builder, "set_" + key)(getattr(self, key))

        if group_shapes is not None and grouping is None:
            grouping = [g for g, size in enumerate(group_shapes) for _ in range(size)]
        if grouping is None:
            self.n_groups_ = 1
            config_builder.set_identical_groups(dim_all)
        else:
            assert dim_all == len(grouping)
            self.n_groups_ = len(set(grouping))
            config_builder.set_group_index([int(g) for g in grouping])

        if X_test is not None or X_rel_test:
            if y_test is None:
                raise RuntimeError("Must specify both (X_test or X_rel_test) and y_test.")
            test_size = check_data_consistency(X_test, X_rel_test)
            assert test_size == np.asarray(y_test).shape[0]
            X_test = _as_csr(X_test, test_size)
            do_test = True
        elif y_test is not None:
            raise RuntimeError("Must specify both (X_test or X_rel_test) and y_test.")
        else:
            do_test = False

        config_builder.set_n_iter(n_iter).set_n_kept_samples(n_kept_samples)
        y = self._process_y(y)
        # The sampler does not depend on the order of the training rows (every conditional is a sum over rows), the
        # device path does: a table sorted by its first one-hot field runs the fused passes. Rows that
        # arrive in another order are sorted by the first stored column here, together with y and the relation maps.
        perm = _device_row_order(X)
        if perm is not None:
            ptr, idx, val = _myfm.permute_csr_rows(X.indptr, X.indices, X.data, perm)
            X = sps.csr_matrix((val, idx, ptr), shape=X.shape)
            y = np.asarray(y)[perm]
            X_rel = [RelationBlock(r.original_to_block_array[perm], r.data) for r in X_rel]
            if self._task_type != TaskType.REGRESSION:
                # the latent draws are made in the CALLER's row order (FMTrainer.hpp:500, OProbitSampler.hpp:243): the sorted
                # table's row that is the caller's row i
                inv = np.empty(perm.shape[0], dtype=np.int64)
                inv[perm] = np.arange(perm.shape[0], dtype=np.int64)
                if self._task_type == TaskType.ORDERED:
                    config_builder.set_cutpoint_groups([(int(np.asarray(y).max()) + 1, inv)])
                else:
                    config_builder.set_latent_row_order(inv)
        config_builder.set_task_type(self._task_type)
        config_builder.set_exact_latent_draws(self.exact_latent_draws and not os.environ.get("MYFM_AMD_PHILOX_LATENT"))
        config = config_builder.build()

        default_callback = callback is None
        if default_callback:
            callback = self._default_callback(callback_default_freq, do_test, X_test, X_rel_test, y_test)

        with _ProgressBar(n_iter) as bar:
            seen = [0]

            def wrapped(i, fm, hyper, history) -> bool:
                should_stop, message = callback(i, fm, hyper, history)
                bar.update(message, i + 1 - seen[0])
                seen[0] = i + 1
                return bool(should_stop)

            if default_callback:
                # the default callback only acts every `callback_default_freq` iterations (base.py:179-205): the trainer calls into
                # Python on those (and the last one) only -- the iterations in between never leave the C++ loop
                wrapped.myfm_every = max(1, int(callback_default_freq))

            from . import distributed as _dist

            if _dist.active():
                # row-sharded over the process group: same data on every rank, each trains on its slice
                from . import _capi

                rank, world = _dist.rank_world()
                n = X.shape[0]
                levels, _ = _capi.column_levels(X) if X.shape[1] else (np.zeros(0, np.int32), 0)
                if X.shape[1] and np.diff(X.indptr).min() >= 1:
                    cuts = _dist.shard_cuts(X.indices[X.indptr[:-1]], world)
                else:
                    cuts = [(n * r) // world for r in range(world + 1)]
                lo, hi = cuts[rank], cuts[rank + 1]
                if self._task_type == TaskType.ORDERED:  # the cutpoint group lists LOCAL rows
                    config_builder.set_cutpoint_groups([(int(np.asarray(y).max()) + 1, np.arange(hi - lo, dtype=np.int64))])
                    config = config_builder.build()
                rel_l = [RelationBlock(r.original_to_block_array[lo:hi], r.data) for r in X_rel]
                y_l = np.ascontiguousarray(np.asarray(y, dtype=REAL)[lo:hi])
                self.predictor_, self.history_ = _myfm.create_train_fm_sharded(
                    self.rank, self.init_stdev, X[lo:hi], rel_l, y_l, self.random_seed, config, wrapped, rank, world, n, lo,
                    np.ascontiguousarray(levels, dtype=np.int32), **_dist.comm_kwargs())
            else:
                self.predictor_, self.history_ = _myfm.create_train_fm(
                    self.rank, self.init_stdev, X, list(X_rel), np.ascontiguousarray(y, dtype=REAL), self.random_seed, config,
                    wrapped)

    # ---- posterior access ---------------------------------------------------------------------------
    def _fetch_predictor(self):
        if self.predictor_ is None:
            raise RuntimeError("Predictor called before fit.")
        return self.predictor_

    @property
    def w0_samples(self):
        if self.predictor_ is None:
            return None
        return np.asarray([fm.w0 for fm in self.predictor_.samples], dtype=REAL)

    @property
    def w_samples(self):
        if self.predictor_ is None:
            return None
        return np.asarray([fm.w for fm in self.predictor_.samples], dtype=REAL)

    @property
    def V_samples(self):
        if self.predictor_ is None:
            return None
        return np.asarray([fm.V for fm in self.predictor_.samples], dtype=REAL)

    def _predict_core(self, X, X_rel=[], n_workers: Optional[int] = None):
        predictor = self._fetch_predictor()
        n = check_data_consistency(X, X_rel)
        X = _as_csr(X, n)
        if n_workers is None:
            return predictor.predict(X, list(X_rel))
        return predictor.predict_parallel(X, list(X_rel), n_workers)

    def get_hyper_trace(self):
        """alpha, mu_w[g], lambda_w[g], mu_V[g,r], lambda_V[g,r] per iteration (gibbs.py:109-142)."""
        import pandas as pd

        if self.n_groups_ is None or self.history_ is None:
            raise RuntimeError("Sampler not run yet.")
        G, K = self.n_groups_, self.rank
        columns = (
            ["alpha"]
            + ["mu_w[{}]".format(g) for g in range(G)]
            + ["lambda_w[{}]".format(g) for g in range(G)]
            + ["mu_V[{},{}]".format(g, r) for g in range(G) for r in range(K)]
            + ["lambda_V[{},{}]".format(g, r) for g in range(G) for r in range(K)]
        )
        rows = []
        for hyper in self.history_.hypers:
            parts = [np.asarray([hyper.alpha])]
            for hp in (hyper.mu_w, hyper.lambda_w, hyper.mu_V, hyper.lambda_V):
                parts.append(np.asarray(hp, dtype=REAL).ravel())  # C-order ravel of (G, K): g outer, r inner
            rows.append(np.concatenate(parts))
        df = pd.DataFrame(np.vstack(rows))
        df.columns = columns
        return df


class MyFMGibbsRegressor(MyFMGibbsBase):
    """Bayesian FM regression by Gibbs sampling (gibbs.py:145-240)."""

    _task_type = TaskType.REGRESSION

    def _prepare_prediction_for_test(self, fm, X, X_rel):
        return fm.predict_score(X, list(X_rel))

    def _status_report(self, fm, hyper) -> str:
        return "alpha = {:.2f} w0 = {:.2f} ".format(hyper.alpha, fm.w0)

    def _measure_score(self, prediction, y):
        y = np.asarray(y)
        out = OrderedDict()
        out["rmse"] = ((y - prediction) ** 2).mean() ** 0.5
        out["mae"] = np.abs(y - prediction).mean()
        return out

    def fit(self, X, y, X_rel=[], X_test=None, y_test=None, X_rel_test=[], n_iter=100, n_kept_samples=None, grouping=None,
            group_shapes=None, callback=None, config_builder=None):
        self._fit(X, y, X_rel=X_rel, X_test=X_test, y_test=y_test, X_rel_test=X_rel_test, n_iter=n_iter,
                  n_kept_samples=n_kept_samples, grouping=grouping, group_shapes=group_shapes, callback=callback,
                  config_builder=config_builder)
        return self

    def predict(self, X, X_rel=[], n_workers: Optional[int] = None):
        """Posterior predictive mean (gibbs.py:219-240)."""
        return self._predict_core(X, X_rel, n_workers=n_workers)


class MyFMGibbsClassifier(MyFMGibbsBase):
    """Bayesian FM probit classification (gibbs.py:243-371)."""

    _task_type = TaskType.CLASSIFICATION

    def _process_y(self, y):
        return np.asarray(y).astype(REAL) * 2 - 1  # base.py:385-386

    def _prepare_prediction_for_test(self, fm, X, X_rel):
        return std_cdf(fm.predict_score(X, list(X_rel)))

    def _status_report(self, fm, hyper) -> str:
        return "w0 = {:.2f} ".format(fm.w0)

    def _measure_score(self, prediction, y):
        y = np.asarray(y)
        out = OrderedDict()
        lp = np.log(prediction + 1e-15)
        l1mp = np.log(1 - prediction + 1e-15)
        gt = y > 0
        out["ll"] = (-lp.dot(gt) - l1mp.dot(~gt)) / max(1, prediction.shape[0])
        out["accuracy"] = np.mean((prediction >= 0.5) == gt)
        return out

    def fit(self, X, y, X_rel=[], X_test=None, y_test=None, X_rel_test=[], n_iter=100, n_kept_samples=None, grouping=None,
            group_shapes=None, callback=None, config_builder=None):
        self._fit(X, y, X_rel=X_rel, X_test=X_test, y_test=y_test, X_rel_test=X_rel_test, n_iter=n_iter,
                  n_kept_samples=n_kept_samples, grouping=grouping, group_shapes=group_shapes, callback=callback,
                  config_builder=config_builder)
        return self

    def predict_proba(self, X, X_rel=[], n_workers: Optional[int] = None):
        return self._predict_core(X, X_rel, n_workers=n_workers)

    def predict(self, X, X_rel=[], n_workers: Optional[int] = None):
        return self.predict_proba(X, X_rel, n_workers=n_workers) > 0.5


def _device_row_order(X):
    """stable order of the rows by their first stored column, or None when they already are (or it does not apply)"""
    import os

    if os.environ.get("MYFM_AMD_KEEP_ROW_ORDER") or X.shape[0] < 2 or X.shape[1] == 0:
        return None
    lens = np.diff(X.indptr)
    if lens.min() < 1:
        return None
    if not X.has_sorted_indices:
        X.sort_indices()
    first = X.indices[X.indptr[:-1]]
    if np.all(first[1:] >= first[:-1]):
        return None
    return _myfm.row_order_by_first_column(X.indptr, X.indices, X.shape[1])  # (stable counting sort)


class MyFMOrderedProbit(MyFMGibbsBase):
    """Bayesian FM ordinal regression (gibbs.py:374-543)."""

    _task_type = TaskType.ORDERED

    def _process_y(self, y):
        y = np.asarray(y)
        assert y.min() >= 0
        return y.astype(REAL)

    def fit(self, X, y, X_rel=[], X_test=None, y_test=None, X_rel_test=[], n_iter=100, n_kept_samples=None, grouping=None,
            group_shapes=None, callback=None, callback_default_freq=5):
        builder = ConfigBuilder()
        y = np.asarray(y)
        n_class = int(y.max()) + 1
        groups = [(n_class, np.arange(y.shape[0], dtype=np.int64))]
        self.n_cutpoint_groups = len(groups)
        builder.set_cutpoint_groups(groups)
        self._fit(X, y, X_rel=X_rel, X_test=X_test, y_test=y_test, X_rel_test=X_rel_test, n_iter=n_iter,
                  n_kept_samples=n_kept_samples, grouping=grouping, group_shapes=group_shapes, callback=callback,
                  config_builder=builder, callback_default_freq=callback_default_freq)
        return self

    def _prepare_prediction_for_test(self, fm, X, X_rel):
        return fm.oprobit_predict_proba(sps.csr_matrix(X, dtype=REAL), list(X_rel), 0)

    def _measure_score(self, prediction, y):
        y = np.asarray(y)
        out = OrderedDict()
        out["accuracy"] = (np.argmax(prediction, axis=1) == y).mean()
        out["log_loss"] = -np.log(prediction[np.arange(prediction.shape[0]), y.astype(np.int64)] + 1e-15).mean()
        return out

    def _status_report(self, fm, hyper) -> str:
        msg = "w0 = {:.2f}, ".format(fm.w0)
        cps = fm.cutpoints
        if len(cps) == 1:
            msg += "cutpoint = {} ".format(["{:.3f}".format(c) for c in list(cps[0])])
        return msg

    def predict_proba(self, X, X_rel=[], n_workers: Optional[int] = None):
        predictor = self._fetch_predictor()
        n = check_data_consistency(X, X_rel)
        X = _as_csr(X, n)
        return predictor.predict_parallel_oprobit(X, list(X_rel), n_workers or 1, 0)

    def predict(self, X, X_rel=[]):
        return self.predict_proba(X, X_rel=X_rel).argmax(axis=1)

    @property
    def cutpoint_samples(self):
        if self.predictor_ is None:
            return None
        return np.asarray([fm.cutpoints[0] for fm in self.predictor_.samples], dtype=REAL)


MyFMRegressor = MyFMGibbsRegressor
MyFMClassifier = MyFMGibbsClassifier
