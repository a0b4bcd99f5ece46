"""LibFM-like per-iteration test scoring (reference: src/myfm/utils/callbacks/libfm.py).

The callbacks keep running sums of the per-iteration test predictions instead of posterior samples
(``n_kept_samples`` can then be tiny), exactly like libFM reports "test RMSE of the running mean".
Same class names, constructor arguments, ``predictions`` / ``result_trace`` attributes and metric
names as the reference. On this backend ``fm.predict_score`` of the live sample scores the test design
on the GPU straight from the device-resident model state (no download of w / V per iteration).
"""
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np
from scipy import sparse as sps

from ..estimators import REAL, check_data_consistency, std_cdf

NAN = float("nan")


class LibFMLikeCallbackBase:
    def __init__(self, n_iter, X_test, X_rel_test, y_test, trace_path: Optional[str] = None):
        self.n_test_data = check_data_consistency(X_test, X_rel_test)
        self.n_iter = n_iter
        self.X_test = (
            sps.csr_matrix(X_test, dtype=REAL) if X_test is not None else sps.csr_matrix((self.n_test_data, 0), dtype=REAL)
        )
        self.X_rel_test = list(X_rel_test)
        self.y_test = np.asarray(y_test)
        self.result_trace: List[Dict[str, float]] = []
        self.trace_path = trace_path
        self.n_samples = 0

    # -- to be provided by the task-specific callbacks --------------------------------------------------
    def _predict_this(self, fm):
        raise NotImplementedError

    def _metrics(self, pred) -> "OrderedDict[str, float]":
        raise NotImplementedError

    def _describe(self, hyper, mean, this, but5) -> str:
        raise NotImplementedError

    def _post(self, arr):
        return arr

    # -- running means: all iterations, and all but the first five (burn-in) ---------------------------------
    def _measure_score(self, i, fm, hyper) -> Tuple[str, Dict[str, float]]:
        this = self._predict_this(fm)
        self.predictions += this
        self.n_samples += 1
        m_mean = self._metrics(self._post(self.predictions / self.n_samples))
        m_this = self._metrics(this)
        if i >= 5:
            self.prediction_all_but_5 += this
            m_but5 = self._metrics(self._post(self.prediction_all_but_5 / (i + 1 - 5)))
        else:
            m_but5 = OrderedDict((k, NAN) for k in m_mean)
        result = OrderedDict()
        for k in m_mean:
            result[k] = m_mean[k]
            result[k + "_this"] = m_this[k]
            result[k + "_all_but_5"] = m_but5[k]
        return self._describe(hyper, m_mean, m_this, m_but5), self._order(result, hyper)

    def _order(self, result, hyper):
        return result

    def __call__(self, i, fm, hyper, history) -> Tuple[bool, Optional[str]]:
        description, trace_result = self._measure_score(i, fm, hyper)
        self.result_trace.append(trace_result)
        if self.trace_path is not None:
            import pandas as pd

            pd.DataFrame(self.result_trace).to_csv(self.trace_path, index=False)
        return False, description


class RegressionCallback(LibFMLikeCallbackBase):
    def __init__(self, n_iter, X_test, y_test, X_rel_test=[], clip_min=None, clip_max=None, trace_path=None):
        super().__init__(n_iter, X_test, X_rel_test, y_test, trace_path=trace_path)
        self.predictions = np.zeros(self.n_test_data)
        self.prediction_all_but_5 = np.zeros(self.n_test_data)
        self.clip_min, self.clip_max = clip_min, clip_max

    def clip_value(self, arr):
        if self.clip_min is not None:
            arr[arr <= self.clip_min] = self.clip_min
        if self.clip_max is not None:
            arr[arr >= self.clip_max] = self.clip_max

    def _post(self, arr):
        self.clip_value(arr)
        return arr

    def _predict_this(self, fm):
        return np.asarray(fm.predict_score(self.X_test, self.X_rel_test))

    def _metrics(self, pred):
        return OrderedDict(rmse=float(((self.y_test - pred) ** 2).mean() ** 0.5))

    def _describe(self, hyper, mean, this, but5):
        return "alpha={0:.4f}, rmse_mean={1:.4f}, rmse_this={2:.4f}, rmse_all_but_5={3:.4f}".format(
            hyper.alpha, mean["rmse"], this["rmse"], but5["rmse"]
        )

    def _order(self, result, hyper):
        out = OrderedDict(alpha=hyper.alpha)
        out.update(result)
        return out


class ClassificationCallback(LibFMLikeCallbackBase):
    def __init__(self, n_iter, X_test, y_test, X_rel_test=[], eps=1e-15, trace_path=None):
        super().__init__(n_iter, X_test, X_rel_test, y_test, trace_path=trace_path)
        self.predictions = np.zeros(self.n_test_data)
        self.prediction_all_but_5 = np.zeros(self.n_test_data)
        self.eps = eps

    def clip_value(self, arr):
        if self.eps is not None:
            arr[arr <= self.eps] = self.eps
            arr[arr >= (1 - self.eps)] = 1 - self.eps

    def _post(self, arr):
        self.clip_value(arr)
        return arr

    def _predict_this(self, fm):
        return std_cdf(fm.predict_score(self.X_test, self.X_rel_test))

    def _metrics(self, p):
        ll = -(np.log(p[self.y_test == 1]).sum() + np.log(1 - p[self.y_test == 0]).sum())
        return OrderedDict(log_loss=ll, accuracy=float((self.y_test == (p >= 0.5)).mean()))

    def _describe(self, hyper, mean, this, but5):
        return "ll_mean={0:.4f}, ll_this={1:.4f}, ll_all_but_5={2:.4f}".format(
            mean["log_loss"], this["log_loss"], but5["log_loss"]
        )

    def _order(self, r, hyper):
        keys = ["log_loss", "log_loss_this", "log_loss_all_but_5", "accuracy", "accuracy_this", "accuracy_all_but_5"]
        return OrderedDict((k, r[k]) for k in keys)


class OrderedProbitCallback(LibFMLikeCallbackBase):
    def __init__(self, n_iter, X_test, y_test, n_class, X_rel_test=[], eps=1e-15, trace_path=None):
        super().__init__(n_iter, X_test, X_rel_test, y_test, trace_path=trace_path)
        self.predictions = np.zeros((self.n_test_data, n_class))
        self.prediction_all_but_5 = np.zeros((self.n_test_data, n_class))
        self.n_class, self.eps = n_class, eps
        self.y_test = self.y_test.astype(np.int32)
        assert self.y_test.min() >= 0 and self.y_test.max() <= n_class - 1

    def _predict_this(self, fm):
        return np.asarray(fm.oprobit_predict_proba(self.X_test, self.X_rel_test, 0))

    def _metrics(self, p):
        ps = p[np.arange(self.y_test.shape[0]), self.y_test].copy()
        ps[ps <= self.eps] = self.eps
        return OrderedDict(
            log_loss=-float(np.log(ps).sum()),
            accuracy=float((self.y_test == p.argmax(axis=1)).mean()),
            rmse=float(((self.y_test - p.dot(np.arange(self.n_class))) ** 2).mean()) ** 0.5,
        )

    def _describe(self, hyper, mean, this, but5):
        return "ll_mean={0:.4f}, ll_this={1:.4f}, ll_all_but_5={2:.4f}".format(
            mean["log_loss"], this["log_loss"], but5["log_loss"]
        )

    def _order(self, r, hyper):
        keys = [m + s for m in ("log_loss", "accuracy", "rmse") for s in ("", "_this", "_all_but_5")]
        return OrderedDict((k, r[k]) for k in keys)
