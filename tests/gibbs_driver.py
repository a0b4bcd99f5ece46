"""Test-side driver: one Gibbs iteration (BaseFMTrainer.hpp:135-152) through the C ABI, with every
random variate taken from an oracle trainer's mt19937 in the reference's draw order (SURVEY 8a
"RNG draw order"). Regression only: there the RNG consumption is independent of the model state,
so an oracle trainer stepping normally and this driver consuming a *clone's* generator stay in
lock-step and the two chains must agree to fp64 round-off.

Test infrastructure; the shipped orchestrator is the C++ one in myfm_amd/csrc/_myfm.cpp.
"""
import numpy as np


class CapiGibbs:
    def __init__(self, ctx, rng_trainer, n_rows, group_index, alpha_0=1.0, beta_0=1.0, gamma_0=1.0, mu_0=0.0, reg_0=1.0,
                 fit_w0=True, fit_linear=True, fused=False):
        self.c = ctx
        self.rng = rng_trainer  # oracle trainer used ONLY as the mt19937 source
        self.N = n_rows
        self.gi = np.asarray(group_index)
        self.G = int(self.gi.max()) + 1
        self.n_g = np.bincount(self.gi, minlength=self.G).astype(np.float64)
        self.a0, self.b0, self.g0, self.m0, self.r0 = alpha_0, beta_0, gamma_0, mu_0, reg_0
        self.fit_w0, self.fit_linear = fit_w0, fit_linear
        self.fused = fused  # update_w0's shift + update_w + update_V through mfm_sweep_wV (one call)
        K, G = ctx.K, self.G
        self.alpha = 1.0
        self.mu_w, self.lam_w = np.zeros(G), np.full(G, 1e-5)
        self.mu_V, self.lam_V = np.zeros((G, K)), np.full((G, K), 1e-5)
        self.w0 = ctx.get_state()[0]

    # ---- random variates: host (oracle generator) or the device stream of csrc/mfm_rng.hpp ----
    def use_device_rng(self, state624, position):
        """hand the generator to the device and describe one iteration's draw order"""
        c, G, K, D = self.c, self.G, self.c.K, self.c.D
        c.rng_seed_mt19937(state624, position)
        ops, n = [], 0
        ops.append((1, 0, 1, n, (self.a0 + self.N) / 2)); n += 1
        if self.fit_w0:
            ops.append((0, 0, 1, n, 0.0)); n += 1
        for g in range(G):
            ops.append((1, 0, 1, n, (self.a0 + self.n_g[g]) / 2)); n += 1
        ops.append((0, 0, G, n, 0.0)); n += G
        if self.fit_linear:
            ops.append((0, 1, D, 0, 0.0))
        if K:
            for f in range(K):
                for g in range(G):
                    ops.append((1, 0, 1, n, (self.a0 + self.n_g[g]) / 2)); n += 1
            ops.append((0, 0, K * G, n, 0.0)); n += K * G
            ops.append((0, 2, K * D, 0, 0.0))
        c.rng_set_program(ops)
        c.rng_prefetch()
        self.dev = True
        self.hv, self.hvi = None, 0

    def _begin(self):
        if getattr(self, "dev", False):
            self.hv, self.hvi = self.c.rng_acquire(), 0
            self.c.rng_prefetch()

    def _hv(self):
        v = self.hv[self.hvi]
        self.hvi += 1
        return v

    def _gamma(self, shape, scale):
        if getattr(self, "dev", False):
            return self._hv() * scale
        return self.rng.rng_gamma(shape, scale)

    def _z(self, n):
        return None if getattr(self, "dev", False) else self.rng.rng_sample_normals(n)

    def _normal(self, quad, first):
        z = self._hv() if getattr(self, "dev", False) else self.rng.rng_sample_normals(1)[0]
        return first / quad + z / np.sqrt(quad)

    def step(self, before_update_e=None):
        c, K, G, D = self.c, self.c.K, self.G, self.c.D
        fuse = self.fused and self.fit_linear and K > 0
        e_shift = 0.0
        self._begin()
        # update_alpha, FMTrainer.hpp:127-145
        se, se2 = c.reduce_e()
        self.alpha = self._gamma((self.a0 + self.N) / 2, 1.0 / ((self.b0 + se2) / 2))
        # update_w0, :218-229
        if self.fit_w0:
            lin = self.alpha * (self.N * self.w0 - se)
            quad = self.alpha * self.N + self.r0
            w0_new = self._normal(quad, lin)
            e_shift = w0_new - self.w0
            self.w0 = w0_new
        else:
            self.w0 = 0.0
        if e_shift != 0.0 and not fuse:
            c.shift_e(e_shift)
        c.set_w0(self.w0)
        # update_lambda_w / update_mu_w, :150-200
        s, ssd = c.group_stats_w(self.mu_w)
        for g in range(G):
            self.lam_w[g] = self._gamma((self.a0 + self.n_g[g]) / 2, 2.0 / (self.b0 + ssd[g]))
        for g in range(G):
            sq = self.lam_w[g] * (self.g0 + self.n_g[g])
            lin = (self.g0 * self.m0 + s[g]) * self.lam_w[g]
            self.mu_w[g] = self._normal(sq, lin)
        # update_w, :231-314
        zw = None
        if fuse:
            zw = self._z(D)  # (drawn here: the reference's stream order)
        elif self.fit_linear:
            c.sweep_w(self.alpha, self.lam_w, self.mu_w, self._z(D))
        else:
            c.zero_w()
        # update_lambda_V / update_mu_V, :202-216 (factor outer, group inner)
        if K:
            s, ssd = c.group_stats_V(self.mu_V)
            for f in range(K):
                for g in range(G):
                    self.lam_V[g, f] = self._gamma((self.a0 + self.n_g[g]) / 2, 2.0 / (self.b0 + ssd[g, f]))
            for f in range(K):
                for g in range(G):
                    sq = self.lam_V[g, f] * (self.g0 + self.n_g[g])
                    lin = (self.g0 * self.m0 + s[g, f]) * self.lam_V[g, f]
                    self.mu_V[g, f] = self._normal(sq, lin)
            # update_V, :316-486
            if fuse:
                c.sweep_wV(self.alpha, e_shift, self.lam_w, self.mu_w, zw, 0, K, self.lam_V, self.mu_V, self._z(K * D))
            else:
                c.sweep_V(0, K, self.alpha, self.lam_V, self.mu_V, self._z(K * D))
        # update_e, :493-497
        if before_update_e is not None:
            before_update_e()
        c.update_e_regression()

    def hyper(self):
        return dict(alpha=self.alpha, mu_w=self.mu_w.copy(), lambda_w=self.lam_w.copy(), mu_V=self.mu_V.copy(),
                    lambda_V=self.lam_V.copy())
