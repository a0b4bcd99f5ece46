// mfm_tasks.hpp -- probit classification and ordered-probit pieces of update_e
// (FMTrainer.hpp:498-521, util.hpp:15-78, OProbitSampler.hpp). Included by mfm_hip.hip.
//
// Random numbers: the reference draws the latent z_t row after row from its single mt19937 inside
// data-dependent rejection loops, which has no parallel equivalent. Here every row owns a
// counter-based Philox4x32-10 stream keyed by (seed, draw_index, row), so results are reproducible
// for a seed and independent of the launch geometry; parity with the reference is distributional.
#pragma once

namespace mfm {

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
    const uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += W0;
    k.y += W1;
  }
  return c;
}

struct RowRng {
  uint2 key;
  uint32_t row, d0, d1, n;
  __device__ RowRng(uint64_t seed, uint64_t draw, uint32_t row_)
      : key(make_uint2((uint32_t)seed, (uint32_t)(seed >> 32))), row(row_), d0((uint32_t)draw), d1((uint32_t)(draw >> 32)), n(0) {}
  // two uniforms in (0, 1)
  __device__ __forceinline__ double2 next2() {
    const uint4 r = philox4x32_10(make_uint4(row, n++, d0, d1), key);
    const double a = ((double)(((uint64_t)r.x << 21) | (r.y >> 11)) + 0.5) * (1.0 / 9007199254740992.0);
    const double b = ((double)(((uint64_t)r.z << 21) | (r.w >> 11)) + 0.5) * (1.0 / 9007199254740992.0);
    return make_double2(a, b);
  }
};

constexpr int TN_MAX_TRIES = 1 << 14;

// util.hpp:15-37 (Robert 2009, Prop. 2.3): z ~ N(0,1) | z > mu_minus
__device__ __forceinline__ double tn_left(RowRng &g, double mu_minus) {
  if (mu_minus < 0) {
    for (int it = 0; it < TN_MAX_TRIES; it++) {
      const double2 u = g.next2();
      const double r = sqrt(-2.0 * log(u.x));
      double s, c;
      sincospi(2.0 * u.y, &s, &c);
      if (r * c > mu_minus) return r * c;
      if (r * s > mu_minus) return r * s;
    }
    return 0.0;
  }
  const double alpha_star = (mu_minus + sqrt(mu_minus * mu_minus + 4)) / 2;
  for (int it = 0; it < TN_MAX_TRIES; it++) {
    const double2 u = g.next2();
    const double z = -log(u.x) / alpha_star + mu_minus;
    const double rho = exp(-(z - alpha_star) * (z - alpha_star) / 2);
    if (u.y < rho) return z;
  }
  return mu_minus;
}
__device__ __forceinline__ double tn_right(RowRng &g, double mu_plus) { return -tn_left(g, -mu_plus); }  // util.hpp:68-71
// util.hpp:39-60
__device__ __forceinline__ double tn_twoside(RowRng &g, double mu_minus, double mu_plus) {
  for (int it = 0; it < TN_MAX_TRIES; it++) {
    const double2 u = g.next2();
    const double z = mu_minus + (mu_plus - mu_minus) * u.x;
    double rho;
    if (mu_minus <= 0 && mu_plus >= 0)
      rho = exp(-z * z / 2);
    else if (mu_plus < 0)
      rho = exp((mu_plus * mu_plus - z * z) / 2);
    else
      rho = exp((mu_minus * mu_minus - z * z) / 2);
    if (u.y < rho) return z;
  }
  return 0.5 * (mu_minus + mu_plus);
}

// FMTrainer.hpp:498-512: eq.x holds the score on entry, e = score - z on exit
__global__ __launch_bounds__(WG) void k_tn_classification(double2 *__restrict__ eq, const double *__restrict__ y, int64_t N,
                                                          uint64_t seed, uint64_t draw, int64_t row_offset) {
  const int64_t t = (int64_t)blockIdx.x * WG + threadIdx.x;
  if (t >= N) return;
  RowRng g(seed ^ ((uint64_t)((t + row_offset) >> 32) * 0x9E3779B97F4A7C15ull), draw, (uint32_t)(t + row_offset));
  const double pred = eq[t].x;
  double n;
  if (y[t] > 0)
    n = pred + tn_left(g, (0.0 - pred));   // util.hpp:61-66 with std = 1
  else
    n = pred + tn_right(g, (0.0 - pred));  // util.hpp:73-78
  eq[t].x = pred - n;
}

// erfcx, same evaluation as the CPU oracle (libm erfc for small x, Laplace continued fraction beyond)
__device__ __forceinline__ double d_erfcx_pos(double x) {
  if (x < 3.0) return exp(x * x) * erfc(x);
  if (x > 5e7) return 0.5641895835477562869 / x;
  // the n-th convergent of x + (1/2)/(x + (2/2)/(x + (3/2)/(x + ...))) by the forward recurrence of its numerators and
  // denominators (all terms positive: no cancellation; x^n stays far inside the double range for these n): two FMAs per
  // level and ONE division, instead of a division per level -- in a wavefront one row in the tail makes all 64 lanes walk
  // the loop
  const int n = (x < 5) ? 90 : (x < 10 ? 50 : 25);
  double a1 = 1.0, a0 = x, b1 = 0.0, b0 = 1.0, hk = 0.0;
  for (int k = 1; k <= n; k++) {
    hk += 0.5;
    const double a = __builtin_fma(x, a0, hk * a1), b = __builtin_fma(x, b0, hk * b1);
    a1 = a0;
    a0 = a;
    b1 = b0;
    b0 = b;
  }
  return 0.5641895835477562869 * b0 / a0;
}
__device__ __forceinline__ double d_erfcx(double x) {
  if (x >= 0) return d_erfcx_pos(x);
  if (x < -26.7) return INFINITY;
  return 2 * exp(x * x) - d_erfcx_pos(-x);
}

constexpr int OPROBIT_LDS_CLASS = 32;  // up to here the per-thread accumulators live in LDS; beyond, in global memory
constexpr int OPROBIT_SLOTS = 6;  // ll, d_hi, d_lo, h_hi, h_lo, h_off per label
constexpr int OPROBIT_BLOCKS = 512;

// OProbitSampler.hpp:402-413 with safe_lcdf / safe_lccdf / safe_ldiff (:111-236), accumulated per
// label: slot 1/3 belong to cutpoint index `label`, slot 2/4 to `label - 1`, slot 5 is the
// off-diagonal (label, label-1).
// Accumulation: every thread owns a private set of n_class x 6 accumulators in LDS, laid out [accumulator][thread]
// (conflict-free, no atomics: the sums do not depend on the execution order); at the end accumulator i is summed
// over the threads in thread order by thread i. blockDim.x = 256 / 128 / 64 for n_class <= 5 / 10 / 32 (dynamic
// LDS n_class * 6 * blockDim.x doubles). The kernel is bound by the fp64 erfcx / exp / log chains, not by the
// accumulation (a wave-level reduction per label measured 2x slower).
// More than OPROBIT_LDS_CLASS classes (OProbitSampler.hpp:36-46 has no bound): the same private accumulators, in global
// memory (gacc: [blocks][n_class * 6][blockDim.x], zeroed by the kernel itself).
__global__ __launch_bounds__(WG) void k_oprobit_eval(const double2 *__restrict__ eq, const double *__restrict__ y,
                                                     const int32_t *__restrict__ rows, int64_t n_rows, int n_class,
                                                     const double *__restrict__ gamma, int want_h,
                                                     double *__restrict__ partial, double *__restrict__ gacc) {
  extern __shared__ double oprobit_lds[];  // gam[n_class rounded up to even], then [n_class * OPROBIT_SLOTS][blockDim.x]
  double *gam = oprobit_lds;
  const int NT = blockDim.x, tid = threadIdx.x;
  double *acc = gacc ? gacc + (size_t)blockIdx.x * n_class * OPROBIT_SLOTS * NT : oprobit_lds + ((n_class + 1) & ~1);
  for (int i = 0; i < n_class * OPROBIT_SLOTS; i++) acc[i * NT + tid] = 0.0;
  for (int i = tid; i < n_class - 1; i += NT) gam[i] = gamma[i];
  __syncthreads();
  const double SQRT2 = 1.4142135623730951, SQRT2PI = 1.4142135623730951 * 1.7724538509055159, PI = 3.141592653589793;
  const double C2 = 2 / SQRT2PI, INV_PI = 1 / PI;  // (one reciprocal of den per row instead of three to five fp64 divisions)
  for (int64_t p = (int64_t)blockIdx.x * NT + tid; p < n_rows; p += (int64_t)gridDim.x * NT) {
    const int64_t t = rows ? rows[p] : p;
    const int label = (int)y[t];
    const double sc = eq[t].x;
    // One evaluation for every row of the wavefront instead of seven divergent branches (each with its own erf / erfcx / exp /
    // log chains: a wavefront of mixed labels used to execute nearly all of them). With x = gamma_l - score (upper bound; none
    // for the last label) and yv = gamma_{l-1} - score (lower bound; none for label 0), E(t) = erfcx(|t| / sqrt 2) and
    // g(t) = exp(-t^2 / 2) (1 - erf(|t| / sqrt 2) = g E), the three regimes of safe_ldiff (:111-181) -- and safe_lcdf /
    // safe_lccdf (:183-236) as its limits yv = -inf / x = +inf -- are
    //   A (yv > 0):       2 (Phi(x) - Phi(yv)) = g(yv) [E(yv) - ef E(x)],  ef = exp((yv^2 - x^2) / 2)     weights (ef, 1)
    //   B (x < 0):                            = g(x)  [E(x) - ef E(yv)],  ef = exp((x^2 - yv^2) / 2)     weights (1, ef)
    //   C (yv <= 0 <= x):                     = 2 - g(x) E(x) - g(yv) E(yv)                               weights (g(x), g(yv))
    // with the common tail  ll = L + log(den / 2),  d_hi = C2 f_hi / den,  d_lo = -C2 f_lo / den  and the Hessian terms below
    // ((f_hi, f_lo) = the weights, L = -yv^2/2, -x^2/2, 0). A and B are the reference's own expressions; C and the one-sided
    // labels replace erf(t / sqrt 2) by 1 - g E (absolute error ~1e-16 in den). Heavy calls per row: 2 erfcx, 2 exp, 1 log.
    const bool has_hi = label < n_class - 1, has_lo = label > 0;
    const double x = has_hi ? gam[label] - sc : 0.0, yv = has_lo ? gam[label - 1] - sc : 0.0;
    const bool cA = has_lo && yv > 0, cB = !cA && has_hi && x < 0, cC = !cA && !cB;
    double Ex = d_erfcx_pos(has_hi ? fabs(x) / SQRT2 : 1.0), Ey = d_erfcx_pos(has_lo ? fabs(yv) / SQRT2 : 1.0);
    Ex = has_hi ? Ex : 0.0;
    Ey = has_lo ? Ey : 0.0;
    const double hx = x * x / 2, hy = yv * yv / 2;
    const double a1 = cA ? hy - hx : (cB ? 0.0 : -hx), a2 = cA ? 0.0 : (cB ? hx - hy : -hy);
    double f_hi = exp(has_hi ? a1 : 0.0), f_lo = exp(has_lo ? a2 : 0.0);
    f_hi = has_hi ? f_hi : 0.0;
    f_lo = has_lo ? f_lo : 0.0;
    const double pa = f_hi * Ex, pb = f_lo * Ey;
    double den = cC ? 2.0 - (pa + pb) : (cA ? pb - pa : pa - pb);
    // A narrow interval around the score (yv <= 0 <= x, both small): 2 - (gE(x) + gE(yv)) cancels to an absolute error of ~1e-16,
    // i.e. a RELATIVE error of 1e-16 / den in ll, the gradient and (squared) the Hessian. There the reference's own form
    // erf(x / sqrt 2) - erf(yv / sqrt 2) (OProbitSampler.hpp:185-196) keeps full relative accuracy.
    if (cC && has_hi && has_lo && fmax(fabs(x), fabs(yv)) < 0.5) den = erf(x / SQRT2) - erf(yv / SQRT2);
    const double id = 1.0 / den, w = INV_PI * id * id;
    double ll = cA ? -hy : (cB ? -hx : 0.0);
    ll += log(den / 2);
    const double d_hi = C2 * f_hi * id, d_lo = -(C2 * f_lo * id);
    double h_hi = 0, h_lo = 0, h_off = 0;
    if (want_h) {
      h_hi = -(SQRT2PI * x * den * f_hi + 2 * (f_hi * f_hi)) * w;
      h_lo = (SQRT2PI * yv * den * f_lo - 2 * (f_lo * f_lo)) * w;
      h_off = 2 * f_hi * f_lo * w;
    }
    double *a = acc + (size_t)label * OPROBIT_SLOTS * NT + tid;
    a[0 * NT] += ll;
    a[1 * NT] += d_hi;
    a[2 * NT] += d_lo;
    if (want_h) {
      a[3 * NT] += h_hi;
      a[4 * NT] += h_lo;
      a[5 * NT] += h_off;
    }
  }
  __syncthreads();
  for (int i = tid; i < n_class * OPROBIT_SLOTS; i += NT) {
    double sum = 0.0;
    for (int k = 0; k < NT; k++) sum += acc[i * NT + k];  // thread order: deterministic
    partial[(int64_t)blockIdx.x * n_class * OPROBIT_SLOTS + i] = sum;
  }
}

// sample_z_given_cutpoint, OProbitSampler.hpp:238-272 (deviation = 1)
__global__ __launch_bounds__(WG) void k_oprobit_sample_z(double2 *__restrict__ eq, const double *__restrict__ y,
                                                         const int32_t *__restrict__ rows, int64_t n_rows, int n_class,
                                                         const double *__restrict__ gamma, uint64_t seed, uint64_t draw,
                                                         int64_t row_offset) {
  const int64_t p = (int64_t)blockIdx.x * WG + threadIdx.x;
  if (p >= n_rows) return;
  const int64_t t = rows ? rows[p] : p;
  RowRng g(seed ^ ((uint64_t)((t + row_offset) >> 32) * 0x9E3779B97F4A7C15ull), draw, (uint32_t)(t + row_offset));
  const int cls = (int)y[t];
  const double pred = eq[t].x;
  double z;
  if (cls == 0)
    z = tn_right(g, gamma[0] - pred) + pred;
  else if (cls == n_class - 1)
    z = tn_left(g, gamma[n_class - 2] - pred) + pred;
  else
    z = tn_twoside(g, gamma[cls - 1] - pred, gamma[cls] - pred) + pred;
  eq[t].x = pred - z;
}

// ---- kernel-level test hooks (include/myfm_hip.h "test hooks"): the device erfcx and the truncated-normal samplers on
// caller-supplied arguments, so that the parity tests can hold them against Faddeeva.cc / util.hpp directly
__global__ void k_test_erfcx(const double *__restrict__ x, double *__restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = d_erfcx(x[i]);
}
// kind 0: left (z > lo), 1: right (z < hi), 2: two-sided (lo < z < hi); draw i uses the stream of row i
__global__ void k_test_tn(int kind, double lo, double hi, uint64_t seed, uint64_t draw, int64_t n, double *__restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  RowRng g(seed ^ ((uint64_t)(t >> 32) * 0x9E3779B97F4A7C15ull), draw, (uint32_t)t);
  out[t] = kind == 0 ? tn_left(g, lo) : (kind == 1 ? tn_right(g, hi) : tn_twoside(g, lo, hi));
}

}  // namespace mfm

extern "C" {

static int test_hook_run(int device, int64_t n, const double *in, double *out, int kind, double lo, double hi, uint64_t seed,
                         uint64_t draw) {
  try {
    if (mfm_device_count() <= 0) throw Error(MFM_ERR_DEVICE, "no HIP device is visible (no CPU fallback)");
    if (n < 0) throw Error(MFM_ERR_INVALID, "negative count");
    MFM_HIP_CHECK(hipSetDevice(device));
    if (n == 0) return MFM_OK;
    DevBuf<double> din, dout;
    dout.alloc((size_t)n);
    if (in) {
      din.alloc((size_t)n);
      MFM_HIP_CHECK(hipMemcpy(din.p, in, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
      hipLaunchKernelGGL(k_test_erfcx, dim3(cdiv(n, 256)), dim3(256), 0, 0, din.p, dout.p, n);
    } else {
      hipLaunchKernelGGL(k_test_tn, dim3(cdiv(n, 256)), dim3(256), 0, 0, kind, lo, hi, seed, draw, n, dout.p);
    }
    MFM_HIP_CHECK(hipGetLastError());
    MFM_HIP_CHECK(hipMemcpy(out, dout.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    return MFM_OK;
  } catch (const mfm::Error &ex) {
    g_global_error = ex.what();
    return ex.code;
  } catch (const std::exception &ex) {
    g_global_error = ex.what();
    return MFM_ERR_RUNTIME;
  }
}

int mfm_test_erfcx(int device, const double *x, int64_t n, double *out) {
  return test_hook_run(device, n, x, out, 0, 0.0, 0.0, 0, 0);
}

int mfm_test_truncated_normal(int device, int32_t kind, double lo, double hi, uint64_t seed, uint64_t draw_index, int64_t n,
                              double *out) {
  if (kind < 0 || kind > 2) {
    g_global_error = "kind must be 0 (left), 1 (right) or 2 (two-sided)";
    return MFM_ERR_INVALID;
  }
  return test_hook_run(device, n, nullptr, out, kind, lo, hi, seed, draw_index);
}

int mfm_update_e_classification(mfm_ctx *ctx, uint64_t seed, uint64_t draw_index) {
  MFM_TRY(ctx)
  ctx->need_final();
  score_train(ctx, false);
  if (ctx->N) {
    TimedLaunch t(ctx->timing, ctx->stream, KC_TN_SAMPLE, 24.0 * ctx->N);
    hipLaunchKernelGGL(k_tn_classification, dim3(cdiv(ctx->N, WG)), dim3(WG), 0, ctx->stream, ctx->eq_rows(), ctx->y.p, ctx->N,
                       seed, draw_index, ctx->row_offset);
    MFM_HIP_CHECK(hipGetLastError());
  }
  MFM_CATCH(ctx)
}

int mfm_oprobit_add_group(mfm_ctx *ctx, int32_t n_class, const int64_t *rows, int64_t n_rows, int32_t *group) {
  MFM_TRY(ctx)
  ctx->need_final();
  if (n_class < 2) throw Error(MFM_ERR_INVALID, "ordered probit needs at least 2 classes");
  std::unique_ptr<mfm_ctx::OGroup> g(new mfm_ctx::OGroup());
  g->n_class = n_class;
  if (rows) {
    std::vector<int32_t> r32((size_t)n_rows);
    for (int64_t i = 0; i < n_rows; i++) {
      if (rows[i] < 0 || rows[i] >= ctx->N) throw Error(MFM_ERR_INVALID, "out of range for cutpoint group config.");
      r32[i] = (int32_t)rows[i];
    }
    g->rows.upload(r32);
    g->n_rows = n_rows;
  } else {
    g->n_rows = ctx->N;
  }
  *group = (int32_t)ctx->ogroups.size();
  ctx->ogroups.push_back(std::move(g));
  {  // partial sums per block + the staging copy of the cutpoints, sized for the largest group
    int cmax = 0;
    for (auto &og : ctx->ogroups) cmax = std::max(cmax, og->n_class);
    const size_t need = (size_t)OPROBIT_BLOCKS * cmax * OPROBIT_SLOTS + cmax;
    if (ctx->opartial.n < need) {
      MFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
      ctx->opartial.alloc(need);
    }
    ctx->opartial_cmax = cmax;
  }
  MFM_CATCH(ctx)
}

int mfm_oprobit_eval(mfm_ctx *ctx, int32_t group, const double *gamma, double *ll, double *dgamma, double *H) {
  MFM_TRY(ctx)
  ctx->need_final();
  materialize_e(ctx);
  if (group < 0 || group >= (int)ctx->ogroups.size()) throw Error(MFM_ERR_INVALID, "bad cutpoint group");
  mfm_ctx::OGroup &g = *ctx->ogroups[group];
  hipStream_t s = ctx->stream;
  const int C = g.n_class, m = C - 1;
  double *dgam = ctx->opartial.p + (size_t)OPROBIT_BLOCKS * ctx->opartial_cmax * OPROBIT_SLOTS;
  ctx->ring.upload(dgam, gamma, (size_t)m * sizeof(double), s);
  const int nt = C <= 5 ? 256 : (C <= 10 ? 128 : 64);
  const bool in_lds = C <= OPROBIT_LDS_CLASS;
  const size_t lds = ((size_t)((C + 1) & ~1) + (in_lds ? (size_t)C * OPROBIT_SLOTS * nt : 0)) * sizeof(double);
  const int nb_cap = in_lds ? OPROBIT_BLOCKS : 128;  // (global accumulators: 3 KB per class and block)
  if (!in_lds) {
    const size_t need = (size_t)nb_cap * C * OPROBIT_SLOTS * nt;
    if (ctx->oacc.n < need) ctx->oacc.alloc(need);
  }
  {
    static DeviceOnce raised;
    if (raised.need() && lds > 64 * 1024) {
      MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_oprobit_eval, hipFuncAttributeMaxDynamicSharedMemorySize, 104 * 1024));
      raised.mark();
    }
  }
  const int nb = (int)std::min<int64_t>(nb_cap, std::max<int64_t>(1, cdiv(g.n_rows, nt)));
  {
    TimedLaunch t(ctx->timing, s, KC_OPROBIT_EVAL, 16.0 * g.n_rows);
    hipLaunchKernelGGL(k_oprobit_eval, dim3(nb), dim3(nt), lds, s, ctx->eq_rows(), ctx->y.p, g.rows.p, g.n_rows, C, dgam,
                       H ? 1 : 0, ctx->opartial.p, in_lds ? (double *)nullptr : ctx->oacc.p);
    MFM_HIP_CHECK(hipGetLastError());
  }
  const size_t cnt = (size_t)nb * C * OPROBIT_SLOTS;
  double *h = (double *)ctx->readback((cnt + 1) / 2 + 1);
  MFM_HIP_CHECK(hipMemcpyAsync(h, ctx->opartial.p, cnt * sizeof(double), hipMemcpyDeviceToHost, s));
  MFM_HIP_CHECK(hipStreamSynchronize(s));
  std::vector<double> acc((size_t)C * OPROBIT_SLOTS, 0.0);
  for (int b = 0; b < nb; b++)
    for (int i = 0; i < C * OPROBIT_SLOTS; i++) acc[i] += h[(size_t)b * C * OPROBIT_SLOTS + i];
  if (ctx->comm.active()) {  // row-sharded: the likelihood terms are sums over all ranks' rows
    ctx->ring.upload(ctx->opartial.p, acc.data(), acc.size() * sizeof(double), s);
    ctx->comm.allreduce(ctx->opartial.p, (int64_t)acc.size());
    MFM_HIP_CHECK(hipMemcpyAsync(h, ctx->opartial.p, acc.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    MFM_HIP_CHECK(hipStreamSynchronize(s));
    for (size_t i = 0; i < acc.size(); i++) acc[i] = h[i];
  }
  *ll = 0;
  for (int k = 0; k < m; k++) dgamma[k] = 0;
  if (H)
    for (int k = 0; k < m * m; k++) H[k] = 0;
  for (int l = 0; l < C; l++) {
    const double *a = acc.data() + (size_t)l * OPROBIT_SLOTS;
    *ll += a[0];
    if (l < m) dgamma[l] += a[1];
    if (l >= 1) dgamma[l - 1] += a[2];
    if (H) {
      if (l < m) H[(size_t)l * m + l] += a[3];
      if (l >= 1) H[(size_t)(l - 1) * m + (l - 1)] += a[4];
      if (l >= 1 && l < m) {
        H[(size_t)l * m + (l - 1)] += a[5];
        H[(size_t)(l - 1) * m + l] += a[5];
      }
    }
  }
  MFM_CATCH(ctx)
}

int mfm_oprobit_sample_z(mfm_ctx *ctx, int32_t group, const double *gamma, uint64_t seed, uint64_t draw_index) {
  MFM_TRY(ctx)
  ctx->need_final();
  materialize_e(ctx);
  if (group < 0 || group >= (int)ctx->ogroups.size()) throw Error(MFM_ERR_INVALID, "bad cutpoint group");
  mfm_ctx::OGroup &g = *ctx->ogroups[group];
  hipStream_t s = ctx->stream;
  double *dgam = ctx->opartial.p + (size_t)OPROBIT_BLOCKS * ctx->opartial_cmax * OPROBIT_SLOTS;
  ctx->ring.upload(dgam, gamma, (size_t)(g.n_class - 1) * sizeof(double), s);
  if (g.n_rows) {
    TimedLaunch t(ctx->timing, s, KC_TN_SAMPLE, 24.0 * g.n_rows);
    hipLaunchKernelGGL(k_oprobit_sample_z, dim3(cdiv(g.n_rows, WG)), dim3(WG), 0, s, ctx->eq_rows(), ctx->y.p, g.rows.p,
                       g.n_rows, g.n_class, dgam, seed, draw_index, ctx->row_offset);
    MFM_HIP_CHECK(hipGetLastError());
  }
  MFM_CATCH(ctx)
}

}  // extern "C"
