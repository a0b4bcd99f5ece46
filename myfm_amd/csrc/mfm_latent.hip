// mfm_latent.hip -- the latent draws of probit classification / ordered probit on the reference's OWN random stream, evaluated in
// parallel on the device (FMTrainer.hpp:498-521, OProbitSampler.hpp:238-272, util.hpp:15-60). See mfm_latent_api.hpp for the
// formulation (one monotone lattice path over (row, quad); coalescing flows). Passes:
//   k_lat_rows     per row of the group: the standardised truncation bounds as a 16-byte record, the acceptance probability p of
//                  one quad (closed form) -> expected quads 1/p and variance (1 - p)/p^2, block sums
//   k_lat_scan     prefix sums of the blocks (expected position of every 1024-th row, its variance)
//   k_lat_quads    per quad j: (u1, -log u1 - 1, log u2, the larger of the polar pair's two normals) for the decisions and
//                  (first normal, -log u1) for the accepted values, from the ring of MT19937 outputs
//   k_lat_windows  per chunk c of Lq quads: the rows [lo, hi] that can be in service at quad c Lq (mean +- k sigma)
//   k_lat_round_ring  every live walker of every chunk walks R quads: t += A(t, j)
//   k_lat_compact  per chunk: walkers that met (equal t, they are sorted) are merged; at sub-chunk boundaries the list
//                  (first entering row of the merged range, current row) is kept as a snapshot
//   k_lat_resident the same two steps for the rest of the chunk in one launch once a chunk's walkers fit a workgroup
//   k_lat_resolve  the true entering row of every chunk: T(c + 1) = map_c(T(c)), T(0) = 0
//   k_lat_final    one thread per sub-chunk walks the ONE true path of its quads from the snapshot's row and writes
//                  e = score - z for the rows it accepts; the thread that serves the last row moves the stream's position
// Acceptance is decided in the log domain on the quad table (u2 < exp(a) <=> log u2 < a) and, inside a relative band of 1e-9
// around equality, by the reference's own expression (exp) -- the same function in the flows and in the final pass, so the two
// cannot disagree; against libm the decision can differ only where exp / log differ in the last place (as for the sweep
// normals, mfm_rng.hpp).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "mfm_common.hpp"
#include "mfm_latent_api.hpp"

namespace mfm {

namespace {

constexpr int LAT_RB = 256;      // rows per block of the row pass (its sums anchor the windows: a search ends with a scan of <= LAT_RB rows)
constexpr int LAT_TILE = 256;    // walkers per workgroup of k_lat_round_ring
constexpr int LAT_RES_NT = 512;  // threads (= walkers at most) of the resident kernel
constexpr int LAT_QW = 4;        // doubles per quad record of the flows
constexpr int LAT_RPAD = 16;     // records after the last row that accept nothing (a walker requests up to 8 consecutive records)

struct LatStatus {
  int32_t fail;        // first failure code (0: none)
  int32_t fail_chunk;
  int64_t end_quads;   // quads consumed by the whole draw (-1: the path did not end inside the prepared quads)
  int64_t walkers;     // sum of the windows
  int64_t snap_need;   // snapshot entries the caps add up to
  double total_m, total_v;
};

__device__ __forceinline__ void lat_fail(LatStatus *st, int code, int chunk) {
  if (atomicCAS(&st->fail, 0, code) == 0) st->fail_chunk = chunk;
}

// ---- acceptance probability of one quad (anchors the windows: must be unbiased to ~1e-7) ----
__device__ __forceinline__ double lat_erfcx(double x) {  // x >= 0
  if (x < 25.0) return exp(x * x) * erfc(x);
  const double r = 1.0 / (x * x);
  return 0.56418958354775628695 / x * (1.0 + r * (-0.5 + r * (0.75 + r * (-1.875 + r * 6.5625))));
}
__device__ __forceinline__ double lat_accept_prob(double A, double B) {
  const double SQRT1_2 = 0.70710678118654752440, SQRT_2PI = 2.50662827463100050242;
  double p;
  if (B != B) {  // polar attempt + tail test: a pair exists with probability pi / 4, both normals fail with Phi(mu)^2
    const double Phi = 0.5 * erfc(-A * SQRT1_2);
    p = 0.78539816339744830962 * (1.0 - Phi * Phi);
  } else if (A > B) {  // translated-exponential proposal (Robert 1995): alpha = A, mu = B
    const double d = A - B;
    p = A * 1.25331413731550025121 * exp(-0.5 * d * d) * lat_erfcx(B * SQRT1_2);
  } else {  // uniform proposal on [a, b]
    const double w = B - A;
    if (!(w > 1e-12)) return 1.0;
    if (A <= 0.0 && B >= 0.0) {
      p = SQRT_2PI * 0.5 * (erf(B * SQRT1_2) - erf(A * SQRT1_2)) / w;
    } else {
      const double x1 = (B < 0.0 ? -B : A) * SQRT1_2, x2 = (B < 0.0 ? -A : B) * SQRT1_2;
      p = SQRT_2PI * 0.5 * (lat_erfcx(x1) - exp((x1 - x2) * (x1 + x2)) * lat_erfcx(x2)) / w;
    }
  }
  if (!(p > 1e-12)) p = 1e-12;
  if (p > 1.0) p = 1.0;
  return p;
}

// ---- the row's standardised bounds (OProbitSampler.hpp:246-270, FMTrainer.hpp:503-511 with std = 1) ----
// record (A, B):  one-sided, mu < 0: (mu, NaN);  one-sided, mu >= 0: (alpha*, mu), alpha* > mu;  two-sided: (a, b), a <= b.
// sgn: the draw is sgn * value (right truncation = -left(-mu), util.hpp:70-73); value + score is the latent z.
__device__ __forceinline__ double2 lat_record(double pred, double yv, int n_class, const double *__restrict__ gamma, int *sgn) {
  double mu;
  if (n_class == 0) {
    if (yv > 0) {
      mu = (0.0 - pred) / 1.0;
      *sgn = 1;
    } else {
      mu = -((0.0 - pred) / 1.0);
      *sgn = -1;
    }
  } else {
    const int cls = (int)yv;
    if (cls == 0) {
      mu = -((gamma[0] - pred) / 1.0);
      *sgn = -1;
    } else if (cls == n_class - 1) {
      mu = (gamma[n_class - 2] - pred) / 1.0;
      *sgn = 1;
    } else {
      *sgn = 1;
      return make_double2((gamma[cls - 1] - pred) / 1.0, (gamma[cls] - pred) / 1.0);
    }
  }
  if (mu < 0) return make_double2(mu, __builtin_nan(""));
  const double alpha = (mu + sqrt(mu * mu + 4)) / 2;
  return make_double2(alpha, mu);
}

// ---- the walkers' form of a row and of a quad ---------------------------------------------------------------------------
// A walker decides "does my row accept this quad" ~10^10 times per draw of 5 10^7 rows: the decision is ONE straight-line expression
// for all three regimes, on a 16-byte row record (wA, wB) and a 32-byte quad record (xT, xE, l2, m):
//   N  (wA, wB) = (mu, NaN)            accepted  <=>  m > mu,                        m = max of the polar pair's two normals (NaN: no pair)
//   T  (wA, wB) = (a, b)               accepted  <=>  l2 < (c - (a + (b - a) xT)^2) / 2,   c = max(a, 0)^2 + min(b, 0)^2 (util.hpp:48-55)
//   E  (wA, wB) = (-0.0, alpha* - mu)  accepted  <=>  l2 < -((alpha* - mu) xE)^2 / 2,      xE = -log(u1) - 1
// E is T's formula with a = -0 (c = 0): z - alpha* = -log(u1) / alpha* + mu - alpha* = (alpha* - mu)(-log(u1) - 1) because
// 1 / alpha* = alpha* - mu = (sqrt(mu^2 + 4) - mu) / 2 -- an identity of real numbers: in doubles the two sides differ by
// ~1e-16 alpha*^2 relative, which the band below absorbs (rows with alpha* > 1000, scores a thousand standard deviations on the
// wrong side of their class, send the whole draw to the sequential loop). A two-sided row whose lower bound is exactly -0.0 is
// stored with +0.0. l2 = log u2: u2 < exp(x) <=> log u2 < x. Within a relative band of 4e-9 around equality (and for inf - inf) the
// reference's own expression decides (lat_accept_exact: exp, the division, the original record).
struct LatQ {
  double xT, xE, l2, m;
};
__device__ __forceinline__ bool lat_decide(double wA, double wB, const LatQ &q, bool &band) {
  const bool isN = wB != wB;
  const bool isE = __double_as_longlong(wA) == (long long)0x8000000000000000ull;
  const double X = isE ? q.xE : q.xT;
  const double w = wB - wA;
  const double zz = __builtin_fma(w, X, wA);
  const double ap = fmax(wA, 0.0), bn = fmin(wB, 0.0);
  const double c = __builtin_fma(ap, ap, bn * bn);
  const double arg = 0.5 * __builtin_fma(-zz, zz, c);
  const double diff = q.l2 - arg;
  band = !isN & !(fabs(diff) > __builtin_fma(fabs(arg), 4e-9, 4e-9));
  return isN ? (q.m > wA) : (diff < 0.0);
}
__device__ __forceinline__ double2 lat_walker_record(double A, double B, bool *too_wide) {
  if (B != B) return make_double2(A, B);
  if (A > B) {
    if (!(A < 1e3)) *too_wide = true;
    return make_double2(-0.0, A - B);
  }
  return make_double2(A == 0.0 ? 0.0 : A, B);
}

// u2 < rho by the reference's own expressions (util.hpp:29-35, :46-59) on the ORIGINAL record (A, B)
__device__ __forceinline__ bool lat_accept_exact(double A, double B, double u1, double nl1, double u2) {
  double rho;
  if (A > B) {
    const double z = nl1 / A + B;
    rho = exp(-(z - A) * (z - A) / 2);
  } else {
    const double z = u1 * (B - A) + A;
    if (A <= 0.0 && B >= 0.0)
      rho = exp(-z * z / 2);
    else if (B < 0.0)
      rho = exp((B * B - z * z) / 2);
    else
      rho = exp((A * A - z * z) / 2);
  }
  return u2 < rho;
}
__device__ __forceinline__ double lat_u2(const uint32_t *__restrict__ raw, uint64_t mask, uint64_t p0, int64_t j) {
  const uint64_t b = p0 + 4ull * (uint64_t)j;
  return canonical(mt_temper(raw[(b + 2) & mask]), mt_temper(raw[(b + 3) & mask]));
}
// what every pass needs to decide one (row, quad) pair, the band included
struct LatCtx {
  const double2 *__restrict__ rec;   // original records (exact path, values)
  const double2 *__restrict__ qx;    // per quad (n1, -log u1)
  const uint32_t *__restrict__ raw;
  uint64_t mask, p0;
};
__device__ __forceinline__ bool lat_accept(const LatCtx &cx, int32_t t, double wA, double wB, const LatQ &q, int64_t j) {
  bool band;
  bool acc = lat_decide(wA, wB, q, band);
  if (band) {  // one lane in 10^8 steps
    const double2 ab = cx.rec[t];
    acc = lat_accept_exact(ab.x, ab.y, q.xT, cx.qx[j].y, lat_u2(cx.raw, cx.mask, cx.p0, j));
  }
  return acc;
}
// the accepted value (util.hpp:21-23, :29, :45) from the original record; n1 / nl1 = the quad's first normal / -log u1
__device__ __forceinline__ double lat_value(double A, double B, const LatQ &q, double n1, double nl1) {
  if (B != B) return n1 > A ? n1 : q.m;  // (the first normal when it passes, else the second -- which then is the larger one)
  if (A > B) return nl1 / A + B;
  return q.xT * (B - A) + A;
}
__device__ __forceinline__ LatQ lat_load_quad(const double *__restrict__ qt, int64_t j) {
  const double *p = qt + (size_t)j * LAT_QW;
  LatQ q;
  q.xT = p[0];
  q.xE = p[1];
  q.l2 = p[2];
  q.m = p[3];
  return q;
}

// ---- row pass ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lat_rows(const int32_t *__restrict__ rows, const double2 *__restrict__ eq,
                                                  const double *__restrict__ y, int64_t n, int n_class,
                                                  const double *__restrict__ gamma, double2 *__restrict__ rec,
                                                  double2 *__restrict__ wrec, float *__restrict__ mf, double *__restrict__ blkM,
                                                  double *__restrict__ blkV, int32_t *__restrict__ bad) {
  __shared__ double s_m[4], s_v[4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  double sm = 0.0, sv = 0.0;
#pragma unroll
  for (int r = 0; r < LAT_RB / 256; r++) {
    const int64_t i = (int64_t)blockIdx.x * LAT_RB + r * 256 + tid;
    if (i < n) {
      const int64_t t = rows ? (int64_t)rows[i] : i;
      int sgn;
      const double2 ab = lat_record(eq[t].x, y[t], n_class, gamma, &sgn);
      rec[i] = ab;
      bool wide = false;
      wrec[i] = lat_walker_record(ab.x, ab.y, &wide);
      if (wide) *bad = 1;
      const double p = lat_accept_prob(ab.x, ab.y);
      const double m = 1.0 / p;
      mf[i] = (float)m;
      sm += m;
      sv += (1.0 - p) * m * m;
    } else if (i < n + LAT_RPAD) {
      rec[i] = wrec[i] = make_double2(__builtin_inf(), __builtin_nan(""));  // the rows after the last one accept nothing
    }
  }
  for (int d = 32; d > 0; d >>= 1) {
    sm += __shfl_down(sm, d, 64);
    sv += __shfl_down(sv, d, 64);
  }
  if (lane == 0) {
    s_m[wid] = sm;
    s_v[wid] = sv;
  }
  __syncthreads();
  if (tid == 0) {
    blkM[blockIdx.x] = (s_m[0] + s_m[1]) + (s_m[2] + s_m[3]);
    blkV[blockIdx.x] = (s_v[0] + s_v[1]) + (s_v[2] + s_v[3]);
  }
}

// exclusive prefix sums of blkM / blkV -> PM / PV [nb + 1]; one workgroup
__global__ __launch_bounds__(1024) void k_lat_scan(const double *__restrict__ blkM, const double *__restrict__ blkV, int64_t nb,
                                                   double *__restrict__ PM, double *__restrict__ PV, LatStatus *__restrict__ st,
                                                   const RngState *__restrict__ rs, uint64_t *__restrict__ pos_out) {
  __shared__ double s_m[1024], s_v[1024];
  const int tid = threadIdx.x;
  const int64_t seg = (nb + 1023) / 1024;
  const int64_t b0 = min(nb, (int64_t)tid * seg), b1 = min(nb, b0 + seg);
  double sm = 0.0, sv = 0.0;
  for (int64_t b = b0; b < b1; b++) {
    sm += blkM[b];
    sv += blkV[b];
  }
  s_m[tid] = sm;
  s_v[tid] = sv;
  __syncthreads();
  if (tid == 0) {  // (1024 additions: nothing next to the row pass)
    double am = 0.0, av = 0.0;
    for (int i = 0; i < 1024; i++) {
      const double m = s_m[i], v = s_v[i];
      s_m[i] = am;
      s_v[i] = av;
      am += m;
      av += v;
    }
    PM[nb] = am;
    PV[nb] = av;
    st->total_m = am;
    st->total_v = av;
    st->fail = 0;
    st->fail_chunk = -1;
    st->end_quads = -1;
    st->walkers = 0;
    st->snap_need = 0;
    pos_out[0] = rs->p_cons;
    pos_out[1] = rs->p_gen;
  }
  __syncthreads();
  double am = s_m[tid], av = s_v[tid];
  for (int64_t b = b0; b < b1; b++) {
    PM[b] = am;
    PV[b] = av;
    am += blkM[b];
    av += blkV[b];
  }
}

// ---- quad table ---------------------------------------------------------------------------------------
// the quad of two canonical variates (u1, u2): what a walker looks at (LatQ) and what only the accepted value needs (n1, -log u1)
__device__ __forceinline__ LatQ lat_make_quad(double u1, double u2, double *n1_out, double *nl1_out) {
  // Marsaglia polar attempt of normal_distribution (random.tcc:1811-1826): returns y * mult first, keeps x * mult
  const double x = 2.0 * u1 - 1.0, yy = 2.0 * u2 - 1.0;
  const double r2 = x * x + yy * yy;
  double n1 = __builtin_nan(""), n2 = __builtin_nan("");
  if (!(r2 > 1.0 || r2 == 0.0)) {
    const double mult = sqrt(-2 * log(r2) / r2);
    n1 = (yy * mult) * 1.0 + 0.0;
    n2 = (x * mult) * 1.0 + 0.0;
  }
  const double nl1 = -log(u1);
  LatQ q;
  q.xT = u1;
  q.xE = nl1 - 1.0;
  q.l2 = log(u2);
  q.m = n1 > n2 ? n1 : n2;  // (NaN when there is no pair)
  *n1_out = n1;
  *nl1_out = nl1;
  return q;
}

// ---- the windows' control variate ---------------------------------------------------------------------------------------
// How far the draw has come after J quads is a sum of J accept / reject decisions; the a-priori windows treat it as a sum of
// independent Bernoulli variables with the rows' own probabilities. But every quad is known before the walk, and much of a
// decision is the QUAD's doing (one polar attempt in five has no pair at all: no one-sided row accepts it). With
//   a(u1, u2) = the fraction of the rows that accept the quad (u1, u2),
// tabulated on a G x G grid over the unit square from a sample of S rows at the cells' centres, the sum over the quads before J of
// tab[cell(quad)] - mean(tab) estimates "accepts so far minus expected" -- its own expectation is EXACTLY zero whatever the table
// holds ((u1, u2) is uniform on the square, the cells have equal area): it can be a better or a worse predictor, never a biased one.
// The first attempt's windows are shifted by it and narrowed to the spread it leaves: rho = residual / Bernoulli variance, the
// sample rows weighted by the quads the path spends on them (1 / p), estimated on the same grid (0.33 for probit classification at
// config 3's shape -- measured on the resolved paths: 0.4 --, 0.55-0.7 for five-class ordered probit). A window that misses the path
// is noticed as before; the second attempt uses the plain windows. Used for probit classification draws of 2^20 rows and more (the
// table, its statistics and the chunk sums cost ~0.15 ms there): walkers 3.8 M -> 2.5 M at config 3's shape, 62.5 -> 68 it/s, second
// attempts 3 in 402 draws as without it (scripts/r06_lat_cv.sh).
constexpr int LAT_CV_G = 128, LAT_CV_S = 256;
struct LatCv {
  double mean, rho;  // mean of the table; variance ratio the first attempt's windows use
  double rho_est;    // ... as estimated
};
// one thread per cell: its centre's quad (kept for the statistics pass) and the fraction of the sample that accepts it
__global__ __launch_bounds__(256) void k_lat_cv_table(const double2 *__restrict__ wrec, int64_t n, float *__restrict__ tab,
                                                      LatQ *__restrict__ cellq) {
  __shared__ double2 s_rec[LAT_CV_S];
  const int S = (int)min((int64_t)LAT_CV_S, n);
  const int64_t stride = n / S;
  for (int k = threadIdx.x; k < S; k += 256) s_rec[k] = wrec[(int64_t)k * stride + stride / 2];
  __syncthreads();
  const int cell = blockIdx.x * 256 + threadIdx.x;
  const int i = cell / LAT_CV_G, j = cell % LAT_CV_G;
  double n1, nl1;
  const LatQ q = lat_make_quad((i + 0.5) / LAT_CV_G, (j + 0.5) / LAT_CV_G, &n1, &nl1);
  cellq[cell] = q;
  int acc = 0;
  for (int k = 0; k < S; k++) {
    bool band;
    acc += lat_decide(s_rec[k].x, s_rec[k].y, q, band) ? 1 : 0;
  }
  tab[cell] = (float)acc / (float)S;
}
// one workgroup per sample row: its acceptance probability on the grid and what the table leaves of its variance
__global__ __launch_bounds__(256) void k_lat_cv_stats(const double2 *__restrict__ wrec, int64_t n, const float *__restrict__ tab,
                                                      const LatQ *__restrict__ cellq, double *__restrict__ row_r,
                                                      double *__restrict__ row_v, LatCv *__restrict__ cv) {
  __shared__ double s_a[256], s_b[256];
  constexpr int NC = LAT_CV_G * LAT_CV_G;
  const int tid = threadIdx.x;
  const int S = (int)min((int64_t)LAT_CV_S, n);
  const int64_t stride = n / S;
  auto reduce2 = [&](double &a, double &b) {  // (fixed tree: the same numbers in every workgroup and every run)
    s_a[tid] = a;
    s_b[tid] = b;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
      if (tid < d) {
        s_a[tid] += s_a[tid + d];
        s_b[tid] += s_b[tid + d];
      }
      __syncthreads();
    }
    a = s_a[0];
    b = s_b[0];
    __syncthreads();
  };
  double tm = 0.0, dummy = 0.0;
  for (int c = tid; c < NC; c += 256) tm += (double)tab[c];
  reduce2(tm, dummy);
  const double tbar = tm / NC;
  if (blockIdx.x == 0 && tid == 0) cv->mean = tbar;
  const double2 r = wrec[(int64_t)blockIdx.x * stride + stride / 2];
  double cnt = 0.0, d2 = 0.0;
  for (int c = tid; c < NC; c += 256) {
    bool band;
    const double a = lat_decide(r.x, r.y, cellq[c], band) ? 1.0 : 0.0;
    const double d = a - (double)tab[c];
    cnt += a;
    d2 += d * d;
  }
  reduce2(cnt, d2);
  if (tid == 0) {
    const double p = cnt / NC;
    const double wgt = 1.0 / fmax(p, 0.02);  // (the path spends 1 / p quads on the row)
    row_v[blockIdx.x] = wgt * p * (1.0 - p);
    row_r[blockIdx.x] = wgt * fmax(0.0, d2 / NC - (p - tbar) * (p - tbar));
  }
}
__global__ __launch_bounds__(256) void k_lat_cv_rho(const double *__restrict__ row_r, const double *__restrict__ row_v, int64_t n,
                                                    double safety, LatCv *__restrict__ cv) {
  __shared__ double s_a[256], s_b[256];
  const int tid = threadIdx.x;
  const int S = (int)min((int64_t)LAT_CV_S, n);
  s_a[tid] = tid < S ? row_r[tid] : 0.0;
  s_b[tid] = tid < S ? row_v[tid] : 0.0;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (tid < d) {
      s_a[tid] += s_a[tid + d];
      s_b[tid] += s_b[tid + d];
    }
    __syncthreads();
  }
  if (tid == 0) {
    const double est = s_b[0] > 0.0 ? s_a[0] / s_b[0] : 1.0;
    cv->rho_est = est;
    cv->rho = fmin(1.0, safety * est + 0.02);  // (never wider than a priori)
  }
}
// the table's sum over a chunk's quads, from the waves' sums (k_lat_quads): fixed order
__global__ __launch_bounds__(256) void k_lat_cv_chunks(const double *__restrict__ wavecv, int64_t waves_per_chunk, int64_t n_waves,
                                                       double *__restrict__ cvsum) {
  __shared__ double s_a[256];
  const int tid = threadIdx.x;
  const int64_t w0 = (int64_t)blockIdx.x * waves_per_chunk, w1 = min(n_waves, w0 + waves_per_chunk);
  double a = 0.0;
  for (int64_t w = w0 + tid; w < w1; w += 256) a += wavecv[w];
  s_a[tid] = a;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (tid < d) s_a[tid] += s_a[tid + d];
    __syncthreads();
  }
  if (tid == 0) cvsum[blockIdx.x] = s_a[0];
}

// ---- quad table ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lat_quads(const RngState *__restrict__ rs, const uint32_t *__restrict__ raw, uint64_t mask,
                                                   int64_t nq, double *__restrict__ qt, double2 *__restrict__ qx,
                                                   const float *__restrict__ cvtab, const LatCv *__restrict__ cv,
                                                   double *__restrict__ wavecv) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double dv = 0.0;
  if (j < nq) {
    const uint64_t b = rs->p_cons + 4ull * (uint64_t)j;
    const uint32_t w0 = mt_temper(raw[(b + 0) & mask]), w1 = mt_temper(raw[(b + 1) & mask]);
    const uint32_t w2 = mt_temper(raw[(b + 2) & mask]), w3 = mt_temper(raw[(b + 3) & mask]);
    const double u1 = canonical(w0, w1), u2 = canonical(w2, w3);
    double n1, nl1;
    const LatQ q = lat_make_quad(u1, u2, &n1, &nl1);
    double *p = qt + (size_t)j * LAT_QW;
    p[0] = q.xT;
    p[1] = q.xE;
    p[2] = q.l2;
    p[3] = q.m;
    qx[j] = make_double2(n1, nl1);
    if (cvtab) {
      const int ci = min(LAT_CV_G - 1, (int)(u1 * LAT_CV_G)), cj = min(LAT_CV_G - 1, (int)(u2 * LAT_CV_G));
      dv = (double)cvtab[ci * LAT_CV_G + cj] - cv->mean;
    }
  }
  if (!cvtab) return;
  // (a wave's 64 quads lie in one chunk: chunks are multiples of 64 quads)
  for (int d = 32; d > 0; d >>= 1) dv += __shfl_down(dv, d, 64);
  if ((threadIdx.x & 63) == 0) wavecv[j >> 6] = dv;
}

// snapshot entries a chunk with W entering rows may write: its walkers at every sub-chunk boundary, about K W / sqrt(quads walked) with
// K = 1.1 (independent rows) ... 3.3 (probit classification with well separated classes: most decisions are the quad's alone)
__host__ __device__ inline int64_t lat_snap_cap(int64_t W, int nsub, int64_t Lq, int subq) {
  return 8 * (int64_t)nsub + (int64_t)(8.0 * (double)W * sqrt((double)Lq) / (double)subq);
}

// ---- windows ------------------------------------------------------------------------------------------
// smallest i in [0, n] whose expected position M_i (quads consumed before row i) is >= target
__device__ int64_t lat_find_row(const double *__restrict__ PM, const float *__restrict__ mf, int64_t n, int64_t nb, double target) {
  if (!(target > 0.0)) return 0;
  if (target >= PM[nb]) return n;
  int64_t lo = 0, hi = nb - 1;  // largest b with PM[b] <= target
  while (lo < hi) {
    const int64_t mid = (lo + hi + 1) >> 1;
    if (PM[mid] <= target)
      lo = mid;
    else
      hi = mid - 1;
  }
  double acc = PM[lo];
  int64_t i = lo * LAT_RB;
  const int64_t e = min(n, (lo + 1) * LAT_RB);
  while (i < e && acc < target) acc += (double)mf[i++];
  return i;
}

__global__ __launch_bounds__(1024) void k_lat_windows(const double *__restrict__ PM, const double *__restrict__ PV,
                                                      const float *__restrict__ mf, int64_t n, int64_t nb, int C, int64_t Lq,
                                                      int subq, double ksig, int32_t *__restrict__ win_lo, int32_t *__restrict__ win_hi,
                                                      int64_t *__restrict__ list_off, int64_t *__restrict__ snap_off,
                                                      int32_t *__restrict__ live, int64_t list_cap, int64_t snap_cap,
                                                      LatStatus *__restrict__ st, const double *__restrict__ cvsum,
                                                      double *__restrict__ cvcum, const LatCv *__restrict__ cv) {
  __shared__ int64_t s_w[1024], s_s[1024];
  const int tid = threadIdx.x;
  const int nsub = (int)(Lq / subq);
  const int per = (C + 1023) / 1024;
  const int c0 = min(C, tid * per), c1 = min(C, c0 + per);
  // control variate (cv != null: the first attempt): accepts beyond the expected number in the quads before every chunk
  if (cv && tid == 0) {
    double acc = 0.0;
    for (int c = 0; c < C; c++) {
      cvcum[c] = acc;
      acc += cvsum[c];
    }
  }
  __syncthreads();
  const double srho = cv ? sqrt(cv->rho) : 1.0;
  int64_t sw = 0, ss = 0;
  for (int c = c0; c < c1; c++) {
    int64_t lo = 0, hi = 0;
    if (c > 0) {
      const double J = (double)c * (double)Lq;
      const int64_t ts = lat_find_row(PM, mf, n, nb, J);
      const int64_t b = min(nb - 1, ts / LAT_RB);
      const double sd = sqrt(PV[b + 1]) * srho;
      const int64_t shift = cv ? (int64_t)llrint(cvcum[c]) : 0;
      lo = min(n, max((int64_t)0, lat_find_row(PM, mf, n, nb, J - ksig * sd) - 2 + shift));
      // (a window that reaches "all rows served" keeps that end where it is: a draw that is over by this chunk enters it at row n)
      const int64_t hi0 = min(n, lat_find_row(PM, mf, n, nb, J + ksig * sd) + 2);
      hi = max(lo, hi0 == n ? n : min(n, hi0 + shift));
    }
    win_lo[c] = (int32_t)lo;
    win_hi[c] = (int32_t)hi;
    const int64_t W = hi - lo + 1;
    live[c] = (int32_t)W;
    sw += W;
    ss += lat_snap_cap(W, nsub, Lq, subq);
  }
  s_w[tid] = sw;
  s_s[tid] = ss;
  __syncthreads();
  if (tid == 0) {
    int64_t aw = 0, as = 0;
    for (int i = 0; i < 1024; i++) {
      const int64_t w = s_w[i], s = s_s[i];
      s_w[i] = aw;
      s_s[i] = as;
      aw += w;
      as += s;
    }
    st->walkers = aw;
    st->snap_need = as;
    if (aw > list_cap) lat_fail(st, 3, -1);
    if (as > snap_cap) lat_fail(st, 2, -1);
    list_off[C] = aw;
    snap_off[C] = as;
  }
  __syncthreads();
  int64_t aw = s_w[tid], as = s_s[tid];
  for (int c = c0; c < c1; c++) {
    list_off[c] = aw;
    snap_off[c] = as;
    const int64_t W = (int64_t)win_hi[c] - win_lo[c] + 1;
    aw += W;
    as += lat_snap_cap(W, nsub, Lq, subq);
  }
}

// (the quad tables are padded by LAT_QPAD records: the look-ahead of the last segment reads past the last quad)
constexpr int LAT_QPAD = 40;

// per chunk: merge the walkers that met, compact (src -> dst), optionally keep the list as snapshot `snap_k`
__global__ __launch_bounds__(1024) void k_lat_compact(const int32_t *__restrict__ win_lo, int32_t *__restrict__ live,
                                                      const int64_t *__restrict__ list_off, const int32_t *__restrict__ scur,
                                                      const int32_t *__restrict__ sfin, int32_t *__restrict__ dcur,
                                                      int32_t *__restrict__ dfin, int first, int snap_k, int nsub,
                                                      const int64_t *__restrict__ snap_off, int64_t *__restrict__ snap_pos,
                                                      int2 *__restrict__ snap, int64_t *__restrict__ snap_idx,
                                                      int32_t *__restrict__ snap_cnt, int32_t *__restrict__ max_live,
                                                      LatStatus *__restrict__ st) {
  __shared__ int s_cnt[16];
  __shared__ int s_out;
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int nl = live[c];
  const int64_t off = list_off[c];
  const int lo = win_lo[c];
  const bool do_snap = snap_k >= 0;
  int64_t sp = 0;
  bool snap_ok = false;
  if (do_snap) {
    sp = snap_k == 0 ? snap_off[c] : snap_pos[c];
    snap_ok = true;
  }
  if (tid == 0) s_out = 0;
  __syncthreads();
  for (int base = 0; base < nl; base += 1024) {
    const int i = base + tid;
    const bool act = i < nl;
    int t = 0, f = 0, prev = -1;
    if (act) {
      t = scur[off + i];
      f = first ? lo + i : sfin[off + i];
      if (i > 0) prev = scur[off + i - 1];
    }
    const bool keep = act && t != prev;
    const unsigned long long b = __ballot(keep);
    if (lane == 0) s_cnt[wid] = __popcll(b);
    __syncthreads();
    int before = s_out, tot = 0;
    for (int w = 0; w < 16; w++) {
      const int cw = s_cnt[w];
      if (w < wid) before += cw;
      tot += cw;
    }
    if (keep) {
      const int r = before + __popcll(b & ((1ull << lane) - 1ull));
      dcur[off + r] = t;
      dfin[off + r] = f;
      if (snap_ok && sp + r < snap_off[c + 1]) snap[sp + r] = make_int2(f, t);
    }
    __syncthreads();
    if (tid == 0) s_out += tot;
    __syncthreads();
  }
  if (tid == 0) {
    const int out = s_out;
    live[c] = out;
    atomicMax(max_live, out);
    if (do_snap) {
      if (sp + out > snap_off[c + 1]) lat_fail(st, 2, c);
      snap_idx[(size_t)c * nsub + snap_k] = sp;
      snap_cnt[(size_t)c * nsub + snap_k] = out;
      snap_pos[c] = sp + out;
    }
  }
}

// ---- the rest of a chunk in one launch: a workgroup per chunk, a walker per thread -------------------------------------
// Waves work on their own 64 walkers without workgroup barriers: after every round a wave merges the walkers that met among its own
// (sorted, so a compare with the lane below); at every sub-chunk boundary the workgroup gathers all walkers, merges across the wave
// borders, writes the snapshot and hands the survivors out again (full waves first).
//
// Memory never sits on the path of a step:
//  * rows. A walker reads its rows in order, one per acceptance, each a dependent 16-byte gather. Every lane keeps the records of
//    rows t + 1 .. t + 8 in a private LDS ring (slot = row & 7). Steps run in groups of four: at the start of a group the lane
//    requests the rows the ring will miss after it (it holds rows up to hi - 1 >= t + 4, a group accepts at most four: rows
//    hi .. t + 8, at most four 16-byte loads), at the end of the group it stores them into the ring.
//  * quads. The 16 quads of a segment (640 bytes, the same for every walker of the chunk) are read by broadcast from a per-wave LDS
//    strip; the next segment's strip is requested at the start of a segment and stored (other buffer) at its end. (Scalar loads
//    would do for a wave-uniform address -- but SMEM returns out of order: with one in flight every wait for an LDS read becomes
//    lgkmcnt(0) and pays the scalar load's latency at every step.)
// The loads are inline asm, their results pass through the `s_waitcnt` that ends the group / segment: nothing is in flight across a
// loop back-edge, where the register allocator may copy registers.
typedef double lat_d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ lat_d2v lat_asm_load(const double2 *p) {
  lat_d2v v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int OFF>
__device__ __forceinline__ lat_d2v lat_asm_load_off(const double2 *p) {  // (the 16-byte record at byte offset OFF: same address register)
  lat_d2v v;
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(v) : "v"(p), "n"(OFF) : "memory");
  return v;
}

// a walker and its row supply
struct LatLane {
  int32_t t, hi;  // current row; rows below hi are in the ring or requested
  double2 r;      // walker record of row t
};
// A lane carries WW walkers (consecutive ones of the sorted list): their steps are independent dependency chains the SIMD can
// interleave, and the step's quad is read once for all of them. Walker w of thread tid owns column tid * WW + w of the ring
// ([RING][NT * WW] records).
// (re)start of a lane's row supply: the current record and the ring's rows t + 1 .. t + RING by plain loads
template <int NT, int RING, int WW>
__device__ __forceinline__ void lat_prime(LatLane (&L)[WW], double2 *ring, int tid, const double2 *__restrict__ wrec, int64_t n) {
#pragma unroll
  for (int w = 0; w < WW; w++) {
    L[w].r = wrec[L[w].t];
#pragma unroll
    for (int k = 1; k <= RING; k++)
      ring[(size_t)((L[w].t + k) & (RING - 1)) * (NT * WW) + tid * WW + w] = wrec[min((int64_t)L[w].t + k, n)];
    L[w].hi = L[w].t + RING + 1;
  }
  // (the compiler's own wait for these loads belongs HERE: placed at the first use of r -- inside the step loop -- it would also wait,
  //  at every step, for the asm loads the group has just issued)
#pragma unroll
  for (int w = 0; w < WW; w++) asm volatile("" : "+v"(L[w].r.x), "+v"(L[w].r.y) : : "memory");
}
// `steps` quads from quad j0 for the 64 * WW walkers of a wave. strip0: the wave's two LDS strips of QS quads; buf / strip_ready:
// which of them holds the quads from j0 on (a caller that continues where the last call ended keeps them).
constexpr int LAT_QS = 16;  // quads per segment
template <int NT, int RING, int WW>
__device__ __forceinline__ void lat_walk_wave(LatLane (&L)[WW], double2 *ring, double *strip0, int &buf, bool &strip_ready, int tid,
                                              int lane, const double2 *__restrict__ wrec, int64_t n, const double *__restrict__ qt,
                                              int64_t jstart, int steps, const LatCtx &cx) {
  constexpr int G = RING / 2, QS = LAT_QS, NC = NT * WW;
  static_assert(G == 4, "groups of four steps");
  for (int seg = 0; seg < steps; seg += QS) {
    const int S = min(QS, steps - seg);
    const int64_t j0 = jstart + seg;
    double *strip = strip0 + buf * (QS * LAT_QW);
    const double2 *qsrc = (const double2 *)(qt + (size_t)j0 * LAT_QW) + (lane < QS * LAT_QW / 2 ? lane : 0);
    if (!strip_ready) {
      lat_d2v q = lat_asm_load(qsrc);
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(q) : : "memory");
      if (lane < QS * LAT_QW / 2) ((double2 *)strip)[lane] = make_double2(q.x, q.y);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    lat_d2v qnext = lat_asm_load(qsrc + QS * LAT_QW / 2);  // (the table is padded: the last look-ahead stays inside it)
    for (int k0 = 0; k0 < S; k0 += G) {
      // rows the ring will miss after this group: hi .. t + RING (it holds rows up to hi - 1 >= t + G). All G records from hi on are
      // requested -- one address, consecutive 16-byte records (the walker records are padded by LAT_RPAD rows that accept nothing) --,
      // the first m of them are stored
      int m[WW];
      lat_d2v l[WW][G];
#pragma unroll
      for (int w = 0; w < WW; w++) {
        m[w] = L[w].t + RING + 1 - L[w].hi;  // 0 .. G
        const double2 *src = wrec + min((int64_t)L[w].hi, n);
        l[w][0] = lat_asm_load_off<0>(src);
        l[w][1] = lat_asm_load_off<16>(src);
        l[w][2] = lat_asm_load_off<32>(src);
        l[w][3] = lat_asm_load_off<48>(src);
      }
      auto one_step = [&](int k) {
        const double2 *qp = (const double2 *)(strip + (k0 + k) * LAT_QW);
        const double2 qa = qp[0], qb = qp[1];
        const LatQ q{qa.x, qa.y, qb.x, qb.y};
#pragma unroll
        for (int w = 0; w < WW; w++) {
          const double2 nx = ring[(size_t)((L[w].t + 1) & (RING - 1)) * NC + tid * WW + w];
          const bool acc = lat_accept(cx, L[w].t, L[w].r.x, L[w].r.y, q, j0 + k0 + k);
          L[w].t += acc ? 1 : 0;
          L[w].r.x = acc ? nx.x : L[w].r.x;
          L[w].r.y = acc ? nx.y : L[w].r.y;
        }
      };
      if (S == QS) {  // (the usual segment: the group's steps as one basic block)
#pragma unroll
        for (int k = 0; k < G; k++) one_step(k);
      } else {
        const int ke = min(G, S - k0);
        for (int k = 0; k < ke; k++) one_step(k);
      }
#pragma unroll
      for (int w = 0; w < WW; w++)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(l[w][0]), "+v"(l[w][1]), "+v"(l[w][2]), "+v"(l[w][3]), "+v"(qnext) : : "memory");
#pragma unroll
      for (int w = 0; w < WW; w++) {
#pragma unroll
        for (int k = 0; k < G; k++)
          if (m[w] > k) ring[(size_t)((L[w].hi + k) & (RING - 1)) * NC + tid * WW + w] = make_double2(l[w][k].x, l[w][k].y);
        L[w].hi += m[w];
      }
    }
    // the next segment's quads into the other strip
    buf ^= 1;
    if (lane < QS * LAT_QW / 2) ((double2 *)(strip0 + buf * (QS * LAT_QW)))[lane] = make_double2(qnext.x, qnext.y);
    strip_ready = S == QS;  // (a short segment ends where the caller's round ends: the next one starts elsewhere)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// every live walker of every chunk: `R` quads from quad c Lq + done, a walker per thread, on the row supply above. grid (tiles, C)
template <int NT, int RING>
__global__ __launch_bounds__(NT) void k_lat_round_ring(const double2 *__restrict__ rec, const double2 *__restrict__ wrec,
                                                       const double *__restrict__ qt, const double2 *__restrict__ qx,
                                                       const uint32_t *__restrict__ raw, uint64_t mask,
                                                       const RngState *__restrict__ rs, int64_t n, int64_t Lq, int done, int R,
                                                       const int32_t *__restrict__ win_lo, const int32_t *__restrict__ live,
                                                       const int64_t *__restrict__ list_off, int32_t *cur, int first) {
  extern __shared__ double2 lat_lds[];
  double2 *ring = lat_lds;                         // [RING][NT]
  double *strips = (double *)(lat_lds + RING * NT);  // [NT / 64][2][QS * LAT_QW]
  const int c = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int i = blockIdx.x * NT + tid;
  const int nl = live[c];
  if (blockIdx.x * NT + wid * 64 >= nl) return;  // (a whole wave past the list)
  const LatCtx cx{rec, qx, raw, mask, rs->p_cons};
  int32_t *pc = cur + list_off[c] + i;
  LatLane L[1];
  L[0].t = i < nl ? (first ? win_lo[c] + i : *pc) : (int32_t)n;
  lat_prime<NT, RING, 1>(L, ring, tid, wrec, n);
  int buf = 0;
  bool ready = false;
  lat_walk_wave<NT, RING, 1>(L, ring, strips + (size_t)wid * (2 * LAT_QS * LAT_QW), buf, ready, tid, lane, wrec, n, qt,
                             (int64_t)c * Lq + done, R, cx);
  if (i < nl) *pc = L[0].t;
}

// the rest of a chunk in one launch: NT threads, WW walkers each (<= NT * WW walkers)
template <int NT, int RING, int WW>
__global__ __launch_bounds__(NT) void k_lat_resident(const double2 *__restrict__ rec, const double2 *__restrict__ wrec,
                                                     const double *__restrict__ qt, const double2 *__restrict__ qx,
                                                     const uint32_t *__restrict__ raw, uint64_t mask,
                                                     const RngState *__restrict__ rs, int64_t n, int64_t Lq, int done0, int R, int subq,
                                                     int nsub, const int32_t *__restrict__ live, const int64_t *__restrict__ list_off,
                                                     const int32_t *__restrict__ scur, const int32_t *__restrict__ sfin,
                                                     const int64_t *__restrict__ snap_off, const int64_t *__restrict__ snap_pos,
                                                     int2 *__restrict__ snap, int64_t *__restrict__ snap_idx,
                                                     int32_t *__restrict__ snap_cnt, LatStatus *__restrict__ st) {
  constexpr int NW = NT / 64, NC = NT * WW, WPW = 64 * WW;  // waves, walkers of the workgroup, walkers of a wave
  constexpr int QS = LAT_QS;
  extern __shared__ double2 lat_lds[];
  double2 *ring = lat_lds;                   // [RING][NC]
  int *s_t0 = (int *)(lat_lds + RING * NC);  // [NC] x 2: t / first_in of the gather
  int *s_f0 = s_t0 + NC;
  int *s_cnt = s_f0 + NC;                    // [16]
  double *strips = (double *)(s_cnt + 16);   // [NW][2][QS * LAT_QW]
  int *s_t1 = (int *)ring, *s_f1 = s_t1 + NC;  // second stage of the redistribution: the ring is refilled after it anyway
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const LatCtx cx{rec, qx, raw, mask, rs->p_cons};
  const int64_t J0 = (int64_t)c * Lq;
  int nl = live[c];
  if (nl > NC) {
    if (tid == 0) lat_fail(st, 3, c);
    return;
  }
  const int64_t off = list_off[c];
  int64_t sp = done0 >= subq ? snap_pos[c] : snap_off[c];
  const int64_t sp_end = snap_off[c + 1];
  // the walkers fill the waves one after the other (wave w: walkers [WPW w, WPW w + wl), lane l: walkers WW l .. WW l + WW - 1 of
  // them): the kernel is bound by instruction issue -- eight quarter-filled waves were 20 % slower than two full ones
  const int w0 = wid * WPW;
  int wl = max(0, min(WPW, nl - w0));
  LatLane L[WW];
  int32_t f[WW];
#pragma unroll
  for (int w = 0; w < WW; w++) {
    const int wi = lane * WW + w;
    L[w].t = wi < wl ? scur[off + w0 + wi] : (int32_t)n;
    f[w] = wi < wl ? sfin[off + w0 + wi] : 0;
  }
  lat_prime<NT, RING, WW>(L, ring, tid, wrec, n);
  double *strip0 = strips + (size_t)wid * (2 * QS * LAT_QW);
  int done = done0, buf = 0;
  bool strip_ready = false;  // strips[buf] holds the quads of the segment that starts at `done`
  // survivors among the first `cnt_in` entries of a sorted list held WW per lane: keep[w], rank[w] (order of the list), their number
  auto survivors = [&](const int (&tt)[WW], int prev_first, int cnt_in, bool (&keep)[WW], int (&rank)[WW]) {
    unsigned long long bal[WW];
#pragma unroll
    for (int w = 0; w < WW; w++) {
      const int wi = lane * WW + w;
      int prev = w == 0 ? __shfl_up(tt[WW - 1], 1, 64) : tt[w - 1];
      if (w == 0 && lane == 0) prev = prev_first;
      keep[w] = wi < cnt_in && tt[w] != prev;
      bal[w] = __ballot(keep[w]);
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    int before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < WW; w++) {
      before += __popcll(bal[w] & lt);
      tot += __popcll(bal[w]);
    }
#pragma unroll
    for (int w = 0; w < WW; w++) {
      rank[w] = before;
      before += keep[w] ? 1 : 0;
    }
    return tot;
  };
  while (done < (int)Lq) {
    const int to_boundary = subq - (done % subq);
    const int Rr = min(R, to_boundary);
    if (wl > 0) {
      lat_walk_wave<NT, RING, WW>(L, ring, strip0, buf, strip_ready, tid, lane, wrec, n, qt, J0 + done, Rr, cx);
      // merge inside the wave: the walkers are sorted, a walker that met the one below it is dropped
      int tt[WW], rank[WW];
      bool keep[WW];
#pragma unroll
      for (int w = 0; w < WW; w++) tt[w] = L[w].t;
      const int cnt = survivors(tt, -1, wl, keep, rank);
      if (cnt != wl) {
#pragma unroll
        for (int w = 0; w < WW; w++)
          if (keep[w]) {
            s_t0[w0 + rank[w]] = L[w].t;
            s_f0[w0 + rank[w]] = f[w];
          }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        wl = cnt;
#pragma unroll
        for (int w = 0; w < WW; w++) {
          const int wi = lane * WW + w;
          L[w].t = wi < wl ? s_t0[w0 + wi] : (int32_t)n;
          f[w] = wi < wl ? s_f0[w0 + wi] : 0;
        }
        __builtin_amdgcn_wave_barrier();
        lat_prime<NT, RING, WW>(L, ring, tid, wrec, n);
      }
    } else {
      strip_ready = false;
    }
    done += Rr;
    if (done % subq == 0) {
      // workgroup-wide: gather, merge across wave borders, snapshot, redistribute
      __syncthreads();
      if (lane == 0) s_cnt[wid] = wl;
      __syncthreads();
      int before = 0, tot = 0;
      for (int k = 0; k < NW; k++) {
        const int cw = s_cnt[k];
        if (k < wid) before += cw;
        tot += cw;
      }
#pragma unroll
      for (int w = 0; w < WW; w++) {
        const int wi = lane * WW + w;
        if (wi < wl) {
          s_t0[before + wi] = L[w].t;
          s_f0[before + wi] = f[w];
        }
      }
      __syncthreads();
      // thread tid looks at entries WW tid .. WW tid + WW - 1 of the gathered list
      int t2[WW], f2[WW], rank2[WW];
      bool keep2[WW];
#pragma unroll
      for (int w = 0; w < WW; w++) {
        const int idx = tid * WW + w;
        t2[w] = idx < tot ? s_t0[idx] : 0;
        f2[w] = idx < tot ? s_f0[idx] : 0;
      }
      const int prev_first = (tid > 0 && tid * WW - 1 < tot) ? s_t0[tid * WW - 1] : -1;  // (lane 0 of a wave: the entry before its first)
      const int cnt_wave = max(0, min(WPW, tot - wid * WPW));
      const int kept = survivors(t2, prev_first, cnt_wave, keep2, rank2);
      __syncthreads();
      if (lane == 0) s_cnt[wid] = kept;
      __syncthreads();
      int before2 = 0, tot2 = 0;
      for (int k = 0; k < NW; k++) {
        const int cw = s_cnt[k];
        if (k < wid) before2 += cw;
        tot2 += cw;
      }
      const int sk = done / subq - 1;
#pragma unroll
      for (int w = 0; w < WW; w++)
        if (keep2[w]) {
          const int rk = before2 + rank2[w];
          s_t1[rk] = t2[w];  // (in the ring's memory: every lane is past its last ring read, and refills the ring below)
          s_f1[rk] = f2[w];
          if (sp + rk < sp_end) snap[sp + rk] = make_int2(f2[w], t2[w]);
        }
      if (tid == 0) {
        if (sp + tot2 > sp_end) lat_fail(st, 2, c);
        snap_idx[(size_t)c * nsub + sk] = sp;
        snap_cnt[(size_t)c * nsub + sk] = tot2;
      }
      sp += tot2;
      __syncthreads();
      nl = tot2;
      wl = max(0, min(WPW, nl - w0));
#pragma unroll
      for (int w = 0; w < WW; w++) {
        const int wi = lane * WW + w;
        L[w].t = wi < wl ? s_t1[w0 + wi] : (int32_t)n;
        f[w] = wi < wl ? s_f1[w0 + wi] : 0;
      }
      __syncthreads();
      lat_prime<NT, RING, WW>(L, ring, tid, wrec, n);
    }
  }
}

// ---- resolution: T(c + 1) = map_c(T(c)) -------------------------------------------------------------------------------
// One workgroup walks the chunks in order. A chunk's final list (a few dozen ... a few hundred entries) is requested one chunk ahead
// and searched in LDS: the serial loop never waits for global memory (a dependent global round trip per chunk and level made this
// launch the longest of the draw after the flows: ~8 us x 511 chunks).
constexpr int LAT_RESOLVE_NT = 512;
__global__ __launch_bounds__(LAT_RESOLVE_NT) void k_lat_resolve(int C, int nsub, int n, const int32_t *__restrict__ win_lo,
                                                                const int32_t *__restrict__ win_hi, const int2 *__restrict__ snap,
                                                                const int64_t *__restrict__ snap_idx, const int32_t *__restrict__ snap_cnt,
                                                                int32_t *__restrict__ Tc, LatStatus *__restrict__ st) {
  constexpr int NT = LAT_RESOLVE_NT;
  __shared__ int s_fin[NT + 1];
  __shared__ int s_T;
  const int tid = threadIdx.x;
  if (st->fail != 0) return;
  auto meta = [&](int c, int64_t &sp, int &cnt, int &lo, int &hi) {
    sp = snap_idx[(size_t)c * nsub + nsub - 1];
    cnt = snap_cnt[(size_t)c * nsub + nsub - 1];
    lo = win_lo[c];
    hi = win_hi[c];
  };
  int64_t sp, sp_n = 0;
  int cnt, lo, hi, cnt_n = 0, lo_n = 0, hi_n = 0;
  meta(0, sp, cnt, lo, hi);
  int2 e = tid < cnt ? snap[sp + tid] : make_int2(0x7fffffff, 0), e_n = e;
  if (C > 1) meta(1, sp_n, cnt_n, lo_n, hi_n);
  int T = 0;
  for (int c = 0; c < C; c++) {
    // the next chunk's list and the metadata of the one after it: in flight while this chunk is searched
    if (c + 1 < C) e_n = tid < cnt_n ? snap[sp_n + tid] : make_int2(0x7fffffff, 0);
    int64_t sp_nn = 0;
    int cnt_nn = 0, lo_nn = 0, hi_nn = 0;
    if (c + 2 < C) meta(c + 2, sp_nn, cnt_nn, lo_nn, hi_nn);
    if (tid == 0) Tc[c] = T;
    if (T < lo || T > hi) {
      if (tid == 0) lat_fail(st, 1, c);
      return;
    }
    if (cnt <= NT) {
      s_fin[tid] = e.x;
      if (tid == 0) s_fin[NT] = 0x7fffffff;
      __syncthreads();
      if (tid < cnt && e.x <= T && (tid + 1 == cnt || s_fin[tid + 1] > T)) s_T = e.y;
    } else {  // (a list longer than the workgroup: straight from global memory)
      for (int i = tid; i < cnt; i += NT) {
        const int2 g = snap[sp + i];
        const bool last = i + 1 == cnt;
        if (g.x <= T && (last || snap[sp + i + 1].x > T)) s_T = g.y;
      }
    }
    __syncthreads();
    T = s_T;
    __syncthreads();
    e = e_n;
    sp = sp_n;
    cnt = cnt_n;
    lo = lo_n;
    hi = hi_n;
    sp_n = sp_nn;
    cnt_n = cnt_nn;
    lo_n = lo_nn;
    hi_n = hi_nn;
  }
  if (tid == 0) {
    Tc[C] = T;
    if (T != n) lat_fail(st, 4, C);  // the last row is not served inside the prepared quads: nothing is written
  }
}

// ---- final pass: the one true path, a thread per sub-chunk ---------------------------------------------------------------
__global__ __launch_bounds__(64) void k_lat_final(const double2 *__restrict__ rec, const double2 *__restrict__ wrec,
                                                  const double *__restrict__ qt, const double2 *__restrict__ qx,
                                                  const uint32_t *__restrict__ raw, uint64_t mask, RngState *__restrict__ rs, int64_t n,
                                                  int64_t Lq, int subq, int nsub, int C, const int32_t *__restrict__ Tc,
                                                  const int2 *__restrict__ snap, const int64_t *__restrict__ snap_idx,
                                                  const int32_t *__restrict__ snap_cnt, const int32_t *__restrict__ rows,
                                                  double2 *__restrict__ eq, const double *__restrict__ y, int n_class,
                                                  LatStatus *__restrict__ st) {
  const int64_t g = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (g >= (int64_t)C * nsub) return;
  if (st->fail != 0) return;
  const int c = (int)(g / nsub), ks = (int)(g % nsub);
  const int T = Tc[c];
  int t = T;
  if (ks > 0) {  // the walker of snapshot ks - 1 that carries entering row T: the last entry with first_in <= T
    const int64_t sp = snap_idx[(size_t)c * nsub + ks - 1];
    int lo = 0, hi = snap_cnt[(size_t)c * nsub + ks - 1] - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (snap[sp + mid].x <= T)
        lo = mid;
      else
        hi = mid - 1;
    }
    t = snap[sp + lo].y;
  }
  if (t >= n) return;
  const LatCtx cx{rec, qx, raw, mask, rs->p_cons};
  const int64_t j0 = (int64_t)c * Lq + (int64_t)ks * subq;
  double2 r = wrec[t];
  // four quads at a time: a thread's 128 bytes of decisions' records are one cache line (its neighbours work 16 KB away: a line fetched
  // for one step would be gone from the caches before the next), the next four are requested before these are worked on
  const double4 *qt4 = (const double4 *)qt;
  double4 qc[4], qn[4];
  double2 xc[4], xn[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    qc[k] = qt4[j0 + k];
    xc[k] = qx[j0 + k];
  }
  for (int s = 0; s < subq; s += 4) {
#pragma unroll
    for (int k = 0; k < 4; k++) {  // (the tables are padded: the look-ahead of the last block stays inside them)
      qn[k] = qt4[j0 + s + 4 + k];
      xn[k] = qx[j0 + s + 4 + k];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const LatQ q{qc[k].x, qc[k].y, qc[k].z, qc[k].w};
      if (lat_accept(cx, t, r.x, r.y, q, j0 + s + k)) {
        const double2 ab = rec[t];
        const double val = lat_value(ab.x, ab.y, q, xc[k].x, xc[k].y);
        const int64_t row = rows ? (int64_t)rows[t] : (int64_t)t;
        const double pred = eq[row].x;
        // the side of a one-sided draw (right truncation = -left(-mu)); z = 1 * draw + score, e = score - z
        int sgn = 1;
        if (n_class == 0)
          sgn = y[row] > 0 ? 1 : -1;
        else if ((int)y[row] == 0)
          sgn = -1;
        const double draw = sgn > 0 ? val : -val;
        const double z = 1.0 * draw + pred;
        eq[row].x = pred - z;
        t++;
        if (t >= n) {
          st->end_quads = j0 + s + k + 1;
          return;
        }
        r = wrec[t];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      qc[k] = qn[k];
      xc[k] = xn[k];
    }
  }
}

// the stream moves past the quads the draw consumed (only when every pass succeeded)
__global__ void k_lat_commit(RngState *__restrict__ rs, LatStatus *__restrict__ st) {
  if (st->fail == 0 && st->end_quads < 0) st->fail = 4;
  if (st->fail == 0) {
    rs->p_cons += 4ull * (uint64_t)st->end_quads;
    if (rs->p_cons > rs->p_gen) rs->error = 1;
  }
}

static double env_double(const char *name, double dflt) {
  const char *e = std::getenv(name);
  return e ? std::atof(e) : dflt;
}
static int env_int(const char *name, int dflt) {
  const char *e = std::getenv(name);
  return e ? std::atoi(e) : dflt;
}

}  // namespace

struct LatentEngine::Impl {
  DevBuf<double2> rec, wrec, qx;
  DevBuf<float> mf;
  DevBuf<double> blkM, blkV, PM, PV, qt;
  DevBuf<int32_t> win_lo, win_hi, live, cur0, fin0, cur1, fin1, snap_cnt, Tc, max_live, bad;
  DevBuf<int64_t> list_off, snap_off, snap_pos, snap_idx;
  DevBuf<int2> snap;
  DevBuf<LatStatus> status;
  DevBuf<uint64_t> pos;
  // the windows' control variate
  DevBuf<float> cvtab;
  DevBuf<LatQ> cvq;
  DevBuf<double> cvsum, cvcum, cvrow, wavecv;
  DevBuf<LatCv> cv;
  LatStatus *h_status = nullptr;  // pinned
  uint64_t *h_pos = nullptr;
  int32_t *h_max = nullptr;       // [0] largest list of a round, [1] the row pass's "too wide" flag
  double ksig = 3.5, ksig_retry = 6.5;
  // geometry of the prepared draw
  int64_t n = -1, Lq = 0;
  int C = 0, nsub = 0, subq = 0;
  double sd = 0;
  bool too_wide = false;
  ~Impl() {
    if (h_status) (void)hipHostFree(h_status);
    if (h_pos) (void)hipHostFree(h_pos);
    if (h_max) (void)hipHostFree(h_max);
  }
  template <class T>
  static void ensure(DevBuf<T> &b, size_t count) {
    if (b.n < count) b.alloc(count + count / 8 + 16);
  }
  int64_t wmax(double k) const { return std::min<int64_t>(n + 1, (int64_t)(2.0 * k * sd) + 8); }
};

LatentEngine::LatentEngine() : im(new Impl()) {}
LatentEngine::~LatentEngine() { delete im; }

constexpr int LAT_MAX_ROUNDS = 8192;

void LatentEngine::prepare(const LatentJob &job, LatentPrep *prep) {
  Impl &m = *im;
  if (job.n >= 0x7ffffff0ll) throw Error(MFM_ERR_INVALID, "exact latent draws: more than 2^31 rows in a group");
  hipStream_t s = job.stream;
  if (!m.h_status) {
    MFM_HIP_CHECK(hipHostMalloc((void **)&m.h_status, sizeof(LatStatus), hipHostMallocDefault));
    MFM_HIP_CHECK(hipHostMalloc((void **)&m.h_pos, 2 * sizeof(uint64_t), hipHostMallocDefault));
    MFM_HIP_CHECK(hipHostMalloc((void **)&m.h_max, 2 * sizeof(int32_t), hipHostMallocDefault));
    m.status.alloc(1);
    m.pos.alloc(2);
    m.max_live.alloc(LAT_MAX_ROUNDS);
    m.bad.alloc(1);
  }
  // windows of +- 3.5 sigma of the a-priori position. Measured over 7 000 draws (scripts/r06_lat_k.sh, r06_lat_soak.sh): one draw in
  // ~100 has a chunk whose window misses the path (the Bernoulli model is conservative: +- 4 sigma missed in 1 draw of 700, not 1 of
  // 30); the rate is flat between 3.0 and 3.5, at 3.5 the second attempts are rarest. A draw that misses is repeated with
  // +- 6.5 sigma (nothing was written), and only if that misses too (1e-8) the caller's sequential loop runs
  m.ksig = env_double("MFM_LAT_KSIGMA", 3.5);
  m.ksig_retry = std::max(m.ksig, env_double("MFM_LAT_KSIGMA_RETRY", 6.5));
  const int64_t n = job.n, nb = (n + LAT_RB - 1) / LAT_RB;
  // (the block that holds row n writes the record of "the row after the last one": one more block when n fills its blocks)
  const int64_t grid = (n + LAT_RPAD + LAT_RB - 1) / LAT_RB;
  Impl::ensure(m.rec, (size_t)n + LAT_RPAD);
  Impl::ensure(m.wrec, (size_t)n + LAT_RPAD);
  Impl::ensure(m.mf, (size_t)n + LAT_RPAD);
  Impl::ensure(m.blkM, (size_t)grid);
  Impl::ensure(m.blkV, (size_t)grid);
  Impl::ensure(m.PM, (size_t)nb + 1);
  Impl::ensure(m.PV, (size_t)nb + 1);
  MFM_HIP_CHECK(hipMemsetAsync(m.bad.p, 0, sizeof(int32_t), s));
  hipLaunchKernelGGL(k_lat_rows, dim3((unsigned)grid), dim3(256), 0, s, job.rows, job.eq, job.y, n, job.n_class, job.gamma, m.rec.p,
                     m.wrec.p, m.mf.p, m.blkM.p, m.blkV.p, m.bad.p);
  hipLaunchKernelGGL(k_lat_scan, dim3(1), dim3(1024), 0, s, m.blkM.p, m.blkV.p, nb, m.PM.p, m.PV.p, m.status.p, job.state, m.pos.p);
  MFM_HIP_CHECK(hipGetLastError());
  MFM_HIP_CHECK(hipMemcpyAsync(m.h_status, m.status.p, sizeof(LatStatus), hipMemcpyDeviceToHost, s));
  MFM_HIP_CHECK(hipMemcpyAsync(m.h_pos, m.pos.p, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
  MFM_HIP_CHECK(hipMemcpyAsync(m.h_max + 1, m.bad.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  MFM_HIP_CHECK(hipStreamSynchronize(s));
  prep->mean_quads = m.h_status->total_m;
  prep->var_quads = m.h_status->total_v;
  prep->p_cons = m.h_pos[0];
  prep->p_gen = m.h_pos[1];
  m.too_wide = m.h_max[1] != 0;
  m.sd = std::sqrt(prep->var_quads);
  const double cap = prep->mean_quads + (m.ksig_retry + 1.0) * m.sd + 256.0;
  // (a row whose bounds are NaN, or whose acceptance probability is ~0, never accepts: the reference would spin for ever)
  if (!(cap == cap) || cap > 64.0 * (double)n + 1e6)
    throw Error(MFM_ERR_RUNTIME, "exact latent draws: a score is not finite, or a truncation region has (almost) no mass");
  // geometry: C chunks of Lq quads, sub-chunks of subq quads
  const int target_chunks = std::max(1, env_int("MFM_LAT_CHUNKS", 512));
  int64_t Lq = ((int64_t)cap + target_chunks - 1) / target_chunks;
  Lq = std::max<int64_t>(Lq, env_int("MFM_LAT_MIN_LQ", 1024));
  // sub-chunks of 512 quads; short chunks (small tables) 128: the final pass walks a sub-chunk sequentially, ~1.5 us per quad when
  // nothing else hides the latency
  m.subq = std::max(64, env_int("MFM_LAT_SUBQ", Lq >= 8192 ? 512 : 128) / 64 * 64);
  Lq = (Lq + m.subq - 1) / m.subq * m.subq;
  if (Lq > 0x3fffffff) throw Error(MFM_ERR_RUNTIME, "exact latent draws: chunk too long");
  m.Lq = Lq;
  m.C = (int)(((int64_t)cap + Lq - 1) / Lq);
  m.nsub = (int)(Lq / m.subq);
  m.n = n;
  prep->q_cap = (int64_t)m.C * Lq;
}

int LatentEngine::run(const LatentJob &job, const LatentPrep &prep, LatentStats *stats) {
  Impl &m = *im;
  hipStream_t s = job.stream;
  const int64_t n = job.n;
  LatentStats S;
  if (n == 0) {
    if (stats) *stats = S;
    return 0;
  }
  if (n != m.n) throw Error(MFM_ERR_RUNTIME, "exact latent draws: run() without the matching prepare()");
  if (m.too_wide) {  // a row a thousand standard deviations on the wrong side of its class: the walkers' shortcut does not hold
    S.status = 5;
    if (stats) *stats = S;
    return 5;
  }
  const bool timing = std::getenv("MFM_LATENT_TIMING") != nullptr;
  hipEvent_t ev[4];
  if (timing)
    for (auto &e : ev) MFM_HIP_CHECK(hipEventCreate(&e));
  const int subq = m.subq, C = m.C, nsub = m.nsub;
  const int64_t Lq = m.Lq, nq = prep.q_cap;
  const int R = std::max(4, std::min(subq, env_int("MFM_LAT_ROUND", 16)) / 4 * 4);
  const int64_t nb = (n + LAT_RB - 1) / LAT_RB;
  Impl::ensure(m.qt, (size_t)(nq + LAT_QPAD) * LAT_QW);
  Impl::ensure(m.qx, (size_t)(nq + LAT_QPAD));
  Impl::ensure(m.win_lo, (size_t)C);
  Impl::ensure(m.win_hi, (size_t)C);
  Impl::ensure(m.live, (size_t)C);
  Impl::ensure(m.list_off, (size_t)C + 1);
  Impl::ensure(m.snap_off, (size_t)C + 1);
  Impl::ensure(m.snap_pos, (size_t)C);
  Impl::ensure(m.snap_idx, (size_t)C * nsub);
  Impl::ensure(m.snap_cnt, (size_t)C * nsub);
  Impl::ensure(m.Tc, (size_t)C + 1);

  if (timing) MFM_HIP_CHECK(hipEventRecord(ev[0], s));
  // (probit classification only: there the quads decide two thirds of the variance; for ordered probit -- two-sided rows, whose
  //  decisions depend on their own interval -- a third, which pays for the table and no more: 56.7 -> 56.2 ... 57.8 it/s at config 3's
  //  shape, 2.437 -> 2.432 ... 2.439 at config 5, against 62.5 -> 68.3 ... 69.4 for classification)
  const bool use_cv = job.n_class == 0 && n >= env_int("MFM_LAT_CV_MIN_ROWS", 1 << 20) && std::getenv("MFM_LAT_NO_CV") == nullptr;
  const int64_t n_waves = (nq + 63) / 64;
  if (use_cv) {
    constexpr int NC = LAT_CV_G * LAT_CV_G;
    Impl::ensure(m.cvtab, (size_t)NC);
    Impl::ensure(m.cvq, (size_t)NC);
    Impl::ensure(m.cvrow, (size_t)2 * LAT_CV_S);
    Impl::ensure(m.cvsum, (size_t)C + 1);
    Impl::ensure(m.cvcum, (size_t)C + 1);
    Impl::ensure(m.wavecv, (size_t)n_waves + 4);
    if (!m.cv.p) m.cv.alloc(1);
    const int S = (int)std::min<int64_t>(LAT_CV_S, n);
    hipLaunchKernelGGL(k_lat_cv_table, dim3(NC / 256), dim3(256), 0, s, m.wrec.p, n, m.cvtab.p, m.cvq.p);
    hipLaunchKernelGGL(k_lat_cv_stats, dim3((unsigned)S), dim3(256), 0, s, m.wrec.p, n, m.cvtab.p, m.cvq.p, m.cvrow.p,
                       m.cvrow.p + LAT_CV_S, m.cv.p);
    hipLaunchKernelGGL(k_lat_cv_rho, dim3(1), dim3(256), 0, s, m.cvrow.p, m.cvrow.p + LAT_CV_S, n, env_double("MFM_LAT_CV_SAFETY", 1.15),
                       m.cv.p);
  }
  hipLaunchKernelGGL(k_lat_quads, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, job.state, job.raw, job.mask, nq, m.qt.p, m.qx.p,
                     use_cv ? m.cvtab.p : (const float *)nullptr, m.cv.p, m.wavecv.p);
  if (use_cv) hipLaunchKernelGGL(k_lat_cv_chunks, dim3((unsigned)C), dim3(256), 0, s, m.wavecv.p, Lq / 64, n_waves, m.cvsum.p);
  if (timing) MFM_HIP_CHECK(hipEventRecord(ev[1], s));
  int rounds_done = 0, handover = 0, attempts = 0;
  int64_t max_live = 0;
  for (double ksig : {m.ksig, m.ksig_retry}) {
    attempts++;
    const int64_t Wmax = m.wmax(ksig);
    const int64_t list_cap = (int64_t)C * Wmax;
    const int64_t snap_cap = (int64_t)C * (lat_snap_cap(Wmax, nsub, Lq, subq) + 1);
    Impl::ensure(m.cur0, (size_t)list_cap);
    Impl::ensure(m.fin0, (size_t)list_cap);
    Impl::ensure(m.cur1, (size_t)list_cap);
    Impl::ensure(m.fin1, (size_t)list_cap);
    Impl::ensure(m.snap, (size_t)snap_cap);
    MFM_HIP_CHECK(hipMemsetAsync(m.max_live.p, 0, sizeof(int32_t) * LAT_MAX_ROUNDS, s));
    MFM_HIP_CHECK(hipMemsetAsync(m.status.p, 0, 2 * sizeof(int32_t), s));  // (fail, fail_chunk)
    // (first attempt: windows shifted by the control variate and narrowed to what it leaves; second: the plain a-priori windows)
    const bool cv_now = use_cv && attempts == 1;
    hipLaunchKernelGGL(k_lat_windows, dim3(1), dim3(1024), 0, s, m.PM.p, m.PV.p, m.mf.p, n, nb, C, Lq, subq, ksig, m.win_lo.p,
                       m.win_hi.p, m.list_off.p, m.snap_off.p, m.live.p, list_cap, snap_cap, m.status.p, m.cvsum.p, m.cvcum.p,
                       cv_now ? m.cv.p : (const LatCv *)nullptr);
    // Rounds over all chunks' walkers as grid-wide launches (16, 16, 32, ... quads up to the first sub-chunk boundary, then a
    // sub-chunk per round) while some chunk still has more walkers than a workgroup of the resident kernel; the host reads the
    // largest list back after every round (4 bytes) to size the next launch and to decide the hand-over.
    int done = 0, round = 0;
    bool first = true;
    int32_t *sc = m.cur0.p, *sf = m.fin0.p, *dc = m.cur1.p, *df = m.fin1.p;
    max_live = Wmax;
    int Rr = R;
    const bool no_resident = std::getenv("MFM_LAT_NO_RESIDENT") != nullptr;
    constexpr int res_nt = LAT_RES_NT;
    while (done < Lq) {
      if (!first && max_live <= res_nt && !no_resident) break;
      if (round >= LAT_MAX_ROUNDS) throw Error(MFM_ERR_RUNTIME, "exact latent draws: too many rounds");
      int step = (int)std::min<int64_t>(Rr, Lq - done);
      if (done % subq + step > subq) step = subq - done % subq;
      const unsigned tiles = (unsigned)((max_live + LAT_TILE - 1) / LAT_TILE);
      {
        const size_t lds_r = (size_t)8 * LAT_TILE * sizeof(double2) + (size_t)(LAT_TILE / 64) * 2 * LAT_QS * LAT_QW * sizeof(double);
        hipLaunchKernelGGL((k_lat_round_ring<LAT_TILE, 8>), dim3(tiles, (unsigned)C), dim3(LAT_TILE), lds_r, s, m.rec.p, m.wrec.p, m.qt.p,
                           m.qx.p, job.raw, job.mask, job.state, n, Lq, done, step, m.win_lo.p, m.live.p, m.list_off.p, sc, first ? 1 : 0);
      }
      done += step;
      const int snap_k = done % subq == 0 ? done / subq - 1 : -1;
      hipLaunchKernelGGL(k_lat_compact, dim3((unsigned)C), dim3(1024), 0, s, m.win_lo.p, m.live.p, m.list_off.p, sc, sf, dc, df,
                         first ? 1 : 0, snap_k, nsub, m.snap_off.p, m.snap_pos.p, m.snap.p, m.snap_idx.p, m.snap_cnt.p,
                         m.max_live.p + round, m.status.p);
      MFM_HIP_CHECK(hipMemcpyAsync(m.h_max, m.max_live.p + round, sizeof(int32_t), hipMemcpyDeviceToHost, s));
      MFM_HIP_CHECK(hipStreamSynchronize(s));
      max_live = m.h_max[0];
      round++;
      std::swap(sc, dc);
      std::swap(sf, df);
      first = false;
      if (done >= subq)
        Rr = subq;
      else if (done >= 2 * Rr)
        Rr = std::min(subq, 2 * Rr);
    }
    rounds_done = round;
    handover = done;
    if (done < Lq) {
      // one walker per lane, 512 threads (the kernel is written for WW walkers per lane: with two, at 256 threads, the independent
      // chains of a lane did not overlap -- 57 against 40 ms at config 5 --, HISTORY.md round 6)
      constexpr int nt = LAT_RES_NT;
      const size_t lds = (size_t)8 * nt * sizeof(double2) + (size_t)(2 * nt + 16) * sizeof(int) +
                         (size_t)(nt / 64) * 2 * LAT_QS * LAT_QW * sizeof(double);
      {
        static DeviceOnce raised;
        if (raised.need()) {
          MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_lat_resident<LAT_RES_NT, 8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
          raised.mark();
        }
      }
      hipLaunchKernelGGL((k_lat_resident<LAT_RES_NT, 8, 1>), dim3((unsigned)C), dim3(nt), lds, s, m.rec.p, m.wrec.p, m.qt.p, m.qx.p, job.raw,
                         job.mask, job.state, n, Lq, done, R, subq, nsub, m.live.p, m.list_off.p, sc, sf, m.snap_off.p, m.snap_pos.p,
                         m.snap.p, m.snap_idx.p, m.snap_cnt.p, m.status.p);
    }
    if (timing && attempts == 1) MFM_HIP_CHECK(hipEventRecord(ev[2], s));
    hipLaunchKernelGGL(k_lat_resolve, dim3(1), dim3(LAT_RESOLVE_NT), 0, s, C, nsub, (int)n, m.win_lo.p, m.win_hi.p, m.snap.p, m.snap_idx.p,
                       m.snap_cnt.p, m.Tc.p, m.status.p);
    hipLaunchKernelGGL(k_lat_final, dim3((unsigned)(((int64_t)C * nsub + 63) / 64)), dim3(64), 0, s, m.rec.p, m.wrec.p, m.qt.p, m.qx.p,
                       job.raw, job.mask, job.state, n, Lq, subq, nsub, C, m.Tc.p, m.snap.p, m.snap_idx.p, m.snap_cnt.p, job.rows, job.eq,
                       job.y, job.n_class, m.status.p);
    hipLaunchKernelGGL(k_lat_commit, dim3(1), dim3(1), 0, s, job.state, m.status.p);
    if (timing && attempts == 1) MFM_HIP_CHECK(hipEventRecord(ev[3], s));
    MFM_HIP_CHECK(hipGetLastError());
    MFM_HIP_CHECK(hipMemcpyAsync(m.h_status, m.status.p, sizeof(LatStatus), hipMemcpyDeviceToHost, s));
    MFM_HIP_CHECK(hipStreamSynchronize(s));
    if (m.h_status->fail == 1 && use_cv && std::getenv("MFM_LAT_CV_DEBUG")) {
      const int fc = m.h_status->fail_chunk;
      int32_t t = 0, l = 0, h = 0;
      double cum = 0;
      MFM_HIP_CHECK(hipMemcpy(&t, m.Tc.p + fc, sizeof(int32_t), hipMemcpyDeviceToHost));
      MFM_HIP_CHECK(hipMemcpy(&l, m.win_lo.p + fc, sizeof(int32_t), hipMemcpyDeviceToHost));
      MFM_HIP_CHECK(hipMemcpy(&h, m.win_hi.p + fc, sizeof(int32_t), hipMemcpyDeviceToHost));
      MFM_HIP_CHECK(hipMemcpy(&cum, m.cvcum.p + fc, sizeof(double), hipMemcpyDeviceToHost));
      std::fprintf(stderr, "[latent cv] attempt %d missed at chunk %d of %d: entering row %d, window [%d, %d] (shift %.1f)\n", attempts, fc, C, t, l, h, cum);
    }
    if (m.h_status->fail != 1 || ksig >= m.ksig_retry) break;  // (only a missed window is worth a second look)
  }
  if (use_cv && std::getenv("MFM_LAT_CV_DEBUG")) {  // how well the control variate tracks the path (chunk by chunk)
    std::vector<int32_t> tc((size_t)C + 1), lo((size_t)C), hi((size_t)C);
    std::vector<double> cum((size_t)C + 1);
    MFM_HIP_CHECK(hipMemcpy(tc.data(), m.Tc.p, sizeof(int32_t) * ((size_t)C + 1), hipMemcpyDeviceToHost));
    MFM_HIP_CHECK(hipMemcpy(lo.data(), m.win_lo.p, sizeof(int32_t) * (size_t)C, hipMemcpyDeviceToHost));
    MFM_HIP_CHECK(hipMemcpy(hi.data(), m.win_hi.p, sizeof(int32_t) * (size_t)C, hipMemcpyDeviceToHost));
    MFM_HIP_CHECK(hipMemcpy(cum.data(), m.cvcum.p, sizeof(double) * (size_t)C, hipMemcpyDeviceToHost));
    const bool shifted = attempts == 1;
    double sxx = 0, syy = 0, sxy = 0, sx = 0, sy = 0;
    int cnt = 0;
    for (int c = 1; c < C; c++) {
      if (tc[c] >= n) break;
      const double mid = 0.5 * ((double)lo[c] + hi[c]) - (shifted ? std::llrint(cum[c]) : 0);  // a-priori centre
      const double d = (double)tc[c] - mid, v = cum[c];
      sx += d; sy += v; sxx += d * d; syy += v * v; sxy += d * v;
      cnt++;
    }
    if (cnt > 2) {
      const double mx = sx / cnt, my = sy / cnt;
      const double vx = sxx / cnt - mx * mx, vy = syy / cnt - my * my, cxy = sxy / cnt - mx * my;
      std::fprintf(stderr, "[latent cv] chunks %d attempts %d: rms deviation from the a-priori centre %.1f, rms control variate %.1f, correlation %.3f, "
                   "rms of (deviation - cv) %.1f; last chunk: deviation %.1f cv %.1f\n", cnt, attempts, std::sqrt(sxx / cnt), std::sqrt(syy / cnt),
                   cxy / std::sqrt(vx * vy + 1e-300), std::sqrt((sxx - 2 * sxy + syy) / cnt),
                   (double)tc[cnt] - (0.5 * ((double)lo[cnt] + hi[cnt]) - (shifted ? std::llrint(cum[cnt]) : 0)), cum[cnt]);
    }
  }
  S.status = m.h_status->fail;
  S.chunks = C;
  S.subs = nsub;
  S.lq = Lq;
  S.quads_used = m.h_status->end_quads;
  S.walkers = m.h_status->walkers;
  S.attempts = attempts;
  if (timing) {
    float a = 0, b = 0, c2 = 0;
    (void)hipEventElapsedTime(&a, ev[0], ev[1]);
    (void)hipEventElapsedTime(&b, ev[1], ev[2]);
    (void)hipEventElapsedTime(&c2, ev[2], ev[3]);
    S.ms_quads = a;
    S.ms_flow = b;
    S.ms_final = c2;
    LatCv hcv{0.0, 1.0, 1.0};
    if (use_cv) MFM_HIP_CHECK(hipMemcpy(&hcv, m.cv.p, sizeof(LatCv), hipMemcpyDeviceToHost));
    std::fprintf(stderr,
                 "[latent] n %lld quads %lld (cap %lld) chunks %d x %lld (sub %d) walkers %lld, %d rounds then resident from quad %d "
                 "(largest list %lld), %d attempt(s), status %d (chunk %d), variance ratio %.3f (used %.3f) | first attempt: quads + "
                 "table %.3f ms, windows+flows %.3f ms, resolve+final %.3f ms\n",
                 (long long)n, (long long)S.quads_used, (long long)prep.q_cap, C, (long long)Lq, subq, (long long)S.walkers, rounds_done,
                 handover, (long long)max_live, attempts, S.status, S.status ? m.h_status->fail_chunk : -1, hcv.rho_est, hcv.rho, a, b,
                 c2);
    for (auto &e : ev) (void)hipEventDestroy(e);
  }
  if (stats) *stats = S;
  return S.status;
}

}  // namespace mfm
