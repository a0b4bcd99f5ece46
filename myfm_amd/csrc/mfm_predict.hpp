// mfm_predict.hpp -- prediction designs: FM::predict_score (FM.hpp:47-136), Predictor::predict /
// predict_parallel / predict_parallel_oprobit (predictor.hpp:35-147), FM::oprobit_predict_proba
// (FM.hpp:137-162). Included by mfm_hip.hip.
#pragma once

namespace mfm {

// out[t] (+)= score | Phi(score)     predictor.hpp:133-143
__global__ __launch_bounds__(WG) void k_accumulate_pred(const double *__restrict__ score, double *__restrict__ out,
                                                        int64_t N, int mode, int first) {
  const int64_t t = (int64_t)blockIdx.x * WG + threadIdx.x;
  if (t >= N) return;
  double v = score[t];
  if (mode == 1) v = (erf(v * 0.70710678118654752440) + 1.0) / 2.0;
  out[t] = first ? v : out[t] + v;
}
// ordered probit class probabilities, FM.hpp:150-161; out is (N, n_cut + 1) row-major
__global__ __launch_bounds__(WG) void k_accumulate_oprobit(const double *__restrict__ score,
                                                           const double *__restrict__ cut, int n_cut,
                                                           double *__restrict__ out, int64_t N, int first) {
  const int64_t t = (int64_t)blockIdx.x * WG + threadIdx.x;
  if (t >= N) return;
  const double sc = score[t];
  double prev = 0.0;
  double *o = out + t * (n_cut + 1);
  for (int c = 0; c < n_cut; c++) {
    const double cdf = (1.0 + erf((cut[c] - sc) * 0.70710678118654752440)) / 2.0;
    const double v = cdf - prev;
    o[c] = first ? v : o[c] + v;
    prev = cdf;
  }
  const double v = 1.0 - prev;
  o[n_cut] = first ? v : o[n_cut] + v;
}
// ---- all samples of a store in ONE pass over the test rows (predictor.hpp:35-147) --------------------------------------
// Vt of a chunk of samples: vt_all[c][j][s] = V_c[s][j] (k_build_vt with the sample as grid.z)
__global__ __launch_bounds__(WG) void k_build_vt_batch(const double *const *__restrict__ wv, int64_t D, int K, int KS,
                                                       double *__restrict__ vt_all) {
  __shared__ double tile[32][33];
  const double *V = wv[blockIdx.z] + D;
  double *Vt = vt_all + (size_t)blockIdx.z * D * KS;
  const int64_t j0 = (int64_t)blockIdx.x * 32;
  const int s0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int s = s0 + r;
    const int64_t j = j0 + tx;
    tile[r][tx] = (s < K && j < D) ? V[(int64_t)s * D + j] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int64_t j = j0 + r;
    const int s = s0 + tx;
    if (j < D && s < KS) Vt[j * KS + s] = tile[tx][r];
  }
}

// The row pass of k_score (same lane groups, same order of every addition: the scores are bit-identical to the per-sample
// pass) with the samples as an INNER loop: the test CSR is read once (the rows' entries stay in L1 across the samples), a
// row's scores never leave the registers, mean score / mean Phi(score) / mean class probabilities are accumulated in sample
// order in registers and written once. MODE 0 / 1 / 2 as k_accumulate_pred / k_accumulate_oprobit. `first` = 0: continue
// the sums a previous chunk of samples left in `out`.
constexpr int PRED_MAX_CLASS = 32;  // ordered-probit classes of the single-pass predictor (more: the per-sample passes)
struct ScoreStoreArgs {
  const double *const *wv;  // [S] sample buffers (w then V)
  const double *vt_all;     // [S][D][KS]
  const double *w0;         // [S]
  const double *cut;        // [S][n_cut] (MODE 2)
  int S, n_cut, first;
  double scale;             // 1 / number of samples when this is the last chunk, else 1
};
template <int GS, int SPL, int MODE, bool UNIT, bool ELL>
__global__ __launch_bounds__(WG) void k_score_store(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                                                    const double *__restrict__ val, ScoreStoreArgs sa, int64_t D, int K, int KS,
                                                    int ell, double *__restrict__ out, int64_t N) {
  constexpr int RU = SCORE_RU;
  constexpr int CPL = MODE == 2 ? (PRED_MAX_CLASS + GS - 1) / GS : 1;  // classes per lane
  const int lig = threadIdx.x % GS;
  const int64_t t0 = (((int64_t)blockIdx.x * WG + threadIdx.x) / GS) * RU;
  int64_t pb[RU];
  int len[RU];
  int maxlen = 0;
#pragma unroll
  for (int u = 0; u < RU; u++) {
    const int64_t t = t0 + u;
    if (t < N) {
      if (ELL) {
        pb[u] = t * ell;
        len[u] = ell;
      } else {
        pb[u] = rowptr[t];
        len[u] = rowptr[t + 1] - (int32_t)pb[u];
      }
    } else {
      pb[u] = 0;
      len[u] = 0;
    }
    maxlen = len[u] > maxlen ? len[u] : maxlen;
  }
  const int n_class = sa.n_cut + 1;
  double acc[RU][CPL];
#pragma unroll
  for (int u = 0; u < RU; u++)
#pragma unroll
    for (int c = 0; c < CPL; c++) {
      acc[u][c] = 0.0;
      if (!sa.first && t0 + u < N) {
        if (MODE == 2) {
          const int cls = lig + c * GS;
          if (cls < n_class) acc[u][c] = out[(t0 + u) * n_class + cls];
        } else if (lig == 0) {
          acc[u][c] = out[t0 + u];
        }
      }
    }
  for (int smp = 0; smp < sa.S; smp++) {
    const double *__restrict__ Vt = sa.vt_all + (size_t)smp * D * KS;
    const double *__restrict__ w = sa.wv[smp];
    double2 a[RU][SPL];
    double b[RU], lin[RU];
#pragma unroll
    for (int u = 0; u < RU; u++) {
      b[u] = 0.0;
      lin[u] = 0.0;
#pragma unroll
      for (int k = 0; k < SPL; k++) a[u][k] = make_double2(0.0, 0.0);
    }
    for (int k = 0; k < maxlen; k++) {
#pragma unroll
      for (int u = 0; u < RU; u++) {
        if (k < len[u]) {
          const int32_t j = colidx[pb[u] + k];
          const double x = UNIT ? 1.0 : val[pb[u] + k];
          const double x2 = x * x;
          if (lig == 0) lin[u] += x * w[j];
          const double2 *row = (const double2 *)(Vt + (int64_t)j * KS);
#pragma unroll
          for (int s = 0; s < SPL; s++) {
            const int pr = lig + s * GS;
            if (2 * pr < K) {
              const double2 v = row[pr];
              a[u][s].x += x * v.x;
              a[u][s].y += x * v.y;
              b[u] += x2 * (v.x * v.x);
              b[u] += x2 * (v.y * v.y);
            }
          }
        }
      }
    }
    const double w0 = sa.w0[smp];
#pragma unroll
    for (int u = 0; u < RU; u++) {
      double part = 0.0;
#pragma unroll
      for (int s = 0; s < SPL; s++) part += a[u][s].x * a[u][s].x + a[u][s].y * a[u][s].y;
      part = 0.5 * (part - b[u]) + lin[u];
#pragma unroll
      for (int m = GS / 2; m >= 1; m >>= 1) part += __shfl_xor(part, m, WAVE);
      // (the butterfly leaves the group's sum in every lane; the per-sample pass takes lane 0's -- all lanes add the same
      //  operands in an order that depends on the lane, so the other lanes take lane 0's value)
      const double score = w0 + __shfl(part, (threadIdx.x & 63) - lig, WAVE);
      if (MODE == 0) {
        acc[u][0] = (smp == 0 && sa.first) ? score : acc[u][0] + score;
      } else if (MODE == 1) {
        const double v = (erf(score * 0.70710678118654752440) + 1.0) / 2.0;
        acc[u][0] = (smp == 0 && sa.first) ? v : acc[u][0] + v;
      } else {
        const double *cut = sa.cut + (size_t)smp * sa.n_cut;
#pragma unroll
        for (int c = 0; c < CPL; c++) {
          const int cls = lig + c * GS;
          if (cls < n_class) {
            const double prev = cls == 0 ? 0.0 : (1.0 + erf((cut[cls - 1] - score) * 0.70710678118654752440)) / 2.0;
            const double v = cls < sa.n_cut ? (1.0 + erf((cut[cls] - score) * 0.70710678118654752440)) / 2.0 - prev : 1.0 - prev;
            acc[u][c] = (smp == 0 && sa.first) ? v : acc[u][c] + v;
          }
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < RU; u++) {
    const int64_t t = t0 + u;
    if (t >= N) continue;
    if (MODE == 2) {
#pragma unroll
      for (int c = 0; c < CPL; c++) {
        const int cls = lig + c * GS;
        if (cls < n_class) out[t * n_class + cls] = acc[u][c] * sa.scale;
      }
    } else if (lig == 0) {
      out[t] = acc[u][0] * sa.scale;
    }
  }
}

__global__ void k_scale(double *__restrict__ x, int64_t n, double s) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] *= s;
}

}  // namespace mfm

namespace mfm {
template <int GS, int SPL, bool UNIT, bool ELL>
static void launch_score_store_t(hipStream_t s, int mode, const DevSparse &X, const ScoreStoreArgs &sa, int64_t D, int K, int KS,
                                 double *out) {
  const int64_t N = X.rows;
  const int64_t groups = (N + SCORE_RU - 1) / SCORE_RU;
  dim3 grid(cdiv(groups * GS, WG)), block(WG);
#define MFM_SS(MODE_)                                                                                                       \
  hipLaunchKernelGGL((k_score_store<GS, SPL, MODE_, UNIT, ELL>), grid, block, 0, s, X.rowptr.p, X.colidx.p, X.rval.p, sa, D, K, KS, \
                     (int)X.ell_width, out, N)
  if (mode == 0)
    MFM_SS(0);
  else if (mode == 1)
    MFM_SS(1);
  else
    MFM_SS(2);
#undef MFM_SS
}
template <int GS, int SPL>
static void launch_score_store_f(hipStream_t s, int mode, const DevSparse &X, const ScoreStoreArgs &sa, int64_t D, int K, int KS,
                                 double *out) {
  const bool ell = X.ell_width >= 0;
  if (X.unit && ell)
    launch_score_store_t<GS, SPL, true, true>(s, mode, X, sa, D, K, KS, out);
  else if (X.unit)
    launch_score_store_t<GS, SPL, true, false>(s, mode, X, sa, D, K, KS, out);
  else
    launch_score_store_t<GS, SPL, false, false>(s, mode, X, sa, D, K, KS, out);
}
// (the lane-group shapes of launch_score: the scores must be those of the per-sample pass bit for bit)
static void launch_score_store(hipStream_t s, int mode, const DevSparse &X, const ScoreStoreArgs &sa, int64_t D, int K, int KS,
                               double *out) {
#define MFM_SCORE(GS, SPL) launch_score_store_f<GS, SPL>(s, mode, X, sa, D, K, KS, out)
  if (K <= 8)
    MFM_SCORE(4, 1);
  else if (K <= 16)
    MFM_SCORE(8, 1);
  else if (K <= 32)
    MFM_SCORE(16, 1);
  else if (K <= 64)
    MFM_SCORE(32, 1);
  else if (K <= 128)
    MFM_SCORE(64, 1);
  else if (K <= 256)
    MFM_SCORE(64, 2);
  else
    MFM_SCORE(64, 4);
#undef MFM_SCORE
}
}  // namespace mfm

struct mfm_design {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  int64_t N = 0, D0 = 0, D = 0;
  DevSparse X;
  std::vector<std::unique_ptr<DevBlock>> blocks;
  // cached per rank
  int K = -1, KS = 0;
  DevBuf<double> w, V, Vt, score, out, cut;
  DevBuf<double> vt_all, w0s;      // single-pass predictor: row-major V of a chunk of samples, their w0
  DevBuf<const double *> wvp;      // ... and the samples' buffers
  PinnedRing ring;
  Timing timing;
  ~mfm_design() {
    if (stream) (void)hipStreamDestroy(stream);
  }
  void use_device() { MFM_HIP_CHECK(hipSetDevice(device)); }
};

// Posterior-sample store (FMTrainer.hpp:71-74 keeps the last n_kept_samples FM copies): the kept samples stay in HBM --
// retention is a device-to-device copy on the training stream, prediction reads them in place, the host sees a sample
// only when somebody asks for its arrays (pickling, w_samples / V_samples).
struct mfm_store {
  int device = 0;
  int64_t D = 0;
  int K = 0;
  std::string err;
  std::vector<double> w0;
  struct Sample {
    double *p = nullptr;  // w[D] then V[K][D] (factor-major, the ctx layout): inside a slab, or `own`
    DevBuf<double> own;
  };
  std::vector<std::unique_ptr<Sample>> wv;     // per sample
  std::vector<std::unique_ptr<Sample>> spare;  // buffers allocated ahead (mfm_store_reserve): no hipMalloc in the loop
  std::vector<std::unique_ptr<DevBuf<double>>> slabs;  // what the reserved samples are cut from (one allocation per <= 4 GB, not one
                                                       // per sample: a hipMalloc costs up to 0.4 ms on some boxes, 295 of them showed
                                                       // up as 12 % of a 300-iteration fit)
  hipEvent_t pushed = nullptr;  // recorded on the training stream behind the latest snapshot: readers wait for IT, not for
                                // the whole device (hipDeviceSynchronize would stall every other stream and context)
  bool pushed_valid = false;
  void use_device() { MFM_HIP_CHECK(hipSetDevice(device)); }
  void wait_pushed_host() {
    if (pushed_valid) MFM_HIP_CHECK(hipEventSynchronize(pushed));
  }
  ~mfm_store() {
    if (pushed) (void)hipEventDestroy(pushed);
  }
};

extern "C" {

int mfm_store_create(int device, int64_t D, int32_t rank, mfm_store **out) {
  *out = nullptr;
  try {
    if (mfm_device_count() <= 0) throw Error(MFM_ERR_DEVICE, "no HIP device is visible (no CPU fallback)");
    if (D < 0 || rank < 0) throw Error(MFM_ERR_INVALID, "negative size");
    std::unique_ptr<mfm_store> st(new mfm_store());
    st->device = device;
    st->D = D;
    st->K = rank;
    *out = st.release();
    return MFM_OK;
  } catch (const mfm::Error &ex) {
    g_global_error = ex.what();
    return ex.code;
  }
}
void mfm_store_destroy(mfm_store *st) {
  if (!st) return;
  (void)hipSetDevice(st->device);
  delete st;
}
const char *mfm_store_last_error(const mfm_store *st) { return st ? st->err.c_str() : g_global_error.c_str(); }
int32_t mfm_store_size(const mfm_store *st) { return (int32_t)st->wv.size(); }

static mfm_store::Sample *store_new_sample(mfm_store *st) {
  std::unique_ptr<mfm_store::Sample> b;
  if (!st->spare.empty()) {
    b = std::move(st->spare.back());
    st->spare.pop_back();
  } else {
    b.reset(new mfm_store::Sample());
    b->own.alloc((size_t)std::max<int64_t>(st->D * (st->K + 1), 1));
    b->p = b->own.p;
  }
  st->wv.push_back(std::move(b));
  return st->wv.back().get();
}

int mfm_store_reserve(mfm_store *st, int32_t n_samples) {
  MFM_TRY(st)
  {
    // the kept samples live in HBM for the Predictor's lifetime: refuse a reservation that would take more than half of what
    // is free now (MFM_STORE_MAX_FRACTION), so that a later fit / predict on the same GPU still finds room -- the trainer
    // then keeps host copies, as it does when an allocation fails
    const int64_t have = (int64_t)(st->wv.size() + st->spare.size());
    const double need = (double)std::max<int64_t>(n_samples - have, 0) * (double)std::max<int64_t>(st->D * (st->K + 1), 1) * sizeof(double);
    size_t free_b = 0, total_b = 0;
    const char *fe = std::getenv("MFM_STORE_MAX_FRACTION");
    const double frac = fe ? std::atof(fe) : 0.5;
    if (need > 0 && hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > frac * (double)free_b)
      throw Error(MFM_ERR_RUNTIME, "sample store: the reservation exceeds MFM_STORE_MAX_FRACTION of the free device memory");
  }
  {
    const size_t per = ((size_t)std::max<int64_t>(st->D * (st->K + 1), 1) + 31) & ~(size_t)31;  // doubles per sample, 256-byte multiples
    const size_t slab_max = std::max<size_t>(((size_t)4 << 30) / (per * sizeof(double)), 1);    // samples per slab
    int64_t missing = (int64_t)n_samples - (int64_t)(st->wv.size() + st->spare.size());
    while (missing > 0) {
      const size_t n = std::min<size_t>((size_t)missing, slab_max);
      std::unique_ptr<DevBuf<double>> slab(new DevBuf<double>());
      slab->alloc(n * per);
      // (handed out back to front by store_new_sample: push them so that the first sample taken is the slab's first)
      for (size_t i = n; i-- > 0;) {
        std::unique_ptr<mfm_store::Sample> b(new mfm_store::Sample());
        b->p = slab->p + i * per;
        st->spare.push_back(std::move(b));
      }
      st->slabs.push_back(std::move(slab));
      missing -= (int64_t)n;
    }
  }
  MFM_CATCH(st)
}

int mfm_store_push_ctx(mfm_store *st, mfm_ctx *ctx) {
  MFM_TRY(st)
  ctx->need_final();
  if (ctx->device != st->device) throw Error(MFM_ERR_INVALID, "store and training context live on different devices");
  if (ctx->D != st->D || ctx->K != st->K) throw Error(MFM_ERR_INVALID, "store and training context differ in size");
  mfm_store::Sample *b = store_new_sample(st);
  const size_t D = (size_t)st->D;
  hipError_t e1 = D ? hipMemcpyAsync(b->p, ctx->w.p, D * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream) : hipSuccess;
  hipError_t e2 = (D && st->K && e1 == hipSuccess)
                      ? hipMemcpyAsync(b->p + D, ctx->V.p, D * st->K * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream)
                      : hipSuccess;
  if (e1 != hipSuccess || e2 != hipSuccess) {  // (keep wv and w0 the same length)
    st->spare.push_back(std::move(st->wv.back()));
    st->wv.pop_back();
    MFM_HIP_CHECK(e1 != hipSuccess ? e1 : e2);
  }
  st->w0.push_back(ctx->w0);
  if (!st->pushed) MFM_HIP_CHECK(hipEventCreateWithFlags(&st->pushed, hipEventDisableTiming));
  MFM_HIP_CHECK(hipEventRecord(st->pushed, ctx->stream));
  st->pushed_valid = true;
  MFM_CATCH(st)
}

int mfm_store_push_host(mfm_store *st, double w0, const double *w, const double *V) {
  MFM_TRY(st)
  mfm_store::Sample *b = store_new_sample(st);
  const size_t D = (size_t)st->D;
  hipError_t e1 = D ? hipMemcpy(b->p, w, D * sizeof(double), hipMemcpyHostToDevice) : hipSuccess;
  hipError_t e2 = (D && st->K && e1 == hipSuccess) ? hipMemcpy(b->p + D, V, D * st->K * sizeof(double), hipMemcpyHostToDevice) : hipSuccess;
  if (e1 != hipSuccess || e2 != hipSuccess) {
    st->spare.push_back(std::move(st->wv.back()));
    st->wv.pop_back();
    MFM_HIP_CHECK(e1 != hipSuccess ? e1 : e2);
  }
  st->w0.push_back(w0);
  MFM_CATCH(st)
}

int mfm_store_get(mfm_store *st, int32_t idx, double *w0, double *w, double *V) {
  MFM_TRY(st)
  if (idx < 0 || idx >= (int)st->wv.size()) throw Error(MFM_ERR_INVALID, "sample index out of range");
  const size_t D = (size_t)st->D;
  st->wait_pushed_host();  // (a push_ctx copy may still be in flight on a training stream)
  if (w0) *w0 = st->w0[idx];
  if (w && D) MFM_HIP_CHECK(hipMemcpy(w, st->wv[idx]->p, D * sizeof(double), hipMemcpyDeviceToHost));
  if (V && D && st->K) MFM_HIP_CHECK(hipMemcpy(V, st->wv[idx]->p + D, D * st->K * sizeof(double), hipMemcpyDeviceToHost));
  MFM_CATCH(st)
}

int mfm_design_create(int device, int64_t N, int64_t D0, const int64_t *indptr, const int32_t *indices, const double *data,
                      mfm_design **out) {
  *out = nullptr;
  try {
    int n = mfm_device_count();
    if (n <= 0)
      throw Error(MFM_ERR_DEVICE,
                  "no HIP device is visible: libmyfm_hip.so has no CPU fallback (prediction runs on MI355X only)");
    if (device < 0 || device >= n) throw Error(MFM_ERR_INVALID, "device index out of range");
    std::unique_ptr<mfm_design> d(new mfm_design());
    d->device = device;
    d->use_device();
    MFM_HIP_CHECK(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    HostCsr X = make_host_csr(N, D0, indptr, indices, data);
    d->N = N;
    d->D0 = D0;
    d->D = D0;
    d->X.upload(X, nullptr);
    *out = d.release();
    return MFM_OK;
  } catch (const mfm::Error &ex) {
    g_global_error = ex.what();
    return ex.code;
  } catch (const std::exception &ex) {
    g_global_error = ex.what();
    return MFM_ERR_RUNTIME;
  }
}

int mfm_design_add_block(mfm_design *d, int64_t B, int64_t Db, const int64_t *indptr, const int32_t *indices,
                         const double *data, const int64_t *original_to_block) {
  MFM_TRY(d)
  HostCsr X = make_host_csr(B, Db, indptr, indices, data);
  std::vector<int32_t> m32((size_t)d->N);
  for (int64_t t = 0; t < d->N; t++) {
    if (original_to_block[t] < 0 || original_to_block[t] >= B)
      throw Error(MFM_ERR_RUNTIME, "index mapping points to non-existing row.");
    m32[t] = (int32_t)original_to_block[t];
  }
  std::unique_ptr<DevBlock> blk(new DevBlock());
  blk->B = B;
  blk->Db = Db;
  blk->nnz = X.nnz();
  blk->col_off = d->D;
  blk->X.upload(X, nullptr);
  blk->map.upload(m32);
  d->D += Db;
  d->K = -1;  // caches must be re-sized
  d->blocks.push_back(std::move(blk));
  MFM_CATCH(d)
}

void mfm_design_destroy(mfm_design *d) {
  if (!d) return;
  (void)hipSetDevice(d->device);
  if (d->stream) (void)hipStreamSynchronize(d->stream);
  delete d;
}

const char *mfm_design_last_error(const mfm_design *d) { return d ? d->err.c_str() : g_global_error.c_str(); }
int64_t mfm_design_dim_all(const mfm_design *d) { return d->D; }
int64_t mfm_design_n_rows(const mfm_design *d) { return d->N; }

int mfm_design_score_ctx(mfm_design *d, mfm_ctx *ctx, double *out) {
  MFM_TRY(d)
  ctx->need_final();
  if (ctx->device != d->device) throw Error(MFM_ERR_INVALID, "design and training context live on different devices");
  if (d->D != ctx->D) {  // FM.hpp:67-73
    throw Error(MFM_ERR_INVALID, "Total feature size mismatch. Should be " + std::to_string(ctx->D) + ", but got " +
                                     std::to_string(d->D) + ".");
  }
  const int rank = ctx->K;
  hipStream_t s = ctx->stream;  // ordered after the sweeps that produced the state
  if (d->K != rank) {
    d->K = rank;
    d->KS = (rank + 1) & ~1;
    d->Vt.alloc_zero((size_t)std::max<int64_t>(d->D * d->KS, 1), s);
    d->score.alloc((size_t)std::max<int64_t>(d->N, 1));
    d->w.alloc((size_t)std::max<int64_t>(d->D, 1));
    d->V.alloc((size_t)std::max<int64_t>(d->D * rank, 1));
    for (auto &B : d->blocks) {
      B->bq.alloc_zero((size_t)B->B * std::max(d->KS, 1), s);
      B->bl.alloc_zero((size_t)B->B, s);
      B->bs.alloc_zero((size_t)B->B, s);
    }
  }
  score_design(s, ctx->timing, 1, d->X, d->blocks, d->D, rank, d->KS, ctx->w0, ctx->w.p, ctx->V.p, d->Vt.p, nullptr, nullptr,
               d->score.p);
  if (d->N) MFM_HIP_CHECK(hipMemcpyAsync(out, d->score.p, (size_t)d->N * sizeof(double), hipMemcpyDeviceToHost, s));
  MFM_HIP_CHECK(hipStreamSynchronize(s));
  MFM_CATCH(d)
}

// Predictor::predict* over samples resident in a store (no per-sample upload, no host synchronisation inside the loop):
// samples [first, first + count).
int mfm_design_predict_store(mfm_design *d, mfm_store *st, int32_t first, int32_t count, int32_t mode, int32_t n_cut,
                             const double *cutpoints, double *out) {
  MFM_TRY(d)
  if (count <= 0) throw Error(MFM_ERR_RUNTIME, "Told to predict but no sample available.");  // predictor.hpp:39-41
  if (first < 0 || first + count > (int)st->wv.size()) throw Error(MFM_ERR_INVALID, "sample range out of bounds");
  if (st->device != d->device) throw Error(MFM_ERR_INVALID, "design and sample store live on different devices");
  if (st->D != d->D) throw Error(MFM_ERR_INVALID, "feature size mismatch!");
  if (mode < 0 || mode > 2) throw Error(MFM_ERR_INVALID, "bad prediction mode");
  if (mode == 2 && n_cut < 1) throw Error(MFM_ERR_RUNTIME, "No cutpoint available for this FM.");
  hipStream_t s = d->stream;
  const int rank = st->K;
  const int64_t N = d->N, D = d->D;
  if (d->K != rank) {
    d->K = rank;
    d->KS = (rank + 1) & ~1;
    d->w.alloc((size_t)std::max<int64_t>(D, 1));
    d->V.alloc((size_t)std::max<int64_t>(D * rank, 1));
    d->Vt.alloc_zero((size_t)std::max<int64_t>(D * d->KS, 1), s);
    d->score.alloc((size_t)std::max<int64_t>(N, 1));
    for (auto &B : d->blocks) {
      B->bq.alloc_zero((size_t)B->B * std::max(d->KS, 1), s);
      B->bl.alloc_zero((size_t)B->B, s);
      B->bs.alloc_zero((size_t)B->B, s);
    }
  }
  const int64_t out_n = mode == 2 ? N * (n_cut + 1) : N;
  if (d->out.n < (size_t)std::max<int64_t>(out_n, 1)) d->out.alloc((size_t)std::max<int64_t>(out_n, 1));
  // the samples' device-to-device copies (training stream) must be complete: this stream waits for the latest one's event
  if (st->pushed_valid) MFM_HIP_CHECK(hipStreamWaitEvent(s, st->pushed, 0));
  if (mode == 2) {
    if (d->cut.n < (size_t)n_cut * count) d->cut.alloc((size_t)n_cut * count);
    MFM_HIP_CHECK(hipMemcpyAsync(d->cut.p, cutpoints, (size_t)n_cut * count * sizeof(double), hipMemcpyHostToDevice, s));
  }
  // designs without relation blocks: every sample in ONE pass over the test rows (k_score_store); chunks of samples whose
  // row-major V copies fit 512 MB (the buffer is allocated per design: a larger one costs more than it saves)
  if (d->blocks.empty() && N > 0 && rank <= 512 && (mode != 2 || n_cut + 1 <= PRED_MAX_CLASS) && !std::getenv("MFM_PREDICT_PER_SAMPLE")) {
    const size_t per = (size_t)std::max<int64_t>(D * d->KS, 1) * sizeof(double);
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)count, ((size_t)512 << 20) / per));
    if (d->vt_all.n < (size_t)chunk * (per / sizeof(double))) d->vt_all.alloc((size_t)chunk * (per / sizeof(double)));
    std::vector<const double *> hp((size_t)count);
    std::vector<double> hw0((size_t)count);
    for (int k = 0; k < count; k++) {
      hp[k] = st->wv[first + k]->p;
      hw0[k] = st->w0[first + k];
    }
    if (d->wvp.n < (size_t)count) d->wvp.alloc((size_t)count);
    if (d->w0s.n < (size_t)count) d->w0s.alloc((size_t)count);
    MFM_HIP_CHECK(hipMemcpyAsync(d->wvp.p, hp.data(), (size_t)count * sizeof(double *), hipMemcpyHostToDevice, s));
    MFM_HIP_CHECK(hipMemcpyAsync(d->w0s.p, hw0.data(), (size_t)count * sizeof(double), hipMemcpyHostToDevice, s));
    MFM_HIP_CHECK(hipStreamSynchronize(s));  // (hp / hw0 are pageable host vectors of this frame)
    for (int c0 = 0; c0 < count; c0 += chunk) {
      const int C = std::min(chunk, count - c0);
      if (rank > 0 && D > 0)
        hipLaunchKernelGGL(k_build_vt_batch, dim3((unsigned)cdiv(D, 32), (unsigned)cdiv(d->KS, 32), (unsigned)C), dim3(WG), 0, s,
                           (const double *const *)d->wvp.p + c0, D, rank, d->KS, d->vt_all.p);
      ScoreStoreArgs sa;
      sa.wv = (const double *const *)d->wvp.p + c0;
      sa.vt_all = d->vt_all.p;
      sa.w0 = d->w0s.p + c0;
      sa.cut = mode == 2 ? d->cut.p + (size_t)c0 * n_cut : nullptr;
      sa.S = C;
      sa.n_cut = mode == 2 ? n_cut : 0;
      sa.first = c0 == 0;
      sa.scale = c0 + C == count ? 1.0 / count : 1.0;
      launch_score_store(s, mode, d->X, sa, D, rank, d->KS, d->out.p);
    }
    MFM_HIP_CHECK(hipGetLastError());
    if (out_n) MFM_HIP_CHECK(hipMemcpyAsync(out, d->out.p, (size_t)out_n * sizeof(double), hipMemcpyDeviceToHost, s));
    MFM_HIP_CHECK(hipStreamSynchronize(s));
    return MFM_OK;
  }
  for (int k = 0; k < count; k++) {
    const double *w = st->wv[first + k]->p, *V = w + D;
    score_design(s, d->timing, 1, d->X, d->blocks, D, rank, d->KS, st->w0[first + k], w, V, d->Vt.p, nullptr, nullptr,
                 d->score.p);
    if (N) {
      if (mode == 2)
        hipLaunchKernelGGL(k_accumulate_oprobit, dim3(cdiv(N, WG)), dim3(WG), 0, s, d->score.p, d->cut.p + (size_t)k * n_cut, n_cut,
                           d->out.p, N, k == 0);
      else
        hipLaunchKernelGGL(k_accumulate_pred, dim3(cdiv(N, WG)), dim3(WG), 0, s, d->score.p, d->out.p, N, mode, k == 0);
      MFM_HIP_CHECK(hipGetLastError());
    }
  }
  if (out_n) {
    hipLaunchKernelGGL(k_scale, dim3(cdiv(out_n, WG)), dim3(WG), 0, s, d->out.p, out_n, 1.0 / count);
    MFM_HIP_CHECK(hipMemcpyAsync(out, d->out.p, (size_t)out_n * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  MFM_HIP_CHECK(hipStreamSynchronize(s));
  MFM_CATCH(d)
}

int mfm_design_predict(mfm_design *d, int32_t rank, int32_t n_samples, const double *w0s, const double *ws,
                       const double *Vs, int32_t mode, int32_t n_cut, const double *cutpoints, double *out) {
  MFM_TRY(d)
  if (n_samples <= 0) throw Error(MFM_ERR_RUNTIME, "Told to predict but no sample available.");  // predictor.hpp:39-41
  if (rank < 0) throw Error(MFM_ERR_INVALID, "rank must be non-negative");
  if (mode < 0 || mode > 2) throw Error(MFM_ERR_INVALID, "bad prediction mode");
  if (mode == 2 && n_cut < 1) throw Error(MFM_ERR_RUNTIME, "No cutpoint available for this FM.");  // FM.hpp:141-143
  hipStream_t s = d->stream;
  const int64_t N = d->N, D = d->D;
  if (d->K != rank) {
    d->K = rank;
    d->KS = (rank + 1) & ~1;
    d->w.alloc((size_t)std::max<int64_t>(D, 1));
    d->V.alloc((size_t)std::max<int64_t>(D * rank, 1));
    d->Vt.alloc_zero((size_t)std::max<int64_t>(D * d->KS, 1), s);
    d->score.alloc((size_t)std::max<int64_t>(N, 1));
    for (auto &B : d->blocks) {
      B->bq.alloc_zero((size_t)B->B * std::max(d->KS, 1), s);
      B->bl.alloc_zero((size_t)B->B, s);
      B->bs.alloc_zero((size_t)B->B, s);
    }
  }
  const int64_t out_n = mode == 2 ? N * (n_cut + 1) : N;
  if (d->out.n < (size_t)std::max<int64_t>(out_n, 1)) d->out.alloc((size_t)std::max<int64_t>(out_n, 1));
  if (mode == 2 && d->cut.n < (size_t)n_cut) d->cut.alloc((size_t)n_cut);
  for (int smp = 0; smp < n_samples; smp++) {
    if (D) d->ring.upload(d->w.p, ws + (size_t)smp * D, (size_t)D * sizeof(double), s);
    if (D && rank) d->ring.upload(d->V.p, Vs + (size_t)smp * D * rank, (size_t)D * rank * sizeof(double), s);
    score_design(s, d->timing, 1, d->X, d->blocks, D, rank, d->KS, w0s[smp], d->w.p, d->V.p, d->Vt.p, nullptr, nullptr,
                 d->score.p);
    if (N) {
      if (mode == 2) {
        d->ring.upload(d->cut.p, cutpoints + (size_t)smp * n_cut, (size_t)n_cut * sizeof(double), s);
        hipLaunchKernelGGL(k_accumulate_oprobit, dim3(cdiv(N, WG)), dim3(WG), 0, s, d->score.p, d->cut.p, n_cut, d->out.p,
                           N, smp == 0);
      } else {
        hipLaunchKernelGGL(k_accumulate_pred, dim3(cdiv(N, WG)), dim3(WG), 0, s, d->score.p, d->out.p, N, mode, smp == 0);
      }
      MFM_HIP_CHECK(hipGetLastError());
    }
  }
  if (out_n) {
    hipLaunchKernelGGL(k_scale, dim3(cdiv(out_n, WG)), dim3(WG), 0, s, d->out.p, out_n, 1.0 / n_samples);
    MFM_HIP_CHECK(hipMemcpyAsync(out, d->out.p, (size_t)out_n * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  MFM_HIP_CHECK(hipStreamSynchronize(s));
  MFM_CATCH(d)
}

}  // extern "C"
