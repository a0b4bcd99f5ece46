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
__global__ void k_scale(double *__restrict__ x, int64_t n, double s) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] *= s;
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
  std::vector<std::unique_ptr<DevBuf<double>>> wv;  // per sample: w[D] then V[K][D] (factor-major, the ctx layout)
  std::vector<std::unique_ptr<DevBuf<double>>> spare;  // buffers allocated ahead (mfm_store_reserve): no hipMalloc in the loop
  void use_device() { MFM_HIP_CHECK(hipSetDevice(device)); }
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

static DevBuf<double> *store_new_sample(mfm_store *st) {
  std::unique_ptr<DevBuf<double>> b;
  if (!st->spare.empty()) {
    b = std::move(st->spare.back());
    st->spare.pop_back();
  } else {
    b.reset(new DevBuf<double>());
    b->alloc((size_t)std::max<int64_t>(st->D * (st->K + 1), 1));
  }
  st->wv.push_back(std::move(b));
  return st->wv.back().get();
}

int mfm_store_reserve(mfm_store *st, int32_t n_samples) {
  MFM_TRY(st)
  while ((int)(st->wv.size() + st->spare.size()) < n_samples) {
    std::unique_ptr<DevBuf<double>> b(new DevBuf<double>());
    b->alloc((size_t)std::max<int64_t>(st->D * (st->K + 1), 1));
    st->spare.push_back(std::move(b));
  }
  MFM_CATCH(st)
}

int mfm_store_push_ctx(mfm_store *st, mfm_ctx *ctx) {
  MFM_TRY(st)
  ctx->need_final();
  if (ctx->device != st->device) throw Error(MFM_ERR_INVALID, "store and training context live on different devices");
  if (ctx->D != st->D || ctx->K != st->K) throw Error(MFM_ERR_INVALID, "store and training context differ in size");
  DevBuf<double> *b = store_new_sample(st);
  const size_t D = (size_t)st->D;
  if (D) MFM_HIP_CHECK(hipMemcpyAsync(b->p, ctx->w.p, D * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  if (D && st->K)
    MFM_HIP_CHECK(hipMemcpyAsync(b->p + D, ctx->V.p, D * st->K * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  st->w0.push_back(ctx->w0);
  MFM_CATCH(st)
}

int mfm_store_push_host(mfm_store *st, double w0, const double *w, const double *V) {
  MFM_TRY(st)
  DevBuf<double> *b = store_new_sample(st);
  const size_t D = (size_t)st->D;
  if (D) MFM_HIP_CHECK(hipMemcpy(b->p, w, D * sizeof(double), hipMemcpyHostToDevice));
  if (D && st->K) MFM_HIP_CHECK(hipMemcpy(b->p + D, V, D * st->K * sizeof(double), hipMemcpyHostToDevice));
  st->w0.push_back(w0);
  MFM_CATCH(st)
}

int mfm_store_get(mfm_store *st, int32_t idx, double *w0, double *w, double *V) {
  MFM_TRY(st)
  if (idx < 0 || idx >= (int)st->wv.size()) throw Error(MFM_ERR_INVALID, "sample index out of range");
  const size_t D = (size_t)st->D;
  MFM_HIP_CHECK(hipDeviceSynchronize());  // (a push_ctx copy may still be in flight on a training stream)
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
  if ((int)d->blocks.size() >= MAX_BLOCKS) throw Error(MFM_ERR_INVALID, "too many relation blocks (max 16)");
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
  MFM_HIP_CHECK(hipDeviceSynchronize());  // the samples' device-to-device copies (training stream) are complete
  if (mode == 2) {
    if (d->cut.n < (size_t)n_cut * count) d->cut.alloc((size_t)n_cut * count);
    MFM_HIP_CHECK(hipMemcpyAsync(d->cut.p, cutpoints, (size_t)n_cut * count * sizeof(double), hipMemcpyHostToDevice, s));
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
