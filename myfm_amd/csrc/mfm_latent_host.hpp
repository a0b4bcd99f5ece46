// mfm_latent_host.hpp -- entry points of the exact latent draws (mfm_latent.hip) and of the host's window into the device random
// stream; included by mfm_hip.hip (needs mfm_ctx and the generator kernels of mfm_rng.hpp).
//
// In latent mode "exact" the trainer's std::mt19937 lives on the device for the whole fit, as for regression. What the reference
// draws from it in a state-dependent order is served from the same stream:
//   * the latent z of classification / ordered probit: mfm_update_e_classification_exact / mfm_oprobit_sample_z_exact (parallel,
//     mfm_latent.hip), or -- when a window of the parallel evaluation misses -- the caller's own sequential loop over
//   * mfm_rng_host_read / mfm_rng_host_advance: the engine outputs from the stream's position on, tempered, for the host
//     (the cutpoint sampler's Metropolis draws, OProbitSampler.hpp:55-72, :378, and the sequential fall-back).
#pragma once

namespace mfm {

__global__ void k_rng_copy_window(const uint32_t *__restrict__ src, uint64_t smask, uint32_t *__restrict__ dst, uint64_t dmask, uint64_t lo,
                                  uint64_t hi) {
  const uint64_t i = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < hi) dst[i & dmask] = src[i & smask];
}
__global__ void k_rng_temper_out(const RngState *__restrict__ st, const uint32_t *__restrict__ raw, uint64_t mask, uint64_t offset, int64_t n,
                                 uint32_t *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = mt_temper(raw[(st->p_cons + offset + (uint64_t)i) & mask]);
}
__global__ void k_rng_advance(RngState *__restrict__ st, uint64_t words) {
  st->p_cons += words;
  if (st->p_cons > st->p_gen) st->error = 1;
}

// engine outputs [p_cons, p_cons + need) exist in the ring when this returns (enqueued on stream s; p_cons / p_gen as the host knows
// them). Grows the ring when it cannot hold them.
static void rng_ensure_generated(mfm_ctx *ctx, hipStream_t s, uint64_t p_cons, uint64_t p_gen, uint64_t need) {
  auto &r = ctx->rng;
  need += 2 * MT_N;
  if (need + 2 * MT_N > r.mask + 1) {  // a larger ring: the outputs not yet consumed move over
    uint64_t cap = r.mask + 1;
    while (cap < need + r.need_gen + 4 * MT_N) cap <<= 1;
    DevBuf<uint32_t> bigger;
    bigger.alloc((size_t)cap);
    MFM_HIP_CHECK(hipStreamSynchronize(r.stream));
    if (p_gen > p_cons)
      hipLaunchKernelGGL(k_rng_copy_window, dim3((unsigned)cdiv((int64_t)(p_gen - p_cons), 256)), dim3(256), 0, s, r.raw.p, r.mask, bigger.p,
                         cap - 1, p_cons, p_gen);
    MFM_HIP_CHECK(hipStreamSynchronize(s));
    r.raw = std::move(bigger);
    r.mask = cap - 1;
  }
  uint64_t have = p_gen - p_cons;
  if (have >= need) return;
  if (r.par_wgs > 1) {
    const uint64_t per_launch = (uint64_t)r.par_wgs * (uint64_t)r.par_blocks * MT_N - 2 * MT_N;
    while (have < need) {
      const uint64_t want = std::min<uint64_t>(need, have + per_launch);
      hipLaunchKernelGGL(k_mt_generate_par<1>, dim3(r.par_wgs), dim3(MT_GEN_THREADS),
                         (MT_JUMP_SPAN * MT_N + 2 * (MT_N + 1)) * sizeof(uint32_t), s, r.state.p, r.state_next.p, r.raw.p, r.mask, want,
                         r.jump.p, r.par_blocks, r.starts.p, want);
      hipLaunchKernelGGL(k_mt_generate_par<2>, dim3(r.par_wgs), dim3(MT_GEN_THREADS), 2 * (MT_N + 1) * sizeof(uint32_t), s, r.state.p,
                         r.state_next.p, r.raw.p, r.mask, want, r.jump.p, r.par_blocks, r.starts.p, want);
      hipLaunchKernelGGL(k_mt_commit, dim3(1), dim3(MT_GEN_THREADS), 0, s, r.state.p, r.state_next.p);
      have = want;
    }
  } else {
    hipLaunchKernelGGL(k_mt_generate, dim3(1), dim3(MT_GEN_THREADS), 0, s, r.state.p, r.raw.p, r.mask, need);
  }
  MFM_HIP_CHECK(hipGetLastError());
}

// the side stream's work (a prefetched set still being produced) must be over before the main stream touches the generator state
static void rng_join_side_stream(mfm_ctx *ctx) {
  auto &r = ctx->rng;
  if (r.stream) MFM_HIP_CHECK(hipStreamSynchronize(r.stream));
}

static int latent_exact_run(mfm_ctx *ctx, int64_t n, const int32_t *rows, int n_class, const double *gamma_dev, int32_t *status) {
  auto &r = ctx->rng;
  if (!r.seeded || !r.programmed) throw Error(MFM_ERR_RUNTIME, "exact latent draws need the device random stream (mfm_rng_seed_mt19937 + mfm_rng_set_program)");
  if (r.produced != r.acquired) throw Error(MFM_ERR_RUNTIME, "exact latent draws: a prefetched random set is in flight (the stream's position is not final)");
  rng_join_side_stream(ctx);
  if (!ctx->latent) ctx->latent.reset(new LatentEngine());
  hipStream_t s = ctx->stream;
  LatentJob job;
  job.stream = s;
  job.n = n;
  job.rows = rows;
  job.eq = ctx->eq_rows();
  job.y = ctx->y.p;
  job.n_class = n_class;
  job.gamma = gamma_dev;
  job.state = r.state.p;
  job.mask = r.mask;
  LatentPrep prep;
  LatentStats st;
  {
    TimedLaunch t(ctx->timing, s, KC_TN_SAMPLE, 24.0 * n);
    ctx->latent->prepare(job, &prep);
    rng_ensure_generated(ctx, s, prep.p_cons, prep.p_gen, 4ull * (uint64_t)prep.q_cap);
    job.raw = r.raw.p;
    job.mask = r.mask;
    prep.p_gen = std::max<uint64_t>(prep.p_gen, prep.p_cons + 4ull * (uint64_t)prep.q_cap);
    ctx->latent->run(job, prep, &st);
  }
  ctx->latent_stats = st;
  *status = st.status;
  if (st.status == 0) {
    ctx->e_is_residual = false;
    r.latent_pending = true;  // the next prefetch starts behind this draw (its event is recorded there)
  }
  return MFM_OK;
}

}  // namespace mfm

extern "C" {

int mfm_set_latent_order(mfm_ctx *ctx, const int64_t *rows, int64_t n) {
  MFM_TRY(ctx)
  ctx->need_final();
  if (n == 0) {
    ctx->latent_order.release();
    return MFM_OK;
  }
  if (n != ctx->N) throw Error(MFM_ERR_INVALID, "the latent row order must list every row once");
  std::vector<int32_t> r32((size_t)n);
  std::vector<uint8_t> seen((size_t)n, 0);
  for (int64_t i = 0; i < n; i++) {
    if (rows[i] < 0 || rows[i] >= n || seen[(size_t)rows[i]]) throw Error(MFM_ERR_INVALID, "the latent row order must list every row once");
    seen[(size_t)rows[i]] = 1;
    r32[(size_t)i] = (int32_t)rows[i];
  }
  MFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  ctx->latent_order.upload(r32);
  MFM_CATCH(ctx)
}

int mfm_update_e_classification_exact(mfm_ctx *ctx, int32_t *status) {
  MFM_TRY(ctx)
  ctx->need_final();
  if (ctx->comm.active()) throw Error(MFM_ERR_RUNTIME, "exact latent draws are not available on row-sharded fits");
  score_train(ctx, false);
  int32_t st = 0;
  if (ctx->N) latent_exact_run(ctx, ctx->N, ctx->latent_order.p, 0, nullptr, &st);
  if (status) *status = st;
  MFM_CATCH(ctx)
}

int mfm_oprobit_sample_z_exact(mfm_ctx *ctx, int32_t group, const double *gamma, int32_t *status) {
  MFM_TRY(ctx)
  ctx->need_final();
  if (ctx->comm.active()) throw Error(MFM_ERR_RUNTIME, "exact latent draws are not available on row-sharded fits");
  materialize_e(ctx);
  if (group < 0 || group >= (int)ctx->ogroups.size()) throw Error(MFM_ERR_INVALID, "bad cutpoint group");
  mfm_ctx::OGroup &g = *ctx->ogroups[group];
  double *dgam = ctx->opartial.p + (size_t)OPROBIT_BLOCKS * ctx->opartial_cmax * OPROBIT_SLOTS;
  ctx->ring.upload(dgam, gamma, (size_t)(g.n_class - 1) * sizeof(double), ctx->stream);
  int32_t st = 0;
  if (g.n_rows) latent_exact_run(ctx, g.n_rows, g.rows.p, g.n_class, dgam, &st);
  if (status) *status = st;
  MFM_CATCH(ctx)
}

int mfm_latent_stats(mfm_ctx *ctx, int64_t *out8) {
  MFM_TRY(ctx)
  const LatentStats &s = ctx->latent_stats;
  out8[0] = s.status;
  out8[1] = s.chunks;
  out8[2] = s.subs;
  out8[3] = s.lq;
  out8[4] = s.quads_used;
  out8[5] = s.walkers;
  out8[6] = s.attempts;
  out8[7] = 0;
  MFM_CATCH(ctx)
}

int mfm_rng_host_read(mfm_ctx *ctx, uint64_t offset, int64_t n, uint32_t *out) {
  MFM_TRY(ctx)
  auto &r = ctx->rng;
  if (!r.seeded || !r.programmed) throw Error(MFM_ERR_RUNTIME, "the device random stream has not been set up");
  if (r.produced != r.acquired) throw Error(MFM_ERR_RUNTIME, "mfm_rng_host_read: a prefetched random set is in flight");
  if (n <= 0) return MFM_OK;
  rng_join_side_stream(ctx);
  hipStream_t s = ctx->stream;
  RngState hdr;
  MFM_HIP_CHECK(hipMemcpyAsync(&hdr, r.state.p, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
  MFM_HIP_CHECK(hipStreamSynchronize(s));
  rng_ensure_generated(ctx, s, hdr.p_cons, hdr.p_gen, offset + (uint64_t)n);
  if (r.host_win.n < (size_t)n) r.host_win.alloc((size_t)n);
  hipLaunchKernelGGL(k_rng_temper_out, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, r.state.p, r.raw.p, r.mask, offset, n, r.host_win.p);
  MFM_HIP_CHECK(hipGetLastError());
  MFM_HIP_CHECK(hipMemcpyAsync(out, r.host_win.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  MFM_HIP_CHECK(hipStreamSynchronize(s));
  MFM_CATCH(ctx)
}

int mfm_rng_host_advance(mfm_ctx *ctx, uint64_t words) {
  MFM_TRY(ctx)
  auto &r = ctx->rng;
  if (!r.seeded) throw Error(MFM_ERR_RUNTIME, "the device random stream has not been set up");
  if (r.produced != r.acquired) throw Error(MFM_ERR_RUNTIME, "mfm_rng_host_advance: a prefetched random set is in flight");
  rng_join_side_stream(ctx);
  if (words) {
    hipLaunchKernelGGL(k_rng_advance, dim3(1), dim3(1), 0, ctx->stream, r.state.p, words);
    r.latent_pending = true;
  }
  MFM_CATCH(ctx)
}

}  // extern "C"
