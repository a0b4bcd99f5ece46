/* myfm_hip.h -- C ABI of libmyfm_hip.so: the MI355X (gfx950) Gibbs-sampler hot path of
 * Bayesian Factorization Machines, as the host layer behind myFM's pybind11 boundary binds it.
 *
 * What this replaces in the reference (paths relative to /root/reference):
 *   the arithmetic of GibbsFMTrainer::update_all (include/myfm/BaseFMTrainer.hpp:135-152),
 *   i.e. include/myfm/FMTrainer.hpp:127-522, the scorer FM::predict_score_write_target
 *   (include/myfm/FM.hpp:54-136) and Predictor::predict* (include/myfm/predictor.hpp:35-147).
 * Who calls it: myfm_amd/csrc/_myfm.cpp -- a pybind11 module with the reference's names
 *   (cpp_source/declare_module.hpp:67-404) whose create_train_fm drives these entry points
 *   exactly where the reference's create_train_fm (declare_module.hpp:30-45) drives Eigen.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer argument is HOST memory owned by the
 *     caller and may be freed as soon as the call returns (the library copies / uploads);
 *   - the ctx owns all device memory and one HIP stream; one host thread per ctx;
 *   - matrices: CSR with int64 indptr, int32 indices, f64 data (scipy layout after
 *     base.py:285-286); dense V is column-major (D, K) like the reference
 *     (definitions.hpp:17), hyper matrices mu_V / lambda_V are column-major (G, K), i.e.
 *     element (g, f) at [f * G + g] (HyperParams.hpp:18-19);
 *   - every call returns MFM_OK or an error code; mfm_last_error() gives the message. The
 *     host layer maps MFM_ERR_INVALID -> ValueError (std::invalid_argument in the
 *     reference) and MFM_ERR_RUNTIME / MFM_ERR_DEVICE -> RuntimeError;
 *   - there is NO CPU fallback: without a usable HIP device mfm_create fails with
 *     MFM_ERR_DEVICE.
 */
#ifndef MYFM_HIP_H_
#define MYFM_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFM_OK 0
#define MFM_ERR_INVALID 1 /* std::invalid_argument in the reference -> ValueError   */
#define MFM_ERR_RUNTIME 2 /* std::runtime_error in the reference    -> RuntimeError */
#define MFM_ERR_DEVICE 3  /* HIP failure / no device                -> RuntimeError */

/* task types: FMLearningConfig.hpp:14 (TASKTYPE) */
#define MFM_TASK_REGRESSION 0
#define MFM_TASK_CLASSIFICATION 1
#define MFM_TASK_ORDERED 2

typedef struct mfm_ctx mfm_ctx;       /* one training problem resident on one GPU            */
typedef struct mfm_design mfm_design; /* a (main CSR, relation blocks) design for prediction */
typedef struct mfm_store mfm_store;   /* posterior samples kept resident in HBM                */

/* ---- library / device ---------------------------------------------------------------- */
const char *mfm_version(void);
/* number of visible HIP devices; 0 when there is none (never fails). */
int mfm_device_count(void);
/* last error of a call that had no ctx (mfm_create, mfm_design_create) on this thread. */
const char *mfm_global_error(void);

/* ---- training context: replaces the BaseFMTrainer ctor, BaseFMTrainer.hpp:58-105 ------ */
int mfm_create(int device, mfm_ctx **out);
void mfm_destroy(mfm_ctx *ctx);
const char *mfm_last_error(const mfm_ctx *ctx);
int mfm_get_device(const mfm_ctx *ctx); /* the HIP device index the context lives on */
/* use an existing hipStream_t (e.g. torch's current stream) instead of the ctx's own. */
int mfm_set_stream(mfm_ctx *ctx, void *hip_stream);
int mfm_synchronize(mfm_ctx *ctx);

/* ---- row-sharded multi-GPU mode (one process per GPU) ---------------------------------
 * Every rank holds a contiguous slice of the training rows (its slice of X, y and of every
 * original_to_block) and a full replica of the model state, hyper-parameters and random variates.
 * `fn(user, dev_buf, count)` must sum `count` doubles at device address `dev_buf` IN PLACE over all
 * ranks, enqueued in order on the ctx stream (e.g. torch.distributed.all_reduce on the RCCL backend with
 * the ctx stream made current, see mfm_set_stream), and return 0. The library calls it once per level
 * of every sweep (2 |level| doubles), once per (block, factor) for the block statistics, and for
 * sum e / sum e^2 and the ordered-probit likelihood terms. Must be called before mfm_finalize.        */
int mfm_set_allreduce(mfm_ctx *ctx, int (*fn)(void *user, void *dev_buf, int64_t count), void *user);
/* ... with the callback provider: this rank's index and the number of ranks. REQUIRED together with mfm_set_allreduce
 * (mfm_comm_init implies it): rank 0 alone contributes the replicated columns to the model synchronisation after a
 * sharded latent sweep. A caller that never says (the older sequence mfm_set_allreduce + mfm_set_row_offset) is taken
 * to be the root exactly when its shard starts at global row 0.                                                       */
int mfm_set_shard(mfm_ctx *ctx, int32_t rank, int32_t world);
/* Native provider: RCCL called from this library on the ctx stream (ncclAllReduce, fp64 sum, in place) -- no callback,
 * no interpreter in the loop. One rank obtains the 128-byte id (ncclGetUniqueId) and hands it to the others by any
 * out-of-band means (e.g. a torch.distributed / MPI broadcast); every rank then calls mfm_comm_init (collective:
 * ncclCommInitRank on the ctx's device) before mfm_finalize. librccl.so is bound at run time.                          */
int mfm_comm_unique_id(void *out128);
int mfm_comm_init(mfm_ctx *ctx, const void *id128, int32_t rank, int32_t world);
/* collectives issued so far and doubles they carried (either provider) */
int mfm_comm_stats(const mfm_ctx *ctx, int64_t *calls, int64_t *doubles);
/* native provider: the communicator's own rank count (ncclCommCount) and the path of the librccl.so that was bound
 * (resolution order: one already mapped into the process, the directory of the HIP runtime in use, the loader's search
 * path, /opt/rocm/lib). n_ranks = 0 and an empty path when the ctx has no native communicator.                         */
int mfm_comm_info(const mfm_ctx *ctx, int32_t *n_ranks, char *path, int64_t path_cap);
/* Row-sharded persistent sweep (two-field one-hot table, shards cut between users): mfm_finalize builds the sweep's layout on every
 * rank and allocates the rank's exchange buffers; the sweep goes live once every rank knows every rank's buffers -- until then the
 * per-factor passes run. Inside the launch a rank writes its item sums into every peer's buffer and raises a flag there (no
 * collective between the launches; SURVEY 8e "one-shot all-reduce"). mfm_peer_info: is a layout waiting (pending), this rank's
 * buffers. mfm_peer_set: all ranks' device pointers, valid on THIS device (ranks in one process, or mapped by the caller).
 * mfm_peer_export / mfm_peer_import: the same through IPC handles for one process per GPU (256 bytes per rank: sums, flags, w, V; exchanged by the
 * caller over any channel). Every rank must make the same calls between the same two sweeps.                                  */
int mfm_peer_info(mfm_ctx *ctx, int32_t *pending, void **sum_buf, void **flag_buf, int64_t *sum_bytes, int64_t *flag_bytes);
int mfm_peer_set(mfm_ctx *ctx, int32_t world, int32_t rank, void *const *sum_bufs, void *const *flag_bufs);
/* optional, after mfm_peer_set: every rank's w / V (mfm_peer_model_info gives this rank's). A first-level coefficient is then written
 * to every replica where and when it is drawn, and the model synchronisation after the launch (an all-reduce of w and V) is dropped. */
int mfm_peer_model_info(mfm_ctx *ctx, void **w_buf, void **V_buf);
int mfm_peer_set_model(mfm_ctx *ctx, int32_t world, int32_t rank, void *const *w_bufs, void *const *V_bufs);
int mfm_peer_export(mfm_ctx *ctx, void *handles256);
int mfm_peer_import(mfm_ctx *ctx, int32_t world, int32_t rank, const void *all_handles);
/* give it up (waiting or live): the per-factor passes from the next sweep on; every rank alike */
int mfm_peer_drop(mfm_ctx *ctx);
/* The level schedule of the main table's columns (mfm_host_column_levels of the GLOBAL design): in the
 * row-sharded mode it must be identical on every rank (a conflict may exist only in another rank's
 * rows), so the caller computes it before sharding. Checked against the local rows at mfm_finalize.
 * Relation-block matrices are replicated, their schedules need no hand-over.                          */
int mfm_set_main_levels(mfm_ctx *ctx, const int32_t *level, int64_t D0);
/* global index of this rank's first row: keys the per-row Philox streams so that the latent draws of
 * classification / ordered probit do not depend on how the rows are sharded.                         */
int mfm_set_row_offset(mfm_ctx *ctx, int64_t first_global_row);

/* main table X (N x D0) and targets y[N]; D0 may be 0 (base.py:230-233). The arrays are validated and copied to the device
 * before the call returns (no host copy is kept); the caller may free them afterwards.                                      */
int mfm_set_main(mfm_ctx *ctx, int64_t N, int64_t D0, const int64_t *indptr, const int32_t *indices,
                 const double *data, const double *y);
/* RelationBlock (definitions.hpp:30-52): block CSR (B x Db) + original_to_block[N].
 * MFM_ERR_RUNTIME "index mapping points to non-existing row." on a bad index (:38-41).    */
int mfm_add_block(mfm_ctx *ctx, int64_t B, int64_t Db, const int64_t *indptr, const int32_t *indices,
                  const double *data, const int64_t *original_to_block);
/* group_index over ALL D = D0 + sum Db features (FMLearningConfig.hpp:83), G groups.      */
int mfm_set_groups(mfm_ctx *ctx, const int32_t *group_index, int64_t D, int32_t G);
/* recomputable = 1 (regression): outside the sweeps the residual is always e = score - y -- mfm_update_e_regression follows every
 * update_V (FMTrainer.hpp:494) -- so a sweep that keeps the residual on chip need not write its copy back; whoever reads the
 * residual between the sweep and the update gets it recomputed (equal up to rounding). Default 0: every sweep leaves its residual. */
int mfm_set_residual_policy(mfm_ctx *ctx, int32_t recomputable);
/* Builds the device-side design: CSC (= X_t, BaseFMTrainer.hpp:61), the conflict-free level
 * schedule of the columns, the residual/q-cache vectors. rank = n_factors (may be 0).      */
int mfm_finalize(mfm_ctx *ctx, int32_t rank);

int64_t mfm_dim_all(const mfm_ctx *ctx);
/* schedule statistics: number of main-table levels, of main-table kernel launches per sweep */
int mfm_plan_info(const mfm_ctx *ctx, int64_t *n_levels_main, int64_t *n_launches_per_sweep);
/* Which fast paths mfm_finalize selected for the main table (diagnostics / tests):
 * bit 0 = q-free latent sweep (short rows: q_train is recomputed, not stored, during update_V),
 * bit 1 = all stored values are 1.0 (values never read), bit 2 = uniform row length (rowptr never read),
 * bit 3 = row-sharded (mfm_set_allreduce), bit 4 = split e / q arrays during update_V,
 * bit 5 = the last level's apply pass also runs the next factor's first level (k_tile_apply_next),
 * bit 6 = row-sharded with the fused tile path (first-level columns swept where their rows live),
 * bit 7 = two-field pass (two one-hot-like levels: one pass over the residual per factor, no q-cache in HBM),
 * bit 8 = persistent latent sweep (residual on chip for update_w + update_V), bit 9 = index-tuple designs (cell passes),
 * bit 10 = a relation block's feature chain (FMTrainer.hpp:276-302, :419-470) runs as the streamed one-launch form,
 * bit 11 = the persistent sweep holds more rows per CU than fit on chip (the rest of the residual is streamed every sweep). */
int mfm_plan_flags(const mfm_ctx *ctx);

/* ---- model state (FM.hpp:164-168) ------------------------------------------------------ */
int mfm_set_state(mfm_ctx *ctx, double w0, const double *w, const double *V);
int mfm_get_state(mfm_ctx *ctx, double *w0, double *w, double *V);
int mfm_set_w0(mfm_ctx *ctx, double w0); /* scalar only; does not touch e (FMTrainer.hpp:219-222) */
int mfm_zero_w(mfm_ctx *ctx);            /* fit_linear == false, FMTrainer.hpp:232-235            */
/* residual e_train and per-factor cache q_train (BaseFMTrainer.hpp:46-47), for tests.      */
int mfm_get_e(mfm_ctx *ctx, double *e);
int mfm_get_q(mfm_ctx *ctx, double *q);
int mfm_set_e(mfm_ctx *ctx, const double *e);

/* ---- the Gibbs iteration, in update_all order (BaseFMTrainer.hpp:135-152) --------------- */
/* sum_t e_t and sum_t e_t^2 : inputs of update_alpha (FMTrainer.hpp:138) and update_w0
 * (:223, sum(w0 - e) = N w0 - sum e).                                                       */
int mfm_reduce_e(mfm_ctx *ctx, double *sum_e, double *sum_e2);
/* e += delta (FMTrainer.hpp:227).                                                           */
int mfm_shift_e(mfm_ctx *ctx, double delta);
/* per-group sufficient statistics of w for update_lambda_w / update_mu_w
 * (FMTrainer.hpp:150-200): sum[g] = sum_{j in g} w_j, ssd[g] = sum_{j in g} (w_j - mu[g])^2 */
int mfm_group_stats_w(mfm_ctx *ctx, const double *mu_w, double *sum, double *ssd);
/* same for every factor of V (FMTrainer.hpp:202-216); all arrays (G, K) column-major.       */
int mfm_group_stats_V(mfm_ctx *ctx, const double *mu_V, double *sum, double *ssd);
/* reduce_e + group_stats_w + group_stats_V with one host synchronisation (same outputs; need_e = 0 skips the
 * residual sums). All three only read state the iteration has not touched yet when it needs them
 * (FMTrainer.hpp:138, :150-216, :223). (G) and (G, K) arrays are column-major like the reference's. */
int mfm_hyper_stats(mfm_ctx *ctx, int32_t need_e, const double *mu_w, const double *mu_V, double *sum_e, double *sum_e2,
                    double *sum_w, double *ssd_w, double *sum_V, double *ssd_V);
/* update_w (FMTrainer.hpp:231-314): main table then relation blocks, feature order preserved
 * for every pair of features that share a row. z[D] = the N(0,1) variates of the D
 * sample_normal calls in reference order (main columns, then each block's columns); z == NULL uses the
 * device-generated variates of the set acquired by mfm_rng_acquire.                          */
int mfm_sweep_w(mfm_ctx *ctx, double alpha, const double *lambda_w, const double *mu_w, const double *z);
/* update_V (FMTrainer.hpp:316-486) for factors [f_begin, f_end): per factor the q-cache build
 * (:320-340), the main-table sweep (:343-376) and the per-block sweeps (:378-482).
 * lambda_V / mu_V: full (G, K) column-major; z: (f_end - f_begin) * D variates, factor-major. */
int mfm_sweep_V(mfm_ctx *ctx, int32_t f_begin, int32_t f_end, double alpha, const double *lambda_V,
                const double *mu_V, const double *z);
/* update_w0's residual shift (FMTrainer.hpp:226: e += e_shift), update_w (:231-314) and update_V (:316-486) of factors
 * [f_begin, f_end) as ONE call: BaseFMTrainer.hpp:143-148 draws lambda_V / mu_V between the two sweeps, but those draws read
 * neither w nor e, so the caller can make them first. Same results as mfm_shift_e + mfm_sweep_w + mfm_sweep_V; on a
 * two-field one-hot table it is one persistent launch. zw / zv: both NULL (the acquired device random set) or both given. */
int mfm_sweep_wV(mfm_ctx *ctx, double alpha, double e_shift, const double *lambda_w, const double *mu_w, const double *zw,
                 int32_t f_begin, int32_t f_end, const double *lambda_V, const double *mu_V, const double *zv);

/* One WHOLE regression iteration (GibbsFMTrainer::update_all, BaseFMTrainer.hpp:135-152) enqueued without a host round trip inside
 * it: the reductions update_alpha / update_w0 / update_lambda_* / update_mu_* need (FMTrainer.hpp:127-229), those conditionals
 * themselves on the device (k_hyper_regression: the trainer's arithmetic, operation for operation, on the unit variates of the
 * acquired random set -- draw program [gamma: alpha][normal: w0 if fit_w0][G gammas: lambda_w][G normals: mu_w][K G gammas:
 * lambda_V][K G normals: mu_V] + the w and V sweep normals), update_w0's shift + update_w + update_V as the persistent launch
 * (:231-486), the request for the set after the next, update_e (:493-497). With the host in the loop (mfm_hyper_stats -> host
 * draws -> mfm_sweep_wV) every iteration paid ~0.1 ms of read-back / upload / dispatch latency between two launches.
 *   prior: alpha_0, beta_0, gamma_0, mu_0, reg_0 of FMLearningConfig, n_total = training rows, fit_w0; n_in_group: [G] features
 *   per group. In: *w0, mu_w, mu_V = the current values. Out: this iteration's draws (the call returns when they have arrived,
 *   i.e. early in the iteration's device work: w / V / e are still being swept, like after mfm_sweep_wV).
 *   Errors raised ON the device by this iteration's own launches (a co-residency time-out of the persistent sweep, an exhausted
 *   random stream) cannot be known when the call returns: they are returned by the NEXT call that waits for the stream -- the next
 *   mfm_regression_iteration, mfm_get_state, mfm_synchronize -- and always before a model leaves the device.
 * mfm_regression_iteration_ready: 1 when the context can do this now (finalized two-field table on the persistent sweep, one GPU,
 * the residual in the sweep's slot order with its sums from the last mfm_update_e_regression, a device random stream programmed
 * as above); else 0 and the caller runs the iteration step by step. */
typedef struct mfm_hyper_prior {
  double alpha_0, beta_0, gamma_0, mu_0, reg_0, n_total;
  int32_t fit_w0, reserved;
} mfm_hyper_prior;
int mfm_regression_iteration_ready(mfm_ctx *ctx);
int mfm_regression_iteration(mfm_ctx *ctx, const mfm_hyper_prior *prior, const double *n_in_group, double *alpha, double *w0,
                             double *lambda_w, double *mu_w, double *lambda_V, double *mu_V);
/* update_e (FMTrainer.hpp:493-522), regression: e = predict_score(X_train) - y.             */
int mfm_update_e_regression(mfm_ctx *ctx);
/* update_e, probit classification (:498-512): e_t = score_t - z_t, z_t ~ TN(score_t, 1) on
 * (0, inf) if y_t > 0 else (-inf, 0) (util.hpp:15-78). The reference consumes its mt19937 in
 * data-dependent rejection loops; here each row draws from a Philox4x32-10 stream keyed by
 * (seed, draw_index, row): parity is distributional.                            */
int mfm_update_e_classification(mfm_ctx *ctx, uint64_t seed, uint64_t draw_index);
/* e = predict_score(X_train) only (initialize_e for the ordered task, FMTrainer.hpp:100).   */
int mfm_score_train(mfm_ctx *ctx);

/* ordered probit (OProbitSampler.hpp): one cutpoint group = a row subset with n_class labels.
 * rows == NULL means all N rows. Returns the group id in *group.                            */
int mfm_oprobit_add_group(mfm_ctx *ctx, int32_t n_class, const int64_t *rows, int64_t n_rows, int32_t *group);
/* log-likelihood, d/dgamma and the gamma-space Hessian accumulators of
 * OprobitSampler::operator() (OProbitSampler.hpp:402-413, 111-236) over the group's rows at
 * cutpoints gamma[n_class-1], on the current e (= score). H is (n_class-1)^2 row-major; pass
 * NULL to skip it.                                                                          */
int mfm_oprobit_eval(mfm_ctx *ctx, int32_t group, const double *gamma, double *ll, double *dgamma, double *H);
/* sample_z_given_cutpoint (OProbitSampler.hpp:238-272): e_t -= z_t, Philox-keyed as above.  */
int mfm_oprobit_sample_z(mfm_ctx *ctx, int32_t group, const double *gamma, uint64_t seed, uint64_t draw_index);

/* ---- device-side random stream (bit-compatible with the reference's std::mt19937 + libstdc++
 * distributions, see csrc/mfm_rng.hpp) ----------------------------------------------------------
 * For regression / classification the engine outputs consumed per Gibbs iteration do not depend on
 * the model state, so the iteration's variates are produced on the GPU ahead of the sweeps:
 *   - mfm_rng_seed_mt19937: hand over the generator (the 624 state words and the position index of
 *     a std::mt19937, e.g. after FM::initialize_weight consumed its share, FM.hpp:34-45);
 *   - mfm_rng_set_program: the draw order of one iteration as a list of ops (SURVEY 8a "RNG draw
 *     order"): NORMALS(count) = that many `sample_normal` variates (FMTrainer.hpp:122-125: a fresh
 *     normal_distribution per draw), GAMMA(shape) = the unit-scale variate of
 *     gamma_distribution(shape, scale) (FMTrainer.hpp:142-143, :164-165; multiply by scale);
 *     dest 0 = "hyper" variates returned to the host in program order, 1 = z of mfm_sweep_w (D),
 *     2 = z of mfm_sweep_V (K * D, factor-major);
 *   - mfm_rng_prefetch: start producing a further iteration's variates on a side stream. Sets are produced in
 *     iteration order; up to two may be in flight besides the acquired one (the trainer requests the set of
 *     iteration t + 2 right after the latent sweep of iteration t is enqueued). Where the persistent sweep fills the
 *     device, a set with another one ahead of it is produced in two parts: its single-workgroup kernels (generator,
 *     hyper draws, the linear term's normals) right away, beside the sweep; its whole-GPU evaluation of the K D
 *     sweep normals with the NEXT mfm_rng_prefetch (or the mfm_rng_acquire that needs it), behind everything
 *     enqueued on the ctx stream by then -- between two sweeps, not starved beside one;
 *   - mfm_rng_acquire: wait for the oldest prefetched set, copy its hyper variates to the host and
 *     make its z buffers the ones mfm_sweep_w / mfm_sweep_V use when called with z == NULL.      */
#define MFM_RNG_NORMALS 0
#define MFM_RNG_GAMMA 1
/* MFM_RNG_LATENT(count = rows): not a draw of the set -- tells the generator that every iteration also consumes about 6 outputs per
 * row through mfm_update_e_classification_exact / mfm_oprobit_sample_z_exact (sizes the ring and the parallel generator).        */
#define MFM_RNG_LATENT 2
typedef struct mfm_rng_op {
  int32_t kind;   /* MFM_RNG_NORMALS | MFM_RNG_GAMMA | MFM_RNG_LATENT      */
  int32_t dest;   /* 0 hyper variates, 1 z_w, 2 z_V                        */
  int64_t count;  /* NORMALS: number of draws; GAMMA: 1                    */
  int64_t offset; /* first index in the destination                        */
  double shape;   /* GAMMA: alpha                                          */
} mfm_rng_op;
/* Optional, no context needed: start computing the parallel generator's jump-ahead polynomials for a problem of this size on a
 * helper thread (cached per process; mfm_finalize asks for the same ones, so a caller that knows (D, rank, groups) before the
 * design is handed over hides the ~0.1-0.3 s they take behind its own setup).                                                */
int mfm_rng_prepare(int64_t n_features, int32_t rank, int32_t n_groups);
int mfm_rng_seed_mt19937(mfm_ctx *ctx, const uint32_t *state624, int32_t position);
int mfm_rng_set_program(mfm_ctx *ctx, const mfm_rng_op *ops, int32_t n_ops);
int mfm_rng_prefetch(mfm_ctx *ctx);
int mfm_rng_acquire(mfm_ctx *ctx, double *hyper_variates, int64_t n_hyper_variates);
/* test hook: the z buffers of the acquired set (zw[D], zv[K*D]); either pointer may be NULL. */
int mfm_rng_get_z(mfm_ctx *ctx, double *zw, double *zv);

/* ---- the draws the reference makes in a state-dependent order, on the SAME device stream ("exact" latent mode) ----
 * The latent z of probit classification (FMTrainer.hpp:498-512) and ordered probit (OProbitSampler.hpp:238-272) consume the
 * generator row after row inside rejection loops (util.hpp:15-60). These two entry points make exactly those draws from the
 * stream's current position -- same variates for the same rows, same number of engine outputs consumed -- evaluated in parallel
 * (csrc/mfm_latent.hip). No set may be in flight (acquire what was prefetched first); the next mfm_rng_prefetch starts where
 * the draw ended. *status: 0 = done. Otherwise NOTHING was drawn or consumed (e still holds the scores) and the caller makes
 * the draws sequentially through mfm_rng_host_read / mfm_rng_host_advance: 1 = the true path left a chunk's window of
 * candidate rows even at +-6.5 sigma (probability ~1e-8 per call), 2 / 3 = scratch space, 4 = more engine outputs needed than
 * prepared, 5 = a score more than 1000 standard deviations on the wrong side of its class.
 * mfm_update_e_classification_exact = FM::predict_score_write_target + the draws (the exact twin of
 * mfm_update_e_classification); mfm_oprobit_sample_z_exact = sample_z_given_cutpoint for one cutpoint group on the current e. */
int mfm_update_e_classification_exact(mfm_ctx *ctx, int32_t *status);
/* The order in which mfm_update_e_classification_exact serves the rows (FMTrainer.hpp:500 walks the caller's rows 0 .. N - 1): entry i =
 * the row of this table that is the caller's row i, for a caller that handed the rows over in another order (sorted for the device
 * paths). n = 0: the table's own order. (Ordered probit: the group's row list of mfm_oprobit_add_group is that order.)              */
int mfm_set_latent_order(mfm_ctx *ctx, const int64_t *rows, int64_t n);
int mfm_oprobit_sample_z_exact(mfm_ctx *ctx, int32_t group, const double *gamma, int32_t *status);
/* diagnostics of the last exact draw: {status, chunks, sub-chunks per chunk, quads per chunk, quads consumed, walkers started,
 * attempts (2: the first attempt's windows of +-3.5 sigma missed the path, the second's +-6.5 sigma held it), reserved} */
int mfm_latent_stats(mfm_ctx *ctx, int64_t *out8);
/* The host's window into the device stream: out[0..n) = the engine outputs (tempered, as std::mt19937::operator() returns them)
 * number offset .. offset + n - 1 counted from the stream's position; mfm_rng_host_advance moves the position by `words` outputs.
 * For the few draws the host makes itself between two sets (the cutpoint sampler's Metropolis step, OProbitSampler.hpp:55-72,
 * :378) and for the sequential fall-back of the two calls above. No set may be in flight.                                      */
int mfm_rng_host_read(mfm_ctx *ctx, uint64_t offset, int64_t n, uint32_t *out);
int mfm_rng_host_advance(mfm_ctx *ctx, uint64_t words);

/* ---- per-kernel timing (HIP events on the ctx stream) for bench.py's roofline block ---- */
int mfm_timing_enable(mfm_ctx *ctx, int on);
/* Restrict the event bracketing to one kernel class (index as in mfm_timing_class_name; < 0: all classes).
 * Bracketing every launch costs ~10 % at config 3; one class is cheap enough for a timed benchmark region. */
int mfm_timing_select(mfm_ctx *ctx, int32_t kernel_class);
int mfm_timing_reset(mfm_ctx *ctx);
int mfm_timing_n_classes(void);
const char *mfm_timing_class_name(int cls);
/* total milliseconds, launch count and algorithmic bytes (SURVEY 8d per-unit figures times
 * the units each launch processed) accumulated for one kernel class since the last reset.   */
int mfm_timing_get(mfm_ctx *ctx, int cls, double *ms_total, int64_t *launches, double *alg_bytes_total);

/* ---- prediction: FM::predict_score (FM.hpp:47-136), Predictor::predict* (predictor.hpp) -- */
int mfm_design_create(int device, int64_t N, int64_t D0, const int64_t *indptr, const int32_t *indices,
                      const double *data, mfm_design **out);
int mfm_design_add_block(mfm_design *d, int64_t B, int64_t Db, const int64_t *indptr, const int32_t *indices,
                         const double *data, const int64_t *original_to_block);
void mfm_design_destroy(mfm_design *d);
const char *mfm_design_last_error(const mfm_design *d);
int64_t mfm_design_dim_all(const mfm_design *d);
int64_t mfm_design_n_rows(const mfm_design *d);
/* mode 0: out[N]  = mean_s score_s                    (Predictor::predict, regression)
 * mode 1: out[N]  = mean_s Phi(score_s)               (classification, predictor.hpp:138-143)
 * mode 2: out[N * (n_cut + 1)] row-major = mean_s ordered-probit class probabilities with
 *         cutpoints[s * n_cut + c]                    (FM.hpp:137-162, predictor.hpp:78-124)
 * w0s[S], ws[S * D], Vs[S * D * K] (each sample's V column-major (D, K)).                   */
int mfm_design_predict(mfm_design *d, int32_t rank, int32_t n_samples, const double *w0s, const double *ws,
                       const double *Vs, int32_t mode, int32_t n_cut, const double *cutpoints, double *out);

/* ---- device-resident posterior samples: the Predictor's `samples` (FMTrainer.hpp:71-74, predictor.hpp:35-147) ------
 * mfm_store_push_ctx appends the live (w0, w, V) of a training context with a device-to-device copy on its stream (no
 * host transfer inside the Gibbs loop); mfm_store_get materialises one sample on the host (w[D], V column-major (D, K));
 * mfm_design_predict_store = mfm_design_predict over samples [first, first + count) read in place (cutpoints[count *
 * n_cut] for mode 2).                                                                                                */
int mfm_store_create(int device, int64_t D, int32_t rank, mfm_store **out);
void mfm_store_destroy(mfm_store *st);
const char *mfm_store_last_error(const mfm_store *st);
int32_t mfm_store_size(const mfm_store *st);
int mfm_store_reserve(mfm_store *st, int32_t n_samples); /* allocate room for n_samples ahead of the loop (optional) */
int mfm_store_push_ctx(mfm_store *st, mfm_ctx *ctx);
int mfm_store_push_host(mfm_store *st, double w0, const double *w, const double *V);
int mfm_store_get(mfm_store *st, int32_t idx, double *w0, double *w, double *V);
int mfm_design_predict_store(mfm_design *d, mfm_store *st, int32_t first, int32_t count, int32_t mode, int32_t n_cut,
                             const double *cutpoints, double *out);

/* FM::predict_score of the LIVE sample (the FM* handed to the per-iteration callback,
 * FMTrainer.hpp:78; utils/callbacks/libfm.py:85): scores design `d` with the (w0, w, V) currently
 * resident in training context `ctx` -- no download / upload of the model state. Same device only.  */
int mfm_design_score_ctx(mfm_design *d, mfm_ctx *ctx, double *out);

/* ---- test hooks: kernel-level access to the special functions / samplers of the classification and
 * ordered-probit tasks, so that parity tests can hold them against the reference's own code ----------- */
/* out[i] = the device erfcx(x[i]) used by mfm_oprobit_eval (reference: cpp_source/Faddeeva.cc erfcx, called from
 * OProbitSampler.hpp:111-236).                                                                           */
int mfm_test_erfcx(int device, const double *x, int64_t n, double *out);
/* n draws of the device truncated-normal samplers (util.hpp:15-78) with unit deviation: kind 0 = left
 * (z > lo, sample_truncated_normal_left), 1 = right (z < hi, :68-71), 2 = two-sided (lo < z < hi, :39-60). Draw i
 * uses the Philox stream keyed (seed, draw_index, row = i) -- the keying of mfm_update_e_classification.        */
int mfm_test_truncated_normal(int device, int32_t kind, double lo, double hi, uint64_t seed, uint64_t draw_index,
                              int64_t n, double *out);

/* ---- host-only helpers (no device needed; exercised by the CPU test-suite) -------------- */
/* Level schedule of the columns of a CSR matrix: level[j] = 0 if no earlier
 * column shares a row with column j, else 1 + max level of those columns. Returns the number
 * of levels in *n_levels.                                                                   */
int mfm_host_column_levels(int64_t n_rows, int64_t n_cols, const int64_t *indptr, const int32_t *indices,
                           int32_t *level, int32_t *n_levels);
/* The plan of the streamed block-feature sweep of a large relation block (FMTrainer.hpp:276-302, :419-470 as ONE pipelined launch,
 * csrc/mfm_chain_plan.hpp) for the chain over ALL n_cols columns of the block's CSC (colptr / rowidx / val: column j -> ascending block
 * rows), steps of cg columns, a window of lw steps, nb row ranges, slot reuse delay rd, at most cap LDS slots -- and an emulation of the
 * launch's data flow on the host (every actor on its own copy of what it can see, in the earliest order its flags allow) against the
 * plain sequential sweep: *max_rel_diff = the largest relative difference of coefficients / records (rounding level when the plan is
 * right). info[8]: built (0 / 1: the hot rows did not fit), slots, most entering / leaving rows of a step, cold entries, hot entries,
 * most hot entries of a column, steps.                                                                                          */
int mfm_cs_plan_selftest(int64_t n_rows, int32_t n_cols, const int64_t *colptr, const int32_t *rowidx, const double *val, int32_t cg,
                         int32_t lw, int32_t nb, int32_t rd, int32_t cap, double *max_rel_diff, int64_t *info);

#ifdef __cplusplus
}
#endif
#endif /* MYFM_HIP_H_ */
