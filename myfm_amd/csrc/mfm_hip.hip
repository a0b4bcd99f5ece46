// mfm_hip.hip -- libmyfm_hip.so: the training context and the C ABI of include/myfm_hip.h.
//
// Host side of the device path: builds the device-resident design (CSR + CSC + conflict-free
// level schedule), launches the kernels of mfm_kernels.hpp / mfm_block_kernels.hpp on one HIP
// stream and exposes them as the entry points the pybind11 host layer (csrc/_myfm.cpp) binds.
#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>

#include <hipcub/hipcub.hpp>

#include "mfm_common.hpp"
#include "mfm_kernels.hpp"
#include "mfm_plan.hpp"
#include "mfm_res_plan.hpp"
#include "mfm_block_kernels.hpp"
#include "mfm_cell.hpp"
#include "mfm_mtjump.hpp"
#include "mfm_rng.hpp"
#include "mfm_latent_api.hpp"

namespace mfm {
static thread_local std::string g_global_error;
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Persistent launches with a hand-written grid barrier (k_mf_resident: one workgroup per CU, up to 160 KB of LDS each) are
// correct only while ALL their workgroups are resident at once. Nothing in a plain launch guarantees that when somebody else
// occupies CUs for as long as the launch lasts -- a second persistent sweep (another context, thread or process on the same
// GPU: joblib cross-validation), or a CU mask under which the device still reports every CU. So a context must CLAIM the CUs of
// its persistent launch here before it may use it: inside a process the claims on a device add up to at most its CU count,
// across processes a device's persistent sweeps belong to the one process that holds an exclusive flock on a per-device lock
// file (released by the kernel when the process ends); a context that gets no claim runs the per-factor passes, which assume
// nothing about co-residency.
struct ResidentBudget {
  std::mutex m;
  struct Dev {
    int claimed = 0;
    int lock_fd = -1;
  };
  std::map<std::string, Dev> dev;  // keyed by PCI bus id
  static ResidentBudget &get() {
    static ResidentBudget b;
    return b;
  }
  static std::string key_of(int device) {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) std::snprintf(bus, sizeof(bus), "dev%d", device);
    for (char *p = bus; *p; p++)
      if (*p == ':' || *p == '.' || *p == '/') *p = '_';
    return bus;
  }
  bool acquire(int device, int n_cu_total, int want, std::string &why) {
    for (const char *v : {"HSA_CU_MASK", "ROC_GLOBAL_CU_MASK", "HSA_CU_MASK_SKIP_INIT"})
      if (const char *e = std::getenv(v))
        if (*e) {
          why = std::string(v) + " is set: the CUs a launch really gets are unknown";
          return false;
        }
    std::lock_guard<std::mutex> g(m);
    const std::string k = key_of(device);
    Dev &d = dev[k];
    if (d.claimed + want > n_cu_total) {
      why = "another context of this process holds the CUs for its persistent sweep";
      return false;
    }
    if (d.lock_fd < 0 && !std::getenv("MFM_RES_NO_PROCESS_LOCK")) {
      const char *dir = std::getenv("MFM_LOCK_DIR");
      const std::string path = std::string(dir && *dir ? dir : "/tmp") + "/myfm_amd_resident_" + k + ".lock";
      // The file is shared by every user of the machine: never follow a planted link (O_NOFOLLOW), make it 0666 whatever the
      // creator's umask was, and if it belongs to somebody else and cannot be opened for writing, open it read-only -- flock
      // works on a read-only descriptor. No descriptor at all: exclusivity cannot be established, so the persistent sweep is
      // refused (the per-factor passes assume nothing); MFM_RES_NO_PROCESS_LOCK=1 is the explicit way to do without the lock.
      int fd = ::open(path.c_str(), O_CREAT | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0666);
      if (fd >= 0)
        (void)::fchmod(fd, 0666);
      else if (errno == EACCES || errno == EPERM || errno == EROFS)
        fd = ::open(path.c_str(), O_RDONLY | O_CLOEXEC | O_NOFOLLOW);
      if (fd < 0) {
        why = "the lock file of this GPU's persistent sweeps cannot be opened (" + path + ": " + std::strerror(errno) + ")";
        static std::atomic<bool> warned{false};
        if (!warned.exchange(true))  // (once per process: the fit still runs, on the per-factor passes)
          std::fprintf(stderr,
                       "myfm_amd: %s -- the persistent sweep is not used (slower per-factor passes instead). Point MFM_LOCK_DIR at a "
                       "writable directory shared by the users of this GPU, or set MFM_RES_NO_PROCESS_LOCK=1 if no other process "
                       "runs fits on it.\n",
                       why.c_str());
        return false;
      }
      if (::flock(fd, LOCK_EX | LOCK_NB) != 0) {
        ::close(fd);
        why = "another process runs a persistent sweep on this GPU (" + path + ")";
        return false;
      }
      d.lock_fd = fd;
    }
    d.claimed += want;
    return true;
  }
  void release(int device, int n) {
    if (n <= 0) return;
    std::lock_guard<std::mutex> g(m);
    Dev &d = dev[key_of(device)];
    d.claimed = std::max(0, d.claimed - n);
    if (d.claimed == 0 && d.lock_fd >= 0) {
      ::flock(d.lock_fd, LOCK_UN);
      ::close(d.lock_fd);
      d.lock_fd = -1;
    }
  }
};
}  // namespace mfm

struct mfm_ctx;
namespace mfm {
static void materialize_e(::mfm_ctx *c);  // (brings the residual back to row order: mfm_ctx::eq_rows)
}
using namespace mfm;

struct mfm_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  bool finalized = false;

  // host-side staging until finalize
  HostCsr hX;  // host copy of the main table: only when a host planner asks for it (host_main)
  bool main_set = false, hX_valid = false;
  const HostCsr &host_main() {
    if (main_set && !hX_valid) {
      hX = X.download();
      hX_valid = true;
    }
    return hX;
  }
  std::vector<double> hy;
  struct HostBlock {
    HostCsr X;
    std::unique_ptr<int32_t[]> map;  // original_to_block, 32-bit (first touched by the threads that fill it)
  };
  std::vector<HostBlock> hblocks;
  std::vector<int32_t> hgroup;
  std::vector<int32_t> hlevels;  // optional caller-provided level schedule of the main table
  int32_t G = 0;

  // device design
  int64_t N = 0, D0 = 0, D = 0;
  int K = 0, KS = 0;
  DevSparse X;
  StepPlan plan_V, plan_W;
  LongScratch ls;
  Comm comm;               // row-sharded multi-GPU mode: set before mfm_finalize
  int64_t row_offset = 0;  // global index of local row 0 (keys the per-row Philox streams)
  std::vector<std::unique_ptr<DevBlock>> blocks;
  DevBuf<double> y;
  // {e_t, q_t} in ROW order. Between the sweeps the residual may live elsewhere -- in the persistent sweep's slot order
  // (e_in_slots), in the cell path's order (e_in_cell), or nowhere at all (e_lost: recomputed on demand) -- so nobody touches the
  // buffer directly: eq_rows() first brings the residual home (materialize_e) and is what every reader and read-modify-writer
  // in row order uses; eq_raw() is for the code that manages those states itself (the scorers that overwrite every residual, the
  // sweeps that take it from / leave it in another order, materialize_e).
  DevBuf<double2> eq_;
  double2 *eq_raw() const { return eq_.p; }
  double2 *eq_rows() {
    mfm::materialize_e(this);
    return eq_.p;
  }
  DevBuf<int32_t> group;
  DevBuf<int32_t> feat_sorted;
  DevBuf<int64_t> group_ptr;

  // model state
  double w0 = 0;
  DevBuf<double> w, V, Vt;

  // per-iteration scratch
  DevBuf<double> z;             // max(1, K) * D
  DevBuf<double> lam, mu;       // (G, K) column-major (or [G] for w)
  DevBuf<double2> red_partial;  // REDUCE_BLOCKS
  DevBuf<double2> red_out;      // 1 + G * max(1,K)
  DevBuf<double> scratch_n;     // N doubles (get/set e,q)
  DevBuf<double2> gs_partial;   // group statistics: per-chunk partials
  DevBuf<double2> hs_out;       // mfm_hyper_stats: [sum e | w groups | V groups]
  DevBuf<double> hs_mu;         // mfm_hyper_stats: [mu_w | mu_V]
  int gs_chunks = 1;            // chunks of the largest group
  DevBuf<double> ec, qc;        // split e / q arrays of the latent sweep (soa), compact residual (qfree)
  bool qfree = false, soa = false, fuse_next = false;
  bool mf = false;              // two-field pass (run_sweep_mf): no q-cache in HBM during update_V
  ResPlan res;                  // ... as one persistent launch with the residual resident on chip (mfm_res.hpp)
  // the persistent sweep's layout was built first, on the device, and took the table: X_t, the level plans and the row tiles of
  // the per-factor passes (their fall-back) are built when a call needs them (ensure_main_plans)
  std::vector<DevBuf<int32_t>> pre_maps;  // the blocks' maps uploaded ahead of the blocks (mfm_finalize only)
  // regression: outside the sweeps the residual IS score - y (update_e recomputes it after every update_V, FMTrainer.hpp:494), so
  // the persistent launch need not write its copy back (mfm_set_residual_policy); whoever asks for it in between gets it recomputed
  bool e_recomputable = false, e_lost = false;
  // the residual is (score of the current model) - y, maintained by the sweeps: true after score_train(subtract_y), false once the
  // caller has installed his own residual (mfm_set_e, mfm_shift_e, the latent draws of classification / ordered probit). Only
  // such a residual may be dropped by the sweep and recomputed on demand.
  bool e_is_residual = false;
  bool res_sharded_pending = false;  // row-sharded: the persistent sweep's layout is built on every rank, waiting for mfm_peer_set
  bool main_lazy = false;
  bool res_refused = false;  // the CUs of the persistent sweep were not ours to take
  int res_plan_cus = 0;      // workgroups the layout was asked for
  void ensure_main_plans();
  bool e_in_cell = false;       // the residual lives in cell.e (cell order): every reader of eq calls materialize_e first
  bool cell_w = false;          // ... and update_w runs on the cell layout too (the generic plans of the main table were not built)
  CellPlan cell;                // update_V of a design of index tuples (one-hot fields + relation blocks): no q-cache (mfm_cell.hpp)
  bool sharded_fused = false;   // row-sharded + fused tile path (run_sweep_soa_sharded)
  int q_stale_factor = -1;      // >= 0: the stored q column is stale, mfm_get_q rebuilds it for this factor first
  BlockOverflow gather_overflow;       // q-cache build: pointers of the relation blocks beyond MAX_BLOCKS
  DevBuf<double> wv_pack, zw_host;  // mfm_sweep_wV: [lambda_w | mu_w | lambda_V | mu_V] of the launch / host-given variates
  // mfm_regression_iteration: the iteration's hyper-parameters drawn on the device: [alpha, w0, e_shift, -][lambda_w][mu_w]
  // [lambda_V][mu_V]; what the host hands in ([w0 | mu_w | mu_V] before the draws), the group sizes, the event behind the read-back
  DevBuf<double> hyp, hyp_in, hyp_ng;
  std::vector<double> hyp_ng_host;
  hipEvent_t hyp_ev = nullptr;
  const double *w0_dev = nullptr;  // non-null while score_train should read the intercept from device memory
  std::vector<double> hs_stage;     // host staging of the packed hyper-parameter copies
  bool slot_sums_valid = false;     // res.sums holds sum e / sum e^2 of the residual that is in slot order right now
  bool res_fills_device = false;  // the persistent sweep takes (nearly) every CU: nothing runs beside it
  int res_claim = 0;              // CUs this context holds in ResidentBudget for its persistent launch
  bool e_in_slots = false;      // the residual after the resident latent sweep lives in res.e_slots (slot order): every
                                // reader of eq calls materialize_e first; update_e overwrites it and just drops the flag
  DevBuf<double> sync_mask;     // [D] 1: this rank contributes the column to the model synchronisation
  PinnedRing ring;
  double2 *h_red = nullptr;  // pinned readback
  size_t h_red_cap = 0;
  Timing timing;

  // device-side random stream (mfm_rng.hpp)
  struct RngEngine {
    bool seeded = false, programmed = false;
    hipStream_t stream = nullptr;
    DevBuf<RngState> state, state_next;
    DevBuf<uint32_t> raw;
    DevBuf<uint32_t> jump;  // jump-ahead polynomials of the parallel generator (mfm_mtjump.hpp)
    uint64_t need_gen = 0;  // outputs the parallel generator keeps ahead of the consumer (several iterations of `need`)
    int par_wgs = 1;        // workgroups of k_mt_generate_par (1: the serial generator)
    int par_blocks = 0;     // ... and the blocks each of them generates (chosen at mfm_finalize from the problem size)
    DevBuf<uint32_t> starts;  // [par_wgs][624] the block before each workgroup's first one (jump launch -> generation launch)
    uint64_t mask = 0, need = 0;
    DevBuf<RngOp> ops;
    int n_ops = 0;
    int64_t n_hv = 0, n_zw = 0, n_zv = 0;
    struct Slot {
      DevBuf<double> hv, zw, zv;
      double *h_hv = nullptr;  // pinned: [n_hv] variates + RngState header (3 x 8 bytes)
      hipEvent_t ready = nullptr, free_ev = nullptr;
      bool free_valid = false;
    } slot[3];
    static constexpr int N_SLOTS = 3;
    hipEvent_t gate = nullptr;  // recorded on the main stream at every prefetch: the side stream starts behind it
    // a set whose wide evaluation has not been enqueued yet (mfm_rng_prefetch): its slot, the op to go on with; -1: none
    int pending_slot = -1, pending_op = 0;
    // latent mode "exact": the main stream moved the stream's position (mfm_latent_host.hpp); the next set starts behind that
    bool latent_pending = false;
    hipEvent_t latent_ev = nullptr;
    DevBuf<uint32_t> host_win;  // staging of mfm_rng_host_read
    int64_t produced = 0, acquired = 0;
    int current = -1;
    // big NORMALS ops run as eval / scan / scatter over the whole GPU
    std::vector<RngOp> h_ops;
    DevBuf<double> cand;
    DevBuf<unsigned long long> masks;
    DevBuf<int> counts;
    DevBuf<NormScratch> nscratch;
    static constexpr int64_t WIDE_COUNT = 262144;  // NORMALS ops beyond this many draws: "wide" (mfm_rng_prefetch)
    static int64_t attempts_for(int64_t count) {
      double a = (double)count * (4.0 / 3.14159265358979) * 1.01 + 8.0 * std::sqrt((double)count + 1.0) + 4096.0;
      int64_t n = (int64_t)a;
      return (n + NORM_CHUNK - 1) / NORM_CHUNK * NORM_CHUNK;
    }
    ~RngEngine() {
      for (auto &sl : slot) {
        if (sl.h_hv) (void)hipHostFree(sl.h_hv);
        if (sl.ready) (void)hipEventDestroy(sl.ready);
        if (sl.free_ev) (void)hipEventDestroy(sl.free_ev);
      }
      if (gate) (void)hipEventDestroy(gate);
      if (latent_ev) (void)hipEventDestroy(latent_ev);
      if (stream) (void)hipStreamDestroy(stream);
    }
  } rng;

  // exact latent draws of classification / ordered probit on the device stream (mfm_latent.hip)
  std::unique_ptr<LatentEngine> latent;
  LatentStats latent_stats;
  DevBuf<int32_t> latent_order;  // classification: the rows in the reference's draw order (empty: the table's own order)

  // ordered probit groups
  struct OGroup {
    int n_class = 0;
    int64_t n_rows = 0;
    DevBuf<int32_t> rows;  // empty => all rows
  };
  std::vector<std::unique_ptr<OGroup>> ogroups;
  int opartial_cmax = 0;  // classes the partial-sum buffer is laid out for
  DevBuf<double> oacc;    // per-thread accumulators of k_oprobit_eval beyond OPROBIT_LDS_CLASS classes

  DevBuf<double> opartial;

  ~mfm_ctx() {
    drop_resident();
    if (h_red) (void)hipHostFree(h_red);
    if (own_stream && stream) (void)hipStreamDestroy(stream);
  }
  void use_device() { MFM_HIP_CHECK(hipSetDevice(device)); }
  // give the persistent sweep up (claim released): every later sweep runs the per-factor passes
  void drop_resident() {
    for (void *m : res.peer_mapped) (void)hipIpcCloseMemHandle(m);
    res.peer_mapped.clear();
    res_sharded_pending = false;
    res.peers_model = false;
    if (res_claim) ResidentBudget::get().release(device, res_claim);
    res_claim = 0;
    res.ready = false;
    res_fills_device = false;
  }
  // Co-resident kernels (k_long_coop, k_cb_persist, k_mf_resident) raise ls.error when a partner did not show up within the spin
  // bound: whatever they computed is then garbage. Called wherever the host waits for the stream anyway. The flag is cleared and
  // the persistent sweep given up, so the context stays usable -- but the chain state of THIS fit is invalid, hence the throw.
  void check_coresident(int flag) {
    if (flag == 0) return;
    (void)hipMemsetAsync(ls.error.p, 0, sizeof(int), stream);
    (void)hipStreamSynchronize(stream);
    const bool had = res.ready;
    drop_resident();
    throw Error(MFM_ERR_RUNTIME,
                std::string("co-resident workgroups timed out waiting for each other (long-column sweep / conflict-batched chain / "
                            "resident latent sweep): the state of this fit is invalid") +
                    (flag == 2 ? " [a peer rank's sums did not arrive inside the row-sharded persistent sweep]" : "") +
                    (had ? "; the persistent sweep is switched off for this context" : ""));
  }
  void sync_and_check() {
    if (!ls.error.p) {
      MFM_HIP_CHECK(hipStreamSynchronize(stream));
      return;
    }
    double2 *h = readback(1);
    *(int *)h = 0;
    MFM_HIP_CHECK(hipMemcpyAsync(h, ls.error.p, sizeof(int), hipMemcpyDeviceToHost, stream));
    MFM_HIP_CHECK(hipStreamSynchronize(stream));
    check_coresident(*(const int *)h);
  }
  void need_final() const {
    if (!finalized) throw Error(MFM_ERR_RUNTIME, "mfm_finalize has not been called");
  }
  double2 *readback(size_t n) {
    if (h_red_cap < n) {
      if (h_red) MFM_HIP_CHECK(hipHostFree(h_red));
      h_red = nullptr;
      MFM_HIP_CHECK(hipHostMalloc((void **)&h_red, n * sizeof(double2), hipHostMallocDefault));
      h_red_cap = n;
    }
    return h_red;
  }
};

#define MFM_TRY(ctx) \
  try {              \
    (ctx)->use_device();
#define MFM_CATCH(ctx)                 \
  return MFM_OK;                       \
  }                                    \
  catch (const mfm::Error &ex) {       \
    (ctx)->err = ex.what();            \
    return ex.code;                    \
  }                                    \
  catch (const std::bad_alloc &) {     \
    (ctx)->err = "host out of memory"; \
    return MFM_ERR_RUNTIME;            \
  }                                    \
  catch (const std::exception &ex) {   \
    (ctx)->err = ex.what();            \
    return MFM_ERR_RUNTIME;            \
  }

// =============================================================================================
// launch helpers
// =============================================================================================
namespace mfm {

static void fill_gather_args(std::vector<std::unique_ptr<DevBlock>> &blocks, BlockGatherArgs &g, BlockOverflow &ov, PinnedRing &ring,
                             hipStream_t s) {
  std::memset(&g, 0, sizeof(g));
  g.n_blocks = (int)blocks.size();
  std::vector<const int32_t *> xm;
  std::vector<const double *> xr;
  std::vector<int> xs;
  for (size_t b = 0; b < blocks.size(); b++) {
    const int32_t *m = blocks[b]->map.p;
    const double *r = blocks[b]->qc.p ? blocks[b]->qc.p : blocks[b]->rec.p;
    const int st = blocks[b]->qc.p ? 1 : BLOCK_REC;
    if (b < (size_t)MAX_BLOCKS) {
      g.map[b] = m;
      g.rec[b] = r;
      g.stride[b] = st;
    } else {
      xm.push_back(m);
      xr.push_back(r);
      xs.push_back(st);
    }
  }
  BlockOverflow::put(ov.map, xm, ring, s);
  BlockOverflow::put(ov.p0, xr, ring, s);
  BlockOverflow::put(ov.stride, xs, ring, s);
  g.xmap = ov.map.p;
  g.xrec = ov.p0.p;
  g.xstride = ov.stride.p;
}

// q = X v_f + sum_b q_B[map_b]   (FMTrainer.hpp:320-340); the blocks' q_B must be in rec already
static bool qbuild_by_rows(const mfm_ctx *c) { return c->X.avg_row_nnz <= 16.0; }
// pending: a block whose re-sync of the previous factor is applied by this pass (q_saved holds its old (q_B, q_S))
static void launch_qbuild(mfm_ctx *c, const double *vf, DevBlock *pending = nullptr) {
  if (c->N == 0) return;
  hipStream_t s = c->stream;
  BlockGatherArgs g;
  fill_gather_args(c->blocks, g, const_cast<mfm_ctx *>(c)->gather_overflow, const_cast<mfm_ctx *>(c)->ring, s);
  TimedLaunch t(c->timing, s, KC_QBUILD, 12.0 * c->X.nnz + 8.0 * c->N + 8.0 * c->D0 + 12.0 * c->N * g.n_blocks);
  if (qbuild_by_rows(c)) {
    const bool ell = c->X.ell_width >= 0;
    dim3 grid(cdiv(c->N, WG)), block(WG);
#define MFM_QB(U, E) \
  hipLaunchKernelGGL((k_qbuild_rows<U, E>), grid, block, 0, s, c->X.rowptr.p, c->X.colidx.p, c->X.rval.p, vf, c->eq_rows(), c->N, \
                     (int)c->X.ell_width, g, pending ? pending->map.p : (const int32_t *)nullptr,                          \
                     pending ? pending->q_saved.p : (const double2 *)nullptr)
    if (c->X.unit) {
      if (ell) MFM_QB(true, true); else MFM_QB(true, false);
    } else {
      if (ell) MFM_QB(false, true); else MFM_QB(false, false);
    }
#undef MFM_QB
  } else {
    hipLaunchKernelGGL(k_qbuild_wave, dim3(cdiv(c->N, WG / WAVE)), dim3(WG), 0, s, c->X.rowptr.p, c->X.colidx.p,
                       c->X.rval.p, vf, c->eq_rows(), c->N, g);
  }
  MFM_HIP_CHECK(hipGetLastError());
}

template <int GS, int SPL, bool UNIT, bool ELL>
static void launch_score_t(hipStream_t s, int mode, const DevSparse &X, const double *Vt, const double *w, double w0, int K,
                           int KS, const double *y, double2 *eq, double *out, const BlockScoreArgs &blk) {
  const int64_t N = X.rows;
  const int64_t groups = (N + SCORE_RU - 1) / SCORE_RU;
  dim3 grid(cdiv(groups * GS, WG)), block(WG);
  if (mode == 0)
    hipLaunchKernelGGL((k_score<GS, SPL, 0, UNIT, ELL>), grid, block, 0, s, X.rowptr.p, X.colidx.p, X.rval.p, Vt, w, w0, K, KS,
                       (int)X.ell_width, y, eq, out, N, blk);
  else
    hipLaunchKernelGGL((k_score<GS, SPL, 1, UNIT, ELL>), grid, block, 0, s, X.rowptr.p, X.colidx.p, X.rval.p, Vt, w, w0, K, KS,
                       (int)X.ell_width, y, eq, out, N, blk);
}

template <int GS, int SPL>
static void launch_score_f(hipStream_t s, int mode, const DevSparse &X, const double *Vt, const double *w, double w0, int K,
                           int KS, const double *y, double2 *eq, double *out, const BlockScoreArgs &blk) {
  const bool ell = X.ell_width >= 0;
  if (X.unit && ell)
    launch_score_t<GS, SPL, true, true>(s, mode, X, Vt, w, w0, K, KS, y, eq, out, blk);
  else if (X.unit)
    launch_score_t<GS, SPL, true, false>(s, mode, X, Vt, w, w0, K, KS, y, eq, out, blk);
  else
    launch_score_t<GS, SPL, false, false>(s, mode, X, Vt, w, w0, K, KS, y, eq, out, blk);
}

// mode 0: eq.x = score (- y)   mode 1: out = score.  A lane owns factor pairs: GS lanes cover 2 GS factors.
static void launch_score(hipStream_t s, int mode, const DevSparse &X, const double *Vt, const double *w, double w0, int K,
                         int KS, const double *y, double2 *eq, double *out, const BlockScoreArgs &blk) {
  if (X.rows == 0) return;
#define MFM_SCORE(GS, SPL) launch_score_f<GS, SPL>(s, mode, X, Vt, w, w0, K, KS, y, eq, out, blk)
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
  else if (K <= 512)
    MFM_SCORE(64, 4);
  else
    throw Error(MFM_ERR_INVALID, "rank > 512 is not supported");
#undef MFM_SCORE
  MFM_HIP_CHECK(hipGetLastError());
}

static void build_vt(hipStream_t s, const double *V, double *Vt, int64_t D, int K, int KS) {
  if (K == 0 || D == 0) return;
  hipLaunchKernelGGL(k_build_vt, dim3(cdiv(D, 32), cdiv(KS, 32)), dim3(WG), 0, s, V, Vt, D, K, KS);
  MFM_HIP_CHECK(hipGetLastError());
}

// score of a design under (w0, w, V): builds Vt and the block-level caches, then one fused pass
static void score_design(hipStream_t s, Timing &tm, int mode, const DevSparse &X, std::vector<std::unique_ptr<DevBlock>> &blocks,
                         int64_t D, int K, int KS, double w0, const double *w, const double *V, double *Vt, const double *y,
                         double2 *eq, double *out) {
  {
    TimedLaunch t(tm, s, KC_BUILD_VT, 16.0 * D * K);
    build_vt(s, V, Vt, D, K, KS);
  }
  BlockScoreArgs blk;
  std::memset(&blk, 0, sizeof(blk));
  blk.n_blocks = (int)blocks.size();
  std::vector<const int32_t *> xm;
  std::vector<const double *> xq, xl, xs;
  for (size_t b = 0; b < blocks.size(); b++) {
    DevBlock &B = *blocks[b];
    TimedLaunch t(tm, s, KC_BLOCK_ROWCACHE, 12.0 * B.nnz + 8.0 * B.B * (K + 2));
    launch_block_score_cache(s, B, Vt, w, K, KS);
    if (b < (size_t)MAX_BLOCKS) {
      blk.map[b] = B.map.p;
      blk.bq[b] = B.bq.p;
      blk.bl[b] = B.bl.p;
      blk.bs[b] = B.bs.p;
    } else {
      xm.push_back(B.map.p);
      xq.push_back(B.bq.p);
      xl.push_back(B.bl.p);
      xs.push_back(B.bs.p);
    }
  }
  if (!xm.empty()) {  // (more than MAX_BLOCKS blocks: their pointers through device arrays kept by the first of them)
    BlockOverflow &ov = blocks[MAX_BLOCKS]->ov;
    MFM_HIP_CHECK(hipStreamSynchronize(s));  // (a pass that reads the arrays may be in flight)
    BlockOverflow::put_sync(ov.map, xm);
    BlockOverflow::put_sync(ov.p0, xq);
    BlockOverflow::put_sync(ov.p1, xl);
    BlockOverflow::put_sync(ov.p2, xs);
    blk.xmap = ov.map.p;
    blk.xbq = ov.p0.p;
    blk.xbl = ov.p1.p;
    blk.xbs = ov.p2.p;
  }
  TimedLaunch t(tm, s, KC_UPDATE_E,
                12.0 * X.nnz + 16.0 * X.rows + 8.0 * D * (K + 1) + (4.0 + 8.0 * (K + 2)) * X.rows * blk.n_blocks);
  launch_score(s, mode, X, Vt, w, w0, K, KS, y, eq, out, blk);
}

static SweepArgs main_args(mfm_ctx *c, double *theta, const double *z, const double *lam, const double *mu, double alpha) {
  SweepArgs a;
  a.colptr = c->X.colptr.p;
  a.rowidx = c->X.rowidx.p;
  a.val = c->X.cval.p;
  a.state = c->eq_raw();
  a.theta = theta;
  a.z = z;
  a.group = c->group.p;
  a.lambda = lam;
  a.mu = mu;
  a.alpha = alpha;
  return a;
}

// the residual the resident latent sweep left in slot order -> eq (only when somebody reads it: in the Gibbs loop update_e
// follows and recomputes it)
static void score_train(mfm_ctx *c, bool subtract_y);
// the persistent launch did not write the residual back (e_recomputable): here it is again, e = score - y
static void ensure_e(mfm_ctx *c) {
  if (!c->e_lost) return;
  c->e_lost = false;
  score_train(c, true);
}
static void materialize_e(mfm_ctx *c) {
  ensure_e(c);
  c->slot_sums_valid = false;  // (whoever asks for the residual in row order may change it)
  if (c->e_in_cell) {  // (the cell path's sweeps leave it in cell order; update_e, which follows in the Gibbs loop, just drops it)
    cell_unpack_e(c->stream, c->cell, c->eq_raw());
    c->e_in_cell = false;
  }
  if (!c->e_in_slots) return;
  const int64_t n_slots = (int64_t)c->res.G * c->res.NT * c->res.R();
  hipLaunchKernelGGL(k_res_unpermute, dim3((unsigned)cdiv(n_slots, 256)), dim3(256), 0, c->stream, c->res.e_slots.p, c->res.perm.p,
                     n_slots, c->eq_raw());
  c->e_in_slots = false;
}

static void score_train(mfm_ctx *c, bool subtract_y) {
  c->e_lost = false;
  c->e_is_residual = subtract_y;
  c->e_in_slots = false;  // (every residual is overwritten)
  c->e_in_cell = false;
  c->slot_sums_valid = false;
  if (c->main_lazy) {
    static const bool no_res_score0 = std::getenv("MFM_NO_RES_SCORE") != nullptr || std::getenv("MFM_NO_MF_SCORE") != nullptr;
    if (subtract_y && c->res.ready && !no_res_score0 && res_score_supported(c->res, c->K)) {
      hipStream_t s = c->stream;
      {
        TimedLaunch t(c->timing, s, KC_BUILD_VT, 16.0 * c->D * c->K);
        build_vt(s, c->V.p, c->Vt.p, c->D, c->K, c->KS);
      }
      run_res_score(s, c->timing, c->res, KC_UPDATE_E, c->Vt.p, c->w.p, c->w0, c->K, c->y.p, c->X.nnz, c->w0_dev);
      c->e_in_slots = true;
      c->slot_sums_valid = true;
      return;
    }
    c->ensure_main_plans();
  }
  if (c->mf && !std::getenv("MFM_NO_MF_SCORE")) {
    // two-field table: scorer on the row tiles of the latent sweep (item rows gathered once per run, not once per row)
    hipStream_t s = c->stream;
    {
      TimedLaunch t(c->timing, s, KC_BUILD_VT, 16.0 * c->D * c->K);
      build_vt(s, c->V.p, c->Vt.p, c->D, c->K, c->KS);
    }
    // regression on a table that takes the persistent sweep: e = score - y straight in the sweep's slot order, with its sums
    static const bool no_res_score = std::getenv("MFM_NO_RES_SCORE") != nullptr;
    if (subtract_y && c->res.ready && !no_res_score && res_score_supported(c->res, c->K)) {
      run_res_score(s, c->timing, c->res, KC_UPDATE_E, c->Vt.p, c->w.p, c->w0, c->K, c->y.p, c->X.nnz, c->w0_dev);
      c->e_in_slots = true;
      c->slot_sums_valid = true;
      return;
    }
    TimedLaunch t(c->timing, s, KC_UPDATE_E, 12.0 * c->X.nnz + 16.0 * c->X.rows + 8.0 * c->D * (c->K + 1));
    SweepArgs a = main_args(c, c->w.p, nullptr, nullptr, nullptr, 0.0);
    const bool done = c->X.unit ? launch_mf_score<true>(s, c->plan_V, a, c->Vt.p, c->w.p, c->w0, c->K, c->KS,
                                                        subtract_y ? c->y.p : nullptr, c->eq_raw())
                                : launch_mf_score<false>(s, c->plan_V, a, c->Vt.p, c->w.p, c->w0, c->K, c->KS,
                                                         subtract_y ? c->y.p : nullptr, c->eq_raw());
    if (done) return;
  }
  static const bool no_cell_score = std::getenv("MFM_NO_CELL_SCORE") != nullptr;
  if (c->cell.ready && !no_cell_score) {
    // index-tuple design: K / FB passes over (accumulator, index record) with the factor tables in LDS (mfm_cell.hpp)
    hipStream_t s = c->stream;
    {
      TimedLaunch t(c->timing, s, KC_BUILD_VT, 16.0 * c->D * c->K);
      build_vt(s, c->V.p, c->Vt.p, c->D, c->K, c->KS);
    }
    std::vector<CellScoreSrc> src(c->cell.fields.size());
    for (size_t k = 0; k < c->cell.fields.size(); k++) {
      const CellField &fd = c->cell.fields[k];
      if (fd.kind == 0) {
        src[k].lin = c->w.p + fd.base;
      } else {
        DevBlock &B = *c->blocks[(size_t)fd.base];
        TimedLaunch t(c->timing, s, KC_BLOCK_ROWCACHE, 12.0 * B.nnz + 8.0 * B.B * (c->K + 2));
        launch_block_score_cache(s, B, c->Vt.p, c->w.p, c->K, c->KS);
        src[k].q = B.bq.p;
        src[k].lin = B.bl.p;
        src[k].ss = B.bs.p;
      }
    }
    cell_score(s, c->timing, c->cell, src, c->V.p, c->Vt.p, c->D, c->K, c->KS, c->w0, subtract_y ? c->y.p : nullptr, c->eq_raw());
    return;
  }
  score_design(c->stream, c->timing, 0, c->X, c->blocks, c->D, c->K, c->KS, c->w0, c->w.p, c->V.p, c->Vt.p,
               subtract_y ? c->y.p : nullptr, c->eq_raw(), nullptr);
}

// update_V (FMTrainer.hpp:315-482) of factors [f_begin, f_end) on the cell layout: per factor one streaming pass per field
// (apply the field before it, statistics of its own), the main fields' draws and the blocks' feature sweeps in between.
static void run_sweep_cell(mfm_ctx *c, int f_begin, int f_end, const double *zbase, double alpha) {
  hipStream_t s = c->stream;
  CellPlan &cp = c->cell;
  Timing &tm = c->timing;
  const int m = (int)cp.fields.size();
  if (!c->e_in_cell) cell_pack_e(s, cp, c->eq_raw());
  c->e_in_cell = true;  // (stays in cell order: materialize_e brings it back when somebody reads eq)
  std::vector<CellSrc> cur((size_t)m);
  cp.touch_all();  // (whatever cell_prep built in an earlier call is stale)
  auto set_cur = [&](int f) {
    cp.touch_all();
    for (int k = 0; k < m; k++) {
      const CellField &fd = cp.fields[k];
      if (fd.kind == 0) {
        cur[k].p = c->V.p + (size_t)f * c->D + fd.base;
        cur[k].stride = 1;
      } else {
        cur[k].p = c->blocks[(size_t)fd.base]->rec.p;  // word 0 of the record: q_B
        cur[k].stride = BLOCK_REC;
      }
    }
  };
  auto on_I = [&](int field) { return cp.streams[cp.fields[field].stream].type == CELL_I; };
  const SweepClasses kc{KC_BLOCK_SWEEP, KC_BLOCK_SWEEP, KC_BLOCK_SWEEP, KC_BLOCK_SWEEP, KC_BLOCK_SWEEP, KC_BLOCK_SWEEP,
                        KC_BLOCK_SWEEP, KC_BLOCK_SWEEP};
  int pending = -1;  // the last field of the factor before: drawn, not yet applied to the rows
  for (int f = f_begin; f < f_end; f++) {
    double *Vf = c->V.p + (size_t)f * c->D;
    const double *zf = zbase + (size_t)(f - f_begin) * c->D;
    const double *lamf = c->lam.p + (size_t)f * c->G;
    const double *muf = c->mu.p + (size_t)f * c->G;
    // the pending field is applied with the tables of ITS factor: take them before the blocks' q_B are rebuilt
    if (pending >= 0) cell_prep(s, tm, cp, cur, true, pending, false, -1, on_I(pending));
    set_cur(f);
    for (auto &B : c->blocks) block_rowcache(s, tm, *B, Vf + B->col_off, true);  // :331-333, :388-393
    for (int k = 0; k < m; k++) {
      const int P = k == 0 ? pending : k - 1;
      const bool sw = k == 0 && pending >= 0;
      const CellField &fd = cp.fields[k];
      DevBlock *B = fd.kind == 1 ? c->blocks[(size_t)fd.base].get() : nullptr;
      double *out_u = B ? B->rec.p + 2 : cp.stat.p;  // a U field's sums: (c, c_S, e, e_q) of the record / (S2, S_eh)
      int out_stride = B ? BLOCK_REC : 2;
      const bool split = P >= 0 && cp.lds_bytes(P, k, sw) > CELL_LDS_BYTES;
      if (sw)
        cell_prep(s, tm, cp, cur, false, -1, true, k, false);
      else
        cell_prep(s, tm, cp, cur, P >= 0, P, true, k, P >= 0 && on_I(P));
      const bool sh = c->comm.active();
      const int ns = B ? 4 : 2;
      if (sh) {  // row-sharded: the sums go through one dense array that is all-reduced
        out_u = cp.dense.p;
        out_stride = ns;
        MFM_HIP_CHECK(hipMemsetAsync(cp.dense.p, 0, (size_t)fd.n * ns * sizeof(double), s));
      }
      if (!split) {
        cell_pass(s, tm, cp, P, k, sw, out_u, out_stride);
      } else {  // (both roles do not fit the LDS together)
        cell_pass(s, tm, cp, P, -1, false, nullptr, 0);
        cell_pass(s, tm, cp, -1, k, false, out_u, out_stride);
      }
      if (sh) {
        cell_stats_dense(s, tm, cp, k, ns, cp.dense.p);
        c->comm.allreduce(cp.dense.p, fd.n * ns);
        if (B) cell_dense_to_rec(s, tm, cp, k, 4, cp.dense.p, B->rec.p, 2);
      }
      if (!B) {
        cell_draw_main(s, tm, cp, k, Vf, zf, c->group.p, lamf, muf, alpha, sh ? cp.dense.p : nullptr);  // :357-369
      } else {
        if (!sh) cell_block_stats(s, tm, cp, k, B->rec.p);  // :401-407
        if (B->q_saved.n < (size_t)B->B) B->q_saved.alloc((size_t)B->B);
        hipLaunchKernelGGL(k_save_q, dim3(cdiv(std::max<int64_t>(B->B, 1), 256)), dim3(256), 0, s, B->rec.p, B->B, B->q_saved.p);
        SweepArgs a = block_args(*B, Vf, zf, c->group.p, lamf, muf, alpha);
        run_plan<PBlockV>(s, tm, B->plan_V, a, c->ls, kc, false);  // :419-470
        cell_block_delta(s, tm, cp, k, B->rec.p, B->q_saved.p);
      }
      cp.touch(k);  // (field k's table changed: the tables of its stream are rebuilt by the next cell_prep)
    }
    pending = m - 1;
  }
  if (pending >= 0) {
    cell_prep(s, tm, cp, cur, true, pending, false, -1, on_I(pending));
    cell_pass(s, tm, cp, pending, -1, false, nullptr, 0);
  }
  c->q_stale_factor = f_end - 1;  // q_train as the reference leaves it (:373, :479): rebuilt when asked for
  MFM_HIP_CHECK(hipGetLastError());
}

// update_w (FMTrainer.hpp:231-313) on the cell layout: per field one pass (apply the field before it: e += d1; sum e per index
// value of its own), the main fields' draws and the blocks' feature sweeps (run_plan<PBlockW> on their records) in between
static void run_sweep_w_cell(mfm_ctx *c, const double *zdev, double alpha) {
  hipStream_t s = c->stream;
  CellPlan &cp = c->cell;
  Timing &tm = c->timing;
  const int m = (int)cp.fields.size();
  const bool sh = c->comm.active();
  if (!cp.cnt_ready) {
    if (c->e_in_cell) materialize_e(c);  // (cell.e is the scratch of the one-off row count)
    cell_counts(s, tm, cp);
    if (sh)  // (a column's rows on all ranks)
      for (int k = 0; k < m; k++)
        if (cp.fields[k].kind == 0) c->comm.allreduce(cp.cnt[k].p, cp.fields[k].n);
  }
  if (!c->e_in_cell) cell_pack_e(s, cp, c->eq_raw());
  c->e_in_cell = true;
  const SweepClasses kc{KC_BLOCK_SWEEP, KC_BLOCK_SWEEP, KC_BLOCK_SWEEP, KC_BLOCK_SWEEP, KC_BLOCK_SWEEP, KC_BLOCK_SWEEP,
                        KC_BLOCK_SWEEP, KC_BLOCK_SWEEP};
  for (int k = 0; k < m; k++) {
    const CellField &fd = cp.fields[k];
    DevBlock *B = fd.kind == 1 ? c->blocks[(size_t)fd.base].get() : nullptr;
    if (B) {
      block_rowcache(s, tm, *B, c->w.p + B->col_off, false);  // q_B = X_B w_B (:265-266); zeroes the record's sums
      if (B->q_saved.n < (size_t)B->B) B->q_saved.alloc((size_t)B->B);
      hipLaunchKernelGGL(k_save_q, dim3(cdiv(std::max<int64_t>(B->B, 1), 256)), dim3(256), 0, s, B->rec.p, B->B, B->q_saved.p);
    }
    double *sums = B ? B->rec.p + 4 : cp.stat1.p;  // e_B of the record (:271) / the draw's input
    int stride = B ? BLOCK_REC : 1;
    if (sh) {
      sums = cp.dense.p;
      stride = 1;
      MFM_HIP_CHECK(hipMemsetAsync(cp.dense.p, 0, (size_t)fd.n * sizeof(double), s));
    }
    cell_pass(s, tm, cp, k - 1, k, false, sums, stride, true);
    cell_sum1(s, tm, cp, k, sums, stride);
    if (sh) {
      c->comm.allreduce(cp.dense.p, fd.n);
      if (B) cell_dense_to_rec(s, tm, cp, k, 1, cp.dense.p, B->rec.p, 4);
    }
    if (!B) {
      cell_draw_main_w(s, tm, cp, k, c->w.p, zdev, c->group.p, c->lam.p, c->mu.p, alpha, sh ? cp.dense.p : nullptr);  // :237-254
    } else {
      SweepArgs a = block_args(*B, c->w.p, zdev, c->group.p, c->lam.p, c->mu.p, alpha);
      run_plan<PBlockW>(s, tm, B->plan_W, a, c->ls, kc, false);  // :276-302
      block_rowcache(s, tm, *B, c->w.p + B->col_off, false);     // :304-305
      cell_block_delta_w(s, tm, cp, k, B->rec.p, B->q_saved.p);  // the rows change by q_B' - q_B (:272-273, :306-311)
    }
  }
  cell_pass(s, tm, cp, m - 1, -1, false, nullptr, 0, true);
  MFM_HIP_CHECK(hipGetLastError());
}

}  // namespace mfm

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

const char *mfm_version(void) { return "myfm_hip 0.1 (gfx950)"; }

int mfm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

const char *mfm_global_error(void) { return g_global_error.c_str(); }

// The library's code object (a few hundred kernels) is loaded by the runtime at the first launch from it -- about 40 ms on
// MI355X. The first context of a process starts that launch on a helper thread, so the load runs beside the host's copies of
// the design (mfm_set_main / mfm_add_block); mfm_finalize waits for it before its own first launch.
__global__ void k_warm(int *p) {
  if (p && threadIdx.x == 0) *p = 1;
}
struct CodeWarmup {
  std::once_flag once;
  std::mutex mu;
  std::thread th;
  void start(int device) {
    std::call_once(once, [&]() {
      if (std::getenv("MFM_NO_WARMUP")) return;
      th = std::thread([device]() {
        hipStream_t st = nullptr;
        if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return;
        hipLaunchKernelGGL(k_warm, dim3(1), dim3(64), 0, st, (int *)nullptr);
        (void)hipStreamSynchronize(st);
        (void)hipStreamDestroy(st);
        (void)hipGetLastError();
      });
    });
  }
  void join() {
    std::lock_guard<std::mutex> lk(mu);
    if (th.joinable()) th.join();
  }
  ~CodeWarmup() { join(); }
  static CodeWarmup &get() {
    static CodeWarmup w;
    return w;
  }
};

int mfm_create(int device, mfm_ctx **out) {
  *out = nullptr;
  try {
    int n = mfm_device_count();
    if (n <= 0)
      throw Error(MFM_ERR_DEVICE,
                  "no HIP device is visible: libmyfm_hip.so has no CPU fallback (the Gibbs hot path runs on MI355X only)");
    if (device < 0 || device >= n) throw Error(MFM_ERR_INVALID, "device index out of range");
    std::unique_ptr<mfm_ctx> c(new mfm_ctx());
    c->device = device;
    c->use_device();
    MFM_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
    CodeWarmup::get().start(device);
    *out = c.release();
    return MFM_OK;
  } catch (const mfm::Error &ex) {
    g_global_error = ex.what();
    return ex.code;
  } catch (const std::exception &ex) {
    g_global_error = ex.what();
    return MFM_ERR_RUNTIME;
  }
}

void mfm_destroy(mfm_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  delete ctx;
}

int mfm_get_device(const mfm_ctx *ctx) { return ctx ? ctx->device : -1; }

const char *mfm_last_error(const mfm_ctx *ctx) { return ctx ? ctx->err.c_str() : g_global_error.c_str(); }

int mfm_set_stream(mfm_ctx *ctx, void *hip_stream) {
  MFM_TRY(ctx)
  if (ctx->own_stream && ctx->stream) {
    MFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    MFM_HIP_CHECK(hipStreamDestroy(ctx->stream));
  }
  ctx->stream = (hipStream_t)hip_stream;
  ctx->own_stream = false;
  ctx->comm.stream = ctx->stream;
  MFM_CATCH(ctx)
}

int mfm_set_allreduce(mfm_ctx *ctx, int (*fn)(void *user, void *dev_buf, int64_t count), void *user) {
  MFM_TRY(ctx)
  if (ctx->finalized) throw Error(MFM_ERR_RUNTIME, "mfm_set_allreduce must be called before mfm_finalize");
  ctx->comm.fn = fn;
  ctx->comm.user = user;
  MFM_CATCH(ctx)
}

int mfm_comm_unique_id(void *out128) {
  try {
    Rccl &r = Rccl::get();
    r.check(r.GetUniqueId(out128), "ncclGetUniqueId");
    return MFM_OK;
  } catch (const std::exception &ex) {
    g_global_error = ex.what();
    return MFM_ERR_RUNTIME;
  }
}

int mfm_comm_init(mfm_ctx *ctx, const void *id128, int32_t rank, int32_t world) {
  MFM_TRY(ctx)
  if (ctx->finalized) throw Error(MFM_ERR_RUNTIME, "mfm_comm_init must be called before mfm_finalize");
  if (world < 1 || rank < 0 || rank >= world) throw Error(MFM_ERR_INVALID, "bad rank / world size");
  if (ctx->comm.nccl) throw Error(MFM_ERR_RUNTIME, "communicator already initialised");
  Rccl &r = Rccl::get();
  mfm_nccl_id id;
  std::memcpy(&id, id128, sizeof(id));
  void *comm = nullptr;
  r.check(r.CommInitRank(&comm, world, id, rank), "ncclCommInitRank");
  ctx->comm.nccl = comm;
  ctx->comm.stream = ctx->stream;
  ctx->comm.rank = rank;
  ctx->comm.world = world;
  ctx->comm.shard_set = true;
  MFM_CATCH(ctx)
}

int mfm_set_shard(mfm_ctx *ctx, int32_t rank, int32_t world) {
  MFM_TRY(ctx)
  if (world < 1 || rank < 0 || rank >= world) throw Error(MFM_ERR_INVALID, "bad rank / world size");
  ctx->comm.rank = rank;
  ctx->comm.world = world;
  ctx->comm.shard_set = true;
  MFM_CATCH(ctx)
}

int mfm_comm_info(const mfm_ctx *ctx, int32_t *n_ranks, char *path, int64_t path_cap) {
  // evidence for a scaling run: the communicator's own rank count (ncclCommCount) and the librccl that carries it
  if (n_ranks) *n_ranks = 0;
  if (path && path_cap > 0) path[0] = 0;
  if (!ctx || !ctx->comm.nccl) return MFM_OK;
  try {
    Rccl &r = Rccl::get();
    int n = 0;
    if (r.CommCount) r.check(r.CommCount(ctx->comm.nccl, &n), "ncclCommCount");
    if (n_ranks) *n_ranks = n;
    if (path && path_cap > 0) {
      std::strncpy(path, r.path.c_str(), (size_t)path_cap - 1);
      path[path_cap - 1] = 0;
    }
    return MFM_OK;
  } catch (const std::exception &ex) {
    g_global_error = ex.what();
    return MFM_ERR_RUNTIME;
  }
}

int mfm_comm_stats(const mfm_ctx *ctx, int64_t *calls, int64_t *doubles) {
  if (calls) *calls = ctx->comm.calls;
  if (doubles) *doubles = ctx->comm.doubles;
  return MFM_OK;
}

int mfm_set_main_levels(mfm_ctx *ctx, const int32_t *level, int64_t D0) {
  MFM_TRY(ctx)
  if (ctx->finalized) throw Error(MFM_ERR_RUNTIME, "mfm_set_main_levels must be called before mfm_finalize");
  ctx->hlevels.assign(level, level + D0);
  MFM_CATCH(ctx)
}

int mfm_set_row_offset(mfm_ctx *ctx, int64_t first_global_row) {
  ctx->row_offset = first_global_row;
  return MFM_OK;
}

int mfm_synchronize(mfm_ctx *ctx) {
  MFM_TRY(ctx)
  ctx->sync_and_check();
  MFM_CATCH(ctx)
}

int mfm_set_main(mfm_ctx *ctx, int64_t N, int64_t D0, const int64_t *indptr, const int32_t *indices, const double *data,
                 const double *y) {
  MFM_TRY(ctx)
  if (ctx->finalized) throw Error(MFM_ERR_RUNTIME, "design already finalized");
  // straight to the device (validated on the way, no host copy: at config 5 the copy, its page faults and giving the 2.4 GB back
  // cost 0.6 s); the host planners that want the table download it (host_main)
  ctx->use_device();
  ctx->hX = HostCsr();
  ctx->hX_valid = false;
  ctx->X = DevSparse();
  ctx->X.upload_raw(N, D0, indptr, indices, data);
  ctx->y.upload(y, (size_t)N);
  ctx->N = N;
  ctx->D0 = D0;
  ctx->main_set = true;
  MFM_CATCH(ctx)
}

int mfm_add_block(mfm_ctx *ctx, int64_t B, int64_t Db, const int64_t *indptr, const int32_t *indices, const double *data,
                  const int64_t *original_to_block) {
  MFM_TRY(ctx)
  if (ctx->finalized) throw Error(MFM_ERR_RUNTIME, "design already finalized");
  mfm_ctx::HostBlock hb;
  hb.X = make_host_csr(B, Db, indptr, indices, data);
  int64_t N = ctx->N;
  if (B >= (int64_t)2147483647) throw Error(MFM_ERR_INVALID, "a relation block must have fewer than 2^31 rows");
  std::atomic<int> bad(0);
  hb.map.reset(new int32_t[(size_t)std::max<int64_t>(N, 1)]);  // (kept as 32-bit indices: half the bytes of the copy, and what the device wants)
  parallel_ranges(N, [&](int64_t lo, int64_t hi) {  // definitions.hpp:38-41
    for (int64_t t = lo; t < hi; t++) {
      if (original_to_block[t] < 0 || original_to_block[t] >= B) bad = 1;
      hb.map[t] = (int32_t)original_to_block[t];
    }
  });
  if (bad) throw Error(MFM_ERR_RUNTIME, "index mapping points to non-existing row.");
  ctx->hblocks.push_back(std::move(hb));
  MFM_CATCH(ctx)
}

int mfm_set_groups(mfm_ctx *ctx, const int32_t *group_index, int64_t D, int32_t G) {
  MFM_TRY(ctx)
  if (ctx->finalized) throw Error(MFM_ERR_RUNTIME, "design already finalized");
  if (G <= 0) throw Error(MFM_ERR_INVALID, "n_groups must be positive");
  ctx->hgroup.assign(group_index, group_index + D);
  std::vector<char> seen((size_t)G, 0);
  for (auto g : ctx->hgroup) {
    if (g < 0 || g >= G) throw Error(MFM_ERR_INVALID, "group index out of range");
    seen[g] = 1;
  }
  for (int g = 0; g < G; g++)  // FMLearningConfig.hpp:33-40
    if (!seen[g]) throw Error(MFM_ERR_INVALID, "No matching index for group index " + std::to_string(g) + " found.");
  ctx->G = G;
  MFM_CATCH(ctx)
}

static int64_t res_min_rows(const mfm_ctx *) {
  return std::getenv("MFM_RES_MIN_ROWS") ? std::atoll(std::getenv("MFM_RES_MIN_ROWS")) : ((int64_t)1 << 20);
}

// The persistent sweep's layout (build(plan, workgroups): the device builder of mfm_res_plan.hpp or the host builder) and the
// claim on its CUs.
static void plan_resident(mfm_ctx *c, const std::function<void(ResPlan &, int)> &build, bool tlog) {
  int n_cu = 0;
  MFM_HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device));
  const int n_cu_dev = n_cu;
  if (const char *e = std::getenv("MFM_RES_CUS")) {
    n_cu = std::max(1, std::min(n_cu, std::atoi(e)));
    build(c->res, n_cu);
  } else {
    // one CU per XCD stays free when the rows still fit the others (config 3: 40 323 of 40 959 slots per workgroup): the small
    // kernels between two launches, the side stream's single-workgroup draws and the slot-order scorer's last workgroup no
    // longer queue behind each other -- 308 -> 314-317 it/s at config 3 (MI355X, profiles/r04_q_res_cus.txt)
    bool done = false;
    if (n_cu >= 64 && c->N / (n_cu - 8) < 40500) {
      build(c->res, n_cu - 8);
      done = c->res.ready;
      if (done) n_cu -= 8;
    }
    if (!done) build(c->res, n_cu);
  }
  c->res_plan_cus = n_cu;
  if (c->res.ready) {
    // all G workgroups must be resident at once: one per CU must fit (registers + LDS), and the CUs must be ours
    int per_cu = 0;
    const hipError_t oe = res_occupancy(c->res, &per_cu);
    std::string why;
    if (oe != hipSuccess || per_cu < 1) {
      c->res.fail("the persistent kernel does not fit a CU (occupancy query)");
    } else if (!ResidentBudget::get().acquire(c->device, n_cu_dev, c->res.G, why)) {
      c->res.fail(why.c_str());
      c->res_refused = true;
    } else {
      c->res_claim = c->res.G;
    }
  }
  c->res_fills_device = c->res.ready && c->res.G > n_cu - n_cu / 4;
  if (tlog)
    std::fprintf(stderr, "[mfm_finalize] resident plan: %s (G=%d RV=%d RL=%d umax=%d runs=%lld lds=%zu)\n",
                 c->res.ready ? "ready" : c->res.why.c_str(), c->res.G, c->res.RV, c->res.RL, c->res.umax, (long long)c->res.n_runs,
                 c->res.lds_bytes);
}

// Before anything else is planned: does the table take the persistent sweep? (the layout from the device CSR, mfm_res_plan.hpp)
static void try_resident_first(mfm_ctx *c, const std::function<void(const char *)> &lap) {
  static const char *const off[] = {"MFM_NO_RESIDENT", "MFM_RES_HOST_PLAN", "MFM_RES_PROF", "MFM_HOST_TRANSPOSE", "MFM_NO_SOA",
                                    "MFM_NO_FUSED_QBUILD", "MFM_NO_FUSED_NEXT", "MFM_NO_MF", "MFM_NO_FUSED_TWO", "MFM_NO_FUSED_STATS",
                                    "MFM_HOST_LEVELS", "MFM_NO_SCATTER", "MFM_TILE_BITS"};
  for (const char *e : off)
    if (std::getenv(e)) return;
  if (std::getenv("MFM_QFREE") && std::atoi(std::getenv("MFM_QFREE"))) return;
  if (c->comm.active() || !c->hblocks.empty() || !c->hlevels.empty() || !c->X.unit || c->X.ell_width != 2 || c->K < 1 ||
      c->N < res_min_rows(c))
    return;
  const bool tlog = std::getenv("MFM_SETUP_TIMING") != nullptr;
  plan_resident(c, [&](ResPlan &rp, int n_cu) { res_plan_build_device(rp, c->X, &c->hgroup, n_cu, c->stream); }, tlog);
  lap("resident plan (device)");
  // (tests: with MFM_PLAN_CHECK everything else is built too and the host builder's layout compared)
  c->main_lazy = c->res.ready && !std::getenv("MFM_PLAN_CHECK") && !std::getenv("MFM_EAGER_PLANS");
}

// The main table's generic structures: X_t (device transpose, copied back for the planner), the level plans of both sweeps, the
// row tiles. Xt: filled here unless the caller already has it (MFM_HOST_TRANSPOSE).
static void plan_main_table(mfm_ctx *c, HostCsr &Xt, const std::function<void(const char *)> &lap) {
    if (!std::getenv("MFM_HOST_TRANSPOSE") || (int64_t)Xt.ptr.size() != c->D0 + 1) {
      Xt = transpose_device(c->X, c->stream);
      lap("transpose (device) + copy back");
    }
    c->plan_V.sharded = c->plan_W.sharded = c->comm.active();
    c->plan_V.given_levels = c->plan_W.given_levels = c->hlevels;
    // scattered levels: LDS row tiles of 2^tile_bits {e, q} records (64 KiB by default: two workgroups per CU)
    // (short tables: 1024-row tiles -- a 4096-row tile leaves most of the 256 CUs without a workgroup and the pass is bound by
    // the life time of one workgroup: ML-100k shape 3270 -> 4040 it/s)
    int tile_bits = (c->N < ((int64_t)1 << 20) && !c->comm.active()) ? 10 : 12;
    if (const char *e = std::getenv("MFM_TILE_BITS")) tile_bits = std::atoi(e);
    if (tile_bits < 9 || tile_bits > 13) tile_bits = 0;  // 0: L2-window path (k_scat_*)
    c->plan_V.tile_bits = c->plan_W.tile_bits = tile_bits;
    // one plan serves the three latent policies of the main table: size the co-resident launch for all of them
    const int coop_v = std::min({coop_capacity<PMainV>(), coop_capacity<PMainVe<false, false>>(), coop_capacity<PMainVe<true, false>>(),
                                 coop_capacity<PMainVsq<false>>(), coop_capacity<PMainVsq<true>>(), coop_capacity<PMainVs>()});
    // Row-sharded fused tile path: which first-level columns need an all-reduce of their statistics? Those
    // with rows on more than one rank -- the same ("special") set on every rank, from two all-reduces of
    // per-column indicators (columns empty on every rank are drawn from the prior by every rank itself).
    bool try_fused = c->comm.active() && c->hblocks.empty() && !c->hlevels.empty() && tile_bits > 0 && c->N > 0 &&
                     !std::getenv("MFM_NO_SHARDED_FUSED") && !std::getenv("MFM_NO_SOA");
    if (c->comm.active()) {
      // the predicate has rank-local inputs (an empty shard, the environment): every rank must enter the collectives
      // below or none -- agree first
      double no = try_fused ? 0.0 : 1.0;
      DevBuf<double> d;
      d.upload(&no, 1);
      c->comm.allreduce(d.p, 1);
      MFM_HIP_CHECK(hipStreamSynchronize(c->stream));
      MFM_HIP_CHECK(hipMemcpy(&no, d.p, sizeof(double), hipMemcpyDeviceToHost));
      try_fused = no == 0.0;
    }
    std::vector<double> col_cnt;
    if (try_fused) {
      const int64_t D0 = c->D0, RB = (int64_t)1 << tile_bits;
      std::vector<double> h((size_t)2 * D0, 0.0);
      col_cnt.resize((size_t)D0);
      for (int64_t j = 0; j < D0; j++) {
        col_cnt[j] = h[j] = (double)(Xt.ptr[j + 1] - Xt.ptr[j]);
        h[D0 + j] = h[j] > (double)RB ? 1.0 : 0.0;
      }
      DevBuf<double> d;
      d.upload(h);
      c->comm.allreduce(d.p, 2 * D0);
      MFM_HIP_CHECK(hipStreamSynchronize(c->stream));
      std::vector<double> g((size_t)2 * D0);
      MFM_HIP_CHECK(hipMemcpy(g.data(), d.p, g.size() * sizeof(double), hipMemcpyDeviceToHost));
      std::vector<double> sh((size_t)D0);
      for (int64_t j = 0; j < D0; j++) sh[j] = (h[j] > 0 && h[j] < g[j]) ? 1.0 : 0.0;
      d.upload(sh);
      c->comm.allreduce(d.p, D0);
      MFM_HIP_CHECK(hipStreamSynchronize(c->stream));
      MFM_HIP_CHECK(hipMemcpy(sh.data(), d.p, sh.size() * sizeof(double), hipMemcpyDeviceToHost));
      std::vector<char> special((size_t)D0, 0);
      for (int64_t j = 0; j < D0; j++)
        special[j] = c->hlevels[j] != 0 ? 0 : sh[j] > 0 ? 1 : g[j] == 0 ? 2 : 0;
      c->plan_V.sharded_tiles = true;
      c->plan_V.special = std::move(special);
    }
    c->plan_V.group_of = &c->hgroup;
    // the row-tile layouts of scattered levels are packed on the device from the device-resident CSC
    DevCscView dev_view = DevCscView{c->X.colptr.p, c->X.rowidx.p, c->X.cval.p, c->stream};
    dev_view.rowptr = c->X.rowptr.p;
    dev_view.colidx = c->X.colidx.p;
    dev_view.n_rows = c->N;
    dev_view.n_cols = c->D0;
    dev_view.ell = (int)c->X.ell_width;
    c->plan_V.dev_csc = c->X.colptr.p ? &dev_view : nullptr;
    c->plan_W.dev_csc = c->plan_V.dev_csc;
    c->plan_V.build(Xt, PMainV::R_W16, PMainV::R_WG, coop_v, true, c->X.unit);
    lap("plan_V");
    if (try_fused) {
      // every rank must take the same path: agree
      double bad = plan_supports_sharded_fused(c->plan_V) ? 0.0 : 1.0;
      DevBuf<double> d;
      d.upload(&bad, 1);
      c->comm.allreduce(d.p, 1);
      MFM_HIP_CHECK(hipStreamSynchronize(c->stream));
      MFM_HIP_CHECK(hipMemcpy(&bad, d.p, sizeof(double), hipMemcpyDeviceToHost));
      if (bad > 0) {
        c->plan_V = StepPlan();
        c->plan_V.sharded = true;
        c->plan_V.given_levels = c->hlevels;
        c->plan_V.tile_bits = tile_bits;
        c->plan_V.group_of = &c->hgroup;
        c->plan_V.dev_csc = c->plan_W.dev_csc;
        c->plan_V.build(Xt, PMainV::R_W16, PMainV::R_WG, coop_v, true, c->X.unit);
      } else {
        c->sharded_fused = true;
        // model synchronisation after the sweep: a non-special first-level column is contributed by the rank
        // that holds its rows, every other column (identical on all ranks) by rank 0 of the communicator
        // (a caller of the older sequence mfm_set_allreduce + mfm_set_row_offset never said which rank it is: the shard
        // that starts at global row 0 is the root then)
        const bool root = c->comm.shard_set ? c->comm.rank == 0 : c->row_offset == 0;
        std::vector<double> mask((size_t)c->D, root ? 1.0 : 0.0);
        for (int64_t j = 0; j < c->D0; j++)
          if (c->hlevels[j] == 0 && c->plan_V.special[j] == 0) mask[j] = col_cnt[j] > 0 ? 1.0 : 0.0;
        c->sync_mask.upload(mask);
      }
    }
    c->plan_W.build(Xt, PMainW::R_W16, PMainW::R_WG, coop_capacity<PMainW>(), true, c->X.unit,
                    std::getenv("MFM_NO_PLAN_TWIN") ? nullptr : &c->plan_V);
    lap("plan_W");
    c->plan_V.dev_csc = c->plan_W.dev_csc = nullptr;  // (the view lives on this frame: build() is its only reader)
    c->ls.reserve_cols(std::max(c->plan_V.max_cols_scat, c->plan_W.max_cols_scat));
    if (c->comm.active()) {
      c->ls.reserve_cols(c->D0);
      c->ls.reserve_stats(c->D0);
    }
}

// which of the main table's sweep forms the plans support (flags of mfm_plan_flags) and their buffers
static void decide_main_paths(mfm_ctx *c) {
  // q-free latent sweep (PMainVe), opt-in (MFM_QFREE=1): pays off when the levels' rows are contiguous; needs
  // short rows, no relation blocks, no sharding, single-pass PAR levels, no row-tile levels
  c->qfree = !c->comm.active() && c->blocks.empty() && c->X.rows > 0 && c->X.avg_row_nnz <= 4.0 &&
             plan_is_single_pass_par(c->plan_V) && std::getenv("MFM_QFREE") && std::atoi(std::getenv("MFM_QFREE"));
  if (c->qfree) c->ec.alloc((size_t)c->N);
  // split e / q layout for update_V (run_plan_soa)
  c->soa = !c->qfree && !c->comm.active() && c->blocks.empty() && c->N > 0 && plan_supports_soa(c->plan_V) &&
           !std::getenv("MFM_NO_SOA") && !std::getenv("MFM_NO_FUSED_QBUILD");
  if (c->sharded_fused) {
    c->ec.alloc((size_t)c->N);
    c->qc.alloc((size_t)c->N);
    // no first-level column straddles a rank boundary (and none is longer than ... any length is fine): the two-field pass
    c->mf = plan_supports_mf(c->plan_V) && c->plan_V.n_special == 0 && !std::getenv("MFM_NO_MF") &&
            !std::getenv("MFM_NO_FUSED_TWO") && !std::getenv("MFM_NO_FUSED_STATS");
    {  // every rank must take the same path
      double no = c->mf ? 0.0 : 1.0;
      DevBuf<double> d;
      d.upload(&no, 1);
      c->comm.allreduce(d.p, 1);
      MFM_HIP_CHECK(hipStreamSynchronize(c->stream));
      MFM_HIP_CHECK(hipMemcpy(&no, d.p, sizeof(double), hipMemcpyDeviceToHost));
      c->mf = no == 0.0;
    }
  }
  if (c->soa) {
    c->ec.alloc((size_t)c->N);
    c->qc.alloc((size_t)c->N);
    c->fuse_next = plan_supports_fused_next(c->plan_V) && !std::getenv("MFM_NO_FUSED_NEXT");
    c->mf = c->fuse_next && plan_supports_mf(c->plan_V) && !std::getenv("MFM_NO_MF") && !std::getenv("MFM_NO_FUSED_TWO") &&
            !std::getenv("MFM_NO_FUSED_STATS");
  }
}

// The jump polynomials of the parallel MT19937 generator for a problem of D features, rank K, G groups, sized for one iteration's
// draws (an upper estimate of the generating workgroups): computed on a helper thread, cached per process. Returns the
// generator's workgroup count (par_blocks; `keep` when the serial generator serves the problem).
static int rng_prefetch_jumps(int64_t D, int K, int G, int keep) {
  if (std::getenv("MFM_RNG_SERIAL")) return keep;
  const double normals = (double)D * (K + 1) + 4.0 * G * (K + 1) + 16;
  const double need = normals * (16.0 / 3.14159265358979) * 1.02 + 6.0 * 2.4 * std::sqrt(normals + 1.0) + 4096.0 * (2 + 2.0 * G * (K + 1)) + 2e6;
  const int64_t blocks = (int64_t)(need / MT_N) + 2;
  if (blocks <= MT_PAR_BLOCKS) return keep;
  const int64_t gblocks = blocks * mt_gen_batch(need);  // (the generator is asked for several iterations at a time)
  const int par_blocks = mt_par_blocks_for(gblocks);
  mtjump::JumpCache::inst().prefetch(par_blocks, (int)((gblocks + par_blocks - 1) / par_blocks) + 1);
  return par_blocks;
}

int mfm_rng_prepare(int64_t n_features, int32_t rank, int32_t n_groups) {
  try {
    if (n_features < 0 || rank < 0 || n_groups < 1) return MFM_ERR_INVALID;
    (void)rng_prefetch_jumps(n_features, rank, n_groups, 0);
    return MFM_OK;
  } catch (...) {
    return MFM_ERR_RUNTIME;
  }
}

void mfm_ctx::ensure_main_plans() {
  if (!main_lazy) return;
  main_lazy = false;
  use_device();
  HostCsr Xt;
  plan_main_table(this, Xt, [](const char *) {});
  ls.reserve(std::max({plan_V.max_hchunks, plan_W.max_hchunks, 1}), std::max({plan_V.max_huge, plan_W.max_huge, 1}));
  decide_main_paths(this);
}

int mfm_finalize(mfm_ctx *ctx, int32_t rank) {
  MFM_TRY(ctx)
  if (ctx->finalized) throw Error(MFM_ERR_RUNTIME, "design already finalized");
  if (rank < 0) throw Error(MFM_ERR_INVALID, "rank must be non-negative");
  mfm_ctx *c = ctx;
  if (!c->main_set) {  // (no main table was given: an empty one)
    c->N = c->D0 = 0;
    c->X.upload(HostCsr(), nullptr);
    c->y.alloc(0);
  }
  c->D = c->D0;
  for (auto &hb : c->hblocks) c->D += hb.X.cols;
  if (c->G == 0) throw Error(MFM_ERR_RUNTIME, "mfm_set_groups has not been called");
  if ((int64_t)c->hgroup.size() != c->D)
    throw Error(MFM_ERR_INVALID, "group_index has " + std::to_string(c->hgroup.size()) + " entries but the design has " +
                                     std::to_string(c->D) + " features");
  if (c->N >= (int64_t)2147483647) throw Error(MFM_ERR_INVALID, "N must be < 2^31 per GPU");
  c->K = rank;
  c->KS = (rank + 1) & ~1;
  // the parallel generator's jump polynomials (mfm_rng_set_program needs them right after this call): start computing
  // them now on a helper thread (a caller that knows the problem's size earlier has already asked: mfm_rng_prepare)
  c->rng.par_blocks = rng_prefetch_jumps(c->D, c->K, c->G, c->rng.par_blocks);
  const bool tlog = std::getenv("MFM_SETUP_TIMING") != nullptr;
  auto tnow = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_prev = tnow();
  CodeWarmup::get().join();
  auto lap = [&](const char *what) {
    if (!tlog) return;
    const double t = tnow();
    std::fprintf(stderr, "[mfm_finalize] %-28s %7.3f s\n", what, t - t_prev);
    t_prev = t;
  };
  lap("(code object ready)");
  // main table
  HostCsr Xt_keep;  // (X_t outlives the planner's scope: the resident layout is built from it once the path is known)
  {
    HostCsr &Xt = Xt_keep;
    if (std::getenv("MFM_HOST_TRANSPOSE")) {
      Xt = transpose_host(c->host_main());
      lap("transpose (host)");
      c->X.upload_csc(Xt);
      lap("upload CSC");
    }
    // a row of unit-valued one-hot fields + relation blocks on one GPU: update_w / update_V / update_e on index tuples, no
    // q-cache (mfm_cell.hpp). Decided first: when it takes the design, X_t, the main table's level plans and row tiles and
    // the blocks' inverse maps are never used and are not built (config 5: 1.3 s of mfm_finalize and 3 GB of HBM).
    {
      const int64_t cell_min_rows = std::getenv("MFM_CELL_MIN_ROWS") ? std::atoll(std::getenv("MFM_CELL_MIN_ROWS")) : ((int64_t)1 << 20);
      // row-sharded: every rank plans its own rows; what must be the same everywhere (fields, streams, the verdict) is summed
      // over the ranks inside the planner
      const bool sh = c->comm.active();
      auto sum_ranks = [&](std::vector<double> &v) {
        DevBuf<double> d;
        d.upload(v);
        c->comm.allreduce(d.p, (int64_t)v.size());
        MFM_HIP_CHECK(hipStreamSynchronize(c->stream));
        MFM_HIP_CHECK(hipMemcpy(v.data(), d.p, v.size() * sizeof(double), hipMemcpyDeviceToHost));
      };
      // (also tables of three or more one-hot fields WITHOUT relation blocks on one GPU: one pass per field instead of the
      //  row-tile passes of run_sweep_soa_multi; two-field tables keep the persistent sweep / the two-field pass)
      const bool flat = c->hblocks.empty() && !sh && c->X.ell_width >= 3 && !std::getenv("MFM_NO_CELL_FLAT");
      bool try_cell = (!c->hblocks.empty() || flat) && c->K > 0 && !std::getenv("MFM_NO_CELL") && (!sh || !std::getenv("MFM_NO_CELL_SHARDED"));
      if (!sh) {
        try_cell = try_cell && c->X.unit && c->X.ell_width >= 1 && c->N >= cell_min_rows;
      } else if (try_cell) {  // (the same decision on every rank: the rows of all of them count, an empty shard has no say)
        std::vector<double> v{(double)c->N, (c->N > 0 && !(c->X.unit && c->X.ell_width >= 1)) ? 1.0 : 0.0, c->comm.shard_set ? 0.0 : 1.0};
        sum_ranks(v);
        try_cell = v[0] >= (double)cell_min_rows && v[1] == 0.0 && v[2] == 0.0;
      }
      if (try_cell) {
        int n_cu = 0;
        MFM_HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device));
        if (const char *e = std::getenv("MFM_CELL_GROUPS")) n_cu = std::max(1, std::atoi(e));
        std::vector<CellBlockIn> bin;
        for (auto &hb : c->hblocks) bin.push_back(CellBlockIn{hb.map.get(), hb.X.rows});
        const int sh_rank = c->comm.shard_set ? c->comm.rank : -1;
        if (std::getenv("MFM_CELL_HOST_PLAN")) {
          if (sh)
            cell_plan_build(c->cell, c->host_main(), bin, n_cu, c->stream, sh_rank, c->comm.world, sum_ranks);
          else
            cell_plan_build(c->cell, c->host_main(), bin, n_cu, c->stream);
        } else {
          // one GPU: the plan is built on the device from the CSR and the blocks' maps (uploaded here, kept for the blocks)
          const size_t nb = c->hblocks.size();
          c->pre_maps.clear();
          c->pre_maps.resize(nb);
          {
            std::vector<std::thread> pool;
            std::vector<std::exception_ptr> errs(nb);
            auto up = [&](size_t b) {
              try {
                MFM_HIP_CHECK(hipSetDevice(c->device));
                c->pre_maps[b].upload(c->hblocks[b].map.get(), (size_t)c->N);
              } catch (...) {
                errs[b] = std::current_exception();
              }
            };
            for (size_t b = 0; b + 1 < nb; b++) pool.emplace_back(up, b);
            if (nb) up(nb - 1);
            for (auto &t : pool) t.join();
            for (auto &e : errs)
              if (e) std::rethrow_exception(e);
          }
          lap("block maps to the device");
          std::vector<CellBlockDev> bdev;
          for (size_t b = 0; b < nb; b++) bdev.push_back(CellBlockDev{c->pre_maps[b].p, c->hblocks[b].X.rows});
          // (row-sharded: every rank plans its own rows on its own device, the agreements are summed over the ranks inside)
          auto host_plan = [&](CellPlan &cp) {
            if (sh)
              cell_plan_build(cp, c->host_main(), bin, n_cu, c->stream, sh_rank, c->comm.world, sum_ranks);
            else
              cell_plan_build(cp, c->host_main(), bin, n_cu, c->stream);
          };
          if (sh)
            cell_plan_build_device(c->cell, c->X, bdev, n_cu, c->stream, sh_rank, c->comm.world, sum_ranks);
          else
            cell_plan_build_device(c->cell, c->X, bdev, n_cu, c->stream);
          if (!c->cell.ready && c->cell.why.rfind("device planner:", 0) == 0) {
            host_plan(c->cell);  // (a shape only the host planner handles: the same verdict on every rank)
          } else if (std::getenv("MFM_PLAN_CHECK")) {  // tests: the host planner must give the same plan, array for array
            lap("cell plan (device)");
            CellPlan chk;
            host_plan(chk);
            if (chk.ready != c->cell.ready) throw Error(MFM_ERR_RUNTIME, "plan check: device and host cell planners disagree: device '" + c->cell.why + "' host '" + chk.why + "'");
            if (chk.ready) {
              const std::string diff = cell_plan_compare(c->cell, chk, c->stream);
              if (!diff.empty()) throw Error(MFM_ERR_RUNTIME, "plan check: device and host cell plans differ (" + diff + ")");
            }
          }
        }
        c->cell_w = c->cell.ready && !std::getenv("MFM_NO_CELL_W");
        if (tlog)
          std::fprintf(stderr, "[mfm_finalize] cell plan: %s (G=%d umax=%lld streams=%zu fields=%zu item32=%d)\n",
                       c->cell.ready ? "ready" : c->cell.why.c_str(), c->cell.G, (long long)c->cell.umax, c->cell.streams.size(),
                       c->cell.fields.size(), (int)c->cell.item32);
        lap("cell plan");
      }
    }
    // two-field unit-valued table on one GPU: the persistent sweep's layout is built FIRST, on the device, straight from the
    // CSR (mfm_res_plan.hpp). When it takes the table, X_t, the level plans and the row tiles of the per-factor passes (its
    // fall-back, and what a stand-alone mfm_sweep_w runs) are built only if a call ever needs them (ensure_main_plans):
    // config 3 0.26 s of mfm_finalize -> 0.1 s.
    if (!c->cell.ready) try_resident_first(c, lap);
    const bool lean = c->cell_w || c->main_lazy;
    if (!lean) plan_main_table(c, Xt, lap);
  }
  c->eq_.alloc_zero((size_t)c->N, c->stream);
  c->group.upload(c->hgroup);
  // features sorted by group (FMLearningConfig.hpp:41-46 group_vs_feature_index)
  {
    std::vector<int64_t> gptr((size_t)c->G + 1, 0);
    for (auto g : c->hgroup) gptr[g + 1]++;
    for (int g = 0; g < c->G; g++) gptr[g + 1] += gptr[g];
    std::vector<int32_t> fs((size_t)c->D);
    std::vector<int64_t> cur(gptr.begin(), gptr.end() - 1);
    for (int64_t j = 0; j < c->D; j++) fs[cur[c->hgroup[j]]++] = (int32_t)j;
    c->group_ptr.upload(gptr);
    c->feat_sorted.upload(fs);
    int64_t big = 1;
    for (int g = 0; g < c->G; g++) big = std::max(big, gptr[g + 1] - gptr[g]);
    c->gs_chunks = (int)((big + GS_CHUNK - 1) / GS_CHUNK);
  }
  // relation blocks
  int64_t off = c->D0;
  int max_chunks = std::max(c->plan_V.max_hchunks, c->plan_W.max_hchunks);
  int max_long = std::max(c->plan_V.max_huge, c->plan_W.max_huge);
  {
    // the blocks' device structures (transpose, plans, inverse maps: O(N) host passes each) are independent: host threads
    std::vector<std::unique_ptr<DevBlock>> built(c->hblocks.size());
    std::vector<std::exception_ptr> errs(c->hblocks.size());
    std::vector<int64_t> offs(c->hblocks.size());
    for (size_t b = 0; b < c->hblocks.size(); b++) {
      offs[b] = off;
      off += c->hblocks[b].X.cols;
    }
    auto build_one = [&](size_t b) {
      try {
        MFM_HIP_CHECK(hipSetDevice(c->device));
        built[b].reset(new DevBlock());
        built[b]->col_off = offs[b];
        built[b]->build(c->hblocks[b].X, c->hblocks[b].map.get(), c->N, c->KS, c->stream, c->cell_w,
                        b < c->pre_maps.size() ? &c->pre_maps[b] : nullptr);
      } catch (...) {
        errs[b] = std::current_exception();
      }
    };
    {
      (void)coop_capacity<PBlockV>();  // (function-local caches: filled before the threads start)
      (void)coop_capacity<PBlockW>();
      std::vector<std::thread> pool;
      const bool par = c->hblocks.size() > 1 && !std::getenv("MFM_SERIAL_BLOCK_BUILD");
      for (size_t b = 0; b < c->hblocks.size(); b++) {
        if (par && b + 1 < c->hblocks.size())
          pool.emplace_back(build_one, b);
        else
          build_one(b);
      }
      for (auto &t : pool) t.join();
    }
    for (size_t b = 0; b < built.size(); b++) {
      if (errs[b]) std::rethrow_exception(errs[b]);
      std::unique_ptr<DevBlock> &B = built[b];
      max_chunks = std::max({max_chunks, B->plan_V.max_hchunks, B->plan_W.max_hchunks});
      max_long = std::max({max_long, B->plan_V.max_huge, B->plan_W.max_huge});
      B->allreduce_fields(c->stream, c->comm, 6, 1);  // cardinality counts the rows of ALL ranks (definitions.hpp:65-68)
      c->blocks.push_back(std::move(B));
    }
  }
  c->ls.reserve(std::max(max_chunks, 1), std::max(max_long, 1));
  // state + scratch
  c->w.alloc_zero((size_t)c->D, c->stream);
  c->V.alloc_zero((size_t)c->D * c->K, c->stream);
  c->Vt.alloc_zero((size_t)c->D * c->KS, c->stream);
  c->z.alloc((size_t)std::max<int64_t>(c->D, 1) * std::max(c->K, 1));
  c->lam.alloc((size_t)c->G * std::max(c->K, 1));
  c->mu.alloc((size_t)c->G * std::max(c->K, 1));
  c->red_partial.alloc(REDUCE_BLOCKS);
  c->red_out.alloc((size_t)1 + (size_t)c->G * std::max(c->K, 1));
  c->scratch_n.alloc((size_t)std::max<int64_t>(c->N, 1));
  decide_main_paths(c);
  // two-field unit-valued table on one GPU: the whole update_V as one persistent launch, residual resident on chip
  // (small tables stay with the per-factor passes: the launch's fixed costs -- two grid barriers per sweep, census, slot-ordered
  //  residual -- outweigh the bytes it saves; ML-100k shape: fit() 3060 it/s resident, 3950 per-factor. MFM_RES_MIN_ROWS)
  if (c->res.ready) {
    // (built first, on the device) tests: the host builder on X_t must give the same layout, array for array
    if (!c->main_lazy && std::getenv("MFM_PLAN_CHECK")) {
      ResPlan chk;
      chk.build(Xt_keep, c->plan_V.h_level, &c->hgroup, c->res_plan_cus);
      const std::string diff = chk.ready ? res_plan_compare(c->res, chk, c->stream) : ("host builder: " + chk.why);
      if (!diff.empty()) throw Error(MFM_ERR_RUNTIME, "plan check: device and host resident layouts differ (" + diff + ")");
      lap("resident plan (host, check)");
    }
  } else if (!c->res_refused && c->soa && c->mf && c->X.unit && !c->comm.active() && c->N >= res_min_rows(c) && !std::getenv("MFM_NO_RESIDENT")) {
    plan_resident(c, [&](ResPlan &rp, int n_cu) { rp.build(Xt_keep, c->plan_V.h_level, &c->hgroup, n_cu); }, tlog);
    lap("resident plan (host)");
  }
  // Row-sharded with the shards cut between users (the two-field pass runs sharded, c->mf): the persistent sweep runs on every rank
  // over its own rows, the ranks' item sums meet inside the launch (mfm_res.hpp, XCH). The layout is built here -- items = the
  // level-1 columns of the GLOBAL design, so every rank numbers them alike -- and goes live when the caller has handed over the
  // peers' exchange buffers (mfm_peer_set); until then, and if any rank cannot take part, the per-factor passes run.
  if (c->comm.active() && c->sharded_fused && c->mf && c->comm.shard_set && c->comm.world <= RES_MAX_PEERS && c->comm.world > 1 &&
      !std::getenv("MFM_NO_RESIDENT") && !std::getenv("MFM_NO_SHARDED_RESIDENT")) {
    const int64_t min_rows = res_min_rows(c);
    const bool want = c->X.unit && c->X.ell_width == 2 && c->N >= std::max<int64_t>(1, min_rows / c->comm.world) && c->K > 0;
    bool mine = false;
    if (want) {
      std::vector<char> draw_empty((size_t)c->D0, 0);
      for (int64_t j = 0; j < c->D0; j++) draw_empty[j] = (size_t)j < c->plan_V.special.size() && c->plan_V.special[j] == 2;
      plan_resident(c, [&](ResPlan &rp, int n_cu) {
        rp.allow_overflow = false;
        res_plan_build_device(rp, c->X, &c->hgroup, n_cu, c->stream, &c->hlevels, &draw_empty);
      }, tlog);
      if (c->res.ready && std::getenv("MFM_PLAN_CHECK")) {
        ResPlan chk;
        chk.allow_overflow = false;
        chk.build(Xt_keep, c->hlevels, &c->hgroup, c->res_plan_cus, &draw_empty);
        const std::string diff = chk.ready ? res_plan_compare(c->res, chk, c->stream) : ("host builder: " + chk.why);
        if (!diff.empty()) throw Error(MFM_ERR_RUNTIME, "plan check: device and host resident layouts differ (sharded; " + diff + ")");
      }
      mine = c->res.ready;
      c->res.ready = false;
    }
    if (mine) {  // the exchange buffers BEFORE the ranks agree: a rank that gets no uncached memory votes against the layout
      try {
        c->res.alloc_exchange(c->comm.world, c->comm.rank, c->stream);
      } catch (const Error &) {
        mine = false;
      }
    }
    double bad = mine ? 0.0 : 1.0;
    {
      DevBuf<double> d;
      d.upload(&bad, 1);
      c->comm.allreduce(d.p, 1);
      MFM_HIP_CHECK(hipStreamSynchronize(c->stream));
      MFM_HIP_CHECK(hipMemcpy(&bad, d.p, sizeof(double), hipMemcpyDeviceToHost));
    }
    if (bad == 0.0) {
      c->res_sharded_pending = true;
    } else {
      c->drop_resident();
    }
    lap("resident plan (row-sharded)");
  }
  Xt_keep = HostCsr();
  MFM_HIP_CHECK(hipStreamSynchronize(c->stream));
  lap("blocks, state, scratch");
  // host copies are no longer needed
  lap("(sync)");
  c->hX = HostCsr();
  c->hX_valid = false;
  c->hy = std::vector<double>();
  c->hblocks.clear();
  c->pre_maps.clear();
  lap("host copies released");
  c->finalized = true;
  MFM_CATCH(ctx)
}

// ---- row-sharded persistent sweep: the ranks' exchange buffers ---------------------------------------------------------------
int mfm_peer_info(mfm_ctx *ctx, int32_t *pending, void **sum_buf, void **flag_buf, int64_t *sum_bytes, int64_t *flag_bytes) {
  MFM_TRY(ctx)
  ctx->need_final();
  const bool p = ctx->res_sharded_pending;
  if (pending) *pending = p ? 1 : 0;
  if (sum_buf) *sum_buf = p ? (void *)ctx->res.xsum.p : nullptr;
  if (flag_buf) *flag_buf = p ? (void *)ctx->res.xflag.p : nullptr;
  if (sum_bytes) *sum_bytes = p ? (int64_t)(ctx->res.xsum.n * sizeof(double)) : 0;
  if (flag_bytes) *flag_bytes = p ? (int64_t)(ctx->res.xflag.n * sizeof(unsigned long long)) : 0;
  MFM_CATCH(ctx)
}

int mfm_peer_set(mfm_ctx *ctx, int32_t world, int32_t rank, void *const *sum_bufs, void *const *flag_bufs) {
  MFM_TRY(ctx)
  ctx->need_final();
  mfm_ctx *c = ctx;
  if (!c->res_sharded_pending) throw Error(MFM_ERR_RUNTIME, "mfm_peer_set: this context has no row-sharded persistent sweep waiting for its peers");
  if (world != c->res.xworld || rank != c->res.xrank) throw Error(MFM_ERR_INVALID, "mfm_peer_set: world / rank differ from the communicator's");
  for (int r = 0; r < world; r++) {
    if (!sum_bufs[r] || !flag_bufs[r]) throw Error(MFM_ERR_INVALID, "mfm_peer_set: null buffer");
    c->res.peer_sum[r] = (double *)sum_bufs[r];
    c->res.peer_flag[r] = (unsigned long long *)flag_bufs[r];
  }
  if (c->res.peer_sum[rank] != c->res.xsum.p || c->res.peer_flag[rank] != c->res.xflag.p)
    throw Error(MFM_ERR_INVALID, "mfm_peer_set: this rank's own entry must be the buffers of mfm_peer_info");
  MFM_HIP_CHECK(hipStreamSynchronize(c->stream));
  c->res.peers_set = true;
  c->res.ready = true;
  c->res_sharded_pending = false;
  MFM_CATCH(ctx)
}

// the peers' replicas of (w, V): with them a first-level coefficient goes to every replica inside the launch and the model
// synchronisation after it is dropped. After mfm_peer_set, before the next sweep; every rank alike.
int mfm_peer_set_model(mfm_ctx *ctx, int32_t world, int32_t rank, void *const *w_bufs, void *const *V_bufs) {
  MFM_TRY(ctx)
  ctx->need_final();
  mfm_ctx *c = ctx;
  if (!c->res.ready || !c->res.peers_set) throw Error(MFM_ERR_RUNTIME, "mfm_peer_set_model: mfm_peer_set comes first");
  if (world != c->res.xworld || rank != c->res.xrank) throw Error(MFM_ERR_INVALID, "mfm_peer_set_model: world / rank differ from the communicator's");
  for (int r = 0; r < world; r++) {
    if (!w_bufs[r] || (c->K > 0 && !V_bufs[r])) throw Error(MFM_ERR_INVALID, "mfm_peer_set_model: null buffer");
    c->res.peer_w[r] = (double *)w_bufs[r];
    c->res.peer_V[r] = (double *)V_bufs[r];
  }
  if (c->res.peer_w[rank] != c->w.p || c->res.peer_V[rank] != c->V.p)
    throw Error(MFM_ERR_INVALID, "mfm_peer_set_model: this rank's own entry must be its own w / V (mfm_peer_model_info)");
  MFM_HIP_CHECK(hipStreamSynchronize(c->stream));
  c->res.peers_model = !std::getenv("MFM_NO_PEER_MODEL");
  MFM_CATCH(ctx)
}

int mfm_peer_model_info(mfm_ctx *ctx, void **w_buf, void **V_buf) {
  MFM_TRY(ctx)
  ctx->need_final();
  if (w_buf) *w_buf = ctx->w.p;
  if (V_buf) *V_buf = ctx->V.p;
  MFM_CATCH(ctx)
}

// one process per GPU: the buffers as four 64-byte IPC handles (sums, flags, w, V) ...
int mfm_peer_export(mfm_ctx *ctx, void *handles256) {
  MFM_TRY(ctx)
  ctx->need_final();
  if (!ctx->res_sharded_pending) throw Error(MFM_ERR_RUNTIME, "mfm_peer_export: no row-sharded persistent sweep waiting for its peers");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  hipIpcMemHandle_t h[4];
  std::memset(h, 0, sizeof h);
  MFM_HIP_CHECK(hipIpcGetMemHandle(&h[0], ctx->res.xsum.p));
  MFM_HIP_CHECK(hipIpcGetMemHandle(&h[1], ctx->res.xflag.p));
  MFM_HIP_CHECK(hipIpcGetMemHandle(&h[2], ctx->w.p));
  if (ctx->K > 0) MFM_HIP_CHECK(hipIpcGetMemHandle(&h[3], ctx->V.p));
  std::memcpy(handles256, h, sizeof h);
  MFM_CATCH(ctx)
}

// ... and every rank's handles ([world][256] bytes, rank order) opened and installed
int mfm_peer_import(mfm_ctx *ctx, int32_t world, int32_t rank, const void *all_handles) {
  MFM_TRY(ctx)
  ctx->need_final();
  mfm_ctx *c = ctx;
  if (!c->res_sharded_pending) throw Error(MFM_ERR_RUNTIME, "mfm_peer_import: no row-sharded persistent sweep waiting for its peers");
  if (world != c->res.xworld || rank != c->res.xrank) throw Error(MFM_ERR_INVALID, "mfm_peer_import: world / rank differ from the communicator's");
  void *sums[RES_MAX_PEERS], *flags[RES_MAX_PEERS], *ws[RES_MAX_PEERS], *Vs[RES_MAX_PEERS];
  for (int r = 0; r < world; r++) {
    if (r == rank) {
      sums[r] = c->res.xsum.p;
      flags[r] = c->res.xflag.p;
      ws[r] = c->w.p;
      Vs[r] = c->V.p;
      continue;
    }
    hipIpcMemHandle_t h[4];
    std::memcpy(h, (const char *)all_handles + (size_t)r * 256, sizeof h);
    MFM_HIP_CHECK(hipIpcOpenMemHandle(&sums[r], h[0], hipIpcMemLazyEnablePeerAccess));
    c->res.peer_mapped.push_back(sums[r]);
    MFM_HIP_CHECK(hipIpcOpenMemHandle(&flags[r], h[1], hipIpcMemLazyEnablePeerAccess));
    c->res.peer_mapped.push_back(flags[r]);
    MFM_HIP_CHECK(hipIpcOpenMemHandle(&ws[r], h[2], hipIpcMemLazyEnablePeerAccess));
    c->res.peer_mapped.push_back(ws[r]);
    Vs[r] = nullptr;
    if (c->K > 0) {
      MFM_HIP_CHECK(hipIpcOpenMemHandle(&Vs[r], h[3], hipIpcMemLazyEnablePeerAccess));
      c->res.peer_mapped.push_back(Vs[r]);
    }
  }
  int rc = mfm_peer_set(ctx, world, rank, sums, flags);
  if (rc != MFM_OK) return rc;
  // The peers' w / V (first-level coefficients written into every replica where they are drawn, no model all-reduce after the
  // launch) only on request, MYFM_PEER_MODEL=1: w and V are ordinary cached device memory which this GPU's later kernels read
  // through its L2 -- that a peer's write over xGMI is seen there rests on the L2 not holding the line (no kernel of an
  // iteration reads a first-level coefficient between its draw and the next launch ... on ONE GPU, where this was tested). Until
  // it has run on two GPUs the default keeps the model all-reduce after the launch (sync_model_sharded): the only memory another
  // GPU writes is then the exchange buffers, which are uncached (MYFM_PEER_MODEL=0 / 1 also separates "exchange wrong" from "model
  // write wrong" on a first real run).
  const char *pm = std::getenv("MYFM_PEER_MODEL");
  if (pm && std::atoi(pm) != 0) {
    rc = mfm_peer_set_model(ctx, world, rank, ws, Vs);
    if (rc != MFM_OK) return rc;
  }
  MFM_CATCH(ctx)
}

// give the row-sharded persistent sweep up (pending or live): the per-factor passes from the next sweep on. Every rank alike.
int mfm_peer_drop(mfm_ctx *ctx) {
  MFM_TRY(ctx)
  ctx->need_final();
  MFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  materialize_e(ctx);
  ctx->drop_resident();
  MFM_CATCH(ctx)
}

int mfm_set_residual_policy(mfm_ctx *ctx, int32_t recomputable) {
  if (!ctx) return MFM_ERR_INVALID;
  ctx->e_recomputable = recomputable != 0;
  return MFM_OK;
}

int64_t mfm_dim_all(const mfm_ctx *ctx) { return ctx->D; }

int mfm_plan_info(const mfm_ctx *ctx, int64_t *n_levels_main, int64_t *n_launches_per_sweep) {
  if (ctx->main_lazy) {  // (a two-field table on the persistent sweep: what its per-factor fall-back would be)
    if (n_levels_main) *n_levels_main = 2;
    if (n_launches_per_sweep) *n_launches_per_sweep = 1;
    return MFM_OK;
  }
  if (n_levels_main) *n_levels_main = (int64_t)ctx->plan_V.n_levels;
  if (n_launches_per_sweep) {
    int64_t n = ctx->plan_V.launches;
    for (auto &B : ctx->blocks) n += B->plan_V.launches + 3;
    *n_launches_per_sweep = n;
  }
  return MFM_OK;
}

int mfm_plan_flags(const mfm_ctx *ctx) {
  bool streamed = false;  // a relation block's feature chain runs as the streamed one-launch form (mfm_chain_stream.hpp)
  for (auto &B : ctx->blocks)
    for (const Step &st : B->plan_V.steps) streamed = streamed || (st.is_chain && st.chain.stream);
  return (streamed ? 1024 : 0) | (ctx->res.ready && ctx->res.RX > 0 ? 2048 : 0) | (ctx->qfree ? 1 : 0) | (ctx->X.unit ? 2 : 0) | (ctx->X.ell_width >= 0 ? 4 : 0) | (ctx->comm.active() ? 8 : 0) |
         (ctx->soa ? 16 : 0) | (ctx->fuse_next ? 32 : 0) | (ctx->sharded_fused ? 64 : 0) | (ctx->mf ? 128 : 0) |
         (ctx->res.ready ? 256 : 0) | (ctx->cell.ready ? 512 : 0);
}

// ---- state ------------------------------------------------------------------------------------
int mfm_set_state(mfm_ctx *ctx, double w0, const double *w, const double *V) {
  MFM_TRY(ctx)
  ctx->need_final();
  ctx->w0 = w0;
  ctx->ring.upload(ctx->w.p, w, (size_t)ctx->D * sizeof(double), ctx->stream);
  if (ctx->K) ctx->ring.upload(ctx->V.p, V, (size_t)ctx->D * ctx->K * sizeof(double), ctx->stream);
  MFM_CATCH(ctx)
}

int mfm_get_state(mfm_ctx *ctx, double *w0, double *w, double *V) {
  MFM_TRY(ctx)
  ctx->need_final();
  *w0 = ctx->w0;
  if (ctx->D)
    MFM_HIP_CHECK(hipMemcpyAsync(w, ctx->w.p, (size_t)ctx->D * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (ctx->K && ctx->D)
    MFM_HIP_CHECK(
        hipMemcpyAsync(V, ctx->V.p, (size_t)ctx->D * ctx->K * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ctx->sync_and_check();  // (a timed-out persistent sweep of the LAST iteration is reported here, not returned as a sample)
  MFM_CATCH(ctx)
}

int mfm_set_w0(mfm_ctx *ctx, double w0) {
  MFM_TRY(ctx)
  // a residual the sweep dropped is recomputed from the model AND this intercept: materialise it first with the intercept it
  // belongs to (a following mfm_shift_e would otherwise count the change twice)
  if (ctx->e_lost && w0 != ctx->w0) ensure_e(ctx);
  ctx->w0 = w0;
  MFM_CATCH(ctx)
}

int mfm_zero_w(mfm_ctx *ctx) {
  MFM_TRY(ctx)
  ctx->need_final();
  if (ctx->D) MFM_HIP_CHECK(hipMemsetAsync(ctx->w.p, 0, (size_t)ctx->D * sizeof(double), ctx->stream));
  MFM_CATCH(ctx)
}

static int get_eq(mfm_ctx *ctx, double *dst, int which) {
  MFM_TRY(ctx)
  ctx->need_final();
  materialize_e(ctx);
  if (ctx->N) {
    hipLaunchKernelGGL(k_get_eq, dim3(cdiv(ctx->N, WG)), dim3(WG), 0, ctx->stream, ctx->eq_rows(), ctx->scratch_n.p, ctx->N,
                       which);
    MFM_HIP_CHECK(
        hipMemcpyAsync(dst, ctx->scratch_n.p, (size_t)ctx->N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  }
  MFM_CATCH(ctx)
}
int mfm_get_e(mfm_ctx *ctx, double *e) { return get_eq(ctx, e, 0); }
int mfm_get_q(mfm_ctx *ctx, double *q) {
  if (ctx->finalized && ctx->q_stale_factor >= 0) {
    // the split-layout sweeps never store q during update_V; nothing on the device path reads it afterwards
    try {
      ctx->use_device();
      // (the blocks' feature sweeps update q_B in their records only: the compact copies the q-cache build gathers from are
      //  rebuilt from the factor's coefficients first)
      for (auto &B : ctx->blocks)
        if (ctx->cell.ready) block_rowcache(ctx->stream, ctx->timing, *B, ctx->V.p + (size_t)ctx->q_stale_factor * ctx->D + B->col_off, true);
      launch_qbuild(ctx, ctx->V.p + (size_t)ctx->q_stale_factor * ctx->D);
      ctx->q_stale_factor = -1;
    } catch (const std::exception &ex) {
      ctx->err = ex.what();
      return MFM_ERR_RUNTIME;
    }
  }
  return get_eq(ctx, q, 1);
}
int mfm_set_e(mfm_ctx *ctx, const double *e) {
  MFM_TRY(ctx)
  ctx->need_final();
  ctx->e_lost = false;  // (every residual is overwritten)
  ctx->e_is_residual = false;  // (the caller's own numbers: never dropped and recomputed)
  materialize_e(ctx);
  if (ctx->N) {
    MFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    MFM_HIP_CHECK(hipMemcpy(ctx->scratch_n.p, e, (size_t)ctx->N * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_set_eq, dim3(cdiv(ctx->N, WG)), dim3(WG), 0, ctx->stream, ctx->eq_rows(), ctx->scratch_n.p, ctx->N, 0);
    MFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  }
  MFM_CATCH(ctx)
}

// ---- reductions / hyper statistics ------------------------------------------------------------
int mfm_reduce_e(mfm_ctx *ctx, double *sum_e, double *sum_e2) {
  MFM_TRY(ctx)
  ctx->need_final();
  materialize_e(ctx);
  hipStream_t s = ctx->stream;
  {
    TimedLaunch t(ctx->timing, s, KC_REDUCE_E, 8.0 * ctx->N);
    hipLaunchKernelGGL(k_reduce_e_partial, dim3(REDUCE_BLOCKS), dim3(WG), 0, s, ctx->eq_rows(), ctx->N, ctx->red_partial.p);
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(WG), 0, s, ctx->red_partial.p, REDUCE_BLOCKS, ctx->red_out.p);
  }
  ctx->comm.allreduce(ctx->red_out.p, 2);
  double2 *h = ctx->readback(2);
  MFM_HIP_CHECK(hipMemcpyAsync(h, ctx->red_out.p, sizeof(double2), hipMemcpyDeviceToHost, s));
  MFM_HIP_CHECK(hipMemcpyAsync(h + 1, ctx->ls.error.p, sizeof(int), hipMemcpyDeviceToHost, s));
  MFM_HIP_CHECK(hipStreamSynchronize(s));
  ctx->check_coresident(*(const int *)(h + 1));
  *sum_e = h[0].x;
  *sum_e2 = h[0].y;
  MFM_CATCH(ctx)
}

int mfm_shift_e(mfm_ctx *ctx, double delta) {
  MFM_TRY(ctx)
  ctx->need_final();
  materialize_e(ctx);
  ctx->e_is_residual = false;  // (shifted by hand: no longer score - y of the stored model and intercept)
  if (ctx->N) {
    TimedLaunch t(ctx->timing, ctx->stream, KC_SHIFT_E, 16.0 * ctx->N);
    hipLaunchKernelGGL(k_shift_e, dim3(cdiv(ctx->N, WG)), dim3(WG), 0, ctx->stream, ctx->eq_rows(), ctx->N, delta);
    MFM_HIP_CHECK(hipGetLastError());
  }
  MFM_CATCH(ctx)
}

static int group_stats(mfm_ctx *ctx, const double *theta, int nf, const double *mu_host, double *sum, double *ssd) {
  MFM_TRY(ctx)
  ctx->need_final();
  hipStream_t s = ctx->stream;
  const int G = ctx->G;
  if (nf > 0) {
    ctx->ring.upload(ctx->mu.p, mu_host, (size_t)G * nf * sizeof(double), s);
    {
      TimedLaunch t(ctx->timing, s, KC_GROUP_STATS, 12.0 * ctx->D * nf);
      const int n_ch = std::max(1, ctx->gs_chunks);
      if (ctx->gs_partial.n < (size_t)G * nf * n_ch) ctx->gs_partial.alloc((size_t)G * nf * n_ch);
      hipLaunchKernelGGL(k_group_stats, dim3(G, nf, n_ch), dim3(WG), 0, s, theta, ctx->D, ctx->feat_sorted.p,
                         ctx->group_ptr.p, ctx->mu.p, G, ctx->gs_partial.p);
      hipLaunchKernelGGL(k_group_stats_final, dim3(cdiv(G * nf, 64)), dim3(64), 0, s, ctx->gs_partial.p, G * nf, n_ch,
                         ctx->red_out.p + 1);
    }
    // (peers write first-level coefficients into this rank's model inside their launches: nobody may start the next one before
    //  every rank has read the model here -- an all-reduce of one double orders that, see mfm_hyper_stats)
    if (ctx->res.peers_model && ctx->comm.active()) ctx->comm.allreduce(ctx->red_out.p, 1);
    double2 *h = ctx->readback((size_t)G * nf + 1);
    MFM_HIP_CHECK(hipMemcpyAsync(h, ctx->red_out.p + 1, (size_t)G * nf * sizeof(double2), hipMemcpyDeviceToHost, s));
    MFM_HIP_CHECK(hipStreamSynchronize(s));
    for (int i = 0; i < G * nf; i++) {
      sum[i] = h[i].x;
      ssd[i] = h[i].y;
    }
  }
  MFM_CATCH(ctx)
}
int mfm_group_stats_w(mfm_ctx *ctx, const double *mu_w, double *sum, double *ssd) {
  return group_stats(ctx, ctx->w.p, 1, mu_w, sum, ssd);
}
int mfm_group_stats_V(mfm_ctx *ctx, const double *mu_V, double *sum, double *ssd) {
  return group_stats(ctx, ctx->V.p, ctx->K, mu_V, sum, ssd);
}

// All the reductions the hyper-parameter updates of one iteration need, with ONE host synchronisation: sum e /
// sum e^2 (update_alpha, update_w0: FMTrainer.hpp:138, :223), the per-group sums of w and of every factor of V
// (update_lambda / update_mu: :150-216). None of them depends on what the iteration does before it uses them
// (the w statistics are taken before update_w, the V statistics before update_V, both with the previous
// iteration's mu), so they can all be taken up front.
int mfm_hyper_stats(mfm_ctx *ctx, int32_t need_e, const double *mu_w, const double *mu_V, double *sum_e, double *sum_e2,
                    double *sum_w, double *ssd_w, double *sum_V, double *ssd_V) {
  MFM_TRY(ctx)
  ctx->need_final();
  mfm_ctx *c = ctx;
  // the slot-order scorer has left sum e / sum e^2 as one partial per workgroup: no pass over the residual, which stays in
  // slot order for the next persistent launch
  const bool slot_sums = need_e && c->e_in_slots && c->slot_sums_valid;
  if (need_e && !slot_sums) materialize_e(ctx);
  hipStream_t s = c->stream;
  const int G = c->G, K = c->K, n_ch = std::max(1, c->gs_chunks);
  const size_t n_out = 1 + (size_t)G * (K + 1);
  if (c->hs_out.n < n_out) c->hs_out.alloc(n_out);
  if (c->hs_mu.n < (size_t)G * (K + 1)) c->hs_mu.alloc((size_t)G * (K + 1));
  if (c->gs_partial.n < (size_t)G * (K + 1) * n_ch) c->gs_partial.alloc((size_t)G * (K + 1) * n_ch);
  {  // (one copy for both mean vectors, ahead of the reduction it does not depend on: every async call costs the host ~10 us)
    c->hs_stage.resize((size_t)G * (K + 1));
    std::memcpy(c->hs_stage.data(), mu_w, (size_t)G * sizeof(double));
    if (K) std::memcpy(c->hs_stage.data() + G, mu_V, (size_t)G * K * sizeof(double));
    c->ring.upload(c->hs_mu.p, c->hs_stage.data(), c->hs_stage.size() * sizeof(double), s);
  }
  // (the group sums of w / V FIRST, the residual sums and their all-reduce behind them: with the row-sharded persistent sweep a
  //  peer writes the first-level coefficients it draws straight into this rank's w / V (mfm_peer_set_model), and it can start its
  //  next launch only behind this all-reduce -- which this rank enters after its group sums have read the model)
  {
    TimedLaunch t(c->timing, s, KC_GROUP_STATS, 12.0 * c->D * (K + 1));
    hipLaunchKernelGGL(k_group_stats, dim3(G, 1, n_ch), dim3(WG), 0, s, c->w.p, c->D, c->feat_sorted.p, c->group_ptr.p,
                       c->hs_mu.p, G, c->gs_partial.p);
    if (K)
      hipLaunchKernelGGL(k_group_stats, dim3(G, K, n_ch), dim3(WG), 0, s, c->V.p, c->D, c->feat_sorted.p, c->group_ptr.p,
                         c->hs_mu.p + G, G, c->gs_partial.p + (size_t)G * n_ch);
    hipLaunchKernelGGL(k_group_stats_final, dim3(cdiv(G * (K + 1), 64)), dim3(64), 0, s, c->gs_partial.p, G * (K + 1), n_ch,
                       c->hs_out.p + 1);
  }
  if (slot_sums) {
    TimedLaunch t(c->timing, s, KC_REDUCE_E, 16.0 * c->res.G);
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(WG), 0, s, c->res.sums.p, c->res.G, c->hs_out.p);
    c->comm.allreduce(c->hs_out.p, 2);
  } else if (need_e) {
    TimedLaunch t(c->timing, s, KC_REDUCE_E, 8.0 * c->N);
    hipLaunchKernelGGL(k_reduce_e_partial, dim3(REDUCE_BLOCKS), dim3(WG), 0, s, c->eq_rows(), c->N, c->red_partial.p);
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(WG), 0, s, c->red_partial.p, REDUCE_BLOCKS, c->hs_out.p);
    c->comm.allreduce(c->hs_out.p, 2);
  }
  if (!need_e && c->res.peers_model && c->comm.active()) c->comm.allreduce(c->hs_out.p, 1);  // (ordering only, see above)
  MFM_HIP_CHECK(hipGetLastError());
  double2 *h = c->readback(n_out + 1);
  MFM_HIP_CHECK(hipMemcpyAsync(h, c->hs_out.p, n_out * sizeof(double2), hipMemcpyDeviceToHost, s));
  MFM_HIP_CHECK(hipMemcpyAsync(h + n_out, c->ls.error.p, sizeof(int), hipMemcpyDeviceToHost, s));
  MFM_HIP_CHECK(hipStreamSynchronize(s));
  c->check_coresident(*(const int *)(h + n_out));
  if (need_e) {
    *sum_e = h[0].x;
    *sum_e2 = h[0].y;
  }
  for (int g = 0; g < G; g++) {
    sum_w[g] = h[1 + g].x;
    ssd_w[g] = h[1 + g].y;
  }
  for (int i = 0; i < G * K; i++) {
    sum_V[i] = h[1 + G + i].x;
    ssd_V[i] = h[1 + G + i].y;
  }
  MFM_CATCH(ctx)
}

// Row-sharded persistent sweep: a first-level coefficient was drawn where its rows live -- make the replicas identical again
// (every rank zeroes what it does not contribute, one all-reduce each for w and for the swept factors of V)
static void sync_model_sharded(mfm_ctx *c, bool with_w, int f_begin, int f_end) {
  hipStream_t s = c->stream;
  if (with_w) {
    hipLaunchKernelGGL(k_mask_rows, dim3(cdiv(c->D, 256)), dim3(256), 0, s, c->w.p, c->sync_mask.p, c->D, c->D);
    c->comm.allreduce(c->w.p, c->D);
  }
  if (f_end > f_begin) {
    const int64_t n = (int64_t)(f_end - f_begin) * c->D;
    double *Vb = c->V.p + (size_t)f_begin * c->D;
    hipLaunchKernelGGL(k_mask_rows, dim3(cdiv(n, 256)), dim3(256), 0, s, Vb, c->sync_mask.p, c->D, n);
    c->comm.allreduce(Vb, n);
  }
}

// ---- sweeps -----------------------------------------------------------------------------------
int mfm_sweep_w(mfm_ctx *ctx, double alpha, const double *lambda_w, const double *mu_w, const double *z) {
  MFM_TRY(ctx)
  ctx->need_final();
  if (!ctx->cell_w) materialize_e(ctx);
  mfm_ctx *c = ctx;
  hipStream_t s = c->stream;
  c->ring.upload(c->lam.p, lambda_w, (size_t)c->G * sizeof(double), s);
  c->ring.upload(c->mu.p, mu_w, (size_t)c->G * sizeof(double), s);
  const double *zdev = c->z.p;
  if (z) {
    c->ring.upload(c->z.p, z, (size_t)c->D * sizeof(double), s);
  } else {
    if (c->rng.current < 0 || c->rng.n_zw != c->D)
      throw Error(MFM_ERR_RUNTIME, "mfm_sweep_w(z = NULL) needs an acquired device random set with D z_w variates");
    zdev = c->rng.slot[c->rng.current].zw.p;
  }
  if (c->cell_w) {
    run_sweep_w_cell(c, zdev, alpha);
    return MFM_OK;
  }
  c->ensure_main_plans();
  SweepArgs a = main_args(c, c->w.p, zdev, c->lam.p, c->mu.p, alpha);
  const SweepClasses kcw{KC_SWEEP_W_LIGHT, KC_SWEEP_W_HEAVY, KC_SWEEP_W_COOP, KC_SWEEP_W_LSTATS, KC_SWEEP_W_LDRAW,
                         KC_SWEEP_W_LAPPLY, KC_SWEEP_W_CHAIN, KC_SWEEP_W_SCAT};
  if (c->comm.active())
    run_plan_sharded<PMainW>(s, c->timing, c->plan_W, a, c->ls, kcw, c->X.unit, c->comm);
  else
    run_plan<PMainW>(s, c->timing, c->plan_W, a, c->ls, kcw, c->X.unit);
  for (auto &B : c->blocks)
    block_sweep_w(s, c->timing, c->ls, *B, c->N, c->eq_rows(), c->w.p, zdev, c->group.p, c->lam.p, c->mu.p, alpha, c->comm);
  MFM_CATCH(ctx)
}

// update_w followed by update_V (BaseFMTrainer.hpp:143-148: the hyper-parameter draws between them read neither w nor e), with
// update_w0's residual shift (:226) in front. On a two-field one-hot table all of it is ONE persistent launch (mfm_res.hpp:
// the linear sweep is the latent sweep with h = 1); everywhere else the three calls in sequence.
int mfm_sweep_wV(mfm_ctx *ctx, double alpha, double e_shift, const double *lambda_w, const double *mu_w, const double *zw,
                 int32_t f_begin, int32_t f_end, const double *lambda_V, const double *mu_V, const double *zv) {
  {
    mfm_ctx *c = ctx;
    const bool fused = c && c->finalized && c->res.ready && f_begin < f_end && !std::getenv("MFM_RES_NO_LINEAR");
    if (!fused) {
      int rc = e_shift != 0.0 ? mfm_shift_e(ctx, e_shift) : MFM_OK;
      if (rc == MFM_OK) rc = mfm_sweep_w(ctx, alpha, lambda_w, mu_w, zw);
      if (rc == MFM_OK) rc = mfm_sweep_V(ctx, f_begin, f_end, alpha, lambda_V, mu_V, zv);
      return rc;
    }
  }
  MFM_TRY(ctx)
  ctx->need_final();
  mfm_ctx *c = ctx;
  // (every argument check before anything is enqueued)
  if (f_begin < 0 || f_end > c->K) throw Error(MFM_ERR_INVALID, "factor range out of bounds");
  if ((zw == nullptr) != (zv == nullptr)) throw Error(MFM_ERR_INVALID, "mfm_sweep_wV: give both variate arrays or none");
  if (!zw && (c->rng.current < 0 || c->rng.n_zw != c->D || c->rng.n_zv != c->D * (int64_t)c->K))
    throw Error(MFM_ERR_RUNTIME, "mfm_sweep_wV(z = NULL) needs an acquired device random set with D + K*D variates");
  ensure_e(c);
  const bool load_slots = c->e_in_slots;  // the residual is already in the launch's slot order: read it there
  c->slot_sums_valid = false;
  hipStream_t s = c->stream;
  // the four hyper-parameter vectors in ONE copy: [lambda_w | mu_w | lambda_V | mu_V]
  const size_t nG = (size_t)c->G, nGK = (size_t)c->G * c->K;
  if (c->wv_pack.n < 2 * nG + 2 * nGK) c->wv_pack.alloc(2 * nG + 2 * nGK);
  c->hs_stage.resize(2 * nG + 2 * nGK);
  std::memcpy(c->hs_stage.data(), lambda_w, nG * sizeof(double));
  std::memcpy(c->hs_stage.data() + nG, mu_w, nG * sizeof(double));
  std::memcpy(c->hs_stage.data() + 2 * nG, lambda_V, nGK * sizeof(double));
  std::memcpy(c->hs_stage.data() + 2 * nG + nGK, mu_V, nGK * sizeof(double));
  c->ring.upload(c->wv_pack.p, c->hs_stage.data(), c->hs_stage.size() * sizeof(double), s);
  const double *d_lam_w = c->wv_pack.p, *d_mu_w = c->wv_pack.p + nG, *d_lam = c->wv_pack.p + 2 * nG,
               *d_mu = c->wv_pack.p + 2 * nG + nGK;
  const double *zwdev, *zbase;
  if (zw) {
    if (!c->zw_host.p) c->zw_host.alloc((size_t)c->D);
    c->ring.upload(c->zw_host.p, zw, (size_t)c->D * sizeof(double), s);
    c->ring.upload(c->z.p, zv, (size_t)c->D * (f_end - f_begin) * sizeof(double), s);
    zwdev = c->zw_host.p;
    zbase = c->z.p;
  } else {
    zwdev = c->rng.slot[c->rng.current].zw.p;
    zbase = c->rng.slot[c->rng.current].zv.p + (size_t)f_begin * c->D;
  }
  const bool lazy_store = !std::getenv("MFM_RES_EAGER_STORE");
  // (regression, all factors swept: update_e follows and recomputes the residual -- the launch's copy would be a dead store, 64 us)
  const bool no_store =
      c->e_recomputable && c->e_is_residual && lazy_store && f_begin == 0 && f_end == c->K && !std::getenv("MFM_RES_ALWAYS_STORE");
  run_sweep_resident(s, c->timing, c->res, KC_SWEEP_V_RESIDENT, c->eq_raw(), c->V.p, c->D, f_begin, f_end, zbase, d_lam, d_mu,
                     c->group.p, c->G, alpha, c->ls.error.p, lazy_store, c->w.p, zwdev, d_lam_w, d_mu_w, e_shift, load_slots, no_store);
  c->e_in_slots = lazy_store && !no_store;
  c->e_lost = no_store;
  c->q_stale_factor = f_end - 1;
  if (c->comm.active() && !c->res.peers_model) sync_model_sharded(c, true, f_begin, f_end);
  MFM_CATCH(ctx)
}

// ---- a whole regression iteration without the host in its loop (include/myfm_hip.h) -------------------------------------------
static bool regression_iteration_ready(mfm_ctx *c) {
  if (!c || !c->finalized || !c->res.ready || c->comm.active() || c->K <= 0 || c->D <= 0) return false;
  if (std::getenv("MFM_RES_NO_LINEAR") || std::getenv("MFM_RES_EAGER_STORE")) return false;
  static const bool no_res_score = std::getenv("MFM_NO_RES_SCORE") != nullptr || std::getenv("MFM_NO_MF_SCORE") != nullptr;
  if (no_res_score || !res_score_supported(c->res, c->K)) return false;
  if (!(c->e_in_slots && c->slot_sums_valid && c->e_is_residual && !c->e_lost)) return false;  // (update_e's slot-order sums)
  const auto &r = c->rng;
  if (!r.programmed || r.produced <= r.acquired || r.n_zw != c->D || r.n_zv != c->D * (int64_t)c->K) return false;
  return true;
}

static void rng_finish_pending(mfm_ctx *ctx, bool gated);  // (with mfm_rng_prefetch, below)

int mfm_regression_iteration_ready(mfm_ctx *ctx) { return regression_iteration_ready(ctx) ? 1 : 0; }

int mfm_regression_iteration(mfm_ctx *ctx, const mfm_hyper_prior *prior, const double *n_in_group, double *alpha, double *w0,
                             double *lambda_w, double *mu_w, double *lambda_V, double *mu_V) {
  MFM_TRY(ctx)
  ctx->need_final();
  mfm_ctx *c = ctx;
  if (!regression_iteration_ready(c)) throw Error(MFM_ERR_RUNTIME, "mfm_regression_iteration: the context is not ready for it");
  auto &r = c->rng;
  const int G = c->G, K = c->K;
  const size_t nG = (size_t)G, nGK = (size_t)G * K, n_hyp = 4 + 2 * nG + 2 * nGK;
  if (r.n_hv != (int64_t)(1 + (prior->fit_w0 ? 1 : 0) + 2 * nG + 2 * nGK))
    throw Error(MFM_ERR_INVALID, "mfm_regression_iteration: the draw program does not have this iteration's hyper variates");
  hipStream_t s = c->stream;
  if (c->hyp.n < n_hyp) c->hyp.alloc(n_hyp);
  if (c->hyp_in.n < 1 + nG + nGK) c->hyp_in.alloc(1 + nG + nGK);
  if (!c->hyp_ev) MFM_HIP_CHECK(hipEventCreateWithFlags(&c->hyp_ev, hipEventDisableTiming));
  if (c->hyp_ng.n < nG || c->hyp_ng_host.size() != nG || std::memcmp(c->hyp_ng_host.data(), n_in_group, nG * sizeof(double)) != 0) {
    c->hyp_ng.alloc(nG);
    c->hyp_ng_host.assign(n_in_group, n_in_group + nG);
    MFM_HIP_CHECK(hipMemcpyAsync(c->hyp_ng.p, c->hyp_ng_host.data(), nG * sizeof(double), hipMemcpyHostToDevice, s));
    MFM_HIP_CHECK(hipStreamSynchronize(s));
  }
  // 1. this iteration's random set: the main stream waits for it (no host wait: it was finished an iteration ago)
  if (r.current >= 0) {
    auto &prev = r.slot[r.current];
    MFM_HIP_CHECK(hipEventRecord(prev.free_ev, s));
    prev.free_valid = true;
  }
  if (r.pending_slot == (int)(r.acquired % mfm_ctx::RngEngine::N_SLOTS)) rng_finish_pending(ctx, false);
  auto &sl = r.slot[r.acquired % mfm_ctx::RngEngine::N_SLOTS];
  MFM_HIP_CHECK(hipStreamWaitEvent(s, sl.ready, 0));
  r.current = (int)(r.acquired % mfm_ctx::RngEngine::N_SLOTS);
  r.acquired++;
  // 2. what the conditionals start from: [w0 | mu_w | mu_V]
  c->hs_stage.resize(1 + nG + nGK);
  c->hs_stage[0] = *w0;
  std::memcpy(c->hs_stage.data() + 1, mu_w, nG * sizeof(double));
  std::memcpy(c->hs_stage.data() + 1 + nG, mu_V, nGK * sizeof(double));
  c->ring.upload(c->hyp_in.p, c->hs_stage.data(), c->hs_stage.size() * sizeof(double), s);
  // 3. the reductions (mfm_hyper_stats without its read-back)
  const int n_ch = std::max(1, c->gs_chunks);
  const size_t n_out = 1 + nG * (K + 1);
  if (c->hs_out.n < n_out) c->hs_out.alloc(n_out);
  if (c->gs_partial.n < nG * (K + 1) * n_ch) c->gs_partial.alloc(nG * (K + 1) * n_ch);
  {
    TimedLaunch t(c->timing, s, KC_GROUP_STATS, 12.0 * c->D * (K + 1));
    hipLaunchKernelGGL(k_group_stats, dim3(G, 1, n_ch), dim3(WG), 0, s, c->w.p, c->D, c->feat_sorted.p, c->group_ptr.p,
                       c->hyp_in.p + 1, G, c->gs_partial.p);
    hipLaunchKernelGGL(k_group_stats, dim3(G, K, n_ch), dim3(WG), 0, s, c->V.p, c->D, c->feat_sorted.p, c->group_ptr.p,
                       c->hyp_in.p + 1 + nG, G, c->gs_partial.p + nG * n_ch);
    hipLaunchKernelGGL(k_group_stats_final, dim3(cdiv(G * (K + 1), 64)), dim3(64), 0, s, c->gs_partial.p, G * (K + 1), n_ch,
                       c->hs_out.p + 1);
  }
  {
    TimedLaunch t(c->timing, s, KC_REDUCE_E, 16.0 * c->res.G);
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(WG), 0, s, c->res.sums.p, c->res.G, c->hs_out.p);
  }
  // 4. the conditionals
  HyperPrior P;
  P.alpha_0 = prior->alpha_0;
  P.beta_0 = prior->beta_0;
  P.gamma_0 = prior->gamma_0;
  P.mu_0 = prior->mu_0;
  P.reg_0 = prior->reg_0;
  P.n_total = prior->n_total;
  P.fit_w0 = prior->fit_w0 ? 1 : 0;
  P.G = G;
  P.K = K;
  P.pad = 0;
  hipLaunchKernelGGL(k_hyper_regression, dim3(1), dim3(256), 0, s, P, c->hs_out.p, sl.hv.p, c->hyp_ng.p, c->hyp_in.p, c->hyp.p);
  // 5. their read-back (behind it on the stream: the sweeps -- the host has the draws while those run)
  double2 *h = c->readback((n_hyp + 3) / 2 + 2);
  double *hd = (double *)h;
  MFM_HIP_CHECK(hipMemcpyAsync(hd, c->hyp.p, n_hyp * sizeof(double), hipMemcpyDeviceToHost, s));
  MFM_HIP_CHECK(hipMemcpyAsync(hd + n_hyp, c->ls.error.p, sizeof(int), hipMemcpyDeviceToHost, s));
  MFM_HIP_CHECK(hipEventRecord(c->hyp_ev, s));
  // 6. update_w0's shift + update_w + update_V: the persistent launch with alpha / e_shift read from c->hyp
  {
    const bool load_slots = c->e_in_slots;
    c->slot_sums_valid = false;
    const double *d_lam_w = c->hyp.p + 4, *d_mu_w = d_lam_w + nG, *d_lam = d_mu_w + nG, *d_mu = d_lam + nGK;
    const bool no_store = c->e_recomputable && c->e_is_residual && !std::getenv("MFM_RES_ALWAYS_STORE");
    run_sweep_resident(s, c->timing, c->res, KC_SWEEP_V_RESIDENT, c->eq_raw(), c->V.p, c->D, 0, K, sl.zv.p, d_lam, d_mu, c->group.p,
                       c->G, 0.0, c->ls.error.p, true, c->w.p, sl.zw.p, d_lam_w, d_mu_w, 0.0, load_slots, no_store, c->hyp.p);
    c->e_in_slots = !no_store;
    c->e_lost = no_store;
    c->q_stale_factor = K - 1;
  }
  // 7. the set after the next (gated behind the launch), 8. update_e with the intercept the device has drawn
  {
    const int rc = mfm_rng_prefetch(ctx);
    if (rc != MFM_OK) return rc;
  }
  c->w0_dev = c->hyp.p + 1;
  try {
    score_train(c, true);
  } catch (...) {
    c->w0_dev = nullptr;
    throw;
  }
  c->w0_dev = nullptr;
  // 9. the draws
  MFM_HIP_CHECK(hipEventSynchronize(c->hyp_ev));
  c->check_coresident(*(const int *)(hd + n_hyp));
  {
    RngState hdr;
    std::memcpy(&hdr, sl.h_hv + r.n_hv, 3 * sizeof(double));
    if (hdr.error) throw Error(MFM_ERR_RUNTIME, "device random stream underflow (generated range exhausted)");
  }
  *alpha = hd[0];
  *w0 = hd[1];
  c->w0 = hd[1];
  std::memcpy(lambda_w, hd + 4, nG * sizeof(double));
  std::memcpy(mu_w, hd + 4 + nG, nG * sizeof(double));
  std::memcpy(lambda_V, hd + 4 + 2 * nG, nGK * sizeof(double));
  std::memcpy(mu_V, hd + 4 + 2 * nG + nGK, nGK * sizeof(double));
  MFM_CATCH(ctx)
}

int mfm_sweep_V(mfm_ctx *ctx, int32_t f_begin, int32_t f_end, double alpha, const double *lambda_V, const double *mu_V,
                const double *z) {
  MFM_TRY(ctx)
  ctx->need_final();
  if (!ctx->cell.ready) materialize_e(ctx);
  mfm_ctx *c = ctx;
  if (f_begin < 0 || f_end > c->K || f_begin > f_end) throw Error(MFM_ERR_INVALID, "factor range out of bounds");
  if (f_begin == f_end) return MFM_OK;
  hipStream_t s = c->stream;
  c->ring.upload(c->lam.p, lambda_V, (size_t)c->G * c->K * sizeof(double), s);
  c->ring.upload(c->mu.p, mu_V, (size_t)c->G * c->K * sizeof(double), s);
  const double *zbase = c->z.p;
  if (z) {
    c->ring.upload(c->z.p, z, (size_t)c->D * (f_end - f_begin) * sizeof(double), s);
  } else {
    if (c->rng.current < 0 || c->rng.n_zv != c->D * (int64_t)c->K)
      throw Error(MFM_ERR_RUNTIME, "mfm_sweep_V(z = NULL) needs an acquired device random set with K*D z_V variates");
    zbase = c->rng.slot[c->rng.current].zv.p + (size_t)f_begin * c->D;
  }
  if (c->main_lazy) {
    if (c->res.ready) {
      const bool lazy_store = !std::getenv("MFM_RES_EAGER_STORE");
      run_sweep_resident(s, c->timing, c->res, KC_SWEEP_V_RESIDENT, c->eq_raw(), c->V.p, c->D, f_begin, f_end, zbase, c->lam.p, c->mu.p,
                         c->group.p, c->G, alpha, c->ls.error.p, lazy_store);
      c->e_in_slots = lazy_store;
      c->q_stale_factor = f_end - 1;
      return MFM_OK;
    }
    c->ensure_main_plans();
  }
  // the first level rebuilds q from the CSR rows itself when it touches every row exactly once
  const bool first_q = !c->comm.active() && c->blocks.empty() && plan_first_level_builds_q(c->plan_V) &&
                       !std::getenv("MFM_NO_FUSED_QBUILD");
  if (c->qfree) {
    // compact e for the duration of the factor loop; q is never materialised inside it
    hipLaunchKernelGGL(k_e_pack, dim3(cdiv(c->N, 256)), dim3(256), 0, s, c->eq_rows(), c->ec.p, c->N);
    const SweepClasses kcv{KC_SWEEP_V_LIGHT, KC_SWEEP_V_HEAVY, KC_SWEEP_V_COOP, KC_SWEEP_V_LSTATS, KC_SWEEP_V_LDRAW,
                           KC_SWEEP_V_LAPPLY, KC_SWEEP_V_CHAIN, KC_SWEEP_V_SCAT};
    for (int f = f_begin; f < f_end; f++) {
      double *Vf = c->V.p + (size_t)f * c->D;
      SweepArgs a = main_args(c, Vf, zbase + (size_t)(f - f_begin) * c->D, c->lam.p + (size_t)f * c->G,
                              c->mu.p + (size_t)f * c->G, alpha);
      a.state = c->ec.p;
      a.r_rowptr = c->X.rowptr.p;
      a.r_colidx = c->X.colidx.p;
      a.r_val = c->X.rval.p;
      a.r_ell = (int)c->X.ell_width;
      run_plan_qfree(s, c->timing, c->plan_V, a, c->ls, kcv, c->X.unit);
    }
    hipLaunchKernelGGL(k_e_unpack, dim3(cdiv(c->N, 256)), dim3(256), 0, s, c->eq_rows(), c->ec.p, c->N);
    c->q_stale_factor = f_end - 1;  // q_train as the reference leaves it (FMTrainer.hpp:373): rebuilt when asked for
    MFM_HIP_CHECK(hipGetLastError());
    return MFM_OK;
  }
  if (c->sharded_fused && c->res.ready) {  // (row-sharded persistent sweep: the peers' buffers are set)
    const bool lazy_store = !std::getenv("MFM_RES_EAGER_STORE");
    run_sweep_resident(s, c->timing, c->res, KC_SWEEP_V_RESIDENT, c->eq_raw(), c->V.p, c->D, f_begin, f_end, zbase, c->lam.p, c->mu.p,
                       c->group.p, c->G, alpha, c->ls.error.p, lazy_store);
    c->e_in_slots = lazy_store;
    c->q_stale_factor = f_end - 1;
    if (!c->res.peers_model) sync_model_sharded(c, false, f_begin, f_end);
    return MFM_OK;
  }
  if (c->sharded_fused) {
    // (no pack / unpack passes: the first level reads e from the interleaved array, the final apply pass writes it back)
    const SweepClasses kcv{KC_SWEEP_V_LIGHT, KC_SWEEP_V_HEAVY, KC_SWEEP_V_COOP, KC_SWEEP_V_LSTATS, KC_SWEEP_V_LDRAW,
                           KC_SWEEP_V_LAPPLY, KC_SWEEP_V_CHAIN, KC_SWEEP_V_SCAT};
    auto args = [&](int f) {
      SweepArgs a = main_args(c, c->V.p + (size_t)f * c->D, zbase + (size_t)(f - f_begin) * c->D,
                              c->lam.p + (size_t)f * c->G, c->mu.p + (size_t)f * c->G, alpha);
      a.state = c->ec.p;
      a.state2 = c->qc.p;
      a.aos = c->eq_rows();
      a.r_rowptr = c->X.rowptr.p;
      a.r_colidx = c->X.colidx.p;
      a.r_val = c->X.rval.p;
      a.r_ell = (int)c->X.ell_width;
      return a;
    };
    if (c->mf) {
      if (c->X.unit)
        run_sweep_mf<true>(s, c->timing, c->plan_V, args, f_begin, f_end, c->ls, kcv, &c->comm);
      else
        run_sweep_mf<false>(s, c->timing, c->plan_V, args, f_begin, f_end, c->ls, kcv, &c->comm);
    } else if (c->plan_V.steps.size() > 2) {
      if (c->X.unit)
        run_sweep_soa_multi<true>(s, c->timing, c->plan_V, args, f_begin, f_end, c->ls, kcv, &c->comm);
      else
        run_sweep_soa_multi<false>(s, c->timing, c->plan_V, args, f_begin, f_end, c->ls, kcv, &c->comm);
    } else if (c->X.unit) {
      run_sweep_soa_sharded<true>(s, c->timing, c->plan_V, args, f_begin, f_end, c->ls, kcv, c->comm);
    } else {
      run_sweep_soa_sharded<false>(s, c->timing, c->plan_V, args, f_begin, f_end, c->ls, kcv, c->comm);
    }
    // make the first-level coefficients identical on every rank again (each was drawn where its rows live)
    {
      const int64_t n = (int64_t)(f_end - f_begin) * c->D;
      double *Vb = c->V.p + (size_t)f_begin * c->D;
      hipLaunchKernelGGL(k_mask_rows, dim3(cdiv(n, 256)), dim3(256), 0, s, Vb, c->sync_mask.p, c->D, n);
      c->comm.allreduce(Vb, n);
    }
    c->q_stale_factor = f_end - 1;  // q_train as the reference leaves it (FMTrainer.hpp:373): rebuilt when asked for
    MFM_HIP_CHECK(hipGetLastError());
    return MFM_OK;
  }
  if (c->soa) {
    // (no pack pass: the first level reads e from the interleaved array; no unpack pass when the last level runs on
    // row tiles: its final apply pass writes e back there)
    const SweepClasses kcv{KC_SWEEP_V_LIGHT, KC_SWEEP_V_HEAVY, KC_SWEEP_V_COOP, KC_SWEEP_V_LSTATS, KC_SWEEP_V_LDRAW,
                           KC_SWEEP_V_LAPPLY, KC_SWEEP_V_CHAIN, KC_SWEEP_V_SCAT};
    auto args = [&](int f) {
      SweepArgs a = main_args(c, c->V.p + (size_t)f * c->D, zbase + (size_t)(f - f_begin) * c->D,
                              c->lam.p + (size_t)f * c->G, c->mu.p + (size_t)f * c->G, alpha);
      a.state = c->ec.p;
      a.state2 = c->qc.p;
      a.aos = c->eq_rows();
      a.r_rowptr = c->X.rowptr.p;
      a.r_colidx = c->X.colidx.p;
      a.r_val = c->X.rval.p;
      a.r_ell = (int)c->X.ell_width;
      return a;
    };
    const bool fuse = c->fuse_next;
    if (c->res.ready) {
      const bool lazy_store = !std::getenv("MFM_RES_EAGER_STORE");
      run_sweep_resident(s, c->timing, c->res, KC_SWEEP_V_RESIDENT, c->eq_raw(), c->V.p, c->D, f_begin, f_end, zbase, c->lam.p, c->mu.p,
                         c->group.p, c->G, alpha, c->ls.error.p, lazy_store);
      c->e_in_slots = lazy_store;
      c->q_stale_factor = f_end - 1;
      return MFM_OK;
    }
    if (c->mf) {
      if (c->X.unit)
        run_sweep_mf<true>(s, c->timing, c->plan_V, args, f_begin, f_end, c->ls, kcv);
      else
        run_sweep_mf<false>(s, c->timing, c->plan_V, args, f_begin, f_end, c->ls, kcv);
    } else if (c->X.unit)
      run_sweep_soa<true>(s, c->timing, c->plan_V, args, f_begin, f_end, c->ls, kcv, fuse);
    else
      run_sweep_soa<false>(s, c->timing, c->plan_V, args, f_begin, f_end, c->ls, kcv, fuse);
    if (!c->plan_V.steps.back().par.tiled)
      hipLaunchKernelGGL(k_e_unpack, dim3(cdiv(c->N, 256)), dim3(256), 0, s, c->eq_rows(), c->ec.p, c->N);
    c->q_stale_factor = f_end - 1;  // q_train as the reference leaves it (FMTrainer.hpp:373): rebuilt when asked for
    MFM_HIP_CHECK(hipGetLastError());
    return MFM_OK;
  }
  if (c->cell.ready) {
    run_sweep_cell(c, f_begin, f_end, zbase, alpha);
    return MFM_OK;
  }
  DevBlock *carry = nullptr;  // last block of the previous factor, re-sync still owed
  for (int f = f_begin; f < f_end; f++) {
    double *Vf = c->V.p + (size_t)f * c->D;
    const double *zf = zbase + (size_t)(f - f_begin) * c->D;
    const double *lamf = c->lam.p + (size_t)f * c->G;
    const double *muf = c->mu.p + (size_t)f * c->G;
    if (carry) {  // the last block of the previous factor owes its re-sync: keep its (q_B, q_S) for the q-cache build below
      if (carry->q_saved.n < (size_t)carry->B) carry->q_saved.alloc((size_t)carry->B);
      hipLaunchKernelGGL(k_save_q, dim3(cdiv(carry->B, 256)), dim3(256), 0, s, carry->rec.p, carry->B, carry->q_saved.p);
    }
    for (auto &B : c->blocks) block_rowcache(s, c->timing, *B, Vf + B->col_off, true);  // :331-333, :388-393
    if (!first_q) launch_qbuild(c, Vf, carry);                                          // :320, :334-337
    carry = nullptr;
    SweepArgs a = main_args(c, Vf, zf, lamf, muf, alpha);
    if (first_q) {
      a.r_rowptr = c->X.rowptr.p;
      a.r_colidx = c->X.colidx.p;
      a.r_val = c->X.rval.p;
      a.r_ell = (int)c->X.ell_width;
    }
    const SweepClasses kcv{KC_SWEEP_V_LIGHT, KC_SWEEP_V_HEAVY, KC_SWEEP_V_COOP, KC_SWEEP_V_LSTATS, KC_SWEEP_V_LDRAW,
                           KC_SWEEP_V_LAPPLY, KC_SWEEP_V_CHAIN, KC_SWEEP_V_SCAT};
    if (c->comm.active())
      run_plan_sharded<PMainV>(s, c->timing, c->plan_V, a, c->ls, kcv, c->X.unit, c->comm);
    else
      run_plan<PMainV>(s, c->timing, c->plan_V, a, c->ls, kcv, c->X.unit, first_q);  // :343-376
    // a block whose successor streams its statistics pass leaves its re-sync to that pass (one read + write of eq less)
    static const bool fuse_resync = !std::getenv("MFM_NO_RESYNC_FUSE");
    DevBlock *pending = nullptr;
    for (size_t bi = 0; bi < c->blocks.size(); bi++) {
      DevBlock &B = *c->blocks[bi];
      const bool last = bi + 1 == c->blocks.size();
      // ... and the last block leaves it to the next factor's q-cache build (thread-per-row form)
      const bool defer = fuse_resync && c->N > 0 &&
                         (last ? (f + 1 < f_end && !first_q && qbuild_by_rows(c))
                               : (c->blocks[bi + 1]->stream_unsync || c->blocks[bi + 1]->split_unsync));
      block_sweep_V(s, c->timing, c->ls, B, c->N, c->eq_rows(), Vf, zf, c->group.p, lamf, muf, alpha, c->comm, pending, defer);  // :378-482
      pending = defer ? &B : nullptr;
    }
    carry = pending;
  }
  MFM_CATCH(ctx)
}

int mfm_update_e_regression(mfm_ctx *ctx) {
  MFM_TRY(ctx)
  ctx->need_final();
  score_train(ctx, true);
  MFM_CATCH(ctx)
}

int mfm_score_train(mfm_ctx *ctx) {
  MFM_TRY(ctx)
  ctx->need_final();
  score_train(ctx, false);
  MFM_CATCH(ctx)
}

// ---- device random stream ---------------------------------------------------------------------------
int mfm_rng_seed_mt19937(mfm_ctx *ctx, const uint32_t *state624, int32_t position) {
  MFM_TRY(ctx)
  auto &r = ctx->rng;
  if (position < 0 || position > MT_N) throw Error(MFM_ERR_INVALID, "mt19937 position out of range");
  if (!r.stream) {
    // the generator runs one iteration ahead next to the sweeps; at the highest priority its workgroups take the CUs the
    // short kernels at the start of an iteration free, so that it is done before the persistent latent sweep (which needs
    // every CU at once: a generator workgroup still running then delays all of its workgroups) is launched
    int lo = 0, hi = 0;
    MFM_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    MFM_HIP_CHECK(hipStreamCreateWithPriority(&r.stream, hipStreamNonBlocking, std::getenv("MFM_RNG_NO_PRIORITY") ? lo : hi));
  }
  MFM_HIP_CHECK(hipStreamSynchronize(r.stream));
  RngState h;
  std::memset(&h, 0, sizeof(h));
  h.mt_pos = position;
  std::memcpy(h.mt, state624, sizeof(uint32_t) * MT_N);
  r.state.alloc(1);
  MFM_HIP_CHECK(hipMemcpy(r.state.p, &h, sizeof(h), hipMemcpyHostToDevice));
  r.seeded = true;
  r.produced = r.acquired = 0;
  r.current = -1;
  r.pending_slot = -1;
  for (auto &sl : r.slot) sl.free_valid = false;
  MFM_CATCH(ctx)
}

int mfm_rng_set_program(mfm_ctx *ctx, const mfm_rng_op *ops, int32_t n_ops) {
  MFM_TRY(ctx)
  auto &r = ctx->rng;
  if (!r.seeded) throw Error(MFM_ERR_RUNTIME, "mfm_rng_seed_mt19937 has not been called");
  if (r.produced != r.acquired) throw Error(MFM_ERR_RUNTIME, "cannot change the draw program while a set is in flight");
  MFM_HIP_CHECK(hipStreamSynchronize(r.stream));
  std::vector<RngOp> h((size_t)n_ops);
  int64_t n_dest[3] = {0, 0, 0};
  double normals = 0, gammas = 0;
  int n_normal_ops = 0;
  int64_t latent_rows = 0;
  int n_kept = 0;
  for (int k = 0; k < n_ops; k++) {
    const mfm_rng_op &o = ops[k];
    if (o.kind == MFM_RNG_LATENT) {  // (not a draw of the set: only sizes the generator, below)
      if (o.count < 0) throw Error(MFM_ERR_INVALID, "bad rng op");
      latent_rows += o.count;
      continue;
    }
    const int i = n_kept++;
    if (o.kind != MFM_RNG_NORMALS && o.kind != MFM_RNG_GAMMA) throw Error(MFM_ERR_INVALID, "bad rng op kind");
    if (o.dest < 0 || o.dest > 2 || o.count < 0 || o.offset < 0) throw Error(MFM_ERR_INVALID, "bad rng op");
    if (o.kind == MFM_RNG_GAMMA && !(o.shape > 0)) throw Error(MFM_ERR_INVALID, "gamma shape must be positive");
    h[i].kind = o.kind;
    h[i].dest = o.dest;
    h[i].count = o.kind == MFM_RNG_GAMMA ? 1 : o.count;
    h[i].offset = o.offset;
    h[i].shape = o.shape;
    n_dest[o.dest] = std::max(n_dest[o.dest], o.offset + h[i].count);
    if (o.kind == MFM_RNG_GAMMA)
      gammas += 1;
    else {
      normals += (double)o.count;
      n_normal_ops++;
    }
  }
  h.resize((size_t)n_kept);
  n_ops = n_kept;
  r.n_hv = n_dest[0];
  r.n_zw = n_dest[1];
  r.n_zv = n_dest[2];
  r.ops.upload(h);
  r.h_ops = h;
  r.n_ops = n_ops;
  {
    int64_t amax = 0;
    for (auto &o : h)
      if (o.kind == MFM_RNG_NORMALS && o.count > 16384) amax = std::max(amax, mfm_ctx::RngEngine::attempts_for(o.count));
    r.cand.alloc((size_t)std::max<int64_t>(amax, 1));
    r.masks.alloc((size_t)std::max<int64_t>(amax / 64, 1));
    r.counts.alloc((size_t)std::max<int64_t>(amax / NORM_CHUNK, 1));
    r.nscratch.alloc(1);
  }
  // engine outputs one iteration can consume: 4 per polar attempt, 4/pi attempts per normal on average
  // (+2 % and a 6-sigma allowance), a batch of look-ahead per NORMALS op, 4096 per gamma draw.
  const double per_normal = 4.0 * 4.0 / 3.14159265358979;
  r.need = (uint64_t)(normals * per_normal * 1.02 + 6.0 * 2.4 * std::sqrt(normals + 1.0) + gammas * 4096.0 +
                      4.0 * RNG_ATT * RNG_CONSUME_THREADS * (n_normal_ops + 1) + 65536.0);
  // the exact latent draws: ~1.3-1.5 quads of 4 outputs per row (what is missing at run time is generated then)
  if (latent_rows > 0) r.need += (uint64_t)(4.0 * (1.6 * (double)latent_rows + 6.0 * std::sqrt((double)latent_rows) + 1024.0));
  for (auto &o : h)  // the evaluation window of a big op reaches past its last accept
    if (o.kind == MFM_RNG_NORMALS && o.count > 16384)
      r.need += (uint64_t)(4 * (mfm_ctx::RngEngine::attempts_for(o.count) - (int64_t)((double)o.count * 1.2732)));
  // more than one workgroup's worth of blocks per iteration: generate in parallel with jump-ahead
  r.par_wgs = 1;
  r.need_gen = r.need;
  {
    const int64_t blocks1 = (int64_t)(r.need / MT_N) + 2;
    if (blocks1 > MT_PAR_BLOCKS && !std::getenv("MFM_RNG_SERIAL")) {
      r.need_gen = r.need * (uint64_t)(latent_rows > (1 << 22) ? 2 : mt_gen_batch((double)r.need));
      const int64_t blocks = (int64_t)(r.need_gen / MT_N) + 2;
      if (r.par_blocks <= 0) r.par_blocks = mt_par_blocks_for(blocks);
      const int wgs = (int)((blocks + r.par_blocks - 1) / r.par_blocks);
      std::vector<uint32_t> tab;
      if (mtjump::JumpCache::inst().get(r.par_blocks, wgs - 1, tab)) {
        r.jump.upload(tab);
        r.state_next.alloc(1);
        r.par_wgs = wgs;
        r.starts.alloc((size_t)wgs * MT_N);
        MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_mt_generate_par<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)((MT_JUMP_SPAN * MT_N + 2 * (MT_N + 1)) * sizeof(uint32_t))));
        MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_mt_generate_par<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)((MT_JUMP_SPAN * MT_N + 2 * (MT_N + 1)) * sizeof(uint32_t))));
      }
    }
  }
  uint64_t cap = 1;
  while (cap < r.need_gen + r.need + 2 * MT_N) cap <<= 1;
  r.raw.alloc((size_t)cap);
  r.mask = cap - 1;
  for (auto &sl : r.slot) {
    sl.hv.alloc((size_t)std::max<int64_t>(r.n_hv, 1));
    sl.zw.alloc((size_t)std::max<int64_t>(r.n_zw, 1));
    sl.zv.alloc((size_t)std::max<int64_t>(r.n_zv, 1));
    if (sl.h_hv) MFM_HIP_CHECK(hipHostFree(sl.h_hv));
    sl.h_hv = nullptr;
    MFM_HIP_CHECK(hipHostMalloc((void **)&sl.h_hv, ((size_t)r.n_hv + 4) * sizeof(double), hipHostMallocDefault));
    if (!sl.ready) MFM_HIP_CHECK(hipEventCreateWithFlags(&sl.ready, hipEventDisableTiming));
    if (!sl.free_ev) MFM_HIP_CHECK(hipEventCreateWithFlags(&sl.free_ev, hipEventDisableTiming));
    sl.free_valid = false;
  }
  r.programmed = true;
  MFM_CATCH(ctx)
}

// the ops [first, n_ops) of a set on the side stream, then the set's read-back and its ready event. stop_at_wide: stop in front of the
// first whole-GPU evaluation and return its index (the caller finishes the set later); otherwise n_ops.
static int rng_enqueue_ops(mfm_ctx *ctx, mfm_ctx::RngEngine::Slot &sl, int first, bool stop_at_wide) {
  auto &r = ctx->rng;
  hipStream_t s = r.stream;
  // small ops (hyper draws) run in one sequential workgroup; big NORMALS ops on the whole GPU
  int i = first;
  while (i < r.n_ops) {
    int j = i;
    while (j < r.n_ops && !(r.h_ops[j].kind == MFM_RNG_NORMALS && r.h_ops[j].count > 16384)) j++;
    if (j > i)
      hipLaunchKernelGGL(k_rng_consume, dim3(1), dim3(RNG_CONSUME_THREADS), 0, s, r.state.p, r.raw.p, r.mask, r.ops.p, i, j,
                         sl.hv.p, sl.zw.p, sl.zv.p);
    if (j < r.n_ops) {
      const RngOp &o = r.h_ops[j];
      if (stop_at_wide && o.count > mfm_ctx::RngEngine::WIDE_COUNT) {
        MFM_HIP_CHECK(hipGetLastError());
        return j;
      }
      const int64_t A = mfm_ctx::RngEngine::attempts_for(o.count);
      const int n_chunks = (int)(A / NORM_CHUNK);
      double *dst = (o.dest == 0 ? sl.hv.p : (o.dest == 1 ? sl.zw.p : sl.zv.p)) + o.offset;
      hipLaunchKernelGGL(k_norm_eval, dim3(n_chunks), dim3(256), 0, s, r.state.p, r.raw.p, r.mask, r.cand.p, r.masks.p,
                         r.counts.p);
      hipLaunchKernelGGL(k_norm_scan, dim3(1), dim3(1024), 0, s, r.state.p, r.counts.p, n_chunks, o.count, r.nscratch.p);
      hipLaunchKernelGGL(k_norm_scatter, dim3(n_chunks), dim3(256), 0, s, r.state.p, r.cand.p, r.masks.p, r.counts.p, o.count,
                         r.nscratch.p, dst);
      j++;
    }
    i = j;
  }
  MFM_HIP_CHECK(hipGetLastError());
  if (r.n_hv)
    MFM_HIP_CHECK(hipMemcpyAsync(sl.h_hv, sl.hv.p, (size_t)r.n_hv * sizeof(double), hipMemcpyDeviceToHost, s));
  MFM_HIP_CHECK(hipMemcpyAsync(sl.h_hv + r.n_hv, r.state.p, 3 * sizeof(double), hipMemcpyDeviceToHost, s));
  MFM_HIP_CHECK(hipEventRecord(sl.ready, s));
  return r.n_ops;
}

// the rest of the set mfm_rng_prefetch left unfinished: its whole-GPU evaluation(s), behind everything enqueued on the main stream so
// far when `gated`
static void rng_finish_pending(mfm_ctx *ctx, bool gated) {
  auto &r = ctx->rng;
  if (r.pending_slot < 0) return;
  if (gated) {
    if (!r.gate) MFM_HIP_CHECK(hipEventCreateWithFlags(&r.gate, hipEventDisableTiming));
    MFM_HIP_CHECK(hipEventRecord(r.gate, ctx->stream));
    MFM_HIP_CHECK(hipStreamWaitEvent(r.stream, r.gate, 0));
  }
  const int slot = r.pending_slot;
  r.pending_slot = -1;
  rng_enqueue_ops(ctx, r.slot[slot], r.pending_op, false);
}

int mfm_rng_prefetch(mfm_ctx *ctx) {
  MFM_TRY(ctx)
  auto &r = ctx->rng;
  if (!r.programmed) throw Error(MFM_ERR_RUNTIME, "mfm_rng_set_program has not been called");
  // three slots: the acquired set stays valid until the next acquire, so at most two further sets may be in flight (three
  // before the first acquire). The trainer keeps two ahead: the set of iteration t + 2 is requested right after the persistent
  // sweep of iteration t is enqueued.
  //
  // Where that sweep fills the device, the order on the side stream matters: a kernel of more than one workgroup that is dispatched
  // while the sweep runs does not move until the sweep ends (per-dispatch timelines, profiles/r06_q_timeline_config3_*.txt), and
  // holds up the stream behind it. A set is a chain of narrow kernels (the generator, the hyper draws, the linear term's normals:
  // ~0.26 ms in a row) and one whole-GPU evaluation (the K D sweep normals: 27 us + scan + scatter on a free device). With one gate in
  // front of the whole set (rounds 4 and 5) the narrow chain filled the gap between two sweeps and the wide evaluation was
  // dispatched into the next sweep: stuck for its 2.6 ms, the sweep 2.4 % slower. So the set is produced in two parts: the narrow
  // chain of set t + 2 with this request, its wide evaluation only with the NEXT one, behind sweep t + 1 -- every gap then starts
  // with a wide evaluation at full width (done ~0.1 ms into a ~0.28 ms gap, in time for sweep t + 2), followed by the next set's
  // narrow chain, of which only the last single-workgroup kernel overlaps the next sweep's start.
  if (r.produced - r.acquired >= (r.current >= 0 ? mfm_ctx::RngEngine::N_SLOTS - 1 : mfm_ctx::RngEngine::N_SLOTS))
    throw Error(MFM_ERR_RUNTIME, "every random set is in use (one acquired, the others in flight): acquire the next one first");
  static const bool no_gate = std::getenv("MFM_RES_NO_GATE") != nullptr;  // (experiments with CUs left free by MFM_RES_CUS)
  static const bool one_part = std::getenv("MFM_RNG_ONE_PART") != nullptr;  // (the whole set behind one gate, as before round 6)
  // (the gate only where the persistent sweep fills the device: a small table's launch leaves most CUs free, and there the
  //  generator should run beside it -- ML-100k shape: 2700 it/s gated, 3000 not)
  const bool gated = ctx->res.ready && ctx->res_fills_device && !no_gate;
  rng_finish_pending(ctx, gated);
  auto &sl = r.slot[r.produced % mfm_ctx::RngEngine::N_SLOTS];
  hipStream_t s = r.stream;
  {  // (timing experiment, wrong results: after the first sets nothing is generated any more -- what the iteration costs without the
     //  random stream's work)
    static const bool reuse = std::getenv("MFM_RNG_DBG_REUSE") != nullptr;
    if (reuse && r.produced >= 2 * mfm_ctx::RngEngine::N_SLOTS) {
      MFM_HIP_CHECK(hipEventRecord(sl.ready, s));
      r.produced++;
      return MFM_OK;
    }
  }
  // two parts only with another finished set ahead of this one (the set is then not needed before the request after this one)
  const bool two_parts = gated && !one_part && r.produced - r.acquired >= 1;
  if (gated && !two_parts) {
    if (!r.gate) MFM_HIP_CHECK(hipEventCreateWithFlags(&r.gate, hipEventDisableTiming));
    MFM_HIP_CHECK(hipEventRecord(r.gate, ctx->stream));
    MFM_HIP_CHECK(hipStreamWaitEvent(s, r.gate, 0));
  }
  if (sl.free_valid) MFM_HIP_CHECK(hipStreamWaitEvent(s, sl.free_ev, 0));
  if (r.latent_pending) {  // the main stream consumed engine outputs (exact latent draws): this set starts where they ended
    if (!r.latent_ev) MFM_HIP_CHECK(hipEventCreateWithFlags(&r.latent_ev, hipEventDisableTiming));
    MFM_HIP_CHECK(hipEventRecord(r.latent_ev, ctx->stream));
    MFM_HIP_CHECK(hipStreamWaitEvent(s, r.latent_ev, 0));
    r.latent_pending = false;
  }
  if (r.par_wgs > 1) {
    if (std::getenv("MFM_RNG_FUSED_JUMP")) {
      hipLaunchKernelGGL(k_mt_generate_par<0>, dim3(r.par_wgs), dim3(MT_GEN_THREADS),
                         (MT_JUMP_SPAN * MT_N + 2 * (MT_N + 1)) * sizeof(uint32_t), s, r.state.p, r.state_next.p, r.raw.p, r.mask,
                         r.need_gen, r.jump.p, r.par_blocks, (uint32_t *)nullptr, r.need);
    } else {
      hipLaunchKernelGGL(k_mt_generate_par<1>, dim3(r.par_wgs), dim3(MT_GEN_THREADS),
                         (MT_JUMP_SPAN * MT_N + 2 * (MT_N + 1)) * sizeof(uint32_t), s, r.state.p, r.state_next.p, r.raw.p, r.mask,
                         r.need_gen, r.jump.p, r.par_blocks, r.starts.p, r.need);
      hipLaunchKernelGGL(k_mt_generate_par<2>, dim3(r.par_wgs), dim3(MT_GEN_THREADS), 2 * (MT_N + 1) * sizeof(uint32_t), s,
                         r.state.p, r.state_next.p, r.raw.p, r.mask, r.need_gen, r.jump.p, r.par_blocks, r.starts.p, r.need);
    }
    hipLaunchKernelGGL(k_mt_commit, dim3(1), dim3(MT_GEN_THREADS), 0, s, r.state.p, r.state_next.p);
  } else {
    hipLaunchKernelGGL(k_mt_generate, dim3(1), dim3(MT_GEN_THREADS), 0, s, r.state.p, r.raw.p, r.mask, r.need);
  }
  const int slot_idx = (int)(r.produced % mfm_ctx::RngEngine::N_SLOTS);
  const int next_op = rng_enqueue_ops(ctx, sl, 0, two_parts);
  if (next_op < r.n_ops) {
    r.pending_slot = slot_idx;
    r.pending_op = next_op;
  }
  r.produced++;
  MFM_CATCH(ctx)
}

int mfm_rng_acquire(mfm_ctx *ctx, double *hyper_variates, int64_t n_hyper_variates) {
  MFM_TRY(ctx)
  auto &r = ctx->rng;
  if (r.produced <= r.acquired) throw Error(MFM_ERR_RUNTIME, "no prefetched random set: call mfm_rng_prefetch first");
  if (n_hyper_variates != r.n_hv) throw Error(MFM_ERR_INVALID, "hyper variate count does not match the draw program");
  // everything that used the previously acquired set has been enqueued on the main stream by now
  if (r.current >= 0) {
    auto &prev = r.slot[r.current];
    MFM_HIP_CHECK(hipEventRecord(prev.free_ev, ctx->stream));
    prev.free_valid = true;
  }
  if (r.pending_slot == (int)(r.acquired % mfm_ctx::RngEngine::N_SLOTS)) rng_finish_pending(ctx, false);
  auto &sl = r.slot[r.acquired % mfm_ctx::RngEngine::N_SLOTS];
  MFM_HIP_CHECK(hipEventSynchronize(sl.ready));
  RngState hdr;
  std::memcpy(&hdr, sl.h_hv + r.n_hv, 3 * sizeof(double));
  if (hdr.error) throw Error(MFM_ERR_RUNTIME, "device random stream underflow (generated range exhausted)");
  if (r.n_hv) std::memcpy(hyper_variates, sl.h_hv, (size_t)r.n_hv * sizeof(double));
  r.current = (int)(r.acquired % mfm_ctx::RngEngine::N_SLOTS);
  r.acquired++;
  MFM_CATCH(ctx)
}

int mfm_rng_get_z(mfm_ctx *ctx, double *zw, double *zv) {
  MFM_TRY(ctx)
  auto &r = ctx->rng;
  if (r.current < 0) throw Error(MFM_ERR_RUNTIME, "no acquired random set");
  auto &sl = r.slot[r.current];
  if (zw && r.n_zw) MFM_HIP_CHECK(hipMemcpy(zw, sl.zw.p, (size_t)r.n_zw * sizeof(double), hipMemcpyDeviceToHost));
  if (zv && r.n_zv) MFM_HIP_CHECK(hipMemcpy(zv, sl.zv.p, (size_t)r.n_zv * sizeof(double), hipMemcpyDeviceToHost));
  MFM_CATCH(ctx)
}

// ---- timing ---------------------------------------------------------------------------------------
int mfm_timing_enable(mfm_ctx *ctx, int on) {
  MFM_TRY(ctx)
  if (!on) ctx->timing.resolve();
  ctx->timing.on = on != 0;
  MFM_CATCH(ctx)
}

int mfm_timing_select(mfm_ctx *ctx, int32_t kernel_class) {
  MFM_TRY(ctx)
  if (kernel_class >= KC_N) throw Error(MFM_ERR_INVALID, "kernel class index out of range");
  ctx->timing.only = kernel_class < 0 ? -1 : kernel_class;
  MFM_CATCH(ctx)
}
int mfm_timing_reset(mfm_ctx *ctx) {
  MFM_TRY(ctx)
  MFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  ctx->timing.reset();
  MFM_CATCH(ctx)
}
int mfm_timing_n_classes(void) { return KC_N; }
const char *mfm_timing_class_name(int cls) { return (cls >= 0 && cls < KC_N) ? kKernelClassNames[cls] : ""; }
int mfm_timing_get(mfm_ctx *ctx, int cls, double *ms_total, int64_t *launches, double *alg_bytes_total) {
  MFM_TRY(ctx)
  if (cls < 0 || cls >= KC_N) throw Error(MFM_ERR_INVALID, "bad kernel class");
  MFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  ctx->timing.resolve();
  *ms_total = ctx->timing.ms[cls];
  *launches = ctx->timing.launches[cls];
  *alg_bytes_total = ctx->timing.bytes[cls];
  MFM_CATCH(ctx)
}

int mfm_host_column_levels(int64_t n_rows, int64_t n_cols, const int64_t *indptr, const int32_t *indices, int32_t *level,
                           int32_t *n_levels) {
  try {
    HostCsr X;
    X.rows = n_rows;
    X.cols = n_cols;
    X.ptr.assign(indptr, indptr + n_rows + 1);
    X.idx.assign(indices, indices + indptr[n_rows]);
    X.val.assign((size_t)indptr[n_rows], 1.0);
    HostCsr Xt = transpose_host(X);
    std::vector<int32_t> lv;
    *n_levels = column_levels(Xt, lv);
    std::copy(lv.begin(), lv.end(), level);
    return MFM_OK;
  } catch (const std::exception &ex) {
    g_global_error = ex.what();
    return MFM_ERR_RUNTIME;
  }
}

}  // extern "C"

#include "mfm_tasks.hpp"    // classification / ordered-probit entry points
#include "mfm_latent_host.hpp"  // ... their exact latent draws on the device stream
#include "mfm_predict.hpp"  // mfm_design_* entry points
