// mfm_chain.hip -- translation unit of the streamed conflict-window chain of large relation blocks (k_cs_stream): plan upload,
// the rings the walker and the row ranges exchange through, the launch, and a C entry point that emulates the plan's data flow
// on the host (tests without a GPU).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "mfm_chain_api.hpp"
#include "mfm_chain_stream.hpp"

namespace mfm {

struct CsStream {
  CsStreamInfo info;
  DevBuf<int32_t> cols, cold_ptr, cold_rc, enter_ptr, enter_row, enter_slot, exit_ptr, exit_row, exit_slot, hot_ptr, hot_slot;
  DevBuf<double> cold_x, hot_x;
  DevBuf<double2> in_ring, out_ring, part, oldnew;
  DevBuf<CsSync> sync;
  mutable DevBuf<int32_t> col_group;             // group index of every chain column (gathered at the first launch)
  mutable const int32_t *col_group_of = nullptr;  // ... from this group array
};

static int env_int(const char *name, int dflt) {
  const char *e = std::getenv(name);
  return e ? std::atoi(e) : dflt;
}

std::shared_ptr<CsStream> cs_stream_build(const HostCsr &csc, const std::vector<int32_t> &run, CsStreamInfo *info) {
  const auto t0 = std::chrono::steady_clock::now();
  CsParams prm;
  prm.NB = std::max(1, std::min(CS_MAX_NB, env_int("MFM_CS_NB", 32)));
  prm.RD = std::max(1, std::min(CS_ER, env_int("MFM_CS_RD", 2)));  // (the walker keeps the hot entry lists of CS_ER steps)
  prm.cap = std::max(16, env_int("MFM_CS_CAP", 1 << 20));
  const int cg_forced = env_int("MFM_CS_CG", 0);
  const int lw_max = std::max(2, std::min(CS_MAX_LW, env_int("MFM_CS_LW", CS_MAX_LW)));
  // (a window of at least two steps: the U wavefront pairs take the steps in turn, and two cold touches of one row in consecutive steps
  //  -- possible only with a window of one -- would then race)
  const int lw_min = std::max(2, std::min(lw_max, env_int("MFM_CS_LW_MIN", 2)));
  // Steps of Cg columns, a window of Lw steps: the walker never waits when the hop walker -> ranges -> walker (~12 us, measured:
  // profiles/r05_*) fits Lw steps, and the hot rows (56 bytes each) + the hot entry lists of CS_ER steps must fit its LDS -- both
  // want Cg * Lw large, the LDS need grows with its square. Per step width the largest window that fits; of those the one with the
  // shortest predicted time per column, the wider step on a tie (fewer hand-overs).
  CsPlanHost P;
  {
    const CsTouches T = cs_touches(csc.ptr.data(), csc.idx.data(), csc.cols, run);
    const int n = (int)run.size();
    struct Cand {
      int cg = 0, lw = 0;
      double per_col = 1e30;
    };
    std::vector<Cand> cands((size_t)CS_MAX_CG + 1);
    auto fits = [&](int cg, int lw, CsNeed &N) {
      N = cs_estimate(T, cg, lw, prm.RD);
      return N.n_slots <= prm.cap && cs_lds_bytes(N.n_slots, cg, N.max_hot_col) <= CS_LDS_MAX && cs_ecap(N.max_hot_col) <= 1024;
    };
    auto search = [&](int cg) {  // the best window for this step width (the LDS need grows with the window: bisection)
      Cand &C = cands[(size_t)cg];
      CsNeed N;
      if (!fits(cg, lw_min, N)) return;
      int lo = lw_min, hi = lw_max;  // lo fits
      while (lo < hi) {
        const int mid = (lo + hi + 1) / 2;
        if (fits(cg, mid, N)) lo = mid; else hi = mid - 1;
      }
      // predicted time per column (calibrated on MI355X, profiles/r05_*): the walker needs 0.62 us + 1.3 ns per hot entry for a
      // column; it never waits when the hop walker -> ranges -> walker (~11 us) fits the window; + the step's hand-over costs
      for (int lw = lo; lw >= std::max(lw_min, lo - 3); lw--) {
        fits(cg, lw, N);
        const double walk = 0.62 + 0.0013 * (double)N.n_hot / std::max(n, 1);
        const double per_col = std::max(walk, (11.0 + walk * cg) / ((double)lw * cg)) + 0.1 / cg;
        if (per_col < C.per_col - 1e-9) C = Cand{cg, lw, per_col};
      }
    };
    std::vector<std::thread> pool;
    for (int cg = CS_MAX_CG; cg >= 1; cg--) {
      if (cg_forced > 0 ? cg != std::max(1, std::min(CS_MAX_CG, cg_forced)) : cg < 2) continue;
      pool.emplace_back(search, cg);
    }
    for (auto &t : pool) t.join();
    Cand best;
    for (int cg = CS_MAX_CG; cg >= 1; cg--)
      if (cands[(size_t)cg].per_col < best.per_col - 1e-9) best = cands[(size_t)cg];  // (the wider step on a tie)
    if (best.cg == 0) return nullptr;
    prm.Cg = best.cg;
    prm.Lw = best.lw;
    P = cs_build_plan(csc.ptr.data(), csc.idx.data(), csc.val.data(), csc.cols, run, prm);
    if (P.ok && cs_lds_bytes(P.n_slots, prm.Cg, P.max_hot_col) > CS_LDS_MAX) P.ok = false;
  }
  if (!P.ok) return nullptr;
  auto st = std::make_shared<CsStream>();
  st->cols.upload(run);
  st->cold_ptr.upload(P.cold_ptr);
  st->cold_rc.upload(P.cold_rc);
  st->cold_x.upload(P.cold_x);
  st->enter_ptr.upload(P.enter_ptr);
  st->enter_row.upload(P.enter_row);
  st->enter_slot.upload(P.enter_slot);
  st->exit_ptr.upload(P.exit_ptr);
  st->exit_row.upload(P.exit_row);
  st->exit_slot.upload(P.exit_slot);
  st->hot_ptr.upload(P.hot_ptr);
  st->hot_slot.upload(P.hot_slot);
  st->hot_x.upload(P.hot_x);
  st->in_ring.alloc((size_t)CS_RING * std::max(P.max_enter, 1) * 4);
  st->out_ring.alloc((size_t)CS_RING * std::max(P.max_exit, 1) * 2);
  st->part.alloc((size_t)CS_RING * P.prm.NB * CS_NWS * CS_MAX_CG);
  st->oldnew.alloc((size_t)CS_RING * CS_MAX_CG);
  st->sync.alloc(1);
  CsStreamInfo &I = st->info;
  I.n_steps = P.n_steps;
  I.Cg = P.prm.Cg;
  I.Lw = P.prm.Lw;
  I.NB = P.prm.NB;
  I.RD = P.prm.RD;
  I.n_slots = P.n_slots;
  I.max_hot_col = P.max_hot_col;
  I.max_enter = P.max_enter;
  I.max_exit = P.max_exit;
  I.n_cold = P.n_cold;
  I.n_hot = P.n_hot;
  I.plan_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (info) *info = I;
  return st;
}

const CsStreamInfo &cs_stream_info(const CsStream &st) { return st.info; }

static __global__ void k_cs_gather_i32(const int32_t *__restrict__ src, const int32_t *__restrict__ idx, int n, int32_t *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = src[idx[i]];
}

void cs_stream_launch(hipStream_t s, const SweepArgs &a, const CsStream &st, bool latent, int *error) {
  const CsStreamInfo &I = st.info;
  const int n = (int)st.cols.n;
  if (st.col_group.n < (size_t)n || st.col_group_of != a.group) {
    st.col_group.alloc((size_t)std::max(n, 1));
    hipLaunchKernelGGL(k_cs_gather_i32, dim3((n + 255) / 256), dim3(256), 0, s, a.group, st.cols.p, n, st.col_group.p);
    st.col_group_of = a.group;
  }
  const size_t lds = cs_lds_bytes(I.n_slots, I.Cg, I.max_hot_col);
  static DeviceOnce raised;
  if (raised.need()) {
    MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_cs_stream<PBlockV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CS_LDS_MAX));
    MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_cs_stream<PBlockW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CS_LDS_MAX));
    raised.mark();
  }
  MFM_HIP_CHECK(hipMemsetAsync(st.sync.p, 0, sizeof(CsSync), s));
  CsArgs g;
  g.n_cols = n;
  g.n_steps = I.n_steps;
  g.Cg = I.Cg;
  g.Lw = I.Lw;
  g.NB = I.NB;
  g.RD = I.RD;
  g.max_enter = I.max_enter;
  g.max_exit = I.max_exit;
  g.n_slots = I.n_slots;
  g.ecap = cs_ecap(I.max_hot_col);
  g.dbg = env_int("MFM_CS_DBG", 0);
  g.cols = st.cols.p;
  g.col_group = st.col_group.p;
  g.cold_ptr = st.cold_ptr.p;
  g.cold_rc = st.cold_rc.p;
  g.cold_x = st.cold_x.p;
  g.enter_ptr = st.enter_ptr.p;
  g.enter_row = st.enter_row.p;
  g.enter_slot = st.enter_slot.p;
  g.exit_ptr = st.exit_ptr.p;
  g.exit_row = st.exit_row.p;
  g.exit_slot = st.exit_slot.p;
  g.hot_ptr = st.hot_ptr.p;
  g.hot_slot = st.hot_slot.p;
  g.hot_x = st.hot_x.p;
  g.in_ring = st.in_ring.p;
  g.out_ring = st.out_ring.p;
  g.part = st.part.p;
  g.oldnew = st.oldnew.p;
  g.sync = st.sync.p;
  g.error = error;
  // MFM_CB_PROF=n: waits and work of the four roles (s_memrealtime of one wavefront each), printed every n launches
  static const int cb_prof = env_int("MFM_CB_PROF", 0);
  static DevBuf<unsigned long long> prof_buf;
  static long prof_launches = 0;
  g.prof = nullptr;
  if (cb_prof > 0) {
    if (!prof_buf.p) {
      prof_buf.alloc(16);
      MFM_HIP_CHECK(hipMemset(prof_buf.p, 0, 16 * sizeof(unsigned long long)));
    }
    g.prof = prof_buf.p;
  }
  // MFM_CS_TRACE=file (with MFM_CB_PROF): raw stamps of every actor and step of the MFM_CS_TRACE_LAUNCH-th launch (default 40)
  static const char *trace_file = std::getenv("MFM_CS_TRACE");
  static const long trace_launch = env_int("MFM_CS_TRACE_LAUNCH", 40);
  static long n_launch = 0;
  DevBuf<unsigned long long> trace_buf;
  g.trace = nullptr;
  const bool tracing = trace_file && g.prof && ++n_launch == trace_launch;
  const size_t trace_n = (size_t)(3 + 2 * I.NB) * I.n_steps * 4;
  if (tracing) {
    trace_buf.alloc_zero(trace_n, s);
    g.trace = trace_buf.p;
  }
  if (latent)
    hipLaunchKernelGGL((k_cs_stream<PBlockV>), dim3(1 + I.NB), dim3(CS_NT), lds, s, a, g);
  else
    hipLaunchKernelGGL((k_cs_stream<PBlockW>), dim3(1 + I.NB), dim3(CS_NT), lds, s, a, g);
  MFM_HIP_CHECK(hipGetLastError());
  if (tracing) {
    std::vector<unsigned long long> h(trace_n);
    MFM_HIP_CHECK(hipStreamSynchronize(s));
    MFM_HIP_CHECK(hipMemcpy(h.data(), trace_buf.p, trace_n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (FILE *f = std::fopen(trace_file, "w")) {
      std::fprintf(f, "# k_cs_stream launch %ld: n_steps %d Cg %d Lw %d NB %d RD %d slots %d; rows: actor step t0 t1 t2 (10 ns ticks; actors: 0 walker, 1 X, "
                      "2 Y, 3.. U per range, %d.. S per range)\n", n_launch, I.n_steps, I.Cg, I.Lw, I.NB, I.RD, I.n_slots, 3 + I.NB);
      for (int ac = 0; ac < 3 + 2 * I.NB; ac++)
        for (int st_ = 0; st_ < I.n_steps; st_++) {
          const unsigned long long *r = &h[((size_t)ac * I.n_steps + st_) * 4];
          std::fprintf(f, "%d %d %llu %llu %llu\n", ac, st_, r[0], r[1], r[2]);
        }
      std::fclose(f);
    }
  }
  if (cb_prof > 0 && ++prof_launches % cb_prof == 0) {
    unsigned long long h[16];
    MFM_HIP_CHECK(hipStreamSynchronize(s));
    MFM_HIP_CHECK(hipMemcpy(h, prof_buf.p, sizeof(h), hipMemcpyDeviceToHost));
    MFM_HIP_CHECK(hipMemset(prof_buf.p, 0, sizeof(h)));
    const double nsx = (double)std::max<unsigned long long>(h[2], 1) * 100.0;
    std::fprintf(stderr,
                 "[k_cs_stream] %ld launches (Cg %d, Lw %d, NB %d, %d slots), us per step -- walker: waits %.2f, walks %.2f | X: waits %.2f, works %.2f | Y: waits %.2f, "
                 "stages %.2f | U (range 0): waits %.2f, works %.2f | S (range 0): waits %.2f, works %.2f\n",
                 prof_launches, I.Cg, I.Lw, I.NB, I.n_slots, h[0] / nsx, h[1] / nsx, h[6] / nsx, h[7] / nsx, h[4] / nsx, h[5] / nsx, h[8] / nsx, h[9] / nsx, h[12] / nsx,
                 h[13] / nsx);
  }
}

}  // namespace mfm

// Host-only check of a plan's data flow (tests/test_chain_plan_cpu.py): builds the plan of the chain over ALL columns of the given
// CSC (column j -> ascending rows) and emulates the launch's four actors on separate copies of what each of them can see, against
// the plain sequential sweep. info[8]: built (0/1), slots, most entering / leaving rows of a step, cold entries, hot entries, most
// hot entries of a column, steps. Returns MFM_OK, or MFM_ERR_INVALID with *max_rel_diff = 1e300 when the emulation got stuck.
extern "C" int mfm_cs_plan_selftest(int64_t n_rows, int32_t n_cols, const int64_t *colptr, const int32_t *rowidx, const double *val,
                                    int32_t cg, int32_t lw, int32_t nb, int32_t rd, int32_t cap, double *max_rel_diff, int64_t *info) {
  using namespace mfm;
  try {
    std::vector<int32_t> run((size_t)n_cols);
    for (int32_t j = 0; j < n_cols; j++) run[j] = j;
    CsParams prm;
    prm.Cg = cg;
    prm.Lw = lw;
    prm.NB = nb;
    prm.RD = rd;
    prm.cap = cap;
    CsPlanHost P = cs_build_plan(colptr, rowidx, val, n_rows, run, prm);
    if (info) {
      info[0] = P.ok ? 1 : 0;
      info[1] = P.n_slots;
      info[2] = P.max_enter;
      info[3] = P.max_exit;
      info[4] = P.n_cold;
      info[5] = P.n_hot;
      info[6] = P.max_hot_col;
      info[7] = P.n_steps;
    }
    if (max_rel_diff) *max_rel_diff = 0.0;
    if (!P.ok) return MFM_OK;
    std::vector<double> q0((size_t)n_rows);
    for (int64_t r = 0; r < n_rows; r++) q0[r] = std::sin(0.7 * (double)r) + 0.1;
    std::string err;
    const double d = cs_emulate(P, colptr, rowidx, val, run, q0, &err);
    if (max_rel_diff) *max_rel_diff = d;
    return err.empty() ? MFM_OK : MFM_ERR_INVALID;
  } catch (...) {
    return MFM_ERR_RUNTIME;
  }
}
