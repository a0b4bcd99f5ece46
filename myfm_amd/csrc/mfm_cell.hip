// mfm_cell.hip -- the cell path of update_V (see mfm_cell.hpp): planner, pass kernel, table / draw / reduce kernels.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <mutex>
#include <thread>

#include "mfm_cell.hpp"
#include "mfm_wave.hpp"

namespace mfm {

static inline int cdiv_c(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------------------------------------
// LDS layout of a pass (offsets in doubles): per LDS stream its value table(s), the pending field's (d1, d2), the accumulators
// of the statistics field, the turn word.   off[0..3] = A, off[4..7] = S, off[8] = DP, off[9] = accumulators, off[10] = turn
size_t CellPlan::lds_bytes(int P, int F, bool sw, int *off) const {
  size_t o = 0;
  int tmp[11];
  if (!off) off = tmp;
  const int sP = P >= 0 ? fields[P].stream : -1, sF = F >= 0 ? fields[F].stream : -1;
  for (int s = 0; s < CELL_MAX_STREAMS; s++) off[s] = off[4 + s] = 0;
  for (size_t s = 0; s < streams.size(); s++) {
    if (streams[s].type == CELL_I) continue;
    const size_t n = streams[s].type == CELL_U ? (size_t)umax : (size_t)streams[s].card;
    const bool pair = P >= 0 && F >= 0 && (sw || (int)s == sP || (int)s == sF);
    off[s] = (int)o;
    o += n;
    if (pair) {
      off[4 + s] = (int)o;
      o += n;
    } else {
      off[4 + s] = off[s];
    }
  }
  off[8] = off[9] = 0;
  if (sP >= 0 && streams[sP].type != CELL_I) {
    o = (o + 1) & ~(size_t)1;
    off[8] = (int)o;
    o += 2 * (streams[sP].type == CELL_U ? (size_t)umax : (size_t)streams[sP].card);
  }
  if (sF >= 0 && streams[sF].type != CELL_I) {
    const size_t ns = fields[F].kind == 0 ? 2 : 4;
    off[9] = (int)o;
    o += ns * (streams[sF].type == CELL_U ? (size_t)umax : (size_t)streams[sF].card);
  }
  off[10] = (int)o;
  o += 2;
  return o * sizeof(double);
}

// ---------------------------------------------------------------------------------------------------------------------------
// planner
template <class F>
static void par_for(int64_t n, int64_t min_per_thread, F f) {
  const int hw = (int)std::thread::hardware_concurrency();
  const int T = (int)std::max<int64_t>(1, std::min<int64_t>({n / std::max<int64_t>(min_per_thread, 1), 16, hw > 0 ? hw : 1}));
  if (T <= 1) {
    f((int64_t)0, n);
    return;
  }
  std::vector<std::thread> pool;
  for (int t = 1; t < T; t++) pool.emplace_back(f, n * t / T, n * (t + 1) / T);
  f((int64_t)0, n / T);
  for (auto &t : pool) t.join();
}

bool cell_plan_build(CellPlan &cp, const HostCsr &X, const std::vector<CellBlockIn> &blocks, int n_cu, hipStream_t s) {
  cp.ready = false;
  cp.streams.clear();
  cp.fields.clear();
  const int64_t N = X.rows;
  cp.N = N;
  if (N <= 0 || N >= (int64_t)2147483647) return cp.fail("no rows");
  const int64_t W = X.ptr[1] - X.ptr[0];
  if (W < 1 || W > CELL_MAX_STREAMS || X.nnz() != N * W) return cp.fail("the main table is not a row of one-hot fields");
  // field p = the p-th stored entry of every row: its column range, unit values, first field sorted
  std::vector<int64_t> lo((size_t)W, (int64_t)1 << 60), hi((size_t)W, -1);
  std::atomic<int> bad(0);
  std::mutex mx;
  par_for(N, 1 << 20, [&](int64_t a, int64_t b) {
    std::vector<int64_t> l((size_t)W, (int64_t)1 << 60), h((size_t)W, -1);
    for (int64_t t = a; t < b; t++) {
      if (X.ptr[t + 1] - X.ptr[t] != W) bad = 1;
      for (int64_t p = 0; p < W; p++) {
        const int64_t c = X.idx[t * W + p];
        l[p] = std::min(l[p], c);
        h[p] = std::max(h[p], c);
        if (X.val[t * W + p] != 1.0) bad = 1;
      }
      if (t > 0 && X.idx[t * W] < X.idx[(t - 1) * W]) bad = 2;
    }
    std::lock_guard<std::mutex> g(mx);
    for (int64_t p = 0; p < W; p++) {
      lo[p] = std::min(lo[p], l[p]);
      hi[p] = std::max(hi[p], h[p]);
    }
  });
  if (bad == 1) return cp.fail("the main table is not a row of unit-valued one-hot fields");
  if (bad == 2) return cp.fail("the rows are not sorted by the first field");
  for (int64_t p = 1; p < W; p++)
    if (lo[p] <= hi[p - 1]) return cp.fail("the one-hot fields' column ranges overlap");
  std::vector<int64_t> base((size_t)W + 1, 0);
  for (int64_t p = 1; p < W; p++) base[p] = lo[p];
  base[W] = X.cols;
  // streams: main fields first, then every block whose map is not one of the streams already there
  struct HostStream {
    int main_p = -1;               // main field position, or
    const int64_t *map = nullptr;  // a block's map
  };
  std::vector<HostStream> hs;
  for (int64_t p = 0; p < W; p++) {
    CellStream st;
    st.card = base[p + 1] - base[p];
    st.fields.push_back((int)cp.fields.size());
    CellField f;
    f.stream = (int)p;
    f.kind = 0;
    f.n = st.card;
    f.base = base[p];
    cp.fields.push_back(f);
    cp.streams.push_back(st);
    HostStream h;
    h.main_p = (int)p;
    hs.push_back(h);
  }
  auto idx_of = [&](const HostStream &h, int64_t t) -> int64_t {
    return h.main_p >= 0 ? (int64_t)X.idx[t * W + h.main_p] - base[h.main_p] : h.map[t];
  };
  for (size_t b = 0; b < blocks.size(); b++) {
    int found = -1;
    for (size_t si = 0; si < hs.size() && found < 0; si++) {
      std::atomic<int> diff(0);
      par_for(N, 1 << 20, [&](int64_t a, int64_t e) {
        for (int64_t t = a; t < e && !diff.load(std::memory_order_relaxed); t++)
          if (idx_of(hs[si], t) != blocks[b].map[t]) diff = 1;
      });
      if (!diff) found = (int)si;
    }
    if (found < 0) {
      if (hs.size() >= (size_t)CELL_MAX_STREAMS) return cp.fail("more index streams than a row record holds");
      HostStream h;
      h.map = blocks[b].map;
      hs.push_back(h);
      cp.streams.push_back(CellStream());
      found = (int)hs.size() - 1;
    }
    if (cp.fields.size() >= (size_t)CELL_MAX_FIELDS) return cp.fail("too many fields");
    CellField f;
    f.stream = found;
    f.kind = 1;
    f.n = blocks[b].B;
    f.base = (int64_t)b;
    cp.streams[found].card = std::max(cp.streams[found].card, blocks[b].B);
    cp.streams[found].fields.push_back((int)cp.fields.size());
    cp.fields.push_back(f);
  }
  // stream types and record slots
  cp.sU = 0;
  cp.sI = -1;
  cp.streams[0].type = CELL_U;
  int n_slots = 1;
  cp.streams[0].slot = 0;
  for (size_t si = 1; si < cp.streams.size(); si++) {
    if (cp.streams[si].card <= CELL_SMALL_MAX) {
      cp.streams[si].type = CELL_C;
      cp.streams[si].slot = n_slots++;
    } else {
      if (cp.sI >= 0) return cp.fail("more than one large scattered index stream");
      cp.sI = (int)si;
      cp.streams[si].type = CELL_I;
    }
  }
  cp.item32 = false;
  if (cp.sI >= 0) {
    if (cp.streams[cp.sI].card <= 65536 && n_slots < 4)
      cp.streams[cp.sI].slot = n_slots++;
    else {
      cp.streams[cp.sI].slot = -1;
      cp.item32 = true;
    }
    if (cp.streams[cp.sI].card >= (int64_t)2147483647) return cp.fail("I stream too large");
  }
  if (n_slots > 4) return cp.fail("more small index streams than a row record holds");
  // groups of consecutive U values, rows balanced; more (smaller) groups until every pass of the sweep fits its LDS
  const int64_t cardU = cp.streams[0].card;
  std::vector<int64_t> grow;  // first row of every group, then N
  std::vector<int32_t> gu0;
  const HostStream &hU = hs[0];
  bool fits = false;
  for (int mult = 1; mult <= 16 && !fits; mult *= 2) {
    const int64_t G0 = (int64_t)std::max(1, n_cu) * mult;
    const int64_t target = (N + G0 - 1) / G0;
    grow.assign(1, 0);
    for (int64_t g = 1; g < G0; g++) {
      int64_t r = std::min(N, g * target);
      while (r < N && r > 0 && idx_of(hU, r) == idx_of(hU, r - 1)) r++;
      if (r > grow.back() && r < N) grow.push_back(r);
    }
    grow.push_back(N);
    const int G = (int)grow.size() - 1;
    gu0.assign((size_t)G + 1, 0);
    for (int g = 1; g < G; g++) gu0[g] = (int32_t)idx_of(hU, grow[g]);
    gu0[G] = (int32_t)cardU;
    int64_t um = 0;
    for (int g = 0; g < G; g++) um = std::max<int64_t>(um, gu0[g + 1] - gu0[g]);
    cp.G = G;
    cp.umax = um;
    if (um > 65535) continue;
    // the passes the sweep will run: (P, F) = (last of the previous factor | none, first), (k - 1, k), (last, none)
    const int m = (int)cp.fields.size();
    size_t worst = 0;
    for (int k = 0; k < m; k++) {
      const int P = k == 0 ? m - 1 : k - 1;
      size_t need = cp.lds_bytes(P, k, k == 0);
      if (need > CELL_LDS_BYTES)  // (split form: apply-only pass, then statistics-only pass)
        need = std::max(cp.lds_bytes(P, -1, false), cp.lds_bytes(-1, k, false));
      worst = std::max(worst, need);
    }
    fits = worst <= CELL_LDS_BYTES;
  }
  if (!fits) return cp.fail("a group's tables do not fit the LDS (a first-field value with too many rows, or too many values per group)");
  const int G = cp.G;
  const int64_t cardI = cp.sI >= 0 ? cp.streams[cp.sI].card : 0;
  if (cardI > 0 && (double)G * (double)cardI * 48.0 > 16e9) return cp.fail("the (group, item) partials would not fit");
  // rows in cell order: inside a group by the I index (stable), chunks cut between two I values
  std::vector<int32_t> perm((size_t)N), chunk0((size_t)G * CELL_NW + 1, 0), steps((size_t)G, 0), item;
  std::vector<uint2> ix((size_t)N);
  if (cp.item32) item.resize((size_t)N);
  std::atomic<int> next_g(0);
  auto work = [&]() {
    std::vector<int32_t> cnt, key;
    for (;;) {
      const int g = next_g.fetch_add(1);
      if (g >= G) break;
      const int64_t R0 = grow[g], R1 = grow[g + 1], L = R1 - R0;
      if (cp.sI >= 0) {
        const HostStream &hI = hs[cp.sI];
        key.resize((size_t)L);
        cnt.assign((size_t)cardI + 1, 0);
        for (int64_t r = 0; r < L; r++) {
          key[r] = (int32_t)idx_of(hI, R0 + r);
          cnt[key[r] + 1]++;
        }
        for (int64_t i = 0; i < cardI; i++) cnt[i + 1] += cnt[i];
        for (int64_t r = 0; r < L; r++) perm[R0 + cnt[key[r]]++] = (int32_t)(R0 + r);
      } else {
        for (int64_t r = 0; r < L; r++) perm[R0 + r] = (int32_t)(R0 + r);
      }
      // records
      for (int64_t p = R0; p < R1; p++) {
        const int64_t t = perm[p];
        uint32_t sl[4] = {0, 0, 0, 0};
        for (size_t si = 0; si < cp.streams.size(); si++) {
          const int64_t v = idx_of(hs[si], t);
          const int slot = cp.streams[si].slot;
          if (si == 0)
            sl[0] = (uint32_t)(v - gu0[g]);
          else if (slot >= 0)
            sl[slot] = (uint32_t)v;
          else
            item[p] = (int32_t)v;
        }
        ix[p] = make_uint2(sl[0] | (sl[1] << 16), sl[2] | (sl[3] << 16));
      }
      // wave chunks
      int64_t longest = 0;
      chunk0[(size_t)g * CELL_NW] = (int32_t)R0;
      for (int j = 1; j <= CELL_NW; j++) {
        int64_t r = j == CELL_NW ? R1 : R0 + L * j / CELL_NW;
        if (cp.sI >= 0 && j < CELL_NW) {
          const HostStream &hI = hs[cp.sI];
          while (r < R1 && r > R0 && idx_of(hI, perm[r]) == idx_of(hI, perm[r - 1])) r++;
        }
        r = std::max<int64_t>(r, chunk0[(size_t)g * CELL_NW + j - 1]);
        if (j < CELL_NW) chunk0[(size_t)g * CELL_NW + j] = (int32_t)r;
        longest = std::max<int64_t>(longest, r - chunk0[(size_t)g * CELL_NW + j - 1]);
      }
      steps[g] = (int32_t)((longest + 64 * CELL_R - 1) / (64 * CELL_R));
    }
  };
  {
    const int hw = (int)std::thread::hardware_concurrency();
    const int T = std::max(1, std::min({hw > 0 ? hw : 1, 16, G}));
    std::vector<std::thread> pool;
    for (int t = 1; t < T; t++) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
  }
  chunk0[(size_t)G * CELL_NW] = (int32_t)N;
  // device
  cp.ix.upload(ix);
  cp.item.upload(item);
  cp.perm.upload(perm);
  cp.chunk0.upload(chunk0);
  cp.grp_u0.upload(gu0);
  cp.grp_steps.upload(steps);
  cp.e.alloc((size_t)N);
  int64_t maxcard = 0;
  for (size_t si = 0; si < cp.streams.size(); si++) {
    const size_t n = (size_t)cp.streams[si].card;
    maxcard = std::max<int64_t>(maxcard, cp.streams[si].card);
    if (cp.streams[si].type == CELL_I) continue;
    cp.QA[si].alloc_zero(n, s);
    cp.QS[si].alloc_zero(n, s);
  }
  if (cp.sI >= 0) {
    cp.packI.alloc_zero((size_t)cardI * 4, s);
    cp.cells2.alloc_zero((size_t)G * cardI * 2, s);
    cp.cells4.alloc_zero((size_t)G * cardI * 4, s);
  }
  int64_t maxC = 0;
  for (auto &st : cp.streams)
    if (st.type == CELL_C) maxC = std::max(maxC, st.card);
  cp.cpart.alloc_zero((size_t)std::max<int64_t>(1, (int64_t)G * maxC * 4), s);
  cp.DP.alloc_zero((size_t)std::max<int64_t>(1, maxcard), s);
  cp.stat.alloc_zero((size_t)std::max<int64_t>(1, cardU * 2), s);
  MFM_HIP_CHECK(hipStreamSynchronize(s));
  cp.ready = true;
  return true;
}

// ---------------------------------------------------------------------------------------------------------------------------
// e between row order (eq[t].x) and cell order
__global__ void k_cell_pack(const double2 *__restrict__ eq, const int32_t *__restrict__ perm, int64_t N, double *__restrict__ e) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < N) e[p] = eq[perm[p]].x;
}
__global__ void k_cell_unpack(const double *__restrict__ e, const int32_t *__restrict__ perm, int64_t N, double2 *__restrict__ eq) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < N) eq[perm[p]].x = e[p];
}
void cell_pack_e(hipStream_t s, CellPlan &cp, const double2 *eq) {
  hipLaunchKernelGGL(k_cell_pack, dim3(cdiv_c(cp.N, 256)), dim3(256), 0, s, eq, cp.perm.p, cp.N, cp.e.p);
  MFM_HIP_CHECK(hipGetLastError());
}
void cell_unpack_e(hipStream_t s, CellPlan &cp, double2 *eq) {
  hipLaunchKernelGGL(k_cell_unpack, dim3(cdiv_c(cp.N, 256)), dim3(256), 0, s, cp.e.p, cp.perm.p, cp.N, eq);
  MFM_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------------------
// tables: dst[i * dst_stride] = sum_k src_k[i * stride_k]   (k ascending: fixed order)
struct CellPrepJob {
  double *dst;
  int dst_stride, n, nsrc;
  const double *src[4];
  int sstride[4], sn[4];  // (a field shorter than its stream's table contributes only where it has values)
};
constexpr int CELL_PREP_JOBS = 12;
struct CellPrepArgs {
  int n_jobs;
  CellPrepJob job[CELL_PREP_JOBS];
};
__global__ __launch_bounds__(256) void k_cell_prep(CellPrepArgs a) {
  const CellPrepJob &j = a.job[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= j.n) return;
  double v = 0.0;
  for (int k = 0; k < j.nsrc; k++)
    if (i < j.sn[k]) v += j.src[k][(int64_t)i * j.sstride[k]];
  j.dst[(int64_t)i * j.dst_stride] = v;
}

void cell_prep(hipStream_t s, Timing &tm, CellPlan &cp, const std::vector<CellSrc> &cur, bool doA, int exA, bool doS, int exS,
               bool dp_to_I) {
  CellPrepArgs a;
  a.n_jobs = 0;
  int maxn = 0;
  auto add = [&](double *dst, int dst_stride, int64_t n, int stream, int exclude) {
    if (a.n_jobs >= CELL_PREP_JOBS) throw Error(MFM_ERR_RUNTIME, "internal: cell_prep job list full");
    CellPrepJob &j = a.job[a.n_jobs++];
    j.dst = dst;
    j.dst_stride = dst_stride;
    j.n = (int)n;
    j.nsrc = 0;
    if (stream >= 0)
      for (int f : cp.streams[stream].fields) {
        if (f == exclude) continue;
        if (j.nsrc >= 4) throw Error(MFM_ERR_RUNTIME, "internal: more than four fields on one index stream");
        j.src[j.nsrc] = cur[f].p;
        j.sstride[j.nsrc] = cur[f].stride;
        j.sn[j.nsrc] = (int)std::min<int64_t>(cp.fields[f].n, n);
        j.nsrc++;
      }
    maxn = std::max(maxn, (int)n);
  };
  for (size_t si = 0; si < cp.streams.size(); si++) {
    const CellStream &st = cp.streams[si];
    if (st.type == CELL_I) {
      if (doA) add(cp.packI.p + 0, 4, st.card, (int)si, exA);
      if (doS) add(cp.packI.p + 1, 4, st.card, (int)si, exS);
      if (dp_to_I) {
        for (int k = 0; k < 2; k++) {
          add(cp.packI.p + 2 + k, 4, st.card, -1, -1);
          CellPrepJob &j = a.job[a.n_jobs - 1];
          j.nsrc = 1;
          j.src[0] = (const double *)cp.DP.p + k;
          j.sstride[0] = 2;
          j.sn[0] = (int)st.card;
        }
      }
    } else {
      if (doA) add(cp.QA[si].p, 1, st.card, (int)si, exA);
      if (doS) add(cp.QS[si].p, 1, st.card, (int)si, exS);
    }
  }
  if (!a.n_jobs || !maxn) return;
  TimedLaunch t(tm, s, KC_CELL_SMALL, 0.0);
  hipLaunchKernelGGL(k_cell_prep, dim3(cdiv_c(maxn, 256), a.n_jobs), dim3(256), 0, s, a);
  MFM_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------------------
// the pass
struct CellPassArgs {
  const uint2 *ix;
  const int32_t *item;
  double *e;
  const int32_t *chunk0, *grp_u0, *grp_steps;
  const double *QA[CELL_MAX_STREAMS], *QS[CELL_MAX_STREAMS];
  const double *packI;
  const double2 *DP;
  int n_streams;
  int type[CELL_MAX_STREAMS], slot[CELL_MAX_STREAMS], card[CELL_MAX_STREAMS], pair[CELL_MAX_STREAMS];
  int ldsA[CELL_MAX_STREAMS], ldsS[CELL_MAX_STREAMS], ldsDP, ldsAcc, ldsTurn;
  int sP, sF, ns;
  double *out;
  int out_stride;
  int n_out;  // U statistics: index values the output has room for (a block may have fewer rows than its stream has values)
  int cardI;
};

__device__ __forceinline__ int cell_slot(uint2 r, int slot) {
  const uint32_t w = slot < 2 ? r.x : r.y;
  return (int)((w >> ((slot & 1) * 16)) & 0xffffu);
}

// FT: stream type of the statistics field (-1: no statistics); NS: its sums (2: main field, 4: block); HASP: a pending field
// is applied; ITEM32: the I index comes from the int32 array
template <int FT, int NS, bool HASP, bool ITEM32>
__global__ __launch_bounds__(CELL_NT) void k_cell_pass(CellPassArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int g = blockIdx.x, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int u0 = a.grp_u0[g], nu = a.grp_u0[g + 1] - u0;
  // ---- tables -> LDS
  for (int s = 0; s < a.n_streams; s++) {
    const int ty = a.type[s];
    if (ty == CELL_I) continue;
    const int n = ty == CELL_U ? nu : a.card[s];
    const int o = ty == CELL_U ? u0 : 0;
    if (a.pair[s] == 1) {
      for (int i = tid; i < n; i += CELL_NT) {
        lds[a.ldsA[s] + i] = a.QA[s][o + i];
        lds[a.ldsS[s] + i] = a.QS[s][o + i];
      }
    } else {  // one table serves both sides (2: an apply-only pass reads the pending side's)
      const double *src = a.pair[s] == 2 ? a.QA[s] : a.QS[s];
      for (int i = tid; i < n; i += CELL_NT) lds[a.ldsS[s] + i] = src[o + i];
    }
  }
  if (HASP && a.type[a.sP] != CELL_I) {
    const int n = a.type[a.sP] == CELL_U ? nu : a.card[a.sP];
    const int o = a.type[a.sP] == CELL_U ? u0 : 0;
    double2 *d = (double2 *)(lds + a.ldsDP);
    for (int i = tid; i < n; i += CELL_NT) d[i] = a.DP[o + i];
  }
  int *turn = (int *)(lds + a.ldsTurn);
  const int nacc = (FT == CELL_U ? nu : (FT == CELL_C ? a.card[a.sF] : 0)) * NS;
  if (FT == CELL_U || FT == CELL_C) {
    for (int i = tid; i < nacc; i += CELL_NT) lds[a.ldsAcc + i] = 0.0;
    if (tid == 0) *turn = 0;
  }
  __syncthreads();

  const int r0 = a.chunk0[g * CELL_NW + wv], r1 = a.chunk0[g * CELL_NW + wv + 1];
  const int steps = a.grp_steps[g];
  const int slotF = FT >= 0 ? a.slot[a.sF] : 0;
  const int slotP = HASP ? a.slot[a.sP] : 0;
  const bool p_on_I = HASP && a.type[a.sP] == CELL_I;
  const double2 *dpl = (const double2 *)(lds + a.ldsDP);
  double *acc = lds + a.ldsAcc;
  const double4 *packI = (const double4 *)a.packI;
  // I statistics: the open run at the end of the previous window (wave-uniform)
  int carry_it = -1;
  double carry[4] = {0.0, 0.0, 0.0, 0.0};

  // software pipeline: the records and residuals of step st + 1 are requested before step st is worked on
  uint2 rec_n[CELL_R];
  double e_n[CELL_R];
  int it_n[CELL_R];
#pragma unroll
  for (int k = 0; k < CELL_R; k++) {
    const int r = r0 + k * 64 + lane;
    rec_n[k] = make_uint2(0, 0);
    e_n[k] = 0.0;
    it_n[k] = 0;
    if (r < r1) {
      rec_n[k] = a.ix[r];
      e_n[k] = __builtin_nontemporal_load(a.e + r);
      if (ITEM32) it_n[k] = a.item[r];
    }
  }
  for (int st = 0; st < steps; st++) {
    const int base = r0 + st * (64 * CELL_R);
    uint2 rec[CELL_R];
    double e[CELL_R];
    int it[CELL_R];
#pragma unroll
    for (int k = 0; k < CELL_R; k++) {
      rec[k] = rec_n[k];
      e[k] = e_n[k];
      it[k] = it_n[k];
    }
    if (st + 1 < steps) {
#pragma unroll
      for (int k = 0; k < CELL_R; k++) {
        const int r = base + (64 * CELL_R) + k * 64 + lane;
        if (r < r1) {
          rec_n[k] = a.ix[r];
          e_n[k] = __builtin_nontemporal_load(a.e + r);
          if (ITEM32) it_n[k] = a.item[r];
        }
      }
    }
    double v[CELL_R][NS > 0 ? NS : 1];
    int idxF[CELL_R];
    bool valid[CELL_R];
#pragma unroll
    for (int k = 0; k < CELL_R; k++) {
      const int r = base + k * 64 + lane;
      valid[k] = r < r1;
      double qa = 0.0, qs = 0.0;
      double2 dI = make_double2(0.0, 0.0);
      int itv = -2;
      if (valid[k]) {
#pragma unroll
        for (int s = 0; s < CELL_MAX_STREAMS; s++) {
          if (s < a.n_streams) {
            if (a.type[s] == CELL_I) {
              itv = ITEM32 ? it[k] : cell_slot(rec[k], a.slot[s]);
              const double4 pk = packI[itv];
              qa += pk.x;
              qs += pk.y;
              dI = make_double2(pk.z, pk.w);
            } else {
              const int i = cell_slot(rec[k], a.slot[s]);
              const double x = lds[a.ldsS[s] + i];
              qs += x;
              if (HASP) qa += a.pair[s] == 1 ? lds[a.ldsA[s] + i] : x;
            }
          }
        }
        if (HASP) {
          const double2 d = p_on_I ? dI : dpl[cell_slot(rec[k], slotP)];
          e[k] += qa * d.x + d.y;
          __builtin_nontemporal_store(e[k], a.e + r);
        }
      }
      it[k] = itv;
      if (FT >= 0) {
        const double h = qs;
        if (NS == 2) {
          v[k][0] = valid[k] ? h * h : 0.0;
          v[k][1] = valid[k] ? e[k] * h : 0.0;
        } else if (NS == 4) {
          v[k][0] = valid[k] ? h : 0.0;
          v[k][1] = valid[k] ? h * h : 0.0;
          v[k][2] = valid[k] ? e[k] : 0.0;
          v[k][3] = valid[k] ? e[k] * h : 0.0;
        }
        idxF[k] = (FT == CELL_I) ? itv : cell_slot(rec[k], slotF);
      }
    }
    if (FT == CELL_U || FT == CELL_C) {
      // the waves add to the group's table in turn (wave after wave, step after step): every sum has a fixed order
      const int my = st * CELL_NW + wv;
      if (lane == 0)
        while (__hip_atomic_load(turn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != my) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < CELL_R; k++)
        if (valid[k]) {
#pragma unroll
          for (int j = 0; j < NS; j++)
            __hip_atomic_fetch_add(&acc[idxF[k] * NS + j], v[k][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) __hip_atomic_store(turn, my + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (FT == CELL_I) {
#pragma unroll
      for (int k = 0; k < CELL_R; k++) {
        const int itk = idxF[k];
        int prev = __shfl_up(itk, 1, 64);
        if (lane == 0) prev = carry_it;
        const bool head = itk != prev;
        if (lane == 0) {
          if (!head) {
#pragma unroll
            for (int j = 0; j < NS; j++) v[k][j] = carry[j] + v[k][j];
          } else if (carry_it >= 0) {
            double *o = a.out + ((int64_t)g * a.cardI + carry_it) * NS;
#pragma unroll
            for (int j = 0; j < NS; j++) o[j] = carry[j];
          }
        }
        int f1 = head ? 1 : 0;
        wave_segscan2(v[k][0], v[k][1], f1);
        if (NS == 4) {
          int f2 = head ? 1 : 0;
          wave_segscan2(v[k][2], v[k][3], f2);
        }
        const int nxt = __shfl_down(head ? 1 : 0, 1, 64);
        if (lane < 63 && nxt && itk >= 0) {
          double *o = a.out + ((int64_t)g * a.cardI + itk) * NS;
#pragma unroll
          for (int j = 0; j < NS; j++) o[j] = v[k][j];
        }
        carry_it = __builtin_amdgcn_readlane(itk, 63);
#pragma unroll
        for (int j = 0; j < NS; j++) carry[j] = readlane_f64(v[k][j], 63);
      }
    }
  }
  if (FT == CELL_I) {
    if (lane == 0 && carry_it >= 0) {
      double *o = a.out + ((int64_t)g * a.cardI + carry_it) * NS;
#pragma unroll
      for (int j = 0; j < NS; j++) o[j] = carry[j];
    }
  }
  if (FT == CELL_U || FT == CELL_C) {
    __syncthreads();
    if (FT == CELL_U) {
      // a group's U values are its own: the sums are complete
      for (int i = tid; i < nacc; i += CELL_NT)
        if (u0 + i / NS < a.n_out) a.out[(int64_t)(u0 + i / NS) * a.out_stride + (i % NS)] = acc[i];
    } else {
      double *o = a.out + (int64_t)g * nacc;
      for (int i = tid; i < nacc; i += CELL_NT) o[i] = acc[i];
    }
  }
}

template <int FT, int NS, bool HASP, bool ITEM32>
static void launch_pass_t(hipStream_t s, int G, size_t lds, const CellPassArgs &a) {
  static DeviceOnce raised;
  if (raised.need()) {
    MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_cell_pass<FT, NS, HASP, ITEM32>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024));
    raised.mark();
  }
  hipLaunchKernelGGL((k_cell_pass<FT, NS, HASP, ITEM32>), dim3(G), dim3(CELL_NT), lds, s, a);
}
template <int FT, int NS>
static void launch_pass_f(hipStream_t s, int G, size_t lds, const CellPassArgs &a, bool hasp, bool item32) {
  if (hasp) {
    if (item32) launch_pass_t<FT, NS, true, true>(s, G, lds, a); else launch_pass_t<FT, NS, true, false>(s, G, lds, a);
  } else {
    if (item32) launch_pass_t<FT, NS, false, true>(s, G, lds, a); else launch_pass_t<FT, NS, false, false>(s, G, lds, a);
  }
}

void cell_pass(hipStream_t s, Timing &tm, CellPlan &cp, int P, int F, bool sw, double *out_u, int out_stride) {
  if (P < 0 && F < 0) return;
  CellPassArgs a;
  std::memset(&a, 0, sizeof(a));
  int off[11];
  const size_t lds = cp.lds_bytes(P, F, sw, off);
  if (lds > CELL_LDS_BYTES) throw Error(MFM_ERR_RUNTIME, "internal: cell pass does not fit the LDS");
  a.ix = cp.ix.p;
  a.item = cp.item.p;
  a.e = cp.e.p;
  a.chunk0 = cp.chunk0.p;
  a.grp_u0 = cp.grp_u0.p;
  a.grp_steps = cp.grp_steps.p;
  a.packI = cp.packI.p;
  a.DP = cp.DP.p;
  a.n_streams = (int)cp.streams.size();
  a.sP = P >= 0 ? cp.fields[P].stream : -1;
  a.sF = F >= 0 ? cp.fields[F].stream : -1;
  for (int si = 0; si < a.n_streams; si++) {
    a.QA[si] = cp.QA[si].p;
    a.QS[si] = cp.QS[si].p;
    a.type[si] = cp.streams[si].type;
    a.slot[si] = cp.streams[si].slot;
    a.card[si] = (int)cp.streams[si].card;
    a.pair[si] = (P >= 0 && F >= 0 && (sw || si == a.sP || si == a.sF)) ? 1 : (F < 0 ? 2 : 0);
    a.ldsA[si] = off[si];
    a.ldsS[si] = off[4 + si];
  }
  a.ldsDP = off[8];
  a.ldsAcc = off[9];
  a.ldsTurn = off[10];
  a.cardI = cp.sI >= 0 ? (int)cp.streams[cp.sI].card : 0;
  const int ft = F >= 0 ? cp.streams[a.sF].type : -1;
  const int ns = F >= 0 ? (cp.fields[F].kind == 0 ? 2 : 4) : 0;
  a.ns = ns;
  if (ft == CELL_U) {
    a.out = out_u;
    a.out_stride = out_stride;
    a.n_out = (int)cp.fields[F].n;
  } else if (ft == CELL_I) {
    a.out = ns == 2 ? cp.cells2.p : cp.cells4.p;
  } else if (ft == CELL_C) {
    a.out = cp.cpart.p;
  }
  // algorithmic bytes: e read (+ written when a field is applied), the index record, the int32 item
  const double bytes = (double)cp.N * (8.0 + (P >= 0 ? 8.0 : 0.0) + 8.0 + (cp.item32 ? 4.0 : 0.0)) +
                       (ft == CELL_I ? (double)cp.G * a.cardI * 8.0 * ns : 0.0);
  TimedLaunch t(tm, s, KC_CELL_PASS, bytes);
  const bool hasp = P >= 0;
  if (ft < 0)
    launch_pass_f<-1, 0>(s, cp.G, lds, a, hasp, cp.item32);
  else if (ft == CELL_U && ns == 2)
    launch_pass_f<CELL_U, 2>(s, cp.G, lds, a, hasp, cp.item32);
  else if (ft == CELL_U)
    launch_pass_f<CELL_U, 4>(s, cp.G, lds, a, hasp, cp.item32);
  else if (ft == CELL_I && ns == 2)
    launch_pass_f<CELL_I, 2>(s, cp.G, lds, a, hasp, cp.item32);
  else if (ft == CELL_I)
    launch_pass_f<CELL_I, 4>(s, cp.G, lds, a, hasp, cp.item32);
  else if (ns == 2)
    launch_pass_f<CELL_C, 2>(s, cp.G, lds, a, hasp, cp.item32);
  else
    launch_pass_f<CELL_C, 4>(s, cp.G, lds, a, hasp, cp.item32);
  MFM_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------------------
// main field: sums over the groups (group order), the draw of FMTrainer.hpp:357-369 with x = 1: S2 = sum q_other^2,
// S1 = -sum e q_other
template <int SRC /* 0: direct [n][2], 1: partials [G][card][2] */>
__global__ __launch_bounds__(256) void k_cell_draw(const double *__restrict__ src, int G, int64_t card, int n, double *__restrict__ Vf,
                                                   const double *__restrict__ zf, const int32_t *__restrict__ group,
                                                   const double *__restrict__ lam, const double *__restrict__ mu, double alpha,
                                                   int64_t base, double2 *__restrict__ DP) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double S2 = 0.0, Seh = 0.0;
  if (SRC == 0) {
    S2 = src[(int64_t)i * 2];
    Seh = src[(int64_t)i * 2 + 1];
  } else {
    for (int g = 0; g < G; g++) {
      const double2 p = *(const double2 *)(src + ((int64_t)g * card + i) * 2);
      S2 += p.x;
      Seh += p.y;
    }
  }
  const int64_t j = base + i;
  const double old = Vf[j];
  const int gi = group[j];
  const double l = lam[gi], m = mu[gi];
  double lin = (-Seh) + S2 * old;  // :358
  double sq = S2 * alpha;          // :360
  lin = lin * alpha;               // :361
  sq += l;                         // :363
  lin += l * m;                    // :364-365
  const double fresh = sample_normal_z(sq, lin, zf[j]);
  Vf[j] = fresh;
  DP[i] = make_double2(fresh - old, 0.0);
}

void cell_draw_main(hipStream_t s, Timing &tm, CellPlan &cp, int F, double *Vf, const double *zf, const int32_t *group, const double *lam,
                    const double *mu, double alpha) {
  const CellField &f = cp.fields[F];
  const CellStream &st = cp.streams[f.stream];
  TimedLaunch t(tm, s, KC_CELL_SMALL, 0.0);
  const int n = (int)f.n;
  if (st.type == CELL_U)
    hipLaunchKernelGGL((k_cell_draw<0>), dim3(cdiv_c(n, 256)), dim3(256), 0, s, cp.stat.p, cp.G, st.card, n, Vf, zf, group, lam, mu, alpha,
                       f.base, cp.DP.p);
  else
    hipLaunchKernelGGL((k_cell_draw<1>), dim3(cdiv_c(n, 256)), dim3(256), 0, s, st.type == CELL_I ? cp.cells2.p : cp.cpart.p, cp.G, st.card,
                       n, Vf, zf, group, lam, mu, alpha, f.base, cp.DP.p);
  MFM_HIP_CHECK(hipGetLastError());
}

// block on I / C: rec[i].{c, c_S, e, e_q} (words 2..5) = sum over the groups, group order (FMTrainer.hpp:401-407)
__global__ __launch_bounds__(256) void k_cell_block_stats(const double *__restrict__ src, int G, int64_t card, int n, double *__restrict__ rec) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (int g = 0; g < G; g++) {
    const double2 *p = (const double2 *)(src + ((int64_t)g * card + i) * 4);
    const double2 a = p[0], b = p[1];
    s0 += a.x;
    s1 += a.y;
    s2 += b.x;
    s3 += b.y;
  }
  double2 *r = (double2 *)rec + (int64_t)i * 4;
  r[1] = make_double2(s0, s1);
  r[2] = make_double2(s2, s3);
}
void cell_block_stats(hipStream_t s, Timing &tm, CellPlan &cp, int F, double *rec) {
  const CellField &f = cp.fields[F];
  const CellStream &st = cp.streams[f.stream];
  if (st.type == CELL_U) return;
  TimedLaunch t(tm, s, KC_CELL_SMALL, 0.0);
  hipLaunchKernelGGL(k_cell_block_stats, dim3(cdiv_c(f.n, 256)), dim3(256), 0, s, st.type == CELL_I ? cp.cells4.p : cp.cpart.p, cp.G, st.card,
                     (int)f.n, rec);
  MFM_HIP_CHECK(hipGetLastError());
}

// block after its feature sweep: the un-sync (:408-415) and re-sync (:473-480) of a row together are
// e += q_other (q' - q) + (q'^2 - q^2) / 2 - (q_S' - q_S) / 2
__global__ __launch_bounds__(256) void k_cell_block_delta(const double *__restrict__ rec, const double2 *__restrict__ saved, int n,
                                                          double2 *__restrict__ DP) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double2 o = saved[i];
  const double2 c = ((const double2 *)rec)[(int64_t)i * 4];
  DP[i] = make_double2(c.x - o.x, (0.5 * c.x * c.x - 0.5 * c.y) - (0.5 * o.x * o.x - 0.5 * o.y));
}
void cell_block_delta(hipStream_t s, Timing &tm, CellPlan &cp, int F, const double *rec, const double2 *saved) {
  const CellField &f = cp.fields[F];
  TimedLaunch t(tm, s, KC_CELL_SMALL, 0.0);
  hipLaunchKernelGGL(k_cell_block_delta, dim3(cdiv_c(f.n, 256)), dim3(256), 0, s, rec, saved, (int)f.n, cp.DP.p);
  MFM_HIP_CHECK(hipGetLastError());
}

}  // namespace mfm
