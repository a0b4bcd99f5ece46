// mfm_cell.hip -- the cell path of update_V (see mfm_cell.hpp): planner, pass kernel, table / draw / reduce kernels.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <functional>
#include <mutex>
#include <thread>

#include <hipcub/hipcub.hpp>

#include "mfm_cell.hpp"
#include "mfm_wave.hpp"

namespace mfm {

static inline int cdiv_c(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------------------------------------
// LDS layout of a pass (offsets in doubles): per LDS stream its value table(s), the pending field's (d1, d2), the accumulators
// of the statistics field, the turn word.   off[0..3] = A, off[4..7] = S, off[8] = DP, off[9] = accumulators, off[10] = turn
size_t CellPlan::lds_bytes(int P, int F, bool sw, int *off, bool linear) const {
  size_t o = 0;
  int tmp[11];
  if (!off) off = tmp;
  const int sP = P >= 0 ? fields[P].stream : -1, sF = F >= 0 ? fields[F].stream : -1;
  for (int s = 0; s < CELL_MAX_STREAMS; s++) off[s] = off[4 + s] = 0;
  for (size_t s = 0; s < streams.size(); s++) {
    if (streams[s].type == CELL_I || linear) continue;  // (the linear sweep needs no q tables)
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
    const size_t ns = linear ? 1 : (fields[F].kind == 0 ? 2 : 4);
    o = (o + 1) & ~(size_t)1;  // (16-byte accesses)
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

// streams and fields of the design (host decisions, no O(N) work of their own): main field p = the p-th entry of every row; a
// block joins the stream whose indices equal its map on every row (differs(b, candidate): the O(N) comparison, on the host or
// on the device), else it opens a stream. src: where every stream's indices come from.
struct CellStreamSrc {
  int main_p = -1;  // main field position, or
  int block = -1;   // the block whose map opened the stream
};
static bool cell_plan_streams(CellPlan &cp, int64_t W, const std::vector<int64_t> &base, const std::vector<int64_t> &Bs,
                              std::vector<CellStreamSrc> &src, const std::function<bool(size_t, const CellStreamSrc &)> &differs) {
  src.clear();
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
    CellStreamSrc h;
    h.main_p = (int)p;
    src.push_back(h);
  }
  for (size_t b = 0; b < Bs.size(); b++) {
    int found = -1;
    for (size_t si = 0; si < src.size() && found < 0; si++)
      if (!differs(b, src[si])) found = (int)si;
    if (found < 0) {
      if (src.size() >= (size_t)CELL_MAX_STREAMS) return cp.fail("more index streams than a row record holds");
      CellStreamSrc h;
      h.block = (int)b;
      src.push_back(h);
      cp.streams.push_back(CellStream());
      found = (int)src.size() - 1;
    }
    if (cp.fields.size() >= (size_t)CELL_MAX_FIELDS) return cp.fail("too many fields");
    CellField f;
    f.stream = found;
    f.kind = 1;
    f.n = Bs[b];
    f.base = (int64_t)b;
    cp.streams[found].card = std::max(cp.streams[found].card, Bs[b]);
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
  return true;
}

// do the groups' tables fit the LDS in every pass of the sweep? (gu0: first U value of every group, then the end)
static bool cell_plan_groups_fit(CellPlan &cp, const std::vector<int32_t> &gu0) {
  const int G = (int)gu0.size() - 1;
  int64_t um = 0;
  for (int g = 0; g < G; g++) um = std::max<int64_t>(um, gu0[g + 1] - gu0[g]);
  cp.G = G;
  cp.umax = um;
  if (um > 65535) return false;
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
  return worst <= CELL_LDS_BYTES;
}

// the plan's device buffers besides the row arrays
static void cell_plan_buffers(CellPlan &cp, hipStream_t s) {
  const int G = cp.G;
  const int64_t cardI = cp.sI >= 0 ? cp.streams[cp.sI].card : 0, cardU = cp.streams[0].card;
  cp.e.alloc((size_t)cp.Npad);
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
    cp.cells1.alloc_zero((size_t)G * cardI, s);
    cp.cells2.alloc_zero((size_t)G * cardI * 2, s);
    cp.cells4.alloc_zero((size_t)G * cardI * 4, s);
  }
  int64_t maxC = 0;
  for (auto &st : cp.streams)
    if (st.type == CELL_C) maxC = std::max(maxC, st.card);
  cp.cpart.alloc_zero((size_t)std::max<int64_t>(1, (int64_t)G * maxC * 4), s);
  cp.DP.alloc_zero((size_t)std::max<int64_t>(1, maxcard), s);
  cp.stat.alloc_zero((size_t)std::max<int64_t>(1, cardU * 2), s);
  cp.stat1.alloc_zero((size_t)std::max<int64_t>(1, maxcard), s);
  cp.dense.alloc_zero((size_t)std::max<int64_t>(1, maxcard) * 4, s);
  cp.cnt_ready = false;
  for (size_t f = 0; f < cp.fields.size(); f++)
    if (cp.fields[f].kind == 0) cp.cnt[f].alloc_zero((size_t)std::max<int64_t>(1, cp.fields[f].n), s);
  MFM_HIP_CHECK(hipStreamSynchronize(s));
}

bool cell_plan_build(CellPlan &cp, const HostCsr &X, const std::vector<CellBlockIn> &blocks, int n_cu, hipStream_t s,
                     int rank, int world, const std::function<void(std::vector<double> &)> &sum_ranks) {
  // rank / world / sum_ranks: row-sharded mode. Every rank plans its own rows, but all of them must see the same fields (column
  // ranges of the GLOBAL design), the same streams and the same verdict: the local findings are summed over the ranks (sum_ranks,
  // every rank's values in its own slot where a minimum / maximum is needed) before they are used. An empty shard takes part in
  // every sum.
  cp.ready = false;
  cp.streams.clear();
  cp.fields.clear();
  const int64_t N = X.rows;
  cp.N = N;
  const bool shared = (bool)sum_ranks;
  auto agree_bad = [&](int local) {  // > 0 on any rank -> the same on every rank
    if (!shared) return local;
    std::vector<double> v(1, local ? 1.0 : 0.0);
    sum_ranks(v);
    return v[0] > 0.0 ? std::max(local, 1) : 0;
  };
  if (N >= (int64_t)2147483647) return cp.fail("too many rows");
  if (!shared && N <= 0) return cp.fail("no rows");
  int64_t W = N > 0 ? X.ptr[1] - X.ptr[0] : 0;
  std::vector<int64_t> base;
  if (shared) {  // (every rank's rows have the same number of entries; an empty shard learns it here)
    if (rank < 0 || rank >= world) return cp.fail("row-sharded: this rank's place among the ranks is not known");
    std::vector<double> v((size_t)world, 0.0);
    v[rank] = (double)W;
    sum_ranks(v);
    int64_t Wg = 0;
    for (double x : v) Wg = std::max<int64_t>(Wg, (int64_t)x);
    for (double x : v)
      if (x != 0.0 && (int64_t)x != Wg) Wg = -1;
    if (N == 0 && Wg > 0) W = Wg;
    if (N > 0 && W != Wg) W = -2;
  }
  int bad0 = (W < 1 || W > CELL_MAX_STREAMS || X.nnz() != N * std::max<int64_t>(W, 0)) ? 1 : 0;
  if (agree_bad(bad0)) return cp.fail("the main table is not a row of one-hot fields");
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
  if (shared) {  // the fields' column ranges of the GLOBAL design: every rank's (lo, hi) in its own slot of a summed vector
    std::vector<double> v((size_t)world * 2 * W, 0.0);
    for (int64_t p = 0; p < W; p++) {
      v[((size_t)rank * W + p) * 2] = (double)lo[p];
      v[((size_t)rank * W + p) * 2 + 1] = (double)hi[p];
    }
    sum_ranks(v);
    for (int r = 0; r < world; r++)
      for (int64_t p = 0; p < W; p++) {
        lo[p] = std::min(lo[p], (int64_t)v[((size_t)r * W + p) * 2]);
        hi[p] = std::max(hi[p], (int64_t)v[((size_t)r * W + p) * 2 + 1]);
      }
  }
  for (int64_t p = 1; p < W && !bad; p++)
    if (lo[p] <= hi[p - 1]) bad = 3;
  base.assign((size_t)W + 1, 0);
  for (int64_t p = 1; p < W; p++) base[p] = lo[p];
  base[W] = X.cols;
  {
    const int b_all = agree_bad(bad.load());
    if (b_all) return cp.fail(bad == 2 ? "the rows are not sorted by the first field"
                                      : "the main table is not a row of unit-valued one-hot fields with disjoint column ranges");
  }
  // streams: main fields first, then every block whose map is not one of the streams already there
  struct HostStream {
    int main_p = -1;               // main field position, or
    const int32_t *map = nullptr;  // a block's map
  };
  std::vector<HostStream> hs;
  auto idx_of = [&](const HostStream &h, int64_t t) -> int64_t {
    return h.main_p >= 0 ? (int64_t)X.idx[t * W + h.main_p] - base[h.main_p] : h.map[t];
  };
  {
    std::vector<int64_t> Bs;
    for (auto &b : blocks) Bs.push_back(b.B);
    std::vector<CellStreamSrc> src;
    const bool ok = cell_plan_streams(cp, W, base, Bs, src, [&](size_t b, const CellStreamSrc &c) {
      HostStream h;
      h.main_p = c.main_p;
      h.map = c.block >= 0 ? blocks[(size_t)c.block].map : nullptr;
      std::atomic<int> diff(0);
      par_for(N, 1 << 20, [&](int64_t a, int64_t e) {
        for (int64_t t = a; t < e && !diff.load(std::memory_order_relaxed); t++)
          if (idx_of(h, t) != blocks[b].map[t]) diff = 1;
      });
      return agree_bad(diff.load()) != 0;  // (equal on EVERY rank's rows)
    });
    if (!ok) return false;
    for (auto &c : src) {
      HostStream h;
      h.main_p = c.main_p;
      h.map = c.block >= 0 ? blocks[(size_t)c.block].map : nullptr;
      hs.push_back(h);
    }
  }
  // groups of consecutive U values, rows balanced; more (smaller) groups until every pass of the sweep fits its LDS
  const int64_t cardU = cp.streams[0].card;
  std::vector<int64_t> grow;  // first row of every group, then N
  std::vector<int32_t> gu0;
  const HostStream &hU = hs[0];
  bool fits = false;
  if (N == 0) {  // (an empty shard: no groups, no launches; it still takes part in every collective)
    grow.assign(1, 0);
    gu0.assign(1, 0);
    cp.G = 0;
    cp.umax = 0;
    fits = true;
  }
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
    if (shared) {  // (a shard's groups cover its own U values only: the sums of the others come from their ranks; the dense
                   //  array the sums go through is zeroed first, so values without a group contribute zero)
      gu0[0] = (int32_t)idx_of(hU, 0);
      gu0[G] = (int32_t)idx_of(hU, N - 1) + 1;
    }
    fits = cell_plan_groups_fit(cp, gu0);
  }
  const int G = cp.G;
  const int64_t cardI = cp.sI >= 0 ? cp.streams[cp.sI].card : 0;
  {
    const int lf = !fits ? 1 : (cardI > 0 && (double)G * (double)cardI * 56.0 > 16e9) ? 2 : 0;
    if (agree_bad(lf))
      return cp.fail(lf == 2 ? "the (group, item) partials would not fit"
                             : "a group's tables do not fit the LDS (a first-field value with too many rows, or too many values per group)");
  }
  // rows in cell order: inside a group by the I index (stable), the group cut into CELL_NW wave chunks between two I values.
  // Physical layout: what a workgroup touches in one step is ONE contiguous block -- position of row r of wave w's chunk =
  // group base + (r / 256) * 4096 + w * 256 + r % 256 (chunks padded to whole steps; pad rows have perm = -1) -- so that HBM
  // sees 32-KiB bursts per array and step, not 4096 independent 2-KiB streams.
  constexpr int64_t WROWS = 64 * CELL_R, SROWS = WROWS * CELL_NW;  // rows of a wave / of the workgroup per step
  std::vector<int32_t> sorted((size_t)N), chunk0((size_t)G * (CELL_NW + 1), 0), steps((size_t)G, 0);
  auto run_groups = [&](auto fn) {
    std::atomic<int> next_g(0);
    auto work = [&]() {
      for (;;) {
        const int g = next_g.fetch_add(1);
        if (g >= G) break;
        fn(g);
      }
    };
    const int hw = (int)std::thread::hardware_concurrency();
    const int T = std::max(1, std::min({hw > 0 ? hw : 1, 16, G}));
    std::vector<std::thread> pool;
    for (int t = 1; t < T; t++) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
  };
  run_groups([&](int g) {
    const int64_t R0 = grow[g], R1 = grow[g + 1], L = R1 - R0;
    if (cp.sI >= 0) {
      const HostStream &hI = hs[cp.sI];
      std::vector<int32_t> key((size_t)L), cnt((size_t)cardI + 1, 0);
      for (int64_t r = 0; r < L; r++) {
        key[r] = (int32_t)idx_of(hI, R0 + r);
        cnt[key[r] + 1]++;
      }
      for (int64_t i = 0; i < cardI; i++) cnt[i + 1] += cnt[i];
      for (int64_t r = 0; r < L; r++) sorted[R0 + cnt[key[r]]++] = (int32_t)(R0 + r);
    } else {
      for (int64_t r = 0; r < L; r++) sorted[R0 + r] = (int32_t)(R0 + r);
    }
    int64_t longest = 0;
    int32_t *c0 = &chunk0[(size_t)g * (CELL_NW + 1)];
    c0[0] = (int32_t)R0;
    for (int j = 1; j <= CELL_NW; j++) {
      int64_t r = j == CELL_NW ? R1 : R0 + L * j / CELL_NW;
      if (cp.sI >= 0 && j < CELL_NW) {
        const HostStream &hI = hs[cp.sI];
        while (r < R1 && r > R0 && idx_of(hI, sorted[r]) == idx_of(hI, sorted[r - 1])) r++;
      }
      r = std::max<int64_t>(r, c0[j - 1]);
      c0[j] = (int32_t)r;
      longest = std::max<int64_t>(longest, r - c0[j - 1]);
    }
    steps[g] = (int32_t)((longest + WROWS - 1) / WROWS);
  });
  std::vector<int32_t> gbase((size_t)G + 1, 0), clen((size_t)G * CELL_NW, 0);
  int64_t npad = 0;
  for (int g = 0; g < G; g++) {
    gbase[g] = (int32_t)npad;
    npad += (int64_t)steps[g] * SROWS;
  }
  if (agree_bad(npad >= (int64_t)2147483647 ? 1 : 0)) return cp.fail("padded row count exceeds 2^31");
  gbase[G] = (int32_t)npad;
  cp.Npad = npad;
  std::vector<int32_t> perm((size_t)npad, -1), item;
  std::vector<uint2> ix((size_t)npad, make_uint2(0, 0));
  if (cp.item32) item.assign((size_t)npad, 0);
  run_groups([&](int g) {
    const int32_t *c0 = &chunk0[(size_t)g * (CELL_NW + 1)];
    for (int w = 0; w < CELL_NW; w++) {
      const int64_t len = c0[w + 1] - c0[w];
      clen[(size_t)g * CELL_NW + w] = (int32_t)len;
      for (int64_t r = 0; r < len; r++) {
        const int64_t t = sorted[c0[w] + r];
        const int64_t p = (int64_t)gbase[g] + (r / WROWS) * SROWS + (int64_t)w * WROWS + (r % WROWS);
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
        perm[p] = (int32_t)t;
      }
    }
  });
  sorted = std::vector<int32_t>();
  // device
  cp.ix.upload(ix);
  cp.item.upload(item);
  cp.perm.upload(perm);
  cp.chunk_len.upload(clen);
  cp.grp_base.upload(gbase);
  cp.grp_u0.upload(gu0);
  cp.grp_steps.upload(steps);
  cell_plan_buffers(cp, s);
  cp.ready = true;
  return true;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same plan built ON THE DEVICE from the device-resident CSR and the blocks' maps (one GPU; the row-sharded planner above
// needs its agreements over the ranks and stays on the host). O(N) work: the fields' column ranges, which block shares which
// stream, the group cuts along the first field, the rows of every group sorted by the I index (ONE stable radix sort of
// (group, I index) keys), the wave chunks cut between two I values, the row records. The host takes the decisions that are
// O(fields) / O(groups) on a few KB copied back -- with the code the host planner uses (cell_plan_streams,
// cell_plan_groups_fit). tests (MFM_PLAN_CHECK): both planners, every array compared (cell_plan_compare).
namespace cpd {

constexpr int TB = 256;
constexpr int MAXB = CELL_MAX_FIELDS;

struct Src {  // where the index streams of a row come from
  int W;                           // entries per row of the main table
  int nb;                          // blocks
  int32_t base[CELL_MAX_STREAMS];  // first column of main field p
  const int32_t *map[MAXB];
};

// per main field the smallest / largest column; red[2 W] counts rows that break the order of the first field
__global__ void k_fields(const int32_t *__restrict__ colidx, int64_t N, int W, int32_t *__restrict__ red) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int p = 0; p < W; p++) {
    int32_t lo = 0x7fffffff, hi = -1;
    if (t < N) lo = hi = colidx[t * W + p];
    for (int off = 32; off > 0; off >>= 1) {
      lo = min(lo, __shfl_xor(lo, off));
      hi = max(hi, __shfl_xor(hi, off));
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(&red[2 * p], lo);
      atomicMax(&red[2 * p + 1], hi);
    }
  }
  if (t > 0 && t < N && colidx[t * W] < colidx[(t - 1) * W]) atomicAdd(&red[2 * W], 1);
}

// diff[b * (W + nb) + c] != 0: block b's map differs on some row from candidate c (c < W: main field c, else block c - W)
__global__ void k_stream_diff(const int32_t *__restrict__ colidx, Src sr, int64_t N, int32_t *__restrict__ diff) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N) return;
  const int nc = sr.W + sr.nb;
  int32_t cand[CELL_MAX_STREAMS + MAXB];
  for (int p = 0; p < sr.W; p++) cand[p] = colidx[t * sr.W + p] - sr.base[p];
  for (int b = 0; b < sr.nb; b++) cand[sr.W + b] = sr.map[b][t];
  for (int b = 0; b < sr.nb; b++)
    for (int c = 0; c < sr.W + b; c++)
      if (cand[c] != cand[sr.W + b] && !diff[b * nc + c]) diff[b * nc + c] = 1;  // (benign race: every writer stores 1)
}

// group cut g: row g * target, moved up to the next change of the first field; its U value there
__global__ void k_grow(const int32_t *__restrict__ colidx, int W, int64_t N, int64_t target, int G0, int32_t cardU,
                       int32_t *__restrict__ out_r, int32_t *__restrict__ out_u) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G0) return;
  int64_t r = min(N, (int64_t)g * target);
  while (r < N && r > 0 && colidx[r * W] == colidx[(r - 1) * W]) r++;
  out_r[g] = (int32_t)r;
  out_u[g] = r < N ? colidx[r * W] : cardU;
}

__device__ __forceinline__ int find_group(const int32_t *cut, int G, int32_t x) {  // cut[g] <= x < cut[g + 1]
  int lo = 0, hi = G;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (cut[mid] <= x)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

// sort keys (group, I index) of the rows; src_p >= 0: the I stream is main field src_p, else the map
__global__ void k_keys(const int32_t *__restrict__ colidx, int W, int src_p, int32_t base, const int32_t *__restrict__ map,
                       const int32_t *__restrict__ grow, int G, int ibits, int64_t N, uint32_t *__restrict__ key, int32_t *__restrict__ val) {
  extern __shared__ int32_t cut[];
  for (int i = threadIdx.x; i <= G; i += blockDim.x) cut[i] = grow[i];
  __syncthreads();
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N) return;
  const int g = find_group(cut, G, (int32_t)t);
  const uint32_t iv = map ? (uint32_t)map[t] : src_p >= 0 ? (uint32_t)(colidx[t * W + src_p] - base) : 0u;
  key[t] = ((uint32_t)g << ibits) | iv;
  val[t] = (int32_t)t;
}

// chunk cut j of group g (one wave each): row R0 + L j / NW of the sorted order, moved up to the next change of the I index
__global__ void k_chunk_cuts(const uint32_t *__restrict__ key, const int32_t *__restrict__ grow, int G, bool has_i, int32_t *__restrict__ c0) {
  const int wv = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (wv >= G * (CELL_NW + 1)) return;
  const int g = wv / (CELL_NW + 1), j = wv % (CELL_NW + 1);
  const int64_t R0 = grow[g], R1 = grow[g + 1], L = R1 - R0;
  int64_t r = j == 0 ? R0 : j == CELL_NW ? R1 : R0 + L * j / CELL_NW;
  if (has_i && j > 0 && j < CELL_NW) {
    for (;;) {
      const int64_t q = r + lane;
      const bool same = q < R1 && q > R0 && key[q] == key[q - 1];
      const unsigned long long m = __ballot(same);
      if (~m) {
        r += __ffsll((long long)~m) - 1;
        break;
      }
      r += 64;
    }
  }
  if (lane == 0) c0[wv] = (int32_t)r;
}

// per group: cuts made monotone, chunk lengths, steps of the longest chunk
__global__ void k_chunk_fin(int32_t *__restrict__ c0, int G, int wrows, int32_t *__restrict__ clen, int32_t *__restrict__ steps) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  int32_t *c = c0 + (int64_t)g * (CELL_NW + 1);
  int32_t longest = 0;
  for (int j = 1; j <= CELL_NW; j++) {
    c[j] = max(c[j], c[j - 1]);
    clen[g * CELL_NW + j - 1] = c[j] - c[j - 1];
    longest = max(longest, c[j] - c[j - 1]);
  }
  steps[g] = (longest + wrows - 1) / wrows;
}

struct RecSrc {  // how a row's record is put together
  int ns;                            // streams
  int slot[CELL_MAX_STREAMS];        // u16 slot of the record (-1: the int32 item array)
  int main_p[CELL_MAX_STREAMS];      // main field position, or -1: map
  int32_t base[CELL_MAX_STREAMS];
  const int32_t *map[CELL_MAX_STREAMS];
};

// the rows in cell order: position of sorted row q, its record, perm
__global__ void k_records(const uint32_t *__restrict__ key, const int32_t *__restrict__ val, const int32_t *__restrict__ colidx, int W, RecSrc rs,
                          const int32_t *__restrict__ grow, const int32_t *__restrict__ c0, const int32_t *__restrict__ gbase,
                          const int32_t *__restrict__ gu0, int G, int ibits, bool has_i, int wrows, int64_t N, uint2 *__restrict__ ix,
                          int32_t *__restrict__ item, int32_t *__restrict__ perm) {
  extern __shared__ int32_t cut[];
  if (!has_i) {
    for (int i = threadIdx.x; i <= G; i += blockDim.x) cut[i] = grow[i];
    __syncthreads();
  }
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= N) return;
  const int g = has_i ? (int)(key[q] >> ibits) : find_group(cut, G, (int32_t)q);
  const int32_t *c = c0 + (int64_t)g * (CELL_NW + 1);
  int w = 0;
  {  // the LAST chunk that starts at or before q (empty chunks share their start with the next one)
    int lo = 0, hi = CELL_NW;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (c[mid] <= (int32_t)q)
        lo = mid;
      else
        hi = mid;
    }
    w = lo;
  }
  const int64_t r = q - c[w];
  const int64_t p = (int64_t)gbase[g] + (r / wrows) * ((int64_t)wrows * CELL_NW) + (int64_t)w * wrows + (r % wrows);
  const int32_t t = has_i ? val[q] : (int32_t)q;
  uint32_t sl[4] = {0, 0, 0, 0};
  for (int si = 0; si < rs.ns; si++) {
    const int32_t v = rs.main_p[si] >= 0 ? colidx[(int64_t)t * W + rs.main_p[si]] - rs.base[si] : rs.map[si][t];
    if (si == 0)
      sl[0] = (uint32_t)(v - gu0[g]);
    else if (rs.slot[si] >= 0)
      sl[rs.slot[si]] = (uint32_t)v;
    else
      item[p] = v;
  }
  ix[p] = make_uint2(sl[0] | (sl[1] << 16), sl[2] | (sl[3] << 16));
  perm[p] = t;
}

static int bits_for(int64_t n) {
  int b = 1;
  while (((int64_t)1 << b) < n) b++;
  return b;
}

template <class T>
static std::vector<T> download(const T *p, size_t n, hipStream_t s) {
  std::vector<T> h(n);
  if (n) MFM_HIP_CHECK(hipMemcpyAsync(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost, s));
  MFM_HIP_CHECK(hipStreamSynchronize(s));
  return h;
}

}  // namespace cpd

bool cell_plan_build_device(CellPlan &cp, const DevSparse &X, const std::vector<CellBlockDev> &blocks, int n_cu, hipStream_t s, int rank,
                            int world, const std::function<void(std::vector<double> &)> &sum_ranks) {
  // rank / world / sum_ranks: row-sharded, as in cell_plan_build -- every rank plans its own rows on its own device; what must be
  // the same everywhere (entries per row, the fields' column ranges of the GLOBAL design, which block shares which stream, the
  // verdicts) is summed over the ranks. Every rank makes the SAME sequence of sums whatever its rows look like (an empty shard too).
  using namespace cpd;
  cp.ready = false;
  cp.streams.clear();
  cp.fields.clear();
  const int64_t N = X.rows;
  cp.N = N;
  const bool shared = (bool)sum_ranks;
  auto agree_bad = [&](int local) {  // > 0 on any rank -> the same on every rank
    if (!shared) return local;
    std::vector<double> v(1, local ? 1.0 : 0.0);
    sum_ranks(v);
    return v[0] > 0.0 ? std::max(local, 1) : 0;
  };
  if (N >= (int64_t)2147483647) return cp.fail("too many rows");
  if (!shared && N <= 0) return cp.fail("no rows");
  if (blocks.size() > (size_t)MAXB) return cp.fail("too many fields");
  int64_t W = N > 0 ? X.ell_width : 0;
  if (shared) {  // (every rank's rows have the same number of entries; an empty shard learns it here)
    if (rank < 0 || rank >= world) return cp.fail("row-sharded: this rank's place among the ranks is not known");
    std::vector<double> v((size_t)world, 0.0);
    v[rank] = (double)std::max<int64_t>(W, 0);
    sum_ranks(v);
    int64_t Wg = 0;
    for (double x : v) Wg = std::max<int64_t>(Wg, (int64_t)x);
    for (double x : v)
      if (x != 0.0 && (int64_t)x != Wg) Wg = -1;
    if (N == 0 && Wg > 0) W = Wg;
    if (N > 0 && W != Wg) W = -2;
  }
  const int bad0 = (W < 1 || W > CELL_MAX_STREAMS || (N > 0 && (!X.unit || X.nnz != N * W))) ? 1 : 0;
  if (agree_bad(bad0)) return cp.fail("the main table is not a row of one-hot fields");
  auto grid = [](int64_t n) { return dim3((unsigned)((n + TB - 1) / TB)); };
  // fields: column ranges of the W entries of a row, order of the first
  std::vector<int64_t> lo((size_t)W, (int64_t)1 << 60), hi((size_t)W, -1);
  int bad = 0;
  if (N > 0) {
    DevBuf<int32_t> red;
    std::vector<int32_t> r0((size_t)2 * W + 1, 0);
    for (int64_t p = 0; p < W; p++) {
      r0[2 * p] = 0x7fffffff;
      r0[2 * p + 1] = -1;
    }
    red.upload(r0);
    hipLaunchKernelGGL(k_fields, grid(N), dim3(TB), 0, s, X.colidx.p, N, (int)W, red.p);
    const std::vector<int32_t> h_red = download(red.p, (size_t)2 * W + 1, s);
    if (h_red[2 * W]) bad = 2;
    for (int64_t p = 0; p < W; p++) {
      lo[p] = h_red[2 * p];
      hi[p] = h_red[2 * p + 1];
    }
  }
  if (shared) {  // the fields' column ranges of the GLOBAL design: every rank's (lo, hi) in its own slot of a summed vector
    std::vector<double> v((size_t)world * 2 * W, 0.0);
    for (int r = 0; r < world; r++)
      for (int64_t p = 0; p < W; p++) {  // (neutral elements in the slots of the others; an empty shard's own slot too)
        v[((size_t)r * W + p) * 2] = 0.0;
        v[((size_t)r * W + p) * 2 + 1] = 0.0;
      }
    for (int64_t p = 0; p < W; p++) {
      v[((size_t)rank * W + p) * 2] = (double)lo[p];
      v[((size_t)rank * W + p) * 2 + 1] = (double)hi[p];
    }
    sum_ranks(v);
    for (int r = 0; r < world; r++)
      for (int64_t p = 0; p < W; p++) {
        lo[p] = std::min(lo[p], (int64_t)v[((size_t)r * W + p) * 2]);
        hi[p] = std::max(hi[p], (int64_t)v[((size_t)r * W + p) * 2 + 1]);
      }
  }
  for (int64_t p = 1; p < W && !bad; p++)
    if (lo[p] <= hi[p - 1]) bad = 3;
  {
    const int b_all = agree_bad(bad);
    if (b_all) return cp.fail(bad == 2 ? "the rows are not sorted by the first field"
                                      : "the main table is not a row of unit-valued one-hot fields with disjoint column ranges");
  }
  std::vector<int64_t> base((size_t)W + 1, 0);
  for (int64_t p = 1; p < W; p++) base[p] = lo[p];
  base[W] = X.cols;
  // streams: which block's map equals which stream on every row (all pairs in one pass; row-sharded: the matrix summed over the
  // ranks -- a pair differs if it differs on any rank's rows)
  Src sr;
  sr.W = (int)W;
  sr.nb = (int)blocks.size();
  for (int64_t p = 0; p < W; p++) sr.base[p] = (int32_t)base[p];
  for (size_t b = 0; b < blocks.size(); b++) sr.map[b] = blocks[b].map;
  const int nc = sr.W + sr.nb;
  std::vector<int32_t> h_diff((size_t)sr.nb * nc, 0);
  if (sr.nb) {
    if (N > 0) {
      DevBuf<int32_t> diff;
      diff.alloc_zero((size_t)sr.nb * nc, s);
      hipLaunchKernelGGL(k_stream_diff, grid(N), dim3(TB), 0, s, X.colidx.p, sr, N, diff.p);
      h_diff = download(diff.p, (size_t)sr.nb * nc, s);
    }
    if (shared) {
      std::vector<double> v(h_diff.begin(), h_diff.end());
      sum_ranks(v);
      for (size_t i = 0; i < v.size(); i++) h_diff[i] = v[i] > 0.0 ? 1 : 0;
    }
  }
  std::vector<CellStreamSrc> src;
  {
    std::vector<int64_t> Bs;
    for (auto &b : blocks) Bs.push_back(b.B);
    if (!cell_plan_streams(cp, W, base, Bs, src, [&](size_t b, const CellStreamSrc &c) {
          return h_diff[b * nc + (c.main_p >= 0 ? c.main_p : sr.W + c.block)] != 0;
        }))
      return false;  // (decided from agreed inputs: the same on every rank)
  }
  // groups of consecutive U values, rows balanced; more (smaller) groups until every pass of the sweep fits its LDS
  const int64_t cardU = cp.streams[0].card;
  std::vector<int32_t> h_grow, gu0;
  bool fits = false;
  if (N == 0) {  // (an empty shard: no groups, no launches; it still takes part in every sum)
    h_grow.assign(1, 0);
    gu0.assign(1, 0);
    cp.G = 0;
    cp.umax = 0;
    fits = true;
  } else {
    DevBuf<int32_t> out_r, out_u;
    const int64_t Gmax = (int64_t)std::max(1, n_cu) * 16;
    out_r.alloc((size_t)Gmax);
    out_u.alloc((size_t)Gmax);
    for (int mult = 1; mult <= 16 && !fits; mult *= 2) {
      const int64_t G0 = (int64_t)std::max(1, n_cu) * mult;
      const int64_t target = (N + G0 - 1) / G0;
      hipLaunchKernelGGL(k_grow, grid(G0), dim3(TB), 0, s, X.colidx.p, (int)W, N, target, (int)G0, (int32_t)cardU, out_r.p, out_u.p);
      const std::vector<int32_t> hr = download(out_r.p, (size_t)G0, s), hu = download(out_u.p, (size_t)G0, s);
      h_grow.assign(1, 0);
      gu0.assign(1, 0);
      for (int64_t g = 1; g < G0; g++)
        if (hr[g] > h_grow.back() && hr[g] < N) {
          h_grow.push_back(hr[g]);
          gu0.push_back(hu[g]);
        }
      h_grow.push_back((int32_t)N);
      gu0.push_back((int32_t)cardU);
      if (shared) {  // (a shard's groups cover its own U values only: the sums of the others come from their ranks)
        int32_t u_first = 0, u_last = 0;
        MFM_HIP_CHECK(hipMemcpy(&u_first, X.colidx.p, sizeof(int32_t), hipMemcpyDeviceToHost));
        MFM_HIP_CHECK(hipMemcpy(&u_last, X.colidx.p + (N - 1) * W, sizeof(int32_t), hipMemcpyDeviceToHost));
        gu0.front() = u_first;
        gu0.back() = u_last + 1;
      }
      fits = cell_plan_groups_fit(cp, gu0);
    }
  }
  const int G = cp.G;
  const int64_t cardI = cp.sI >= 0 ? cp.streams[cp.sI].card : 0;
  {
    const int lf = !fits ? 1 : (cardI > 0 && (double)G * (double)cardI * 56.0 > 16e9) ? 2 : 0;
    if (agree_bad(lf))
      return cp.fail(lf == 2 ? "the (group, item) partials would not fit"
                             : "a group's tables do not fit the LDS (a first-field value with too many rows, or too many values per group)");
  }
  const bool has_i = cp.sI >= 0;
  // (the key width is judged on the most groups any rank may form, so that every rank takes the same planner)
  const int ibits = has_i ? bits_for(cardI) : 0, gbits = bits_for(shared ? std::max<int64_t>(G, (int64_t)std::max(1, n_cu) * 16) : G);
  if (ibits + gbits > 32) return cp.fail("device planner: (group, item) sort key wider than 32 bits");
  constexpr int WROWS = 64 * CELL_R;
  DevBuf<int32_t> grow, d_gu0, c0, steps;
  grow.upload(h_grow);
  d_gu0.upload(gu0);
  // the rows of every group by I index: one stable radix sort of (group, I index)
  DevBuf<uint32_t> key, key2;
  DevBuf<int32_t> val, val2;
  DevBuf<char> tmp;
  if (has_i && N > 0) {
    key.alloc((size_t)N);
    key2.alloc((size_t)N);
    val.alloc((size_t)N);
    val2.alloc((size_t)N);
    const CellStreamSrc &ci = src[(size_t)cp.sI];
    hipLaunchKernelGGL(k_keys, grid(N), dim3(TB), (size_t)(G + 1) * sizeof(int32_t), s, X.colidx.p, (int)W, ci.main_p,
                       ci.main_p >= 0 ? (int32_t)base[ci.main_p] : 0, ci.block >= 0 ? blocks[(size_t)ci.block].map : nullptr, grow.p, G, ibits,
                       N, key.p, val.p);
    size_t bytes = 0;
    MFM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, key.p, key2.p, val.p, val2.p, (int)N, 0, ibits + gbits, s));
    tmp.alloc(bytes);
    MFM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp.p, bytes, key.p, key2.p, val.p, val2.p, (int)N, 0, ibits + gbits, s));
  }
  c0.alloc((size_t)G * (CELL_NW + 1));
  steps.alloc((size_t)G);
  cp.chunk_len.alloc((size_t)G * CELL_NW);
  if (G > 0) {
    hipLaunchKernelGGL(k_chunk_cuts, grid((int64_t)G * (CELL_NW + 1) * 64), dim3(TB), 0, s, key2.p, grow.p, G, has_i, c0.p);
    hipLaunchKernelGGL(k_chunk_fin, grid(G), dim3(TB), 0, s, c0.p, G, WROWS, cp.chunk_len.p, steps.p);
  }
  const std::vector<int32_t> h_steps = download(steps.p, (size_t)G, s);
  std::vector<int32_t> gbase((size_t)G + 1, 0);
  int64_t npad = 0;
  for (int g = 0; g < G; g++) {
    gbase[g] = (int32_t)npad;
    npad += (int64_t)h_steps[g] * WROWS * CELL_NW;
  }
  if (agree_bad(npad >= (int64_t)2147483647 ? 1 : 0)) return cp.fail("padded row count exceeds 2^31");
  gbase[G] = (int32_t)npad;
  cp.Npad = npad;
  cp.grp_base.upload(gbase);
  cp.grp_u0.upload(gu0);
  cp.grp_steps.upload(h_steps);
  cp.perm.alloc((size_t)npad);
  cp.ix.alloc((size_t)npad);
  MFM_HIP_CHECK(hipMemsetAsync(cp.perm.p, 0xff, (size_t)npad * sizeof(int32_t), s));
  MFM_HIP_CHECK(hipMemsetAsync(cp.ix.p, 0, (size_t)npad * sizeof(uint2), s));
  if (cp.item32) {
    cp.item.alloc((size_t)npad);
    MFM_HIP_CHECK(hipMemsetAsync(cp.item.p, 0, (size_t)npad * sizeof(int32_t), s));
  } else {
    cp.item = DevBuf<int32_t>();
  }
  RecSrc rs;
  rs.ns = (int)cp.streams.size();
  for (int si = 0; si < rs.ns; si++) {
    rs.slot[si] = cp.streams[si].slot;
    rs.main_p[si] = src[si].main_p;
    rs.base[si] = src[si].main_p >= 0 ? (int32_t)base[src[si].main_p] : 0;
    rs.map[si] = src[si].block >= 0 ? blocks[(size_t)src[si].block].map : nullptr;
  }
  if (N > 0)
    hipLaunchKernelGGL(k_records, grid(N), dim3(TB), has_i ? 0 : (size_t)(G + 1) * sizeof(int32_t), s, key2.p, val2.p, X.colidx.p, (int)W, rs,
                       grow.p, c0.p, cp.grp_base.p, cp.grp_u0.p, G, ibits, has_i, WROWS, N, cp.ix.p, cp.item.p, cp.perm.p);
  MFM_HIP_CHECK(hipGetLastError());
  cell_plan_buffers(cp, s);  // (synchronises: the sort buffers go out of scope)
  cp.ready = true;
  return true;
}

// tests (MFM_PLAN_CHECK): every array of two plans of the same design
std::string cell_plan_compare(const CellPlan &a, const CellPlan &b, hipStream_t s) {
  if (a.N != b.N || a.Npad != b.Npad || a.G != b.G || a.sU != b.sU || a.sI != b.sI || a.item32 != b.item32 || a.umax != b.umax ||
      a.streams.size() != b.streams.size() || a.fields.size() != b.fields.size())
    return "scalars";
  for (size_t i = 0; i < a.streams.size(); i++)
    if (a.streams[i].type != b.streams[i].type || a.streams[i].slot != b.streams[i].slot || a.streams[i].card != b.streams[i].card ||
        a.streams[i].fields != b.streams[i].fields)
      return "streams";
  for (size_t i = 0; i < a.fields.size(); i++)
    if (a.fields[i].stream != b.fields[i].stream || a.fields[i].kind != b.fields[i].kind || a.fields[i].n != b.fields[i].n ||
        a.fields[i].base != b.fields[i].base)
      return "fields";
  auto same = [&](const void *p, size_t np, const void *q, size_t nq, size_t elem) {
    if (np != nq) return false;
    std::vector<char> x(np * elem), y(nq * elem);
    if (np) {
      MFM_HIP_CHECK(hipMemcpyAsync(x.data(), p, np * elem, hipMemcpyDeviceToHost, s));
      MFM_HIP_CHECK(hipMemcpyAsync(y.data(), q, nq * elem, hipMemcpyDeviceToHost, s));
    }
    MFM_HIP_CHECK(hipStreamSynchronize(s));
    return x == y;
  };
#define MFM_CP_CMP(f) \
  if (!same(a.f.p, a.f.n, b.f.p, b.f.n, sizeof(*a.f.p))) return #f;
  MFM_CP_CMP(ix)
  MFM_CP_CMP(item)
  MFM_CP_CMP(perm)
  MFM_CP_CMP(chunk_len)
  MFM_CP_CMP(grp_base)
  MFM_CP_CMP(grp_u0)
  MFM_CP_CMP(grp_steps)
#undef MFM_CP_CMP
  return "";
}

// ---------------------------------------------------------------------------------------------------------------------------
// e between row order (eq[t].x) and cell order
__global__ void k_cell_pack(const double2 *__restrict__ eq, const int32_t *__restrict__ perm, int64_t N, double *__restrict__ e) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < N) {
    const int t = perm[p];
    e[p] = t >= 0 ? eq[t].x : 0.0;  // (pad rows of the step layout)
  }
}
__global__ void k_cell_unpack(const double *__restrict__ e, const int32_t *__restrict__ perm, int64_t N, double2 *__restrict__ eq) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < N) {
    const int t = perm[p];
    if (t >= 0) eq[t].x = e[p];
  }
}
void cell_pack_e(hipStream_t s, CellPlan &cp, const double2 *eq) {
  if (cp.Npad == 0) return;
  hipLaunchKernelGGL(k_cell_pack, dim3(cdiv_c(cp.Npad, 256)), dim3(256), 0, s, eq, cp.perm.p, cp.Npad, cp.e.p);
  MFM_HIP_CHECK(hipGetLastError());
}
void cell_unpack_e(hipStream_t s, CellPlan &cp, double2 *eq) {
  if (cp.Npad == 0) return;
  hipLaunchKernelGGL(k_cell_unpack, dim3(cdiv_c(cp.Npad, 256)), dim3(256), 0, s, cp.e.p, cp.perm.p, cp.Npad, eq);
  MFM_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------------------
// tables: dst[i * dst_stride] = sum_k src_k[i * stride_k]   (k ascending: fixed order)
struct CellPrepJob {
  double *dst;
  int dst_stride, n, nsrc;
  const double *src[4];
  int sstride[4], sn[4];  // (a field shorter than its stream's table contributes only where it has values)
  double coef[4];
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
    if (i < j.sn[k]) v += j.coef[k] * j.src[k][(int64_t)i * j.sstride[k]];
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
        j.coef[j.nsrc] = 1.0;
        j.nsrc++;
      }
    maxn = std::max(maxn, (int)n);
  };
  // a table is rebuilt only when its stream's sources changed (cell_touch) or another field is left out of it
  auto fresh = [&](int side, size_t si, int ex) {
    const int ex_here = (ex >= 0 && cp.fields[ex].stream == (int)si) ? ex : -1;
    if (cp.have_ver[side][si] == cp.ver[si] && cp.have_ex[side][si] == ex_here) return true;
    cp.have_ver[side][si] = cp.ver[si];
    cp.have_ex[side][si] = ex_here;
    return false;
  };
  for (size_t si = 0; si < cp.streams.size(); si++) {
    const CellStream &st = cp.streams[si];
    const bool needA = doA && !fresh(0, si, exA), needS = doS && !fresh(1, si, exS);
    if (st.type == CELL_I) {
      if (needA) add(cp.packI.p + 0, 4, st.card, (int)si, exA);
      if (needS) add(cp.packI.p + 1, 4, st.card, (int)si, exS);
      if (dp_to_I) {
        for (int k = 0; k < 2; k++) {
          add(cp.packI.p + 2 + k, 4, st.card, -1, -1);
          CellPrepJob &j = a.job[a.n_jobs - 1];
          j.nsrc = 1;
          j.src[0] = (const double *)cp.DP.p + k;
          j.sstride[0] = 2;
          j.sn[0] = (int)st.card;
          j.coef[0] = 1.0;
        }
      }
    } else {
      if (needA) add(cp.QA[si].p, 1, st.card, (int)si, exA);
      if (needS) add(cp.QS[si].p, 1, st.card, (int)si, exS);
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
  const int32_t *chunk_len, *grp_base, *grp_u0, *grp_steps;
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
  int linear;  // update_w: no q tables; statistics = sum e (NS = 1), the update is e += d1
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
  for (int s = 0; s < (a.linear ? 0 : a.n_streams); s++) {
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
  const int nval = FT == CELL_U ? nu : (FT == CELL_C ? a.card[a.sF] : 0);  // index values of the statistics field in this group
  const int nacc = nval * NS;
  if (FT == CELL_U || FT == CELL_C) {
    for (int i = tid; i < nacc; i += CELL_NT) lds[a.ldsAcc + i] = 0.0;
    if (tid == 0) *turn = 0;
  }
  __syncthreads();

  // row r of this wave's chunk lives at r0 + (r / 256) * 4096 + r % 256 (one contiguous block per workgroup and step)
  const int len = a.chunk_len[g * CELL_NW + wv];
  const int r0 = a.grp_base[g] + wv * (64 * CELL_R);
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

  // software pipeline: the records and residuals of step st + 1 are requested while step st is worked on -- AFTER the I-table
  // gathers of step st, so that waiting for those (vmcnt counts in order) does not wait for the prefetch
  constexpr int WROWS = 64 * CELL_R, SROWS = WROWS * CELL_NW;
  const bool has_I = a.cardI > 0;
  int slotI = 0;
#pragma unroll
  for (int s = 0; s < CELL_MAX_STREAMS; s++)
    if (s < a.n_streams && a.type[s] == CELL_I) slotI = a.slot[s];
  uint2 rec_n[CELL_R];
  double e_n[CELL_R];
  int it_n[CELL_R];
#pragma unroll
  for (int k = 0; k < CELL_R; k++) {
    const int lr = k * 64 + lane, pos = r0 + lr;
    rec_n[k] = make_uint2(0, 0);
    e_n[k] = 0.0;
    it_n[k] = 0;
    if (lr < len) {
      rec_n[k] = a.ix[pos];
      e_n[k] = __builtin_nontemporal_load(a.e + pos);
      if (ITEM32) it_n[k] = a.item[pos];
    }
  }
  for (int st = 0; st < steps; st++) {
    const int base = r0 + st * SROWS, lbase = st * WROWS;
    uint2 rec[CELL_R];
    double e[CELL_R];
    int it[CELL_R];
    bool valid[CELL_R];
    double4 pk[CELL_R];
#pragma unroll
    for (int k = 0; k < CELL_R; k++) {
      rec[k] = rec_n[k];
      e[k] = e_n[k];
      valid[k] = lbase + k * 64 + lane < len;
      it[k] = -2;
      pk[k] = make_double4(0.0, 0.0, 0.0, 0.0);
      if (has_I && valid[k]) {
        it[k] = ITEM32 ? it_n[k] : cell_slot(rec[k], slotI);
        if (!a.linear) {
          pk[k] = packI[it[k]];
        } else if (p_on_I) {  // (update_w: no packed table is built, the pending field's d1 comes straight from DP)
          const double2 dd = a.DP[it[k]];
          pk[k].z = dd.x;
          pk[k].w = dd.y;
        }
      }
    }
    if (st + 1 < steps) {
#pragma unroll
      for (int k = 0; k < CELL_R; k++) {
        const int lr = lbase + WROWS + k * 64 + lane, pos = base + SROWS + k * 64 + lane;
        if (lr < len) {
          rec_n[k] = a.ix[pos];
          e_n[k] = __builtin_nontemporal_load(a.e + pos);
          if (ITEM32) it_n[k] = a.item[pos];
        }
      }
    }
    double v[CELL_R][NS > 1 ? NS : 2];
    int idxF[CELL_R];
#pragma unroll
    for (int k = 0; k < CELL_R; k++) {
      const int pos = base + k * 64 + lane;
      double qa = pk[k].x, qs = pk[k].y;  // (fixed order of the sum: I first, then the LDS streams in stream order)
      if (valid[k]) {
#pragma unroll
        for (int s = 0; s < CELL_MAX_STREAMS; s++) {
          if (s < a.n_streams && a.type[s] != CELL_I && !a.linear) {
            const int i = cell_slot(rec[k], a.slot[s]);
            const double x = lds[a.ldsS[s] + i];
            qs += x;
            if (HASP) qa += a.pair[s] == 1 ? lds[a.ldsA[s] + i] : x;
          }
        }
        if (HASP) {
          const double2 d = p_on_I ? make_double2(pk[k].z, pk[k].w) : dpl[cell_slot(rec[k], slotP)];
          e[k] += a.linear ? d.x : qa * d.x + d.y;
          __builtin_nontemporal_store(e[k], a.e + pos);
        }
      }
      if (FT >= 0) {
        const double h = qs;
        if (NS == 1) {
          v[k][0] = valid[k] ? e[k] : 0.0;
          v[k][1] = 0.0;
        } else if (NS == 2) {
          v[k][0] = valid[k] ? h * h : 0.0;
          v[k][1] = valid[k] ? e[k] * h : 0.0;
        } else if (NS == 4) {
          v[k][0] = valid[k] ? h : 0.0;
          v[k][1] = valid[k] ? h * h : 0.0;
          v[k][2] = valid[k] ? e[k] : 0.0;
          v[k][3] = valid[k] ? e[k] * h : 0.0;
        }
        idxF[k] = (FT == CELL_I) ? it[k] : cell_slot(rec[k], slotF);
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
          for (int j = 0; j < NS; j++)  // (one table per sum: a wave's 64 adds of sum j spread over 16 bank pairs, not 4 as [index][sum])
            __hip_atomic_fetch_add(&acc[j * nval + idxF[k]], v[k][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
        if (u0 + i / NS < a.n_out) a.out[(int64_t)(u0 + i / NS) * a.out_stride + (i % NS)] = acc[(i % NS) * nval + i / NS];
    } else {
      double *o = a.out + (int64_t)g * nacc;
      for (int i = tid; i < nacc; i += CELL_NT) o[i] = acc[(i % NS) * nval + i / NS];
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

void cell_pass(hipStream_t s, Timing &tm, CellPlan &cp, int P, int F, bool sw, double *out_u, int out_stride, bool linear) {
  if ((P < 0 && F < 0) || cp.G == 0) return;  // (G = 0: an empty shard)
  CellPassArgs a;
  std::memset(&a, 0, sizeof(a));
  int off[11];
  const size_t lds = cp.lds_bytes(P, F, sw, off, linear);
  a.linear = linear ? 1 : 0;
  if (lds > CELL_LDS_BYTES) throw Error(MFM_ERR_RUNTIME, "internal: cell pass does not fit the LDS");
  a.ix = cp.ix.p;
  a.item = cp.item.p;
  a.e = cp.e.p;
  a.chunk_len = cp.chunk_len.p;
  a.grp_base = cp.grp_base.p;
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
  const int ns = F >= 0 ? (linear ? 1 : (cp.fields[F].kind == 0 ? 2 : 4)) : 0;
  a.ns = ns;
  if (ft == CELL_U) {
    a.out = out_u;
    a.out_stride = out_stride;
    a.n_out = (int)cp.fields[F].n;
  } else if (ft == CELL_I) {
    a.out = ns == 1 ? cp.cells1.p : (ns == 2 ? cp.cells2.p : cp.cells4.p);
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
  else if (ns == 1 && ft == CELL_U)
    launch_pass_f<CELL_U, 1>(s, cp.G, lds, a, hasp, cp.item32);
  else if (ns == 1 && ft == CELL_I)
    launch_pass_f<CELL_I, 1>(s, cp.G, lds, a, hasp, cp.item32);
  else if (ns == 1)
    launch_pass_f<CELL_C, 1>(s, cp.G, lds, a, hasp, cp.item32);
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
// Sums of the (group, index) partials over the groups, fixed association: 256 threads = 32 index values x 8 slices of the
// group range; a slice adds its groups in group order, the eight slice sums are added in slice order.
template <int NS>
__device__ __forceinline__ bool cell_group_sums(const double *__restrict__ src, int G, int64_t card, int n, double (&out)[NS]) {
  __shared__ double part[8][32][NS];
  const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + o;
  const int gs = (G + 7) / 8;
  double acc[NS];
#pragma unroll
  for (int j = 0; j < NS; j++) acc[j] = 0.0;
  if (i < n) {
    const int g1 = min(G, (sl + 1) * gs);
    for (int g = sl * gs; g < g1; g++) {
      const double2 *p = (const double2 *)(src + ((int64_t)g * card + i) * NS);
#pragma unroll
      for (int j = 0; j < NS / 2; j++) {
        const double2 x = p[j];
        acc[2 * j] += x.x;
        acc[2 * j + 1] += x.y;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NS; j++) part[sl][o][j] = acc[j];
  __syncthreads();
  if (sl != 0 || i >= n) return false;
#pragma unroll
  for (int j = 0; j < NS; j++) {
    double t = part[0][o][j];
#pragma unroll
    for (int k = 1; k < 8; k++) t += part[k][o][j];
    out[j] = t;
  }
  return true;
}

// main field: the draw of FMTrainer.hpp:357-369 with x = 1 (S2 = sum q_other^2, S1 = -sum e q_other), V and (d1, d2) = (v' - v, 0)
__device__ __forceinline__ void cell_draw_one(int i, double S2, double Seh, double *__restrict__ Vf, const double *__restrict__ zf,
                                              const int32_t *__restrict__ group, const double *__restrict__ lam,
                                              const double *__restrict__ mu, double alpha, int64_t base, double2 *__restrict__ DP) {
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
__global__ __launch_bounds__(256) void k_cell_draw_direct(const double *__restrict__ stat, int n, double *__restrict__ Vf,
                                                          const double *__restrict__ zf, const int32_t *__restrict__ group,
                                                          const double *__restrict__ lam, const double *__restrict__ mu, double alpha,
                                                          int64_t base, double2 *__restrict__ DP) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  cell_draw_one(i, stat[(int64_t)i * 2], stat[(int64_t)i * 2 + 1], Vf, zf, group, lam, mu, alpha, base, DP);
}
__global__ __launch_bounds__(256) void k_cell_draw_groups(const double *__restrict__ src, int G, int64_t card, int n,
                                                          double *__restrict__ Vf, const double *__restrict__ zf,
                                                          const int32_t *__restrict__ group, const double *__restrict__ lam,
                                                          const double *__restrict__ mu, double alpha, int64_t base,
                                                          double2 *__restrict__ DP) {
  double sum[2];
  if (!cell_group_sums<2>(src, G, card, n, sum)) return;
  cell_draw_one(blockIdx.x * 32 + (threadIdx.x & 31), sum[0], sum[1], Vf, zf, group, lam, mu, alpha, base, DP);
}

void cell_draw_main(hipStream_t s, Timing &tm, CellPlan &cp, int F, double *Vf, const double *zf, const int32_t *group, const double *lam,
                    const double *mu, double alpha, const double *dense) {
  const CellField &f = cp.fields[F];
  const CellStream &st = cp.streams[f.stream];
  TimedLaunch t(tm, s, KC_CELL_SMALL, 0.0);
  const int n = (int)f.n;
  if (dense)  // (row-sharded: the sums of all ranks, [n][2])
    hipLaunchKernelGGL(k_cell_draw_direct, dim3(cdiv_c(n, 256)), dim3(256), 0, s, dense, n, Vf, zf, group, lam, mu, alpha, f.base, cp.DP.p);
  else if (st.type == CELL_U)
    hipLaunchKernelGGL(k_cell_draw_direct, dim3(cdiv_c(n, 256)), dim3(256), 0, s, cp.stat.p, n, Vf, zf, group, lam, mu, alpha, f.base,
                       cp.DP.p);
  else
    hipLaunchKernelGGL(k_cell_draw_groups, dim3(cdiv_c(n, 32)), dim3(256), 0, s, st.type == CELL_I ? cp.cells2.p : cp.cpart.p, cp.G,
                       st.card, n, Vf, zf, group, lam, mu, alpha, f.base, cp.DP.p);
  MFM_HIP_CHECK(hipGetLastError());
}

// block on I / C: rec[i].{c, c_S, e, e_q} (words 2..5) = sum over the groups (FMTrainer.hpp:401-407)
__global__ __launch_bounds__(256) void k_cell_block_stats(const double *__restrict__ src, int G, int64_t card, int n, double *__restrict__ rec) {
  double sum[4];
  if (!cell_group_sums<4>(src, G, card, n, sum)) return;
  double2 *r = (double2 *)rec + (int64_t)(blockIdx.x * 32 + (threadIdx.x & 31)) * 4;
  r[1] = make_double2(sum[0], sum[1]);
  r[2] = make_double2(sum[2], sum[3]);
}
void cell_block_stats(hipStream_t s, Timing &tm, CellPlan &cp, int F, double *rec) {
  const CellField &f = cp.fields[F];
  const CellStream &st = cp.streams[f.stream];
  if (st.type == CELL_U) return;
  TimedLaunch t(tm, s, KC_CELL_SMALL, 0.0);
  hipLaunchKernelGGL(k_cell_block_stats, dim3(cdiv_c(f.n, 32)), dim3(256), 0, s, st.type == CELL_I ? cp.cells4.p : cp.cpart.p, cp.G, st.card,
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

// ---------------------------------------------------------------------------------------------------------------------------
// Row-sharded mode: every rank runs the passes over its own rows; a field's sums then go through ONE dense array
// [index values][sums] that is all-reduced over the ranks (RCCL, by the caller) before the replicated draw / feature sweep.
template <int NS>
__global__ __launch_bounds__(256) void k_cell_group_sums_dense(const double *__restrict__ src, int G, int64_t card, int n,
                                                               double *__restrict__ dense) {
  double sum[NS];
  if (!cell_group_sums<NS>(src, G, card, n, sum)) return;
  double *o = dense + (int64_t)(blockIdx.x * 32 + (threadIdx.x & 31)) * NS;
#pragma unroll
  for (int j = 0; j < NS; j++) o[j] = sum[j];
}
void cell_stats_dense(hipStream_t s, Timing &tm, CellPlan &cp, int F, int ns, double *dense) {
  const CellField &f = cp.fields[F];
  const CellStream &st = cp.streams[f.stream];
  if (st.type == CELL_U) return;  // (the pass wrote its groups' values into the dense array itself)
  if (ns == 1) {
    cell_sum1(s, tm, cp, F, dense, 1);
    return;
  }
  TimedLaunch t(tm, s, KC_CELL_SMALL, 0.0);
  const double *src = st.type == CELL_I ? (ns == 2 ? cp.cells2.p : cp.cells4.p) : cp.cpart.p;
  if (ns == 2)
    hipLaunchKernelGGL(k_cell_group_sums_dense<2>, dim3(cdiv_c(f.n, 32)), dim3(256), 0, s, src, cp.G, st.card, (int)f.n, dense);
  else
    hipLaunchKernelGGL(k_cell_group_sums_dense<4>, dim3(cdiv_c(f.n, 32)), dim3(256), 0, s, src, cp.G, st.card, (int)f.n, dense);
  MFM_HIP_CHECK(hipGetLastError());
}
__global__ __launch_bounds__(256) void k_cell_dense_to_rec(const double *__restrict__ dense, int n, int ns, double *__restrict__ rec, int w0) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * ns) return;
  rec[(int64_t)(i / ns) * 8 + w0 + (i % ns)] = dense[i];
}
void cell_dense_to_rec(hipStream_t s, Timing &tm, CellPlan &cp, int F, int ns, const double *dense, double *rec, int w0) {
  const CellField &f = cp.fields[F];
  TimedLaunch t(tm, s, KC_CELL_SMALL, 0.0);
  hipLaunchKernelGGL(k_cell_dense_to_rec, dim3(cdiv_c(f.n * ns, 256)), dim3(256), 0, s, dense, (int)f.n, ns, rec, w0);
  MFM_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------------------
// update_w on the cell layout (FMTrainer.hpp:231-313). A one-hot main column i with x = 1: S2 = n_i (the rows it occurs in: static),
// S1 = sum_t (e_t - w_old) = sum e - n_i w_old (:242-248); a block needs e_B = sum e per block row (:271) and changes its rows by
// q_B' - q_B (:272-273 and :306-311 together). So the linear sweep is the same pass with ONE sum per index value and no q tables.
__global__ __launch_bounds__(256) void k_cell_sum1(const double *__restrict__ src, int G, int64_t card, int n, double *__restrict__ dst,
                                                   int dst_stride) {
  // [G][card] partials -> dst[i * dst_stride], 32 values x 8 slices of the group range, fixed association
  __shared__ double part[8][32];
  const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + o;
  const int gs = (G + 7) / 8;
  double acc = 0.0;
  if (i < n)
    for (int g = sl * gs; g < min(G, (sl + 1) * gs); g++) acc += src[(int64_t)g * card + i];
  part[sl][o] = acc;
  __syncthreads();
  if (sl != 0 || i >= n) return;
  double t = part[0][o];
#pragma unroll
  for (int k = 1; k < 8; k++) t += part[k][o];
  dst[(int64_t)i * dst_stride] = t;
}
void cell_sum1(hipStream_t s, Timing &tm, CellPlan &cp, int F, double *dst, int dst_stride) {
  const CellField &f = cp.fields[F];
  const CellStream &st = cp.streams[f.stream];
  if (st.type == CELL_U) return;  // (the pass wrote the sums itself)
  TimedLaunch t(tm, s, KC_CELL_SMALL, 0.0);
  hipLaunchKernelGGL(k_cell_sum1, dim3(cdiv_c(f.n, 32)), dim3(256), 0, s, st.type == CELL_I ? cp.cells1.p : cp.cpart.p, cp.G, st.card,
                     (int)f.n, dst, dst_stride);
  MFM_HIP_CHECK(hipGetLastError());
}
__global__ void k_cell_fill_ones(const int32_t *__restrict__ perm, int64_t N, double *__restrict__ e) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < N) e[p] = perm[p] >= 0 ? 1.0 : 0.0;
}
// n_i of every main column: the statistics pass over a residual of ones (once per plan; cp.e is scratch here)
void cell_counts(hipStream_t s, Timing &tm, CellPlan &cp) {
  if (cp.cnt_ready) return;
  if (cp.Npad > 0) hipLaunchKernelGGL(k_cell_fill_ones, dim3(cdiv_c(cp.Npad, 256)), dim3(256), 0, s, cp.perm.p, cp.Npad, cp.e.p);
  for (size_t F = 0; F < cp.fields.size(); F++) {
    if (cp.fields[F].kind != 0) continue;
    cell_pass(s, tm, cp, -1, (int)F, false, cp.cnt[F].p, 1, true);
    cell_sum1(s, tm, cp, (int)F, cp.cnt[F].p, 1);
  }
  cp.cnt_ready = true;
  MFM_HIP_CHECK(hipGetLastError());
}
__global__ __launch_bounds__(256) void k_cell_draw_w(const double *__restrict__ se, const double *__restrict__ cnt, int n,
                                                     double *__restrict__ w, const double *__restrict__ z,
                                                     const int32_t *__restrict__ group, const double *__restrict__ lam,
                                                     const double *__restrict__ mu, double alpha, int64_t base, double2 *__restrict__ DP) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t j = base + i;
  const double old = w[j], S2 = cnt[i];
  const double S1 = se[i] - S2 * old;  // sum_t x (e_t - x w_old), x = 1   (:242-248)
  const int gi = group[j];
  const double l = lam[gi], m = mu[gi];
  const double sq = l + alpha * S2;        // :250
  const double lin = -alpha * S1 + l * m;  // :251
  const double fresh = sample_normal_z(sq, lin, z[j]);
  w[j] = fresh;
  DP[i] = make_double2(fresh - old, 0.0);  // e += x (w' - w)   (:252)
}
void cell_draw_main_w(hipStream_t s, Timing &tm, CellPlan &cp, int F, double *w, const double *z, const int32_t *group, const double *lam,
                      const double *mu, double alpha, const double *se) {
  const CellField &f = cp.fields[F];
  TimedLaunch t(tm, s, KC_CELL_SMALL, 0.0);
  hipLaunchKernelGGL(k_cell_draw_w, dim3(cdiv_c(f.n, 256)), dim3(256), 0, s, se ? se : cp.stat1.p, cp.cnt[F].p, (int)f.n, w, z, group, lam, mu, alpha,
                     f.base, cp.DP.p);
  MFM_HIP_CHECK(hipGetLastError());
}
__global__ __launch_bounds__(256) void k_cell_block_delta_w(const double *__restrict__ rec, const double2 *__restrict__ saved, int n,
                                                            double2 *__restrict__ DP) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  DP[i] = make_double2(rec[(int64_t)i * 8] - saved[i].x, 0.0);
}
void cell_block_delta_w(hipStream_t s, Timing &tm, CellPlan &cp, int F, const double *rec, const double2 *saved) {
  const CellField &f = cp.fields[F];
  TimedLaunch t(tm, s, KC_CELL_SMALL, 0.0);
  hipLaunchKernelGGL(k_cell_block_delta_w, dim3(cdiv_c(f.n, 256)), dim3(256), 0, s, rec, saved, (int)f.n, cp.DP.p);
  MFM_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------------------
// update_e on the cell layout (FMTrainer.hpp:493-497 -> FM.hpp:54-136). With the row a tuple of indices
//     score_t = w0 + sum_s LS_s[idx_s] + 1/2 sum_f (sum_s Q_s,f[idx_s])^2,
//     Q_s,f = sum of the stream's fields' factor-f tables (V rows of a main field, X_B V_B rows of a block: FM.hpp:104-106),
//     LS_s  = sum of their linear terms (w, X_B w_B: :81) - 1/2 sum of their sum_f sum_l x^2 v^2 terms (:121-127),
// so the scorer is K / FB passes over (accumulator, index record), FB factors per pass with the tables in LDS, instead of one
// pass that gathers three or four K-vectors per row from L2 / HBM (k_score at N = 50 M, rank 64: 68 GB of traffic, 34 ms).
struct CellScoreArgs {
  const uint2 *ix;
  const int32_t *item;
  double *acc;
  const int32_t *chunk_len, *grp_base, *grp_u0, *grp_steps;
  const double *Q[CELL_MAX_STREAMS];   // [card][KS]
  const double *LS[CELL_MAX_STREAMS];  // [card]
  int n_streams, type[CELL_MAX_STREAMS], slot[CELL_MAX_STREAMS], card[CELL_MAX_STREAMS], lds_off[CELL_MAX_STREAMS];
  int KS, f0, nf;  // factors [f0, f0 + nf) of this pass, nf <= FB (nf = 0: the initial pass, acc = w0 + sum LS)
  double w0;
  int cardI;
};

template <int FB, bool ITEM32>
__global__ __launch_bounds__(CELL_NT) void k_cell_score(CellScoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int g = blockIdx.x, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int u0 = a.grp_u0[g], nu = a.grp_u0[g + 1] - u0;
  constexpr int TW = FB > 0 ? FB : 1;  // doubles per table entry in LDS
  for (int s = 0; s < a.n_streams; s++) {
    if (a.type[s] == CELL_I) continue;
    const int n = a.type[s] == CELL_U ? nu : a.card[s];
    const int o = a.type[s] == CELL_U ? u0 : 0;
    double *t = lds + a.lds_off[s];
    if (FB == 0) {
      for (int i = tid; i < n; i += CELL_NT) t[i] = a.LS[s][o + i];
    } else {
      for (int i = tid; i < n * FB; i += CELL_NT) {
        const int r = i / FB, j = i - r * FB;
        t[i] = j < a.nf ? a.Q[s][(int64_t)(o + r) * a.KS + a.f0 + j] : 0.0;
      }
    }
  }
  __syncthreads();
  constexpr int WROWS = 64 * CELL_R, SROWS = WROWS * CELL_NW;
  const int len = a.chunk_len[g * CELL_NW + wv];
  const int r0 = a.grp_base[g] + wv * WROWS;
  const int steps = a.grp_steps[g];
  int sI = -1;
#pragma unroll
  for (int s = 0; s < CELL_MAX_STREAMS; s++)
    if (s < a.n_streams && a.type[s] == CELL_I) sI = s;
  // the same software pipeline as the sweep's pass: records and accumulators of step st + 1 are requested after the I-table
  // gathers of step st
  uint2 rec_n[CELL_R];
  double acc_n[CELL_R];
  int it_n[CELL_R];
#pragma unroll
  for (int k = 0; k < CELL_R; k++) {
    const int lr = k * 64 + lane, pos = r0 + lr;
    rec_n[k] = make_uint2(0, 0);
    acc_n[k] = 0.0;
    it_n[k] = 0;
    if (lr < len) {
      rec_n[k] = a.ix[pos];
      if (FB > 0) acc_n[k] = __builtin_nontemporal_load(a.acc + pos);
      if (ITEM32) it_n[k] = a.item[pos];
    }
  }
  for (int st = 0; st < steps; st++) {
    uint2 rec[CELL_R];
    double acc[CELL_R];
    bool valid[CELL_R];
    double q[CELL_R][TW];
#pragma unroll
    for (int k = 0; k < CELL_R; k++) {
      rec[k] = rec_n[k];
      acc[k] = acc_n[k];
      valid[k] = st * WROWS + k * 64 + lane < len;
#pragma unroll
      for (int j = 0; j < TW; j++) q[k][j] = 0.0;
      if (sI >= 0 && valid[k]) {
        const int it = ITEM32 ? it_n[k] : cell_slot(rec[k], a.slot[sI]);
        if (FB == 0) {
          q[k][0] = a.LS[sI][it];
        } else {
          const double *src = a.Q[sI] + (int64_t)it * a.KS + a.f0;
#pragma unroll
          for (int j = 0; j < TW; j++) q[k][j] = j < a.nf ? src[j] : 0.0;
        }
      }
    }
    if (st + 1 < steps) {
#pragma unroll
      for (int k = 0; k < CELL_R; k++) {
        const int lr = (st + 1) * WROWS + k * 64 + lane, pos = r0 + (st + 1) * SROWS + k * 64 + lane;
        if (lr < len) {
          rec_n[k] = a.ix[pos];
          if (FB > 0) acc_n[k] = __builtin_nontemporal_load(a.acc + pos);
          if (ITEM32) it_n[k] = a.item[pos];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < CELL_R; k++) {
      if (!valid[k]) continue;
      const int pos = r0 + st * SROWS + k * 64 + lane;
#pragma unroll
      for (int s = 0; s < CELL_MAX_STREAMS; s++)
        if (s < a.n_streams && a.type[s] != CELL_I) {
          const double *t = lds + a.lds_off[s] + cell_slot(rec[k], a.slot[s]) * TW;
#pragma unroll
          for (int j = 0; j < TW; j++) q[k][j] += t[j];
        }
      if (FB == 0) {
        __builtin_nontemporal_store(a.w0 + q[k][0], a.acc + pos);
      } else {
        double v = acc[k];
#pragma unroll
        for (int j = 0; j < TW; j++) v += 0.5 * (q[k][j] * q[k][j]);
        __builtin_nontemporal_store(v, a.acc + pos);
      }
    }
  }
}

// per feature: sum_f v_jf^2 from the factor-major V (coalesced across the features)
__global__ __launch_bounds__(256) void k_cell_rowsumsq(const double *__restrict__ V, int64_t D, int K, double *__restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= D) return;
  double s = 0.0;
  for (int f = 0; f < K; f++) {
    const double v = V[(int64_t)f * D + j];
    s += v * v;
  }
  out[j] = s;
}
__global__ void k_cell_unpack_score(const double *__restrict__ acc, const int32_t *__restrict__ perm, int64_t N,
                                    const double *__restrict__ y, double2 *__restrict__ eq) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < N) {
    const int t = perm[p];
    if (t >= 0) eq[t].x = y ? acc[p] - y[t] : acc[p];
  }
}

template <int FB>
static void launch_score_t(hipStream_t s, int G, size_t lds, const CellScoreArgs &a, bool item32) {
  static DeviceOnce raised;
  if (raised.need()) {
    MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_cell_score<FB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_cell_score<FB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    raised.mark();
  }
  if (item32)
    hipLaunchKernelGGL((k_cell_score<FB, true>), dim3(G), dim3(CELL_NT), lds, s, a);
  else
    hipLaunchKernelGGL((k_cell_score<FB, false>), dim3(G), dim3(CELL_NT), lds, s, a);
}

void cell_score(hipStream_t s, Timing &tm, CellPlan &cp, const std::vector<CellScoreSrc> &src, const double *V, const double *Vt, int64_t D,
                int K, int KS, double w0, const double *y, double2 *eq) {
  // tables: Q_s [card][KS], LS_s [card]
  if (cp.vss.n < (size_t)D) cp.vss.alloc((size_t)std::max<int64_t>(D, 1));
  for (size_t si = 0; si < cp.streams.size(); si++) {
    const size_t need = (size_t)cp.streams[si].card * (size_t)std::max(KS, 1);
    if (cp.scoreQ[si].n < need) cp.scoreQ[si].alloc(need);
    if (cp.scoreLS[si].n < (size_t)cp.streams[si].card) cp.scoreLS[si].alloc((size_t)cp.streams[si].card);
  }
  {
    TimedLaunch t(tm, s, KC_CELL_SMALL, 0.0);
    if (K > 0) hipLaunchKernelGGL(k_cell_rowsumsq, dim3(cdiv_c(D, 256)), dim3(256), 0, s, V, D, K, cp.vss.p);
    CellPrepArgs a;
    a.n_jobs = 0;
    int64_t maxn = 0;
    for (size_t si = 0; si < cp.streams.size(); si++) {
      const CellStream &st = cp.streams[si];
      CellPrepJob &jq = a.job[a.n_jobs++];
      CellPrepJob &jl = a.job[a.n_jobs++];
      jq.dst = cp.scoreQ[si].p;
      jq.dst_stride = 1;
      jq.n = (int)(st.card * KS);
      jq.nsrc = 0;
      jl.dst = cp.scoreLS[si].p;
      jl.dst_stride = 1;
      jl.n = (int)st.card;
      jl.nsrc = 0;
      if (st.card * KS >= (int64_t)2147483647) throw Error(MFM_ERR_RUNTIME, "cell scorer: table too large");
      if (st.fields.size() > 2) throw Error(MFM_ERR_RUNTIME, "internal: more than two fields on one index stream");
      for (int f : st.fields) {
        const CellField &fd = cp.fields[f];
        const CellScoreSrc &sc = src[f];
        const double *q = fd.kind == 0 ? Vt + fd.base * KS : sc.q;
        const double *lin = sc.lin;
        const double *ss = fd.kind == 0 ? cp.vss.p + fd.base : sc.ss;
        if (K > 0) {
          jq.src[jq.nsrc] = q;
          jq.sstride[jq.nsrc] = 1;
          jq.sn[jq.nsrc] = (int)(std::min<int64_t>(fd.n, st.card) * KS);
          jq.coef[jq.nsrc] = 1.0;
          jq.nsrc++;
        }
        jl.src[jl.nsrc] = lin;
        jl.sstride[jl.nsrc] = 1;
        jl.sn[jl.nsrc] = (int)std::min<int64_t>(fd.n, st.card);
        jl.coef[jl.nsrc] = 1.0;
        jl.nsrc++;
        if (K > 0) {
          jl.src[jl.nsrc] = ss;
          jl.sstride[jl.nsrc] = 1;
          jl.sn[jl.nsrc] = (int)std::min<int64_t>(fd.n, st.card);
          jl.coef[jl.nsrc] = -0.5;
          jl.nsrc++;
        }
      }
      maxn = std::max<int64_t>(maxn, std::max<int64_t>(jq.n, jl.n));
    }
    hipLaunchKernelGGL(k_cell_prep, dim3(cdiv_c(maxn, 256), a.n_jobs), dim3(256), 0, s, a);
  }
  CellScoreArgs a;
  std::memset(&a, 0, sizeof(a));
  a.ix = cp.ix.p;
  a.item = cp.item.p;
  a.acc = cp.e.p;
  a.chunk_len = cp.chunk_len.p;
  a.grp_base = cp.grp_base.p;
  a.grp_u0 = cp.grp_u0.p;
  a.grp_steps = cp.grp_steps.p;
  a.n_streams = (int)cp.streams.size();
  a.KS = KS;
  a.w0 = w0;
  a.cardI = cp.sI >= 0 ? (int)cp.streams[cp.sI].card : 0;
  size_t per_factor = 0;  // LDS doubles per factor of a pass
  for (int si = 0; si < a.n_streams; si++) {
    a.Q[si] = cp.scoreQ[si].p;
    a.LS[si] = cp.scoreLS[si].p;
    a.type[si] = cp.streams[si].type;
    a.slot[si] = cp.streams[si].slot;
    a.card[si] = (int)cp.streams[si].card;
    if (a.type[si] != CELL_I) per_factor += a.type[si] == CELL_U ? (size_t)cp.umax : (size_t)cp.streams[si].card;
  }
  auto layout = [&](int tw) {
    size_t o = 0;
    for (int si = 0; si < a.n_streams; si++) {
      if (a.type[si] == CELL_I) continue;
      a.lds_off[si] = (int)o;
      o += (size_t)tw * (a.type[si] == CELL_U ? (size_t)cp.umax : (size_t)cp.streams[si].card);
      o = (o + 1) & ~(size_t)1;
    }
    return (o + 2) * sizeof(double);
  };
  int FB = 4;  // (8 factors' q of 6 rows per lane do not fit 128 VGPRs)
  while (FB > 1 && layout(FB) > CELL_LDS_BYTES) FB /= 2;
  if (layout(1) > CELL_LDS_BYTES) throw Error(MFM_ERR_RUNTIME, "internal: cell scorer tables do not fit the LDS");
  const double row_bytes = (double)cp.N * (8.0 + 8.0 + 8.0 + (cp.item32 ? 4.0 : 0.0));
  if (cp.G == 0) return;  // (an empty shard: no rows to score)
  {
    TimedLaunch t(tm, s, KC_UPDATE_E, row_bytes - 8.0 * cp.N);
    a.f0 = 0;
    a.nf = 0;
    launch_score_t<0>(s, cp.G, layout(1), a, cp.item32);
  }
  for (int f0 = 0; f0 < K; f0 += FB) {
    TimedLaunch t(tm, s, KC_UPDATE_E, row_bytes);
    a.f0 = f0;
    a.nf = std::min(FB, K - f0);
    const size_t lds = layout(FB);
    if (FB == 4) launch_score_t<4>(s, cp.G, lds, a, cp.item32);
    else if (FB == 2) launch_score_t<2>(s, cp.G, lds, a, cp.item32);
    else launch_score_t<1>(s, cp.G, lds, a, cp.item32);
  }
  {
    TimedLaunch t(tm, s, KC_UPDATE_E, 28.0 * cp.N);
    hipLaunchKernelGGL(k_cell_unpack_score, dim3(cdiv_c(cp.Npad, 256)), dim3(256), 0, s, cp.e.p, cp.perm.p, cp.Npad, y, eq);
  }
  MFM_HIP_CHECK(hipGetLastError());
}

}  // namespace mfm
