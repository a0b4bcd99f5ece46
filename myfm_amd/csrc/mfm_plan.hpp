// mfm_plan.hpp -- the conflict-free execution plan of a column sweep.
//
// The reference updates the features of a table strictly in index order (FMTrainer.hpp:343, :419).
// Two features whose columns share no row touch disjoint state and commute, so the plan assigns
// level(j) = 1 + max level of the earlier columns sharing a row with j and runs the levels in order:
//   PAR step   one level with enough work: its columns run concurrently, binned by length
//              W1 (<= 64 entries) / W4 (<= 256): wavefront per column, "light" launch
//              W16 (<= 1024, wavefront) / WG (<= 256 * R_WG, workgroup): "heavy" launch
//              LONG: chunks of 256 * R_WG entries, co-resident, single pass (k_long_coop), launched in
//                    rounds that never exceed the device's resident-workgroup capacity
//              HUGE: more chunks than one round can hold (e.g. a dense column): two-pass
//   CHAIN step a run of consecutive tiny levels (dense / multi-hot columns, small blocks): one
//              workgroup walks their columns one after the other inside a single launch.
// The draws are identical to the sequential order given the same per-feature variates.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include <dlfcn.h>
#include <hipcub/hipcub.hpp>

#include "mfm_chain_api.hpp"
#include "mfm_common.hpp"
#include "mfm_kernels.hpp"
#include "mfm_mf_kernels.hpp"
#include "mfm_res.hpp"

struct mfm_nccl_id {
  char internal[128];  // ncclUniqueId (NCCL_UNIQUE_ID_BYTES)
};

namespace mfm {

struct ParLevel {
  DevBuf<int32_t> cols_w1, cols_w4, cols_w16, cols_wg, cols_long, cols_huge;
  int n_w1 = 0, n_w4 = 0, n_w16 = 0, n_wg = 0, n_long = 0, n_huge = 0;
  int64_t nnz_light = 0, nnz_heavy = 0, nnz_long = 0, nnz_huge = 0;
  // LONG: co-resident single pass
  DevBuf<ChunkDesc> lchunks;
  DevBuf<int32_t> lchunk_ptr;
  DevBuf<double> lpartial;
  DevBuf<unsigned long long> arrive;
  mutable unsigned long long epoch = 0;
  std::vector<std::pair<int, int>> rounds;  // (first chunk, number of chunks)
  std::vector<int64_t> round_nnz;
  // HUGE: two-pass
  DevBuf<ChunkDesc> hchunks;
  DevBuf<int32_t> hchunk_ptr;
  int n_hchunks = 0;
  // all columns of the level (for the sharded mode's draw kernel) and their index range
  DevBuf<int32_t> cols_all;
  int n_all = 0, jmin = 0, jmax = -1;
  // scattered level: row-blocked two-pass path (k_scat_*); replaces all the bins above
  bool scattered = false;
  int64_t n_ent = 0;
  int n_cols = 0, n_runs = 0;
  DevBuf<int2> ent;          // (row, column), sorted by (row block, column, row)
  DevBuf<double> ent_val;    // same order (empty when the matrix is unit-valued)
  DevBuf<int32_t> run_base;  // per 64-entry tile: index of its first run
  DevBuf<int32_t> scols, slot_ptr, slot_idx;
  DevBuf<int32_t> slot_pos;  // row-tile variant: run (stream order) -> slot (column-major)
  DevBuf<double2> slots;
  // row-tile variant (k_tile_*): LDS-staged {e, q} tiles, 4-byte packed entries
  bool tiled = false;
  int tile_bits = 0, n_tiles = 0;
  DevBuf<uint32_t> tent;      // padded per tile to whole 64-entry wave tiles
  DevBuf<int32_t> tile_ptr;   // [n_tiles + 1], in wave tiles
  DevBuf<int32_t> tile_row0;  // [n_tiles + 1] first row of every tile (fixed 2^tile_bits grid, or StepPlan::h_tile_start)
  bool covers_rows_once = false;  // every row of the table has exactly one entry in this level
  bool first_and_once = false;    // ... and it is the first step of the plan: the level can rebuild q itself
  bool contig = false;            // every column of the level covers a contiguous row range (StepPlan::col_row0)
  int64_t nnz_total() const { return nnz_light + nnz_heavy + nnz_long + nnz_huge; }
  // The row-tile form of the SAME level of the same matrix built by another plan (the w and V sweeps of a table walk
  // identical tiles): views of its arrays instead of a second build. The sweeps never overlap, so `slots` is shared too.
  void borrow_tiled(const ParLevel &o) {
    scattered = o.scattered;
    tiled = o.tiled;
    tile_bits = o.tile_bits;
    n_tiles = o.n_tiles;
    n_ent = o.n_ent;
    n_cols = o.n_cols;
    n_runs = o.n_runs;
    covers_rows_once = o.covers_rows_once;
    tent.borrow(o.tent);
    ent_val.borrow(o.ent_val);
    tile_ptr.borrow(o.tile_ptr);
    tile_row0.borrow(o.tile_row0);
    run_base.borrow(o.run_base);
    scols.borrow(o.scols);
    slot_ptr.borrow(o.slot_ptr);
    slot_pos.borrow(o.slot_pos);
    slots.borrow(o.slots);
  }
};

struct DevCscView {
  const int64_t *colptr = nullptr;
  const int32_t *rowidx = nullptr;
  const double *cval = nullptr;
  hipStream_t stream = nullptr;
  // the same table by rows (level schedule on the device)
  const int32_t *rowptr = nullptr;
  const int32_t *colidx = nullptr;
  int64_t n_rows = 0, n_cols = 0;
  int ell = -1;  // >= 0: every row has exactly this many entries (rowptr not read)
};

// ---- conflict batches of a chain run built ON THE DEVICE (ChainRun::build_batched is the host form and the checker) ---
// Device-scope accesses of the scratch counters (other waves of the workgroup wrote them through L2: never a stale L1 line)
__device__ __forceinline__ int32_t cbb_ld(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cbb_st(int32_t *p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Batch boundaries: ONE workgroup walks the run's columns in order (the greedy is sequential: a column joins the batch while the
// rows touched by more than one of its columns -- and their entries -- stay within the LDS budget), the entries of a column spread
// over the threads. cnt[n_rows]: zero on entry and on exit. bstart[0 .. n_batches]: first column of every batch, then n_run.
__global__ __launch_bounds__(1024) void k_cbb_bounds(const int64_t *__restrict__ colptr, const int32_t *__restrict__ rowidx,
                                                      const int32_t *__restrict__ run, int n_run, int hot_cap, int32_t *cnt,
                                                      int32_t *__restrict__ bstart, int32_t *__restrict__ n_batches) {
  __shared__ int ra[16], re[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int c = 0, n_hot = 0, n_ent = 0, nb = 0;
  auto close = [&](int c1) {  // the batch [c, c1): counters back to zero
    for (int k = c; k < c1; k++) {
      const int64_t b = colptr[run[k]], e = colptr[run[k] + 1];
      for (int64_t p = b + tid; p < e; p += 1024) cbb_st(&cnt[rowidx[p]], 0);
    }
    if (tid == 0) bstart[nb] = c;
    nb++;
    c = c1;
    n_hot = n_ent = 0;
    __syncthreads();
  };
  for (int e = 0; e < n_run; e++) {
    if (e - c == CHAINB_MAXCOLS) close(e);
    const int64_t b = colptr[run[e]], en = colptr[run[e] + 1];
    int add = 0, ent = 0;
    for (int64_t p = b + tid; p < en; p += 1024) {
      const int k = cbb_ld(&cnt[rowidx[p]]);
      add += k == 1;
      ent += k == 1 ? 2 : (k > 1 ? 1 : 0);
    }
    for (int off = 32; off > 0; off >>= 1) {
      add += __shfl_xor(add, off);
      ent += __shfl_xor(ent, off);
    }
    if (lane == 0) {
      ra[wv] = add;
      re[wv] = ent;
    }
    __syncthreads();
    add = ent = 0;
    for (int w = 0; w < 16; w++) {
      add += ra[w];
      ent += re[w];
    }
    __syncthreads();
    if (e > c && (n_hot + add > hot_cap || n_ent + ent > (5 * hot_cap) / 2)) {
      close(e);
      add = ent = 0;  // (no counter is set any more: the column opens the next batch)
    }
    for (int64_t p = b + tid; p < en; p += 1024) {
      int32_t *q = &cnt[rowidx[p]];
      cbb_st(q, cbb_ld(q) + 1);  // (a column's rows are distinct: one writer per counter)
    }
    n_hot += add;
    n_ent += ent;
    __syncthreads();
  }
  close(n_run);
  if (tid == 0) {
    bstart[nb] = n_run;
    *n_batches = nb;
  }
}

// Per batch (a workgroup takes batches b = blockIdx.x, + gridDim.x, ... with its own scratch counters): rows touched by more than
// one column of the batch = hot, numbered in ascending row order; every column's entries split, in order, into cold (row, local
// column, value) and hot (slot, value). WRITE = false: only the counts (hot rows per batch, cold / hot entries per column).
constexpr int CBB_MAX_HOT = 2048;
template <bool WRITE>
__global__ __launch_bounds__(1024) void k_cbb_batches(const int64_t *__restrict__ colptr, const int32_t *__restrict__ rowidx,
                                                       const double *__restrict__ cval, const int32_t *__restrict__ run,
                                                       const int32_t *__restrict__ bstart, int nb, int32_t *cnt_scr, int32_t *slot_scr,
                                                       int64_t n_rows, int32_t *__restrict__ n_hot_out, int32_t *__restrict__ ccnt,
                                                       int32_t *__restrict__ hcnt, const int32_t *__restrict__ cptr,
                                                       const int32_t *__restrict__ hptr, const int32_t *__restrict__ hot_row0,
                                                       int32_t *__restrict__ crow, int32_t *__restrict__ clcol, double *__restrict__ cx,
                                                       int32_t *__restrict__ hslot, double *__restrict__ hx, int32_t *__restrict__ hrows,
                                                       int *__restrict__ err) {
  __shared__ int32_t hl[CBB_MAX_HOT];
  __shared__ int n_hl;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int32_t *cnt = cnt_scr + (int64_t)blockIdx.x * n_rows, *slot = slot_scr + (int64_t)blockIdx.x * n_rows;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int b = blockIdx.x; b < nb; b += gridDim.x) {
    const int c0 = bstart[b], c1 = bstart[b + 1];
    if (tid == 0) n_hl = 0;
    for (int i = tid; i < CBB_MAX_HOT; i += 1024) hl[i] = 0x7fffffff;
    __syncthreads();
    for (int k = c0 + wv; k < c1; k += 16) {
      const int64_t pb = colptr[run[k]], pe = colptr[run[k] + 1];
      for (int64_t p = pb + lane; p < pe; p += 64) atomicAdd(&cnt[rowidx[p]], 1);
    }
    __syncthreads();
    for (int k = c0 + wv; k < c1; k += 16) {
      const int64_t pb = colptr[run[k]], pe = colptr[run[k] + 1];
      for (int64_t p = pb + lane; p < pe; p += 64) {
        const int32_t r = rowidx[p];
        if (cbb_ld(&cnt[r]) > 1 && atomicCAS(&slot[r], -1, -2) == -1) {
          const int i = atomicAdd(&n_hl, 1);
          if (i < CBB_MAX_HOT) hl[i] = r;
        }
      }
    }
    __syncthreads();
    const int nh = min(n_hl, CBB_MAX_HOT);
    if (n_hl > CBB_MAX_HOT && tid == 0) *err = 1;
    // hot rows in ascending order: bitonic sort of the padded list
    for (int size = 2; size <= CBB_MAX_HOT; size <<= 1)
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int t = tid; t < CBB_MAX_HOT / 2; t += 1024) {
          const int i = 2 * t - (t & (stride - 1)), j = i + stride;
          const bool up = (i & size) == 0;
          const int32_t x = hl[i], y = hl[j];
          if ((x > y) == up) {
            hl[i] = y;
            hl[j] = x;
          }
        }
        __syncthreads();
      }
    for (int i = tid; i < nh; i += 1024) {
      cbb_st(&slot[hl[i]], i);
      if (WRITE) hrows[hot_row0[b] + i] = hl[i];
    }
    if (!WRITE && tid == 0) n_hot_out[b] = nh;
    __syncthreads();
    for (int k = c0 + wv; k < c1; k += 16) {
      const int64_t pb = colptr[run[k]], pe = colptr[run[k] + 1];
      int cb = 0, hb = 0;
      for (int64_t p0 = pb; p0 < pe; p0 += 64) {
        const int64_t p = p0 + lane;
        const bool valid = p < pe;
        const int32_t r = valid ? rowidx[p] : 0;
        const bool hot = valid && cbb_ld(&cnt[r]) > 1;
        const unsigned long long m = __ballot(hot), v = __ballot(valid);
        if (WRITE) {
          if (hot) {
            const int q = hptr[k] + hb + __popcll(m & lt);
            hslot[q] = cbb_ld(&slot[r]);
            hx[q] = cval[p];
          } else if (valid) {
            const int q = cptr[k] + cb + __popcll(~m & v & lt);
            crow[q] = r;
            clcol[q] = k - c0;
            cx[q] = cval[p];
          }
        }
        hb += __popcll(m);
        cb += __popcll(~m & v);
      }
      if (!WRITE && lane == 0) {
        ccnt[k] = cb;
        hcnt[k] = hb;
      }
    }
    __syncthreads();
    for (int k = c0 + wv; k < c1; k += 16) {
      const int64_t pb = colptr[run[k]], pe = colptr[run[k] + 1];
      for (int64_t p = pb + lane; p < pe; p += 64) cbb_st(&cnt[rowidx[p]], 0);
    }
    for (int i = tid; i < nh; i += 1024) cbb_st(&slot[hl[i]], -1);
    __syncthreads();
  }
}

// Row-bucketed copies for the grid form (k_cb_step / k_cb_persist): per batch the cold entries ordered by (row range, class,
// column) -- class = (statistics near / far, update near / far): is the entry's row touched by the batch before / after? -- and the
// hot slots' row ranges. ONE workgroup, batch by batch (the neighbours' rows are stamped as the host form stamps them).
__global__ __launch_bounds__(1024) void k_cbb_buckets(const ChainBatch *__restrict__ bt, int nb, const int32_t *__restrict__ crow,
                                                       const int32_t *__restrict__ clcol, const double *__restrict__ cx,
                                                       const int32_t *__restrict__ hrows, int64_t n_rows, int split, int32_t *seen_prev,
                                                       int32_t *seen_next, int32_t *__restrict__ bptr, int32_t *__restrict__ hbptr,
                                                       int32_t *__restrict__ bcls, int32_t *__restrict__ brow, int32_t *__restrict__ blcol,
                                                       double *__restrict__ bx) {
  constexpr int NB = CB_BUCKETS, NK = NB * 4;
  __shared__ int cntk[NK + 1], cur[NK], wcnt[16][NK], hc[NB + 1];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const int64_t nr = n_rows > 0 ? n_rows : 1;
  auto stamp = [&](int32_t *seen, int b) {
    const ChainBatch B = bt[b];
    for (int p = B.cold_b + tid; p < B.cold_e; p += 1024) cbb_st(&seen[crow[p]], b);
    for (int k = tid; k < B.n_hot; k += 1024) cbb_st(&seen[hrows[B.hot_row0 + k]], b);
  };
  for (int b = 0; b < nb; b++) {
    const ChainBatch B = bt[b];
    const int cb = B.cold_b, ce = B.cold_e;
    if (b + 1 < nb) stamp(seen_next, b + 1);
    for (int i = tid; i <= NK; i += 1024) cntk[i] = 0;
    for (int i = tid; i <= NB; i += 1024) hc[i] = 0;
    __syncthreads();
    auto key_of = [&](int32_t r) {
      const bool sn = b == 0 || cbb_ld(&seen_prev[r]) == b - 1 || !(split & 1);
      const bool un = b + 1 == nb || cbb_ld(&seen_next[r]) == b + 1 || !(split & 2);
      const int ord = sn ? (un ? 1 : 2) : (un ? 0 : 3);
      return (int)(((int64_t)r * NB) / nr) * 4 + ord;
    };
    for (int p = cb + tid; p < ce; p += 1024) atomicAdd(&cntk[key_of(crow[p]) + 1], 1);
    for (int k = tid; k < B.n_hot; k += 1024) atomicAdd(&hc[(int)(((int64_t)hrows[B.hot_row0 + k] * NB) / nr) + 1], 1);
    __syncthreads();
    if (tid == 0) {
      cur[0] = cb;
      for (int k = 1; k < NK; k++) cur[k] = cur[k - 1] + cntk[k];
      int32_t *bp = bptr + (int64_t)b * (NB + 1);
      for (int k = 0; k < NB; k++) {
        bp[k] = cur[k * 4];
        for (int q = 0; q < 3; q++) bcls[((int64_t)b * NB + k) * 3 + q] = cur[k * 4 + q + 1];
      }
      bp[NB] = ce;
      int32_t *hp = hbptr + (int64_t)b * (NB + 1);
      hp[0] = 0;
      for (int k = 0; k < NB; k++) hp[k + 1] = hp[k] + hc[k + 1];  // (slots are in ascending row order)
    }
    __syncthreads();
    // entries arrive ordered by column: a stable scatter by key gives (range, class, column) order
    for (int base = cb; base < ce; base += 1024) {
      const int p = base + tid;
      const bool valid = p < ce;
      const int32_t r = valid ? crow[p] : 0;
      const int key = valid ? key_of(r) : -1;
      wcnt[wv][lane] = 0;
      int rank = 0;
      unsigned long long rem = __ballot(valid);
      while (rem) {
        const int l0 = __ffsll((long long)rem) - 1;
        const int k0 = __shfl(key, l0);
        const unsigned long long m = __ballot(valid && key == k0);
        if (valid && key == k0) rank = __popcll(m & lt);
        if (lane == l0) wcnt[wv][k0] = __popcll(m);
        rem &= ~m;
      }
      __syncthreads();
      if (valid) {
        int pos = cur[key] + rank;
        for (int w = 0; w < wv; w++) pos += wcnt[w][key];
        brow[pos] = r;
        blcol[pos] = clcol[p];
        bx[pos] = cx[p];
      }
      __syncthreads();
      if (tid < NK) {
        int t = 0;
        for (int w = 0; w < 16; w++) t += wcnt[w][tid];
        cur[tid] += t;
      }
      __syncthreads();
    }
    stamp(seen_prev, b);
    __syncthreads();
  }
}

struct ChainRun {
  DevBuf<ChainDesc> desc;  // (first CSC entry, length, column) per chain column, for k_chain_lds
  DevBuf<int32_t> cols;
  int n_cols = 0;
  int64_t nnz = 0;
  // conflict-batched form (k_chain_batched): per batch the rows touched by more than one of its columns ("hot")
  // are staged in LDS; entries are split into cold (sorted by column) and hot (slot into the batch's hot rows)
  bool batched = false;
  int n_batches = 0, max_hot = 0, max_hot_ent = 0;
  DevBuf<ChainBatch> batches;
  std::vector<ChainBatch> h_batches;   // host copies: the grid form (k_cb_*) launches batch by batch
  std::vector<int32_t> h_cold_cnt;     // cold entries per batch
  int64_t n_cold = 0;
  DevBuf<int32_t> cold_ptr, cold_row, cold_lcol, hot_ptr, hot_slot, hot_rows;
  DevBuf<double> cold_x, hot_x;
  // row-bucketed copies for k_cb_step: per batch the cold entries ordered by (row range, column) and the hot slots' row ranges
  bool bucketed = false;
  DevBuf<int32_t> bk_ptr, bk_row, bk_lcol, hbk_ptr, bk_cls;
  DevBuf<double> bk_x;
  mutable DevBuf<int32_t> col_group;             // group index of every chain column (filled at the first launch)
  mutable const int32_t *col_group_of = nullptr;  // ... from this group array
  // the streamed conflict-window form (k_cs_stream, mfm_chain_stream.hpp; round 5): the whole run as one pipelined launch without
  // batch boundaries -- built when the grid form applies and a window fits the walker's LDS; the two sweeps' plans share it
  std::shared_ptr<CsStream> stream;

  std::vector<int32_t> h_cols;  // the run's columns (a twin plan of the same matrix asks: the same run?)
  // the other sweep's plan of the SAME matrix has this run already: non-owning views of its arrays (o outlives this)
  void borrow(const ChainRun &o) {
    desc.borrow(o.desc);
    cols.borrow(o.cols);
    n_cols = o.n_cols;
    nnz = o.nnz;
    batched = o.batched;
    n_batches = o.n_batches;
    max_hot = o.max_hot;
    max_hot_ent = o.max_hot_ent;
    batches.borrow(o.batches);
    h_batches = o.h_batches;
    h_cold_cnt = o.h_cold_cnt;
    n_cold = o.n_cold;
    cold_ptr.borrow(o.cold_ptr);
    cold_row.borrow(o.cold_row);
    cold_lcol.borrow(o.cold_lcol);
    hot_ptr.borrow(o.hot_ptr);
    hot_slot.borrow(o.hot_slot);
    hot_rows.borrow(o.hot_rows);
    cold_x.borrow(o.cold_x);
    hot_x.borrow(o.hot_x);
    bucketed = o.bucketed;
    bk_ptr.borrow(o.bk_ptr);
    bk_row.borrow(o.bk_row);
    bk_lcol.borrow(o.bk_lcol);
    hbk_ptr.borrow(o.hbk_ptr);
    bk_cls.borrow(o.bk_cls);
    bk_x.borrow(o.bk_x);
    stream = o.stream;
    h_cols = o.h_cols;
  }

  // the same structures built on the device from the device CSC (dv.colptr / rowidx / cval); false: not built (the caller takes
  // the host form)
  bool build_batched_device(const DevCscView &dv, const std::vector<int32_t> &run, int hot_cap) {
    const int64_t n_rows = dv.n_rows;
    const int n_run = (int)run.size();
    if (!dv.colptr || !dv.rowidx || !dv.cval || !cols.p || n_run < 1 || n_rows < 1 || hot_cap > CBB_MAX_HOT) return false;
    if (n_rows * 8 > ((int64_t)1 << 30)) return false;  // (scratch counters: 8 bytes per row and workgroup)
    hipStream_t s = dv.stream;
    auto dl = [&](const int32_t *p, size_t n) {
      std::vector<int32_t> h(n);
      if (n) MFM_HIP_CHECK(hipMemcpyAsync(h.data(), p, n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
      MFM_HIP_CHECK(hipStreamSynchronize(s));
      return h;
    };
    // 1. batch boundaries
    DevBuf<int32_t> cnt1, bstart, nbat;
    cnt1.alloc_zero((size_t)n_rows, s);
    bstart.alloc((size_t)n_run + 2);
    nbat.alloc_zero(1, s);
    hipLaunchKernelGGL(k_cbb_bounds, dim3(1), dim3(1024), 0, s, dv.colptr, dv.rowidx, cols.p, n_run, hot_cap, cnt1.p, bstart.p, nbat.p);
    const int nb = dl(nbat.p, 1)[0];
    const std::vector<int32_t> h_bstart = dl(bstart.p, (size_t)nb + 1);
    // 2. per batch: counts, then the lists
    const int nwg = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)nb, 64, ((int64_t)1 << 28) / (8 * n_rows)}));
    DevBuf<int32_t> cnt_scr, slot_scr, d_nhot, ccnt, hcnt;
    DevBuf<int> err;
    cnt_scr.alloc_zero((size_t)nwg * n_rows, s);
    slot_scr.alloc((size_t)nwg * n_rows);
    MFM_HIP_CHECK(hipMemsetAsync(slot_scr.p, 0xff, (size_t)nwg * n_rows * sizeof(int32_t), s));
    d_nhot.alloc((size_t)nb);
    ccnt.alloc((size_t)n_run);
    hcnt.alloc((size_t)n_run);
    err.alloc_zero(1, s);
    hipLaunchKernelGGL((k_cbb_batches<false>), dim3(nwg), dim3(1024), 0, s, dv.colptr, dv.rowidx, dv.cval, cols.p, bstart.p, nb, cnt_scr.p,
                       slot_scr.p, n_rows, d_nhot.p, ccnt.p, hcnt.p, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                       nullptr, err.p);
    const std::vector<int32_t> h_nhot = dl(d_nhot.p, (size_t)nb), h_cc = dl(ccnt.p, (size_t)n_run), h_hc = dl(hcnt.p, (size_t)n_run);
    {
      int h_err = 0;
      MFM_HIP_CHECK(hipMemcpy(&h_err, err.p, sizeof(int), hipMemcpyDeviceToHost));
      if (h_err) return false;
    }
    std::vector<int32_t> cptr((size_t)n_run + 1, 0), hptr((size_t)n_run + 1, 0), hot_row0((size_t)nb + 1, 0);
    for (int k = 0; k < n_run; k++) {
      const int64_t c = (int64_t)cptr[k] + h_cc[k], h = (int64_t)hptr[k] + h_hc[k];
      if (c >= (int64_t)2147483647 || h >= (int64_t)2147483647) return false;
      cptr[k + 1] = (int32_t)c;
      hptr[k + 1] = (int32_t)h;
    }
    std::vector<ChainBatch> bt((size_t)nb);
    max_hot = max_hot_ent = 0;
    for (int b = 0; b < nb; b++) {
      hot_row0[b + 1] = hot_row0[b] + h_nhot[b];
      ChainBatch &B = bt[b];
      B.col0 = h_bstart[b];
      B.ncols = h_bstart[b + 1] - h_bstart[b];
      B.hot_row0 = hot_row0[b];
      B.n_hot = h_nhot[b];
      B.cold_b = cptr[B.col0];
      B.cold_e = cptr[B.col0 + B.ncols];
      B.hot_b = hptr[B.col0];
      B.hot_e = hptr[B.col0 + B.ncols];
      max_hot = std::max(max_hot, B.n_hot);
      max_hot_ent = std::max(max_hot_ent, B.hot_e - B.hot_b);
    }
    n_batches = nb;
    h_batches = bt;
    h_cold_cnt.clear();
    for (const ChainBatch &B : bt) h_cold_cnt.push_back(B.cold_e - B.cold_b);
    n_cold = cptr[n_run];
    const int64_t n_hot_ent = hptr[n_run];
    batches.upload(bt.data(), bt.size());
    cold_ptr.upload(cptr);
    hot_ptr.upload(hptr);
    DevBuf<int32_t> d_hot_row0;
    d_hot_row0.upload(hot_row0);
    cold_row.alloc((size_t)n_cold);
    cold_lcol.alloc((size_t)n_cold);
    cold_x.alloc((size_t)n_cold);
    hot_slot.alloc((size_t)n_hot_ent);
    hot_x.alloc((size_t)n_hot_ent);
    hot_rows.alloc((size_t)hot_row0[nb]);
    hipLaunchKernelGGL((k_cbb_batches<true>), dim3(nwg), dim3(1024), 0, s, dv.colptr, dv.rowidx, dv.cval, cols.p, bstart.p, nb, cnt_scr.p,
                       slot_scr.p, n_rows, nullptr, nullptr, nullptr, cold_ptr.p, hot_ptr.p, d_hot_row0.p, cold_row.p, cold_lcol.p, cold_x.p,
                       hot_slot.p, hot_x.p, hot_rows.p, err.p);
    MFM_HIP_CHECK(hipGetLastError());
    batched = true;
    bucketed = false;
    // 3. the grid form's row-bucketed copies
    const int64_t grid_min = std::getenv("MFM_CHAIN_GRID_MIN") ? std::atoll(std::getenv("MFM_CHAIN_GRID_MIN")) : 4096;
    if (n_batches > 0 && n_cold / n_batches >= grid_min && !std::getenv("MFM_NO_CHAIN_GRID") && !std::getenv("MFM_NO_CB_MERGE")) {
      constexpr int NB = CB_BUCKETS;
      const int split = std::getenv("MFM_NO_CB_PERSIST") ? 0 : std::getenv("MFM_CB_SPLIT") ? std::atoi(std::getenv("MFM_CB_SPLIT")) : 3;
      DevBuf<int32_t> seen_prev, seen_next;
      seen_prev.alloc((size_t)n_rows);
      seen_next.alloc((size_t)n_rows);
      MFM_HIP_CHECK(hipMemsetAsync(seen_prev.p, 0xff, (size_t)n_rows * sizeof(int32_t), s));
      MFM_HIP_CHECK(hipMemsetAsync(seen_next.p, 0xff, (size_t)n_rows * sizeof(int32_t), s));
      bk_ptr.alloc((size_t)n_batches * (NB + 1));
      hbk_ptr.alloc((size_t)n_batches * (NB + 1));
      bk_cls.alloc((size_t)n_batches * NB * 3);
      bk_row.alloc((size_t)n_cold);
      bk_lcol.alloc((size_t)n_cold);
      bk_x.alloc((size_t)n_cold);
      hipLaunchKernelGGL(k_cbb_buckets, dim3(1), dim3(1024), 0, s, batches.p, n_batches, cold_row.p, cold_lcol.p, cold_x.p, hot_rows.p, n_rows,
                         split, seen_prev.p, seen_next.p, bk_ptr.p, hbk_ptr.p, bk_cls.p, bk_row.p, bk_lcol.p, bk_x.p);
      MFM_HIP_CHECK(hipGetLastError());
      MFM_HIP_CHECK(hipStreamSynchronize(s));
      bucketed = true;
    }
    MFM_HIP_CHECK(hipStreamSynchronize(s));
    return true;
  }
  // tests (MFM_PLAN_CHECK): every array of two builds of the same run
  std::string compare_batched(const ChainRun &o, hipStream_t s) const {
    if (batched != o.batched || n_batches != o.n_batches || max_hot != o.max_hot || max_hot_ent != o.max_hot_ent || n_cold != o.n_cold ||
        bucketed != o.bucketed || h_cold_cnt != o.h_cold_cnt)
      return "scalars";
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
#define MFM_CB_CMP(f) \
  if (!same(f.p, f.n, o.f.p, o.f.n, sizeof(*f.p))) return #f;
    MFM_CB_CMP(batches)
    MFM_CB_CMP(cold_ptr)
    MFM_CB_CMP(cold_row)
    MFM_CB_CMP(cold_lcol)
    MFM_CB_CMP(cold_x)
    MFM_CB_CMP(hot_ptr)
    MFM_CB_CMP(hot_slot)
    MFM_CB_CMP(hot_x)
    MFM_CB_CMP(hot_rows)
    MFM_CB_CMP(bk_ptr)
    MFM_CB_CMP(hbk_ptr)
    MFM_CB_CMP(bk_cls)
    MFM_CB_CMP(bk_row)
    MFM_CB_CMP(bk_lcol)
    MFM_CB_CMP(bk_x)
#undef MFM_CB_CMP
    return "";
  }

  void build_batched(const HostCsr &csc, const std::vector<int32_t> &run, int hot_cap) {
    const int64_t n_rows = csc.cols;
    std::vector<int32_t> cnt((size_t)n_rows, 0), slot_of((size_t)n_rows, -1);
    std::vector<ChainBatch> bt;
    std::vector<int32_t> cptr{0}, crow, clcol, hptr{0}, hslot, hrows;
    std::vector<double> cx, hx;
    size_t c = 0;
    while (c < run.size()) {
      // grow the batch while the number of hot rows stays within the LDS budget
      size_t e = c;
      int n_hot = 0, n_hot_ent = 0;
      std::vector<int32_t> touched;
      while (e < run.size() && (int)(e - c) < CHAINB_MAXCOLS) {
        const int32_t j = run[e];
        int add = 0, add_ent = 0;
        for (int64_t p = csc.ptr[j]; p < csc.ptr[j + 1]; p++) {
          const int k = cnt[csc.idx[p]];
          add += k == 1;
          add_ent += k == 1 ? 2 : (k > 1 ? 1 : 0);
        }
        if (e > c && (n_hot + add > hot_cap || n_hot_ent + add_ent > (5 * hot_cap) / 2)) break;
        for (int64_t p = csc.ptr[j]; p < csc.ptr[j + 1]; p++) {
          if (cnt[csc.idx[p]]++ == 0) touched.push_back(csc.idx[p]);
        }
        n_hot += add;
        n_hot_ent += add_ent;
        e++;
      }
      ChainBatch B;
      B.col0 = (int32_t)c;
      B.ncols = (int32_t)(e - c);
      B.hot_row0 = (int32_t)hrows.size();
      int ns = 0;
      {  // hot slots in ascending row order (k_cb_step splits them by row range)
        std::vector<int32_t> hr;
        for (int32_t r : touched)
          if (cnt[r] > 1) hr.push_back(r);
        std::sort(hr.begin(), hr.end());
        for (int32_t r : hr) {
          slot_of[r] = ns++;
          hrows.push_back(r);
        }
      }
      B.n_hot = ns;
      for (size_t k = c; k < e; k++) {
        const int32_t j = run[k];
        for (int64_t p = csc.ptr[j]; p < csc.ptr[j + 1]; p++) {
          const int32_t r = csc.idx[p];
          if (cnt[r] > 1) {
            hslot.push_back(slot_of[r]);
            hx.push_back(csc.val[p]);
          } else {
            crow.push_back(r);
            clcol.push_back((int32_t)(k - c));
            cx.push_back(csc.val[p]);
          }
        }
        cptr.push_back((int32_t)crow.size());
        hptr.push_back((int32_t)hslot.size());
      }
      B.cold_e = (int32_t)crow.size();
      B.hot_e = (int32_t)hslot.size();
      B.cold_b = cptr[c];
      B.hot_b = hptr[c];
      for (int32_t r : touched) {
        cnt[r] = 0;
        slot_of[r] = -1;
      }
      max_hot = std::max(max_hot, ns);
      max_hot_ent = std::max(max_hot_ent, hptr.back() - hptr[hptr.size() - 1 - (e - c)]);
      bt.push_back(B);
      c = e;
    }
    n_batches = (int)bt.size();
    h_batches = bt;
    h_cold_cnt.clear();
    for (const ChainBatch &B : bt) h_cold_cnt.push_back(cptr[B.col0 + B.ncols] - cptr[B.col0]);
    n_cold = (int64_t)crow.size();
    batches.upload(bt.data(), bt.size());
    cold_ptr.upload(cptr);
    cold_row.upload(crow);
    cold_lcol.upload(clcol);
    cold_x.upload(cx);
    hot_ptr.upload(hptr);
    hot_slot.upload(hslot);
    hot_x.upload(hx);
    hot_rows.upload(hrows);
    batched = true;
    // the grid form (k_cb_*) also gets the row-bucketed copies
    const int64_t grid_min = std::getenv("MFM_CHAIN_GRID_MIN") ? std::atoll(std::getenv("MFM_CHAIN_GRID_MIN")) : 4096;
    if (n_batches > 0 && n_cold / n_batches >= grid_min && !std::getenv("MFM_NO_CHAIN_GRID") && !std::getenv("MFM_NO_CB_MERGE")) {
      constexpr int NB = CB_BUCKETS;
      auto bucket = [&](int32_t r) { return (int)(((int64_t)r * NB) / std::max<int64_t>(n_rows, 1)); };
      std::vector<int32_t> bptr((size_t)n_batches * (NB + 1), 0), hbptr((size_t)n_batches * (NB + 1), 0);
      std::vector<int32_t> bcls((size_t)n_batches * NB * 3, 0);
      std::vector<int32_t> brow(crow.size()), blcol(crow.size());
      std::vector<double> bx(crow.size());
      // Classes of a cold entry of batch b (k_cb_persist): "statistics near" = its row is touched by batch b - 1 (every entry of
      // the first batch), "update near" = its row is touched by batch b + 1 (every entry of the last batch). Inside a (batch,
      // range) the entries are ordered by class -- (far, near), (near, near), (near, far), (far, far) as (statistics, update) --
      // and by column inside a class, so that "update near", "statistics near" and "update far" are contiguous and "statistics
      // far" is the two ends.
      std::vector<int32_t> seen_prev((size_t)n_rows, -1), seen_next((size_t)n_rows, -1);
      // (the two-launch form, MFM_NO_CB_PERSIST, walks a range as ONE column-ordered run: a single class)
      const int split = std::getenv("MFM_NO_CB_PERSIST") ? 0 : std::getenv("MFM_CB_SPLIT") ? std::atoi(std::getenv("MFM_CB_SPLIT")) : 3;  // (bit 0: statistics, bit 1: update)
      auto stamp = [&](std::vector<int32_t> &seen, int b) {
        const ChainBatch &B = bt[b];
        for (int p = B.cold_b; p < B.cold_e; p++) seen[crow[p]] = b;
        for (int k = 0; k < B.n_hot; k++) seen[hrows[B.hot_row0 + k]] = b;
      };
      for (int b = 0; b < n_batches; b++) {
        const ChainBatch &B = bt[b];
        const int cb = B.cold_b, ce = B.cold_e;
        if (b + 1 < n_batches) stamp(seen_next, b + 1);
        auto order_of = [&](int32_t r) {
          const bool sn = b == 0 || seen_prev[r] == b - 1 || !(split & 1), un = b + 1 == n_batches || seen_next[r] == b + 1 || !(split & 2);
          return sn ? (un ? 1 : 2) : (un ? 0 : 3);
        };
        int cntb[NB * 4 + 1] = {0};
        for (int p = cb; p < ce; p++) cntb[bucket(crow[p]) * 4 + order_of(crow[p]) + 1]++;
        int32_t *bp = bptr.data() + (size_t)b * (NB + 1);
        int cur[NB * 4];
        cur[0] = cb;
        for (int k = 1; k < NB * 4; k++) cur[k] = cur[k - 1] + cntb[k];
        for (int k = 0; k < NB; k++) {
          bp[k] = cur[k * 4];
          for (int q = 0; q < 3; q++) bcls[((size_t)b * NB + k) * 3 + q] = cur[k * 4 + q + 1];
        }
        bp[NB] = ce;
        for (int p = cb; p < ce; p++) {  // (entries arrive ordered by column: stable => (range, class, column) order)
          const int q = cur[bucket(crow[p]) * 4 + order_of(crow[p])]++;
          brow[q] = crow[p];
          blcol[q] = clcol[p];
          bx[q] = cx[p];
        }
        stamp(seen_prev, b);
        int32_t *hp = hbptr.data() + (size_t)b * (NB + 1);
        int hc[NB + 1] = {0};
        for (int k = 0; k < B.n_hot; k++) hc[bucket(hrows[B.hot_row0 + k]) + 1]++;
        hp[0] = 0;
        for (int k = 0; k < NB; k++) hp[k + 1] = hp[k] + hc[k + 1];  // (slots are in ascending row order)
      }
      bk_cls.upload(bcls);
      bk_ptr.upload(bptr);
      hbk_ptr.upload(hbptr);
      bk_row.upload(brow);
      bk_lcol.upload(blcol);
      bk_x.upload(bx);
      bucketed = true;
    }
  }
};

struct Step {
  bool is_chain = false;
  ParLevel par;
  ChainRun chain;
};

// ---- row-tile layout of a scattered level built ON THE DEVICE --------------------------------------------------
// The same structure build_tiled() makes on the host (entries of the level sorted by (tile, column, row), padded per tile to
// whole wavefronts, runs per wave tile, column-major slot positions), from the device-resident CSC: a stable radix sort of
// (tile << cbits | column) keys over the level's entries in (column, row) order, lower bounds for the tile / column
// boundaries, an exclusive scan of the runs per wave tile, a second stable sort of the runs by column.
// Level schedule (level(j) = 1 + max level of earlier columns sharing a row with j) as the least fixed point of
// level[c_k] >= level[c_{k-1}] + 1 over consecutive stored columns of every row: row-parallel relaxation passes with atomicMax
// until nothing changes -- as many passes as there are levels: two or three for one-hot designs (deep schedules: k_dp_level_seq).
// flags[0]: something changed; flags[1]: a row's column indices are not ascending (host schedule instead).
__global__ void k_dp_level_relax(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, int64_t N, int ell,
                                 int32_t *__restrict__ level, int *__restrict__ flags) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N) return;
  const int64_t b = ell >= 0 ? t * ell : rowptr[t], e = ell >= 0 ? b + ell : rowptr[t + 1];
  int prev_c = -1, prev_l = 0;
  for (int64_t p = b; p < e; p++) {
    const int c = colidx[p];
    if (c < prev_c) flags[1] = 1;
    if (c == prev_c) continue;
    int l = level[c];
    if (prev_c >= 0 && l < prev_l + 1) {
      l = prev_l + 1;
      atomicMax(&level[c], l);
      flags[0] = 1;
    }
    prev_c = c;
    prev_l = l;
  }
}
// Deep schedules: the reference order itself, one workgroup walking the columns in index order -- level(j) = 1 + the
// largest level an earlier column left on one of j's rows -- with the column's entries spread over the threads.
__global__ __launch_bounds__(1024) void k_dp_level_seq(const int64_t *__restrict__ colptr, const int32_t *__restrict__ rowidx, int n_cols,
                                                        int32_t *rowlevel, int32_t *__restrict__ level) {
  __shared__ int32_t red[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int j = 0; j < n_cols; j++) {
    const int64_t b = colptr[j], e = colptr[j + 1];
    int32_t m = -1;
    for (int64_t p = b + tid; p < e; p += 1024) m = max(m, cbb_ld(&rowlevel[rowidx[p]]));  // (written by other waves: device scope)
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off));
    if (lane == 0) red[wv] = m;
    __syncthreads();
    m = red[lane & 15];
    for (int off = 8; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off));
    const int32_t lv = m + 1;
    for (int64_t p = b + tid; p < e; p += 1024) cbb_st(&rowlevel[rowidx[p]], lv);
    if (tid == 0) level[j] = lv;
    __syncthreads();
  }
}
static inline bool column_levels_device(const DevCscView &dv, std::vector<int32_t> &level, int32_t &n_levels) {
  if (!dv.colidx || dv.n_cols <= 0 || dv.n_rows <= 0 || (dv.ell < 0 && !dv.rowptr)) return false;
  DevBuf<int32_t> lv;
  DevBuf<int> fl;
  lv.alloc((size_t)dv.n_cols);
  fl.alloc(2);
  MFM_HIP_CHECK(hipMemsetAsync(lv.p, 0, (size_t)dv.n_cols * sizeof(int32_t), dv.stream));
  MFM_HIP_CHECK(hipMemsetAsync(fl.p, 0, 2 * sizeof(int), dv.stream));
  // (a pass settles at least one more level of every dependency path: the shallow schedules of one-hot designs are done after two
  //  or three. A schedule that is still moving after 8 passes is a deep one -- multi-hot relation blocks: up to a level per
  //  column -- and goes to the column-sequential kernel below: one launch, exact, a few microseconds per column.)
  bool done = false;
  for (int pass = 0, batch = 2; pass < 8 && !done; pass += batch, batch = 6) {
    for (int k = 0; k < batch; k++) {  // fl[0]: did the LAST pass of the batch change anything?
      if (k == batch - 1) MFM_HIP_CHECK(hipMemsetAsync(fl.p, 0, sizeof(int), dv.stream));
      hipLaunchKernelGGL(k_dp_level_relax, dim3((unsigned)((dv.n_rows + 255) / 256)), dim3(256), 0, dv.stream, dv.rowptr, dv.colidx,
                         dv.n_rows, dv.ell, lv.p, fl.p);
    }
    int h[2] = {0, 0};
    MFM_HIP_CHECK(hipMemcpyAsync(h, fl.p, 2 * sizeof(int), hipMemcpyDeviceToHost, dv.stream));
    MFM_HIP_CHECK(hipStreamSynchronize(dv.stream));
    if (h[1]) return false;
    done = h[0] == 0;
  }
  if (!done) {
    if (!dv.colptr || !dv.rowidx || dv.n_cols > ((int64_t)1 << 17)) return false;  // (host schedule)
    DevBuf<int32_t> rowlevel;
    rowlevel.alloc((size_t)dv.n_rows);
    MFM_HIP_CHECK(hipMemsetAsync(rowlevel.p, 0xff, (size_t)dv.n_rows * sizeof(int32_t), dv.stream));
    hipLaunchKernelGGL(k_dp_level_seq, dim3(1), dim3(1024), 0, dv.stream, dv.colptr, dv.rowidx, (int)dv.n_cols, rowlevel.p, lv.p);
    MFM_HIP_CHECK(hipStreamSynchronize(dv.stream));
  }
  level.resize((size_t)dv.n_cols);
  MFM_HIP_CHECK(hipMemcpy(level.data(), lv.p, (size_t)dv.n_cols * sizeof(int32_t), hipMemcpyDeviceToHost));
  n_levels = 0;
  for (int32_t l : level) n_levels = std::max(n_levels, l + 1);
  return true;
}
// First look at a level (one wavefront per column of it, on the device CSC): entries that do not follow their predecessor's
// row directly (0: every column is a contiguous row range), entries 8 or more rows after it (the scatter decision), rows seen
// more than once (seen != null: does the level touch every row exactly once?). out[3], zeroed by the caller.
__global__ __launch_bounds__(WG) void k_dp_level_scan(const int64_t *__restrict__ colptr, const int32_t *__restrict__ rowidx,
                                                       const int32_t *__restrict__ cols, int n_cols, int32_t *__restrict__ seen,
                                                       unsigned long long *__restrict__ out) {
  const int c = blockIdx.x * (WG / WAVE) + (threadIdx.x >> 6);
  if (c >= n_cols) return;
  const int lane = threadIdx.x & 63;
  const int64_t b = colptr[cols[c]], e = colptr[cols[c] + 1];
  unsigned long long gap = 0, far = 0, twice = 0;
  for (int64_t p = b + lane; p < e; p += WAVE) {
    const int32_t r = rowidx[p];
    if (p > b) {
      const int32_t d = r - rowidx[p - 1];
      gap += d != 1;
      far += d >= 8;
    }
    if (seen) twice += atomicAdd(&seen[r], 1) > 0;
  }
  for (int off = 32; off > 0; off >>= 1) {
    gap += __shfl_xor(gap, off);
    far += __shfl_xor(far, off);
    twice += __shfl_xor(twice, off);
  }
  if (lane == 0) {
    if (gap) atomicAdd(&out[0], gap);
    if (far) atomicAdd(&out[1], far);
    if (twice) atomicAdd(&out[2], twice);
  }
}
struct LevelScan {
  bool valid = false;
  int64_t gaps = 0, far = 0, twice = 0;
};
static inline LevelScan level_scan_device(const DevCscView &dv, const int32_t *d_cols, int n_cols, bool want_once) {
  LevelScan r;
  if (!dv.colptr || !dv.rowidx || n_cols <= 0) return r;
  DevBuf<unsigned long long> out;
  DevBuf<int32_t> seen;
  out.alloc_zero(3, dv.stream);
  if (want_once) seen.alloc_zero((size_t)std::max<int64_t>(dv.n_rows, 1), dv.stream);
  hipLaunchKernelGGL(k_dp_level_scan, dim3((unsigned)((n_cols + WG / WAVE - 1) / (WG / WAVE))), dim3(WG), 0, dv.stream, dv.colptr, dv.rowidx,
                     d_cols, n_cols, seen.p, out.p);
  unsigned long long h[3] = {0, 0, 0};
  MFM_HIP_CHECK(hipMemcpyAsync(h, out.p, sizeof h, hipMemcpyDeviceToHost, dv.stream));
  MFM_HIP_CHECK(hipStreamSynchronize(dv.stream));
  r.valid = true;
  r.gaps = (int64_t)h[0];
  r.far = (int64_t)h[1];
  r.twice = (int64_t)h[2];
  return r;
}
__device__ __forceinline__ int dp_upper_tile(const int32_t *tstart, int nb, int r) {  // tile b with tstart[b] <= r < tstart[b + 1]
  int lo = 0, hi = nb;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tstart[mid] <= r) lo = mid; else hi = mid;
  }
  return lo;
}
// one wavefront per level column: key / payload of its entries at the column's offset in the level's entry list
__global__ __launch_bounds__(WG) void k_dp_keys(const int64_t *__restrict__ colptr, const int32_t *__restrict__ rowidx,
                                                 const int32_t *__restrict__ cols, const int32_t *__restrict__ lvl_ptr, int n_cols,
                                                 const int32_t *__restrict__ tstart, int nb, int cbits,
                                                 uint64_t *__restrict__ key, uint32_t *__restrict__ pos) {
  const int c = blockIdx.x * (WG / WAVE) + (threadIdx.x >> 6);
  if (c >= n_cols) return;
  const int lane = threadIdx.x & 63;
  const int64_t b = colptr[cols[c]], e = colptr[cols[c] + 1];
  const int32_t o = lvl_ptr[c];
  for (int64_t p = b + lane; p < e; p += WAVE) {
    const int t = dp_upper_tile(tstart, nb, rowidx[p]);
    key[o + (p - b)] = ((uint64_t)t << cbits) | (uint32_t)c;
    pos[o + (p - b)] = (uint32_t)p;
  }
}
template <class T>
__global__ void k_dp_lower_bound(const T *__restrict__ sorted, int64_t n, int shift, int n_q, int32_t *__restrict__ out) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;  // first position whose (key >> shift) >= q
  if (q >= n_q) return;
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)(sorted[mid] >> shift) < (int64_t)q) lo = mid + 1; else hi = mid;
  }
  out[q] = (int32_t)lo;
}
__global__ void k_dp_fill(const uint64_t *__restrict__ key, const uint32_t *__restrict__ pos, int64_t n, int cbits, int tile_bits,
                          const int32_t *__restrict__ tile_begin, const int32_t *__restrict__ tptr, const int32_t *__restrict__ tstart,
                          const int32_t *__restrict__ rowidx, const double *__restrict__ cval, uint32_t *__restrict__ tent,
                          double *__restrict__ ev) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint64_t k = key[p];
  const int b = (int)(k >> cbits);
  const uint32_t c = (uint32_t)(k & (((uint64_t)1 << cbits) - 1));
  const int64_t q = (int64_t)tptr[b] * WAVE + (p - tile_begin[b]);
  const uint32_t cp = pos[p];
  tent[q] = (c << tile_bits) | (uint32_t)(rowidx[cp] - tstart[b]);
  if (ev) ev[q] = cval[cp];
}
// runs of one column inside a wave tile: count (pass 0) / column of every run at run_base[t] + rank (pass 1)
__global__ __launch_bounds__(WG) void k_dp_runs(const uint32_t *__restrict__ tent, int64_t n_wt, int tile_bits,
                                                 int32_t *__restrict__ run_cnt, const int32_t *__restrict__ run_base,
                                                 int32_t *__restrict__ run_col) {
  const int64_t t = (int64_t)blockIdx.x * (WG / WAVE) + (threadIdx.x >> 6);
  if (t >= n_wt) return;
  const int lane = threadIdx.x & 63;
  const uint32_t u = tent[t * WAVE + lane];
  const uint32_t prev = __shfl_up(u, 1, WAVE);
  const bool head = u != TILE_PAD && (lane == 0 || (prev >> tile_bits) != (u >> tile_bits));
  const unsigned long long m = __ballot(head);
  if (run_col) {
    if (head) run_col[run_base[t] + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)(u >> tile_bits);
  } else if (lane == 0) {
    run_cnt[t] = __popcll(m);
  }
}
__global__ void k_dp_iota(uint32_t *__restrict__ p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (uint32_t)i;
}
__global__ void k_dp_invert(const uint32_t *__restrict__ sidx, int64_t n, int32_t *__restrict__ slot_pos) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) slot_pos[sidx[k]] = (int32_t)k;
}
__global__ void k_dp_row_once(const uint64_t *__restrict__ key, const uint32_t *__restrict__ pos, int64_t n,
                              const int32_t *__restrict__ rowidx, int32_t *__restrict__ cnt, int *__restrict__ dup) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  if (atomicAdd(&cnt[rowidx[pos[p]]], 1) != 0) *dup = 1;
}

constexpr size_t CHAIN_LDS_MAX = 156 * 1024;  // of the CU's 160 KiB

struct StepPlan {
  std::vector<Step> steps;
  int n_levels = 0;
  int64_t n_state_rows = 0;  // rows of the table whose state the sweeps touch (= csc.cols)
  int max_hchunks = 0, max_huge = 0;
  int64_t launches = 0;
  DevBuf<int32_t> col_row0;  // first row of every column (used by levels with contiguous columns)
  // Row tiles aligned to the columns of a contiguous first level (every such column of <= 2^tile_bits rows lies
  // inside one tile; longer columns get tiles of their own): lets the apply pass of the last level run the
  // first level of the NEXT factor on the tile while it is in LDS (k_tile_apply_next).
  std::vector<int32_t> h_tile_start;
  std::vector<int32_t> h_level;  // level of every column (kept for a twin plan)
  DevBuf<int32_t> fuse_cols, fuse_col_ptr;  // first-level columns inside each tile, in row order
  DevBuf<int4> fuse_desc;                   // per entry of fuse_cols: {column, length, first row - tile start, group}
  const std::vector<int32_t> *group_of = nullptr;  // (set by the owner before build: group index per column)
  DevBuf<int32_t> solo_tiles;               // tiles of special first-level columns (row-sharded mode): swept by
  int n_solo_tiles = 0;                     // run_level_sharded, statistics by k_tile_stats on this list
  // first-level columns longer than a tile (complete on this rank): two passes over their tiles
  DevBuf<int32_t> long_cols, long_tile_ptr, long_tiles, solo_col, tile_long_idx;
  DevBuf<double2> long_partial, oldnew_long;
  int n_long_cols = 0, n_long_tiles = 0;
  bool aligned_tiles = false;
  // two-field pass (mfm_mf_kernels.hpp): per 64-row chunk of every tile the first-level column heads, per column of
  // fuse_desc its partials
  DevBuf<MfChunk> mf_chunk;
  DevBuf<int32_t> mf_chunk_ptr;
  DevBuf<int2> mf_upart;
  bool mf_ready = false;
  int mf_max_users = 0;  // most first-level columns with rows in one tile

  // a level is "tiny" when running it as its own launches cannot fill the device anyway
  static bool tiny(size_t n_cols, int64_t nnz) { return n_cols <= 8 && nnz <= 16384; }

  // r_w16: entries per lane of the 16-wide wavefront bin (0: the policy has none); r_wg: entries per
  // thread of the workgroup bin; coop_max: workgroups that are safely co-resident on the device
  int64_t max_cols_scat = 0;

  // Row-blocked layout for a level whose columns jump between far-apart rows. Returns false (and leaves
  // L untouched) when the level is small or its columns are mostly contiguous.
  // tile_bits > 0: row-tile variant (tiles of 2^tile_bits rows staged in LDS, packed + padded entries)
  static bool build_scattered(const HostCsr &csc, const std::vector<int32_t> &cols, int64_t lnnz, bool unit, ParLevel &L,
                              int tile_bits = 0, const std::vector<int32_t> *bounds = nullptr, const DevCscView *dev = nullptr,
                              const LevelScan *scan = nullptr) {
    int64_t min_nnz = 1 << 16;  // (measured: the tile path and the fusions it enables win from ~10^5 entries per level on)
    if (const char *e = std::getenv("MFM_SCATTER_MIN_NNZ")) min_nnz = std::atoll(e);
    if (lnnz < min_nnz) return false;
    if (const char *e = std::getenv("MFM_NO_SCATTER"))
      if (std::atoi(e)) return false;
    int64_t far = 0;
    if (scan && scan->valid) {
      far = scan->far;  // (counted on the device: level_scan_device)
    } else {
      for (int32_t j : cols)
        for (int64_t p = csc.ptr[j] + 1; p < csc.ptr[j + 1]; p++) far += (csc.idx[p] - csc.idx[p - 1]) >= 8;
    }
    if ((double)far < 0.5 * (double)lnnz) return false;
    const int64_t N = csc.cols;
    if (tile_bits > 0) {
      // worthwhile only when the level touches most rows (the apply pass rewrites whole tiles), and the
      // packed entry must hold the column's position in the level
      if (2 * lnnz < N || (int64_t)cols.size() >= ((int64_t)1 << (32 - tile_bits)) - 1) tile_bits = 0;
    }
    if (tile_bits > 0) {
      if (dev && dev->colptr && !std::getenv("MFM_HOST_TILE_PACK") &&
          build_tiled_device(*dev, csc, cols, lnnz, unit, L, tile_bits, bounds)) {
        if (std::getenv("MFM_PLAN_CHECK")) {  // tests: the device layout must be the host layout, array by array
          ParLevel H;
          if (!build_tiled(csc, cols, lnnz, unit, H, tile_bits, bounds)) throw Error(MFM_ERR_RUNTIME, "plan check: host layout failed");
          auto same = [](const auto &a, const auto &b, size_t n, const char *what) {
            typedef typename std::remove_pointer<decltype(a.p)>::type T;
            std::vector<T> x(n), y(n);
            if (n) {
              MFM_HIP_CHECK(hipMemcpy(x.data(), a.p, n * sizeof(T), hipMemcpyDeviceToHost));
              MFM_HIP_CHECK(hipMemcpy(y.data(), b.p, n * sizeof(T), hipMemcpyDeviceToHost));
            }
            if (std::memcmp(x.data(), y.data(), n * sizeof(T)) != 0)
              throw Error(MFM_ERR_RUNTIME, std::string("plan check: device and host tile layouts differ in ") + what);
          };
          if (L.n_runs != H.n_runs || L.n_tiles != H.n_tiles || L.n_ent != H.n_ent || L.covers_rows_once != H.covers_rows_once)
            throw Error(MFM_ERR_RUNTIME, "plan check: device and host tile layouts differ in their sizes");
          const size_t n_pad = H.tent.n;
          same(L.tent, H.tent, n_pad, "tent");
          if (!unit) same(L.ent_val, H.ent_val, n_pad, "ent_val");
          same(L.tile_ptr, H.tile_ptr, (size_t)H.n_tiles + 1, "tile_ptr");
          same(L.run_base, H.run_base, H.run_base.n, "run_base");
          same(L.slot_ptr, H.slot_ptr, cols.size() + 1, "slot_ptr");
          same(L.slot_pos, H.slot_pos, (size_t)H.n_runs, "slot_pos");
        }
        return true;
      }
      return build_tiled(csc, cols, lnnz, unit, L, tile_bits, bounds);
    }
    int64_t RB = SCAT_RB;
    if (const char *e = std::getenv("MFM_SCAT_RB")) RB = std::max<int64_t>(1024, std::atoll(e));
    const int64_t nb = (N + RB - 1) / RB;
    std::vector<int64_t> bptr((size_t)nb + 1, 0);
    for (int32_t j : cols)
      for (int64_t p = csc.ptr[j]; p < csc.ptr[j + 1]; p++) bptr[csc.idx[p] / RB + 1]++;
    for (int64_t b = 0; b < nb; b++) bptr[b + 1] += bptr[b];
    std::vector<int2> ent((size_t)lnnz);
    std::vector<double> ev;
    if (!unit) ev.resize((size_t)lnnz);
    {
      std::vector<int64_t> cur(bptr.begin(), bptr.end() - 1);
      for (int32_t j : cols)  // ascending column, ascending row inside: (block, column, row) order
        for (int64_t p = csc.ptr[j]; p < csc.ptr[j + 1]; p++) {
          const int64_t q = cur[csc.idx[p] / RB]++;
          ent[q] = make_int2(csc.idx[p], j);
          if (!unit) ev[q] = csc.val[p];
        }
    }
    // runs: maximal stretches of one column inside a 64-entry tile
    const int64_t n_tiles = (lnnz + WAVE - 1) / WAVE;
    std::vector<int32_t> run_base((size_t)n_tiles + 1, 0);
    std::vector<int32_t> run_col;
    run_col.reserve((size_t)(lnnz / 4));
    for (int64_t t = 0; t < n_tiles; t++) {
      run_base[t] = (int32_t)run_col.size();
      const int64_t b = t * WAVE, e = std::min<int64_t>(lnnz, b + WAVE);
      for (int64_t p = b; p < e; p++)
        if (p == b || ent[p].y != ent[p - 1].y) run_col.push_back(ent[p].y);
    }
    run_base[n_tiles] = (int32_t)run_col.size();
    // per column: its runs in stream order
    std::vector<int32_t> local((size_t)csc.rows, -1);
    for (size_t c = 0; c < cols.size(); c++) local[cols[c]] = (int32_t)c;
    std::vector<int32_t> sptr(cols.size() + 1, 0), sidx(run_col.size());
    for (int32_t j : run_col) sptr[local[j] + 1]++;
    for (size_t c = 0; c < cols.size(); c++) sptr[c + 1] += sptr[c];
    {
      std::vector<int32_t> cur(sptr.begin(), sptr.end() - 1);
      for (size_t r = 0; r < run_col.size(); r++) sidx[cur[local[run_col[r]]]++] = (int32_t)r;
    }
    {
      std::vector<char> seen((size_t)N, 0);
      bool once = lnnz == N;
      for (int64_t q = 0; once && q < lnnz; q++) {
        once = !seen[ent[q].x];
        seen[ent[q].x] = 1;
      }
      L.covers_rows_once = once;
    }
    L.scattered = true;
    L.n_ent = lnnz;
    L.n_cols = (int)cols.size();
    L.n_runs = (int)run_col.size();
    L.ent.upload(ent.data(), ent.size());
    if (!unit) L.ent_val.upload(ev);
    L.run_base.upload(run_base);
    L.scols.upload(cols);
    L.slot_ptr.upload(sptr);
    L.slot_idx.upload(sidx);
    L.slots.alloc((size_t)std::max<size_t>(run_col.size(), 1));
    return true;
  }

  const DevCscView *dev_csc = nullptr;  // (set by the owner before build: the table's CSC on the device -> tile packing there)

  static bool build_tiled_device(const DevCscView &dv, const HostCsr &csc, const std::vector<int32_t> &cols, int64_t lnnz,
                                 bool unit, ParLevel &L, int tile_bits, const std::vector<int32_t> *bounds) {
    const int64_t N = csc.cols, RB = (int64_t)1 << tile_bits;
    hipStream_t s = dv.stream;
    std::vector<int32_t> grid;
    if (!bounds || bounds->size() < 2) {
      for (int64_t r = 0; r < N; r += RB) grid.push_back((int32_t)r);
      grid.push_back((int32_t)N);
      bounds = &grid;
    }
    const std::vector<int32_t> &tstart = *bounds;
    const int nb = (int)tstart.size() - 1, ncols = (int)cols.size();
    int cbits = 1, bbits = 1;
    while (((int64_t)1 << cbits) < ncols) cbits++;
    while (((int64_t)1 << bbits) < nb) bbits++;
    if (cbits + bbits > 62 || lnnz >= ((int64_t)1 << 31) || lnnz == 0) return false;
    std::vector<int32_t> lptr((size_t)ncols + 1, 0);
    for (int c = 0; c < ncols; c++) lptr[c + 1] = lptr[c] + (int32_t)(csc.ptr[cols[c] + 1] - csc.ptr[cols[c]]);
    DevBuf<int32_t> d_cols, d_lptr, d_tstart, d_tbegin, d_tptr;
    d_cols.upload(cols);
    d_lptr.upload(lptr);
    d_tstart.upload(tstart);
    DevBuf<uint64_t> key, key2;
    DevBuf<uint32_t> pos, pos2;
    key.alloc((size_t)lnnz);
    key2.alloc((size_t)lnnz);
    pos.alloc((size_t)lnnz);
    pos2.alloc((size_t)lnnz);
    hipLaunchKernelGGL(k_dp_keys, dim3((ncols + WG / WAVE - 1) / (WG / WAVE)), dim3(WG), 0, s, dv.colptr, dv.rowidx, d_cols.p, d_lptr.p,
                       ncols, d_tstart.p, nb, cbits, key.p, pos.p);
    size_t tmp_bytes = 0;
    MFM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, key.p, key2.p, pos.p, pos2.p, (int)lnnz, 0, cbits + bbits, s));
    DevBuf<char> tmp;
    tmp.alloc(tmp_bytes);
    MFM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, key.p, key2.p, pos.p, pos2.p, (int)lnnz, 0, cbits + bbits, s));
    // tile boundaries in the sorted order -> padded wave-tile offsets (host: nb + 1 integers)
    d_tbegin.alloc((size_t)nb + 1);
    hipLaunchKernelGGL((k_dp_lower_bound<uint64_t>), dim3((nb + 256) / 256), dim3(256), 0, s, key2.p, lnnz, cbits, nb + 1, d_tbegin.p);
    std::vector<int32_t> tbegin((size_t)nb + 1), tptr((size_t)nb + 1, 0);
    MFM_HIP_CHECK(hipMemcpyAsync(tbegin.data(), d_tbegin.p, ((size_t)nb + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MFM_HIP_CHECK(hipStreamSynchronize(s));
    for (int b = 0; b < nb; b++) {
      const int64_t t = (int64_t)tptr[b] + ((int64_t)(tbegin[b + 1] - tbegin[b]) + WAVE - 1) / WAVE;
      if (t >= (int64_t)1 << 30) return false;
      tptr[b + 1] = (int32_t)t;
    }
    const int64_t n_wt = tptr[nb], n_pad = n_wt * WAVE;
    d_tptr.upload(tptr);
    L.tent.alloc((size_t)std::max<int64_t>(n_pad, 1));
    MFM_HIP_CHECK(hipMemsetAsync(L.tent.p, 0xff, (size_t)n_pad * sizeof(uint32_t), s));
    if (!unit) {
      L.ent_val.alloc((size_t)std::max<int64_t>(n_pad, 1));
      MFM_HIP_CHECK(hipMemsetAsync(L.ent_val.p, 0, (size_t)n_pad * sizeof(double), s));
    }
    hipLaunchKernelGGL(k_dp_fill, dim3((unsigned)((lnnz + 255) / 256)), dim3(256), 0, s, key2.p, pos2.p, lnnz, cbits, tile_bits,
                       d_tbegin.p, d_tptr.p, d_tstart.p, dv.rowidx, unit ? nullptr : dv.cval, L.tent.p, unit ? nullptr : L.ent_val.p);
    // runs per wave tile -> run_base (exclusive scan), column of every run
    DevBuf<int32_t> run_cnt;
    run_cnt.alloc((size_t)n_wt + 1);
    MFM_HIP_CHECK(hipMemsetAsync(run_cnt.p, 0, ((size_t)n_wt + 1) * sizeof(int32_t), s));
    const unsigned g_wt = (unsigned)((n_wt + WG / WAVE - 1) / (WG / WAVE));
    hipLaunchKernelGGL(k_dp_runs, dim3(std::max(g_wt, 1u)), dim3(WG), 0, s, L.tent.p, n_wt, tile_bits, run_cnt.p, nullptr, nullptr);
    L.run_base.alloc((size_t)n_wt + 1);
    size_t scan_bytes = 0;
    MFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, run_cnt.p, L.run_base.p, (int)(n_wt + 1), s));
    DevBuf<char> tmp2;
    tmp2.alloc(scan_bytes);
    MFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp2.p, scan_bytes, run_cnt.p, L.run_base.p, (int)(n_wt + 1), s));
    int32_t n_runs = 0;
    MFM_HIP_CHECK(hipMemcpyAsync(&n_runs, L.run_base.p + n_wt, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MFM_HIP_CHECK(hipStreamSynchronize(s));
    DevBuf<int32_t> run_col, run_col2;
    DevBuf<uint32_t> ridx, sidx;
    run_col.alloc((size_t)std::max(n_runs, 1));
    run_col2.alloc((size_t)std::max(n_runs, 1));
    ridx.alloc((size_t)std::max(n_runs, 1));
    sidx.alloc((size_t)std::max(n_runs, 1));
    hipLaunchKernelGGL(k_dp_runs, dim3(std::max(g_wt, 1u)), dim3(WG), 0, s, L.tent.p, n_wt, tile_bits, nullptr, L.run_base.p, run_col.p);
    // slots in column-major order: the runs sorted by column (stable: stream order inside a column)
    L.slot_ptr.alloc((size_t)ncols + 1);
    L.slot_pos.alloc((size_t)std::max(n_runs, 1));
    if (n_runs > 0) {
      hipLaunchKernelGGL(k_dp_iota, dim3((unsigned)((n_runs + 255) / 256)), dim3(256), 0, s, ridx.p, (int64_t)n_runs);
      size_t sb = 0;
      MFM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, sb, run_col.p, run_col2.p, ridx.p, sidx.p, n_runs, 0, cbits, s));
      DevBuf<char> tmp3;
      tmp3.alloc(sb);
      MFM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp3.p, sb, run_col.p, run_col2.p, ridx.p, sidx.p, n_runs, 0, cbits, s));
      hipLaunchKernelGGL((k_dp_lower_bound<int32_t>), dim3((ncols + 256) / 256), dim3(256), 0, s, run_col2.p, (int64_t)n_runs, 0, ncols + 1,
                         L.slot_ptr.p);
      hipLaunchKernelGGL(k_dp_invert, dim3((unsigned)((n_runs + 255) / 256)), dim3(256), 0, s, sidx.p, (int64_t)n_runs, L.slot_pos.p);
      MFM_HIP_CHECK(hipStreamSynchronize(s));  // (the sort's temporaries are released on return)
    } else {
      MFM_HIP_CHECK(hipMemsetAsync(L.slot_ptr.p, 0, ((size_t)ncols + 1) * sizeof(int32_t), s));
    }
    // does the level touch every row exactly once?
    bool once = lnnz == N;
    if (once) {
      DevBuf<int32_t> cnt;
      DevBuf<int> dup;
      cnt.alloc((size_t)N);
      dup.alloc(1);
      MFM_HIP_CHECK(hipMemsetAsync(cnt.p, 0, (size_t)N * sizeof(int32_t), s));
      MFM_HIP_CHECK(hipMemsetAsync(dup.p, 0, sizeof(int), s));
      hipLaunchKernelGGL(k_dp_row_once, dim3((unsigned)((lnnz + 255) / 256)), dim3(256), 0, s, key2.p, pos2.p, lnnz, dv.rowidx, cnt.p, dup.p);
      int h = 0;
      MFM_HIP_CHECK(hipMemcpyAsync(&h, dup.p, sizeof(int), hipMemcpyDeviceToHost, s));
      MFM_HIP_CHECK(hipStreamSynchronize(s));
      once = h == 0;
    }
    MFM_HIP_CHECK(hipGetLastError());
    MFM_HIP_CHECK(hipStreamSynchronize(s));
    L.covers_rows_once = once;
    L.scattered = true;
    L.tiled = true;
    L.tile_bits = tile_bits;
    L.n_tiles = nb;
    L.n_ent = lnnz;
    L.n_cols = ncols;
    L.n_runs = n_runs;
    L.tile_ptr.upload(tptr);
    L.tile_row0.upload(tstart);
    L.scols.upload(cols);
    L.slots.alloc((size_t)std::max<size_t>((size_t)n_runs, 1));
    MFM_HIP_CHECK(hipMemset(L.slots.p, 0, std::max<size_t>((size_t)n_runs, 1) * sizeof(double2)));
    return true;
  }

  // bounds: tile boundaries (row starts, last = N), every tile at most 2^tile_bits rows; null: fixed grid
  static bool build_tiled(const HostCsr &csc, const std::vector<int32_t> &cols, int64_t lnnz, bool unit, ParLevel &L,
                          int tile_bits, const std::vector<int32_t> *bounds = nullptr) {
    const int64_t N = csc.cols, RB = (int64_t)1 << tile_bits;
    std::vector<int32_t> grid;
    if (!bounds || bounds->size() < 2) {
      for (int64_t r = 0; r < N; r += RB) grid.push_back((int32_t)r);
      grid.push_back((int32_t)N);
      bounds = &grid;
    }
    const std::vector<int32_t> &tstart = *bounds;
    const int64_t nb = (int64_t)tstart.size() - 1;
    std::vector<int32_t> row_tile((size_t)N);
    for (int64_t b = 0; b < nb; b++)
      for (int64_t r = tstart[b]; r < tstart[b + 1]; r++) row_tile[r] = (int32_t)b;
    std::vector<int64_t> cnt((size_t)nb + 1, 0);
    for (int32_t j : cols)
      for (int64_t p = csc.ptr[j]; p < csc.ptr[j + 1]; p++) cnt[row_tile[csc.idx[p]] + 1]++;
    std::vector<int32_t> tptr((size_t)nb + 1, 0);  // in wave tiles
    for (int64_t b = 0; b < nb; b++) {
      const int64_t t = (int64_t)tptr[b] + (cnt[b + 1] + WAVE - 1) / WAVE;
      if (t >= (int64_t)1 << 30) return false;
      tptr[b + 1] = (int32_t)t;
    }
    const int64_t n_wt = tptr[nb], n_pad = n_wt * WAVE;
    std::vector<uint32_t> ent((size_t)n_pad, TILE_PAD);
    std::vector<double> ev;
    if (!unit) ev.assign((size_t)n_pad, 0.0);
    {
      std::vector<int64_t> cur((size_t)nb);
      for (int64_t b = 0; b < nb; b++) cur[b] = (int64_t)tptr[b] * WAVE;
      for (size_t c = 0; c < cols.size(); c++) {  // ascending column, ascending row inside: (tile, column, row)
        const int32_t j = cols[c];
        for (int64_t p = csc.ptr[j]; p < csc.ptr[j + 1]; p++) {
          const int64_t r = csc.idx[p], b = row_tile[r], q = cur[b]++;
          ent[q] = ((uint32_t)c << tile_bits) | (uint32_t)(r - tstart[b]);
          if (!unit) ev[q] = csc.val[p];
        }
      }
    }
    // runs: maximal stretches of one column inside a wave tile
    std::vector<int32_t> run_base((size_t)n_wt + 1, 0), run_col;
    run_col.reserve((size_t)(lnnz / 4));
    for (int64_t t = 0; t < n_wt; t++) {
      run_base[t] = (int32_t)run_col.size();
      for (int64_t p = t * WAVE; p < (t + 1) * WAVE && ent[p] != TILE_PAD; p++) {
        const int32_t c = (int32_t)(ent[p] >> tile_bits);
        if (p == t * WAVE || (int32_t)(ent[p - 1] >> tile_bits) != c) run_col.push_back(c);
      }
    }
    run_base[n_wt] = (int32_t)run_col.size();
    std::vector<int32_t> sptr(cols.size() + 1, 0), sidx(run_col.size());
    for (int32_t c : run_col) sptr[c + 1]++;
    for (size_t c = 0; c < cols.size(); c++) sptr[c + 1] += sptr[c];
    {
      std::vector<int32_t> cur(sptr.begin(), sptr.end() - 1);
      for (size_t r = 0; r < run_col.size(); r++) sidx[cur[run_col[r]]++] = (int32_t)r;
    }
    {
      std::vector<char> seen((size_t)N, 0);
      bool once = lnnz == N;
      for (int32_t j : cols)
        for (int64_t p = csc.ptr[j]; once && p < csc.ptr[j + 1]; p++) {
          once = !seen[csc.idx[p]];
          seen[csc.idx[p]] = 1;
        }
      L.covers_rows_once = once;
    }
    L.scattered = true;
    L.tiled = true;
    L.tile_bits = tile_bits;
    L.n_tiles = (int)nb;
    L.n_ent = lnnz;
    L.n_cols = (int)cols.size();
    L.n_runs = (int)run_col.size();
    L.tent.upload(ent);
    if (!unit) L.ent_val.upload(ev);
    L.tile_ptr.upload(tptr);
    L.tile_row0.upload(tstart);
    L.run_base.upload(run_base);
    L.scols.upload(cols);
    L.slot_ptr.upload(sptr);
    {
      // slots in column-major order (a column's slots contiguous, in stream order): the statistics pass
      // scatters one 16-byte store per run, the draw streams. (Measured alternatives on config 3: stream-order
      // slots + gathering draw 84 + 109 us, tile-group-major slots 99 + 52 us, this layout 103 + 30 us.)
      std::vector<int32_t> pos(sidx.size());
      for (size_t k = 0; k < sidx.size(); k++) pos[sidx[k]] = (int32_t)k;
      L.slot_pos.upload(pos);
    }
    L.slots.alloc((size_t)std::max<size_t>(run_col.size(), 1));
    MFM_HIP_CHECK(hipMemset(L.slots.p, 0, std::max<size_t>(run_col.size(), 1) * sizeof(double2)));
    return true;
  }

  // bins the columns `cols` of one level by length into L (see the file header)
  void bin_columns(const HostCsr &csc, const std::vector<int32_t> &cols, ParLevel &L, int64_t cap_w1, int64_t cap_w4,
                   int64_t cap_w16, int64_t cap_wg, int coop_max) {
    std::vector<int32_t> w1, w4, w16, wg, lg, hg, lptr, hptr;
    std::vector<ChunkDesc> lch, hch;
    for (int32_t j : cols) {
      const int64_t len = csc.ptr[j + 1] - csc.ptr[j];
      if (len <= cap_w1) {
        w1.push_back(j);
        L.nnz_light += len;
      } else if (len <= cap_w4) {
        w4.push_back(j);
        L.nnz_light += len;
      } else if (len <= cap_w16) {
        w16.push_back(j);
        L.nnz_heavy += len;
      } else if (len <= cap_wg) {
        wg.push_back(j);
        L.nnz_heavy += len;
      } else {
        const int64_t nch = (len + cap_wg - 1) / cap_wg;
        const bool coop = nch <= coop_max;
        std::vector<ChunkDesc> &ch = coop ? lch : hch;
        std::vector<int32_t> &ptr = coop ? lptr : hptr;
        std::vector<int32_t> &lst = coop ? lg : hg;
        ptr.push_back((int32_t)ch.size());
        for (int64_t b = 0; b < len; b += cap_wg)
          ch.push_back(ChunkDesc{csc.ptr[j] + b, (int32_t)std::min<int64_t>(cap_wg, len - b), (int32_t)lst.size()});
        lst.push_back(j);
        (coop ? L.nnz_long : L.nnz_huge) += len;
      }
    }
    lptr.push_back((int32_t)lch.size());
    hptr.push_back((int32_t)hch.size());
    L.n_w1 = (int)w1.size();
    L.n_w4 = (int)w4.size();
    L.n_w16 = (int)w16.size();
    L.n_wg = (int)wg.size();
    L.n_long = (int)lg.size();
    L.n_huge = (int)hg.size();
    L.cols_w1.upload(w1);
    L.cols_w4.upload(w4);
    L.cols_w16.upload(w16);
    L.cols_wg.upload(wg);
    L.cols_long.upload(lg);
    L.cols_huge.upload(hg);
    L.lchunks.upload(lch.data(), lch.size());
    L.lchunk_ptr.upload(lptr);
    L.hchunks.upload(hch.data(), hch.size());
    L.hchunk_ptr.upload(hptr);
    L.n_hchunks = (int)hch.size();
    if (!lch.empty()) {
      L.lpartial.alloc(2 * lch.size());
      L.arrive.alloc(lg.size());
      MFM_HIP_CHECK(hipMemset(L.arrive.p, 0, lg.size() * sizeof(unsigned long long)));
      // rounds: whole columns, at most coop_max chunks each
      int first = 0;
      int64_t rn = 0;
      for (size_t c = 0; c < lg.size(); c++) {
        const int cb = lptr[c], ce = lptr[c + 1];
        if (ce - first > coop_max) {
          L.rounds.emplace_back(first, cb - first);
          L.round_nnz.push_back(rn);
          first = cb;
          rn = 0;
        }
        rn += csc.ptr[lg[c] + 1] - csc.ptr[lg[c]];
      }
      L.rounds.emplace_back(first, (int)lch.size() - first);
      L.round_nnz.push_back(rn);
    }
  }

  void build_aligned_tiles(const HostCsr &csc, const std::vector<int32_t> &cols, int64_t RB) {
    std::vector<int32_t> order(cols);
    const int64_t nnz_all = csc.ptr[csc.rows];
    auto first_row = [&](int32_t x) { return csc.ptr[x] < csc.ptr[x + 1] ? (int64_t)csc.idx[csc.ptr[x]] : (int64_t)-1; };
    // empty columns sort first: they join the first tile as zero-length columns (drawn from the prior there)
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return first_row(x) < first_row(y); });
    (void)nnz_all;
    std::vector<int32_t> fcols, fptr, lcols, lptr, ltiles, solo, empties;
    h_tile_start.clear();
    int64_t cur_rows = 0, next_row = 0, cur_cols = 0;
    int tb = 0;
    while (((int64_t)1 << tb) < RB) tb++;
    const int64_t col_cap = mf_user_cap(tb);  // columns with rows per tile (k_mf_pass keeps their scalars in LDS)
    auto open_tile = [&](int64_t row) {
      h_tile_start.push_back((int32_t)row);
      fptr.push_back((int32_t)fcols.size());
      cur_rows = 0;
      cur_cols = 0;
    };
    for (int32_t j : order) {
      const int64_t len = csc.ptr[j + 1] - csc.ptr[j];
      const bool is_special = !special.empty() && special[j] == 1;
      if (len == 0) {
        // special: swept through special_level on every rank; sharded and not empty everywhere: its rows live on
        // another rank; otherwise it is drawn from the prior by whichever tile it is handed to below
        if (!(is_special || (sharded_tiles && special[j] != 2))) empties.push_back(j);
        continue;
      }
      const int64_t r0 = csc.idx[csc.ptr[j]];
      if (r0 != next_row) {  // (cannot happen: the level covers every row once with contiguous columns)
        h_tile_start.clear();
        return;
      }
      if (len > RB || is_special) {
        if (!is_special) {
          lcols.push_back(j);
          lptr.push_back((int32_t)ltiles.size());
        }
        for (int64_t r = r0; r < r0 + len; r += RB) {
          const int32_t tile_id = (int32_t)h_tile_start.size();
          open_tile(r);
          if (is_special) {
            solo.push_back(tile_id);
          } else {
            ltiles.push_back(tile_id);
          }
        }
        cur_rows = RB;  // closed
      } else {
        if (h_tile_start.empty() || cur_rows + len > RB || cur_cols + 1 > col_cap) open_tile(r0);
        fcols.push_back(j);
        cur_rows += len;
        cur_cols += 1;
      }
      next_row = r0 + len;
    }
    h_tile_start.push_back((int32_t)next_row);
    fptr.push_back((int32_t)fcols.size());
    lptr.push_back((int32_t)ltiles.size());
    if (next_row != csc.cols) {
      h_tile_start.clear();
      return;
    }
    if (!empties.empty()) {
      // columns without entries (features that never occur): spread evenly over the tiles that run first-level
      // columns (a tile without any stays a tile of a long / special column)
      const size_t nt = h_tile_start.size() - 1;
      std::vector<size_t> hosts;
      for (size_t b = 0; b < nt; b++)
        if (fptr[b + 1] > fptr[b]) hosts.push_back(b);
      if (hosts.empty()) {  // (only long / special columns: no fused pass to ride on)
        h_tile_start.clear();
        return;
      }
      std::vector<std::vector<int32_t>> extra(nt);
      for (size_t k = 0; k < empties.size(); k++) extra[hosts[k % hosts.size()]].push_back(empties[k]);
      std::vector<int32_t> f2, p2;
      for (size_t b = 0; b < nt; b++) {
        p2.push_back((int32_t)f2.size());
        f2.insert(f2.end(), fcols.begin() + fptr[b], fcols.begin() + fptr[b + 1]);
        f2.insert(f2.end(), extra[b].begin(), extra[b].end());
      }
      p2.push_back((int32_t)f2.size());
      fcols.swap(f2);
      fptr.swap(p2);
    }
    fuse_cols.upload(fcols);
    fuse_col_ptr.upload(fptr);
    {
      std::vector<int4> desc(fcols.size());
      for (size_t b = 0; b + 1 < fptr.size(); b++)
        for (int32_t k = fptr[b]; k < fptr[b + 1]; k++) {
          const int32_t j = fcols[k];
          const int64_t len = csc.ptr[j + 1] - csc.ptr[j];
          desc[k] = make_int4(j, (int)len, len ? (int)(csc.idx[csc.ptr[j]] - h_tile_start[b]) : 0,
                              group_of && (size_t)j < group_of->size() ? (*group_of)[j] : 0);
        }
      fuse_desc.upload(desc.data(), desc.size());
      // chunk heads / partial indices for the two-field pass
      const size_t nt_ = h_tile_start.size() - 1;
      std::vector<int32_t> cptr(nt_ + 1, 0);
      for (size_t b = 0; b < nt_; b++) cptr[b + 1] = cptr[b] + (h_tile_start[b + 1] - h_tile_start[b] + WAVE - 1) / WAVE;
      std::vector<MfChunk> chunks((size_t)cptr[nt_], MfChunk{1ull, 0, 0});
      std::vector<int2> upart(fcols.size(), make_int2(0, 0));
      for (size_t b = 0; b < nt_; b++) {
        MfChunk *C = chunks.data() + cptr[b];
        const int nch = cptr[b + 1] - cptr[b];
        if (fptr[b + 1] == fptr[b]) continue;  // tile of a long / special column
        for (int c = 0; c < nch; c++) C[c].heads = 0ull;
        int ul = 0;
        for (int32_t k = fptr[b]; k < fptr[b + 1]; k++, ul++) {
          if (desc[k].y == 0) continue;  // (never-occurring columns follow the ones with rows)
          const int lr0 = desc[k].z;
          C[lr0 >> 6].heads |= 1ull << (lr0 & 63);
          for (int c = lr0 >> 6; c <= (lr0 + desc[k].y - 1) >> 6; c++)
            if (c > (lr0 >> 6) || (lr0 & 63) == 0) C[c].ubase = ul;  // owner of the chunk's first row
        }
        int pidx = 0;
        for (int c = 0; c < nch; c++) {
          C[c].pbase = pidx;
          pidx += __builtin_popcountll(C[c].heads | 1ull);
        }
        for (int32_t k = fptr[b]; k < fptr[b + 1]; k++) {
          if (desc[k].y == 0) continue;
          const int lr0 = desc[k].z, ca = lr0 >> 6, cb = (lr0 + desc[k].y - 1) >> 6;
          const int seg = __builtin_popcountll(C[ca].heads & ((2ull << (lr0 & 63)) - 2ull));
          upart[k] = make_int2(C[ca].pbase + seg, cb - ca + 1);
        }
      }
      mf_max_users = 0;
      for (size_t b = 0; b < nt_; b++) {
        int cnt = 0;
        for (int32_t k = fptr[b]; k < fptr[b + 1]; k++) cnt += desc[k].y > 0;
        mf_max_users = std::max(mf_max_users, cnt);
      }
      mf_chunk.upload(chunks.data(), chunks.size());
      mf_chunk_ptr.upload(cptr);
      mf_upart.upload(upart.data(), upart.size());
      mf_ready = true;
    }
    {
      const size_t nt = h_tile_start.size() - 1;
      std::vector<int32_t> sc(nt, -1), tl(nt, -1);
      for (size_t l = 0; l < lcols.size(); l++)
        for (int32_t t = lptr[l]; t < lptr[l + 1]; t++) {
          sc[ltiles[t]] = lcols[l];
          tl[ltiles[t]] = (int32_t)l;
        }
      n_solo_tiles = (int)solo.size();
      solo_tiles.upload(solo);
      n_long_cols = (int)lcols.size();
      n_long_tiles = (int)ltiles.size();
      long_cols.upload(lcols);
      long_tile_ptr.upload(lptr);
      long_tiles.upload(ltiles);
      solo_col.upload(sc);
      tile_long_idx.upload(tl);
      long_partial.alloc(std::max<size_t>(nt, 1));
      oldnew_long.alloc(std::max<size_t>(lcols.size(), 1));
    }
    aligned_tiles = true;
  }

  // Row-sharded mode with the fused tile path (run_sweep_soa_sharded): the plan may use row tiles; first-level
  // columns flagged `special` (rows on more than one rank -- the same set on every rank) get tiles of their own and are swept through `special_level` with an
  // all-reduce of their statistics; all other first-level columns with local rows are complete on this rank.
  bool sharded_tiles = false;
  std::vector<char> special;  // per column of the table (sharded_tiles only): 1 = special, 2 = empty on every rank
                              // (drawn from the prior by every rank itself, like a locally complete column)
  ParLevel special_level;
  ParLevel local_level;       // the other first-level columns with local rows: complete here, swept without communication
  DevBuf<int32_t> special_cols;
  int n_special = 0;

  int tile_bits = 0;     // > 0: scattered levels use the row-tile path with tiles of 2^tile_bits rows
  bool sharded = false;  // row-sharded multi-GPU mode: no chains (every column needs an all-reduce), no coop
  bool block_plan = false;  // the plan of a relation block: only those launch the streamed chain (run_plan_t, PBlockV / PBlockW)

  // Row-sharded mode: the schedule must be the same on every rank, so it is computed on the GLOBAL design
  // by the caller and handed in; here it is only checked against the local rows.
  std::vector<int32_t> given_levels;
  static void check_levels(const HostCsr &csc, const std::vector<int32_t> &level) {
    if ((int64_t)level.size() != csc.rows) throw Error(MFM_ERR_INVALID, "column level array has the wrong length");
    std::vector<int32_t> rowlevel((size_t)csc.cols, -1);
    for (int64_t j = 0; j < csc.rows; j++) {
      if (level[j] < 0) throw Error(MFM_ERR_INVALID, "negative column level");
      for (int64_t p = csc.ptr[j]; p < csc.ptr[j + 1]; p++) {
        int32_t &rl = rowlevel[csc.idx[p]];
        if (rl >= level[j])
          throw Error(MFM_ERR_INVALID, "column levels are not a valid schedule: column " + std::to_string(j) +
                                           " shares a row with an earlier column of the same or a later level");
        rl = level[j];
      }
    }
  }

  void build(const HostCsr &csc, int r_w16, int r_wg, int coop_max, bool allow_scatter = false, bool unit = false,
             const StepPlan *twin = nullptr) {
    // twin: a plan already built for the SAME matrix with the same settings (the other sweep of the table): its level
    // assignment and its row-tile levels are reused
    const int coop_local = coop_max;  // (sharded_tiles: locally complete long columns still run co-resident)
    if (sharded) coop_max = 0;
    n_state_rows = csc.cols;
    {
      std::vector<int32_t> r0((size_t)csc.rows, 0);
      for (int64_t j = 0; j < csc.rows; j++)
        if (csc.ptr[j + 1] > csc.ptr[j]) r0[j] = csc.idx[csc.ptr[j]];
      col_row0.upload(r0);
    }
    const int64_t cap_w1 = WAVE, cap_w4 = 4 * WAVE, cap_w16 = (int64_t)r_w16 * WAVE, cap_wg = (int64_t)r_wg * WG;
    std::vector<int32_t> level;
    if (!given_levels.empty() || (sharded && csc.rows > 0)) {
      if (given_levels.empty())
        throw Error(MFM_ERR_RUNTIME, "row-sharded mode needs the column levels of the GLOBAL design (mfm_set_main_levels)");
      check_levels(csc, given_levels);
      level = given_levels;
      n_levels = 0;
      for (auto l : level) n_levels = std::max(n_levels, l + 1);
    } else if (twin && !twin->h_level.empty() && (int64_t)twin->h_level.size() == csc.rows) {
      level = twin->h_level;
      n_levels = twin->n_levels;
    } else if (dev_csc && !std::getenv("MFM_HOST_LEVELS") && column_levels_device(*dev_csc, level, n_levels)) {
      if (std::getenv("MFM_PLAN_CHECK")) {  // tests: the device schedule must be the host schedule
        std::vector<int32_t> hl;
        const int32_t hn = column_levels(csc, hl);
        if (hn != n_levels || hl != level) throw Error(MFM_ERR_RUNTIME, "plan check: device and host level schedules differ");
      }
    } else {
      n_levels = column_levels(csc, level);
    }
    h_level = level;
    std::vector<std::vector<int32_t>> by_level((size_t)n_levels);
    for (int64_t j = 0; j < csc.rows; j++) by_level[level[j]].push_back((int32_t)j);
    steps.clear();
    launches = 0;
    std::vector<int32_t> run;
    int64_t run_nnz = 0;
    auto flush_run = [&]() {
      if (run.empty()) return;
      steps.emplace_back();
      Step &s = steps.back();
      s.is_chain = true;
      {  // the twin has the same run at the same place: its arrays serve both sweeps
        const size_t si = steps.size() - 1;
        const Step *tw = twin && si < twin->steps.size() && twin->steps[si].is_chain ? &twin->steps[si] : nullptr;
        if (tw && !sharded && !twin->sharded && tw->chain.h_cols == run && !std::getenv("MFM_NO_CHAIN_TWIN")) {
          s.chain.borrow(tw->chain);
          launches += 1;
          run.clear();
          run_nnz = 0;
          return;
        }
      }
      s.chain.h_cols = run;
      s.chain.n_cols = (int)run.size();
      s.chain.nnz = run_nnz;
      s.chain.cols.upload(run);
      {
        std::vector<ChainDesc> d;
        for (int32_t j : run) d.push_back(ChainDesc{csc.ptr[j], (int32_t)(csc.ptr[j + 1] - csc.ptr[j]), j});
        s.chain.desc.upload(d.data(), d.size());
      }
      // state too large for the LDS chain of any policy: also keep the conflict-batched form
      // (a main-table plan never launches the stream: building it, and skipping the conflict batches for it, would leave a long main
      //  chain to the single-workgroup kernel)
      const bool stream_allowed =
          block_plan && !std::getenv("MFM_NO_CB_STREAM") && !std::getenv("MFM_NO_CB_PERSIST") && !std::getenv("MFM_NO_CHAIN_GRID");
      auto try_stream = [&]() {
        CsStreamInfo ci;
        s.chain.stream = cs_stream_build(csc, run, &ci);
        if (std::getenv("MFM_SETUP_TIMING")) {
          if (s.chain.stream)
            std::fprintf(stderr, "[plan] streamed chain of %zu columns: steps of %d columns, window %d steps, %d ranges, %d LDS slots, %lld cold + %lld "
                                 "hot entries (<= %d hot per column), <= %d entering / %d leaving rows per step, planned in %.3f s\n",
                         run.size(), ci.Cg, ci.Lw, ci.NB, ci.n_slots, ci.n_cold, ci.n_hot, ci.max_hot_col, ci.max_enter, ci.max_exit, ci.plan_seconds);
          else
            std::fprintf(stderr, "[plan] streamed chain of %zu columns: no window fits the LDS, conflict batches kept\n", run.size());
        }
      };
      const bool chain_batched_wanted =
          (csc.cols > 1900 || std::getenv("MFM_CHAIN_FORCE_BATCHED")) && run.size() >= 2 && !std::getenv("MFM_NO_CHAIN_BATCHED");
      // Long columns (the cold parts of a batch would run on the whole GPU: relation blocks with 10^5..10^6 rows): the streamed form is
      // tried FIRST, and when it can be built the conflict batches -- then never launched -- are not built at all (0.06 s of device
      // work and ~100 MB per big block at config 5). The checker mode builds both: the batches' device / host comparison stays tested.
      bool stream_tried = false;
      if (chain_batched_wanted && stream_allowed && !std::getenv("MFM_PLAN_CHECK") && !std::getenv("MFM_CHAIN_FORCE_BATCHED") &&
          run_nnz >= (int64_t)256 * (int64_t)run.size()) {
        try_stream();
        stream_tried = true;
      }
      if (chain_batched_wanted && !s.chain.stream) {
        const int hot_cap = std::getenv("MFM_CHAIN_HOT_CAP") ? std::max(64, std::atoi(std::getenv("MFM_CHAIN_HOT_CAP"))) : 1200;
        // on the device when the matrix's CSC is there (relation blocks, the main table): the host form is the fall-back and,
        // under MFM_PLAN_CHECK, the checker
        const bool on_dev = dev_csc && !std::getenv("MFM_HOST_CHAIN_BATCHES") && s.chain.build_batched_device(*dev_csc, run, hot_cap);
        if (std::getenv("MFM_SETUP_TIMING"))
          std::fprintf(stderr, "[plan] conflict batches of a %zu-column chain run: %s (%d batches, %lld cold entries)\n", run.size(),
                       on_dev ? "device" : "host", on_dev ? s.chain.n_batches : -1, on_dev ? (long long)s.chain.n_cold : -1ll);
        if (!on_dev) {
          s.chain.build_batched(csc, run, hot_cap);
        } else if (std::getenv("MFM_PLAN_CHECK")) {
          ChainRun chk;
          chk.build_batched(csc, run, hot_cap);
          const std::string diff = s.chain.compare_batched(chk, dev_csc->stream);
          if (!diff.empty()) throw Error(MFM_ERR_RUNTIME, "plan check: device and host conflict batches differ (" + diff + ")");
        }
        // chains whose cold parts run on the whole GPU: the streamed form (unless it was tried above)
        if (s.chain.bucketed && stream_allowed && !stream_tried) try_stream();
      }
      launches += 1;
      run.clear();
      run_nnz = 0;
    };
    for (int32_t l = 0; l < n_levels; l++) {
      if (by_level[l].empty()) continue;
      int64_t lnnz = 0;
      for (int32_t j : by_level[l]) lnnz += csc.ptr[j + 1] - csc.ptr[j];
      if (!sharded && tiny(by_level[l].size(), lnnz)) {
        for (int32_t j : by_level[l]) run.push_back(j);
        run_nnz += lnnz;
        continue;
      }
      flush_run();
      steps.emplace_back();
      ParLevel &L = steps.back().par;
      L.cols_all.upload(by_level[l]);
      L.n_all = (int)by_level[l].size();
      L.jmin = *std::min_element(by_level[l].begin(), by_level[l].end());
      L.jmax = *std::max_element(by_level[l].begin(), by_level[l].end());
      {
        // the twin built this level on row tiles with the same boundaries: borrow it
        const size_t si = steps.size() - 1;
        const ParLevel *tw = twin && si < twin->steps.size() && !twin->steps[si].is_chain ? &twin->steps[si].par : nullptr;
        if (allow_scatter && tw && tw->tiled && tw->tile_bits == tile_bits && tw->n_cols == (int)by_level[l].size() &&
            tw->n_ent == lnnz && twin->aligned_tiles == aligned_tiles && twin->h_tile_start == h_tile_start && !sharded) {
          L.borrow_tiled(*tw);
          launches += 3;
          max_cols_scat = std::max<int64_t>(max_cols_scat, csc.rows);
          continue;
        }
      }
      // first look at the level -- contiguous columns? far-apart rows? every row exactly once? -- on the device when the CSC is there
      const bool want_once = steps.size() == 1 && lnnz == csc.cols;
      LevelScan scan;
      if (dev_csc && !std::getenv("MFM_HOST_LEVEL_SCAN")) {
        scan = level_scan_device(*dev_csc, L.cols_all.p, L.n_all, want_once);
        if (scan.valid && std::getenv("MFM_PLAN_CHECK")) {  // tests: the host's counts
          int64_t gaps = 0, far = 0, twice = 0;
          std::vector<char> seen(want_once ? (size_t)csc.cols : 0, 0);
          for (int32_t j : by_level[l])
            for (int64_t p = csc.ptr[j]; p < csc.ptr[j + 1]; p++) {
              if (p > csc.ptr[j]) {
                gaps += csc.idx[p] != csc.idx[p - 1] + 1;
                far += (csc.idx[p] - csc.idx[p - 1]) >= 8;
              }
              if (want_once) {
                twice += seen[csc.idx[p]];
                seen[csc.idx[p]] = 1;
              }
            }
          if (gaps != scan.gaps || far != scan.far || twice != scan.twice)
            throw Error(MFM_ERR_RUNTIME, "plan check: device and host level scans differ");
        }
      }
      if (allow_scatter && build_scattered(csc, by_level[l], lnnz, unit, L, (sharded && !sharded_tiles) ? 0 : tile_bits,
                                           aligned_tiles ? &h_tile_start : nullptr, dev_csc, &scan)) {
        launches += 3;
        max_cols_scat = std::max<int64_t>(max_cols_scat, csc.rows);
        continue;
      }
      if (want_once && scan.valid) {
        L.first_and_once = scan.twice == 0;
      } else if (want_once) {  // first step: does it touch every row exactly once?
        std::vector<char> seen((size_t)csc.cols, 0);
        bool once = true;
        for (int32_t j : by_level[l])
          for (int64_t p = csc.ptr[j]; once && p < csc.ptr[j + 1]; p++) {
            once = !seen[csc.idx[p]];
            seen[csc.idx[p]] = 1;
          }
        L.first_and_once = once;
      }
      bin_columns(csc, by_level[l], L, cap_w1, cap_w4, cap_w16, cap_wg, coop_max);
      {
        bool contig = !std::getenv("MFM_NO_CONTIG");
        if (scan.valid) contig = contig && scan.gaps == 0;
        for (size_t c = 0; !scan.valid && contig && c < by_level[l].size(); c++) {
          const int32_t j = by_level[l][c];
          for (int64_t p = csc.ptr[j] + 1; contig && p < csc.ptr[j + 1]; p++) contig = csc.idx[p] == csc.idx[p - 1] + 1;
        }
        L.contig = contig;
      }
      if (steps.size() == 1 && L.contig && L.first_and_once && allow_scatter && (!sharded || sharded_tiles) && tile_bits > 0 &&
          !std::getenv("MFM_NO_ALIGNED_TILES"))
        build_aligned_tiles(csc, by_level[l], (int64_t)1 << tile_bits);
      if (steps.size() == 1 && sharded_tiles && aligned_tiles) {
        std::vector<int32_t> sp, loc;
        for (int32_t j : by_level[l]) {
          if (!special.empty() && special[j] == 1)
            sp.push_back(j);
          else if (csc.ptr[j + 1] > csc.ptr[j] || (!special.empty() && special[j] == 2))
            loc.push_back(j);
        }
        local_level.cols_all.upload(loc);
        local_level.n_all = (int)loc.size();
        local_level.contig = L.contig;
        bin_columns(csc, loc, local_level, cap_w1, cap_w4, cap_w16, cap_wg, coop_local);
        max_hchunks = std::max(max_hchunks, local_level.n_hchunks);
        max_huge = std::max(max_huge, local_level.n_huge);
        n_special = (int)sp.size();
        special_cols.upload(sp);
        special_level.cols_all.upload(sp);
        special_level.n_all = n_special;
        special_level.contig = L.contig;
        if (n_special) {
          special_level.jmin = sp.front();
          special_level.jmax = sp.back();
          bin_columns(csc, sp, special_level, cap_w1, cap_w4, cap_w16, cap_wg, 0);
          max_hchunks = std::max(max_hchunks, special_level.n_hchunks);
          max_huge = std::max(max_huge, special_level.n_huge);
        }
      }
      max_hchunks = std::max(max_hchunks, L.n_hchunks);
      max_huge = std::max(max_huge, L.n_huge);
      launches += ((L.n_w1 + L.n_w4) ? 1 : 0) + ((L.n_w16 + L.n_wg) ? 1 : 0) + (int64_t)L.rounds.size() + (L.n_huge ? 3 : 0);
    }
    flush_run();
  }
};

// scratch shared by every sweep of a ctx
struct LongScratch {
  DevBuf<double2> partial, oldnew;
  DevBuf<int> error;  // set by k_long_coop on a spin timeout
  DevBuf<double2> oldnew_col;  // scattered levels / sharded mode: (old, new) per column of the matrix
  DevBuf<double2> S_col;       // sharded mode: per-column statistics (all-reduced over the ranks)
  DevBuf<double> vnext_col;    // fused apply pass: next factor's coefficient per column of the last level
  DevBuf<double2> S_compact;   // sharded fused path: statistics of the special first-level columns
  DevBuf<double> told_col;     // multi-level fused flow: current coefficient per column of the level whose statistics are taken
  std::vector<DevBuf<double>> vnext_lvl;  // ... and per tile level: next factor's coefficient per column (MULTIQ)
  DevBuf<double2> cb_part, cb_oldnew;  // grid-batched chains: per-workgroup column partials, (old, new) per column of a batch
  DevBuf<double> cb_colpack;           // ... the batch's per-column scalars (old coefficient, variate, lambda, mu), packed
  DevBuf<double2> cb_hot, cb_hot2;     // ... and the batch's hot records, packed (k_cb_stats -> k_cb_hot -> k_cb_apply; two: k_cb_step
                                       //     reads the previous batch's while it packs the next one's)
  DevBuf<CbSync> cb_sync;              // ... the counters of the one-launch form (k_cb_persist), zeroed before every launch
  DevBuf<double2> dv_col;      // two-field pass: (delta of this factor, coefficient of the next) per second-level column
  void reserve_cols(int64_t n_cols) {
    if ((size_t)n_cols > oldnew_col.n) {
      oldnew_col.alloc((size_t)n_cols);
      vnext_col.alloc((size_t)n_cols);
      told_col.alloc((size_t)n_cols);
    }
  }
  void reserve_stats(int64_t n_cols) {
    if ((size_t)n_cols > S_col.n) S_col.alloc((size_t)n_cols);
  }
  void reserve(int max_chunks, int max_long) {
    if ((size_t)max_chunks > partial.n) partial.alloc((size_t)max_chunks);
    if ((size_t)max_long > oldnew.n) oldnew.alloc((size_t)max_long);
    if (!error.p) {
      error.alloc(1);
      MFM_HIP_CHECK(hipMemset(error.p, 0, sizeof(int)));
    }
  }
};

static inline int xcd_swizzle_enabled() {
  static int v = -1;
  if (v < 0) {
    const char *e = std::getenv("MFM_XCD_SWIZZLE");
    v = e ? std::atoi(e) : 1;
  }
  return v;
}

struct SweepClasses {
  int light, heavy, coop, hstats, hdraw, happly, chain, scat;
};

// the first (non-scattered, non-chain) level can rebuild the q-cache itself (PMainVq)
static inline bool plan_first_level_builds_q(const StepPlan &plan) {
  if (plan.steps.empty()) return false;
  const Step &first = plan.steps.front();
  return !first.is_chain && !first.par.scattered && first.par.first_and_once && first.par.n_huge == 0;
}

template <class P, bool UNIT>
static void launch_binned_level(hipStream_t s, Timing &tm, const ParLevel &L, const SweepArgs &a_in, LongScratch &ls,
                                const SweepClasses &kc, const int32_t *col_row0, int parts = 3) {
  // parts: bit 0 = the long (co-resident) columns, bit 1 = the wavefront / workgroup bins
  SweepArgs a = a_in;
  a.row0 = L.contig ? col_row0 : nullptr;
  // the long columns first: they are the critical path of the level
  if (L.n_long && (parts & 1)) {
    L.epoch++;
    CoopArgs ca;
    ca.chunks = L.lchunks.p;
    ca.lcols = L.cols_long.p;
    ca.chunk_ptr = L.lchunk_ptr.p;
    ca.partial = L.lpartial.p;
    ca.arrive = L.arrive.p;
    ca.epoch = L.epoch;
    ca.error = ls.error.p;
    for (size_t r = 0; r < L.rounds.size(); r++) {
      TimedLaunch t(tm, s, kc.coop, P::BYTES * L.round_nnz[r]);
      hipLaunchKernelGGL((k_long_coop<P, UNIT>), dim3(L.rounds[r].second), dim3(WG), 0, s, a, ca, L.rounds[r].first);
    }
  }
  if (!(parts & 2)) return;
  if (L.n_wg + L.n_w16) {
    TimedLaunch t(tm, s, kc.heavy, P::BYTES * L.nnz_heavy);
    hipLaunchKernelGGL((k_level_heavy<P, UNIT>), dim3(L.n_wg + (L.n_w16 + 3) / 4), dim3(WG), 0, s, a, L.cols_wg.p, L.n_wg,
                       L.cols_w16.p, L.n_w16, xcd_swizzle_enabled());
  }
  if (L.n_w4 + L.n_w1) {
    TimedLaunch t(tm, s, kc.light, P::BYTES * L.nnz_light);
    hipLaunchKernelGGL((k_level_light<P, UNIT>), dim3((L.n_w4 + 3) / 4 + (L.n_w1 + 3) / 4), dim3(WG), 0, s, a, L.cols_w4.p,
                       L.n_w4, L.cols_w1.p, L.n_w1, xcd_swizzle_enabled());
  }
}

// threads of a row-tile workgroup: TILE_K rows per thread (tile_bits 9..13: 64..1024 threads)
static inline int tile_threads(int tile_bits) { return (1 << tile_bits) / TILE_K; }

// every step is a PAR level without two-pass (huge) columns: what the q-free policy (PMainVe) supports
static inline bool plan_is_single_pass_par(const StepPlan &plan) {
  for (const Step &st : plan.steps)
    if (st.is_chain || (!st.par.scattered && st.par.n_huge > 0) || st.par.tiled) return false;
  return !plan.steps.empty();
}

// PA: policy of the apply pass of scattered levels (differs from P only for the q-free policy)
template <class P, bool UNIT, class PA = P>
static void run_plan_t(hipStream_t s, Timing &tm, const StepPlan &plan, const SweepArgs &a, LongScratch &ls,
                       const SweepClasses &kc, bool first_builds_q = false) {
  for (const Step &st : plan.steps) {
    if (st.is_chain) {
      TimedLaunch t(tm, s, kc.chain, P::BYTES * st.chain.nnz);
      const size_t lds_bytes = (size_t)plan.n_state_rows * (P::REC_DOUBLES > 2 ? P::REC_DOUBLES + 2 : P::REC_DOUBLES) * sizeof(double);
      const bool force_batched = st.chain.batched && std::getenv("MFM_CHAIN_FORCE_BATCHED");
      if (plan.n_state_rows > 0 && lds_bytes <= CHAIN_LDS_MAX && !force_batched) {
        hipLaunchKernelGGL((k_chain_lds<P>), dim3(1), dim3(WAVE), lds_bytes, s, a, st.chain.desc.p, st.chain.n_cols,
                           plan.n_state_rows, (int)P::REC_DOUBLES);
      } else if (st.chain.stream && ls.error.p && (std::is_same<P, PBlockV>::value || std::is_same<P, PBlockW>::value)) {
        // one pipelined launch for the whole run (mfm_chain_stream.hpp)
        if constexpr (std::is_same<P, PBlockV>::value || std::is_same<P, PBlockW>::value)
          cs_stream_launch(s, a, *st.chain.stream, std::is_same<P, PBlockV>::value, ls.error.p);
      } else if (st.chain.batched) {
        const ChainRun &C = st.chain;
        constexpr int rec2_l = P::REC_DOUBLES > 2 ? P::REC_DOUBLES / 2 + 1 : P::REC_DOUBLES / 2;
        const int mhe = std::max(C.max_hot_ent, 1);
        const size_t lds_b = (size_t)std::max(C.max_hot, 1) * rec2_l * sizeof(double2) + 5 * CHAINB_MAXCOLS * sizeof(double) +
                             (size_t)CHAINB_MAXCOLS * (CHAINB_NT / WAVE) * sizeof(double2) + (size_t)mhe * 12 +
                             (CHAINB_MAXCOLS + 2) * sizeof(int);
        static DeviceOnce raised;
        if (raised.need()) {
          MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_chain_batched<P>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)CHAIN_LDS_MAX));
          MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_cb_hot<P>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)CHAIN_LDS_MAX));
          MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_cb_persist<P>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)CHAIN_LDS_MAX));
          raised.mark();
        }
        // batches with many cold entries: cold statistics / updates as grid launches (one CU cannot stream them)
        const int64_t grid_min = std::getenv("MFM_CHAIN_GRID_MIN") ? std::atoll(std::getenv("MFM_CHAIN_GRID_MIN")) : 4096;
        if (C.n_batches > 0 && C.n_cold / C.n_batches >= grid_min && !std::getenv("MFM_NO_CHAIN_GRID")) {
          constexpr int GMAX = 64;  // workgroups of a cold launch
          if (ls.cb_part.n < (size_t)GMAX * CHAINB_MAXCOLS) ls.cb_part.alloc((size_t)GMAX * CHAINB_MAXCOLS);
          if (ls.cb_oldnew.n < (size_t)CHAINB_MAXCOLS) ls.cb_oldnew.alloc((size_t)CHAINB_MAXCOLS);
          const size_t hot16 = (size_t)std::max(C.max_hot, 1) * (P::REC_DOUBLES / 2);
          if (ls.cb_hot.n < hot16) ls.cb_hot.alloc(hot16);
          if (C.col_group.n < (size_t)C.n_cols || C.col_group_of != a.group) {  // group index per chain column (gathered once)
            C.col_group.alloc((size_t)std::max(C.n_cols, 1));
            hipLaunchKernelGGL(k_gather_i32, dim3((C.n_cols + 255) / 256), dim3(256), 0, s, a.group, C.cols.p, C.n_cols, C.col_group.p);
            C.col_group_of = a.group;
          }
          const size_t lds_h = (size_t)std::max(C.max_hot, 1) * rec2_l * sizeof(double2) + 5 * CHAINB_MAXCOLS * sizeof(double) +
                               (size_t)CHAINB_MAXCOLS * sizeof(double2) + (size_t)mhe * 12 + (CHAINB_MAXCOLS + 2) * sizeof(int);
          if (C.bucketed) {
            // two launches per batch: k_cb_step = cold update of the batch before + cold statistics of this one, by row range
            if (ls.cb_hot2.n < hot16) ls.cb_hot2.alloc(hot16);
            if (ls.cb_colpack.n < (size_t)8 * CHAINB_MAXCOLS) ls.cb_colpack.alloc((size_t)8 * CHAINB_MAXCOLS);
            if (ls.cb_part.n < (size_t)CB_BUCKETS * CHAINB_MAXCOLS) ls.cb_part.alloc((size_t)CB_BUCKETS * CHAINB_MAXCOLS);
            static const bool persist = !std::getenv("MFM_NO_CB_PERSIST");
            if (persist && ls.error.p) {
              // the whole batch sequence as one launch: the hot walker + CB_BUCKETS row-range workgroups, counters instead of
              // kernel boundaries
              if (!ls.cb_sync.p) ls.cb_sync.alloc(1);
              MFM_HIP_CHECK(hipMemsetAsync(ls.cb_sync.p, 0, sizeof(CbSync), s));
              CbPersistArgs g;
              g.batches = C.batches.p;
              g.n_batches = C.n_batches;
              g.cols = C.cols.p;
              g.col_group = C.col_group.p;
              g.bk_ptr = C.bk_ptr.p;
              g.bk_row = C.bk_row.p;
              g.bk_lcol = C.bk_lcol.p;
              g.hbk_ptr = C.hbk_ptr.p;
              g.bk_cls = C.bk_cls.p;
              g.hot_rows = C.hot_rows.p;
              g.bk_x = C.bk_x.p;
              g.hot_ptr = C.hot_ptr.p;
              g.hot_slot = C.hot_slot.p;
              g.hot_x = C.hot_x.p;
              g.max_hot = std::max(C.max_hot, 1);
              g.max_hot_ent = mhe;
              g.oldnew_g = ls.cb_oldnew.p;
              g.part_g = ls.cb_part.p;
              g.pack[0] = ls.cb_hot.p;
              g.pack[1] = ls.cb_hot2.p;
              g.colpack = ls.cb_colpack.p;
              g.sync = ls.cb_sync.p;
              g.error = ls.error.p;
              static const int cb_dbg = std::getenv("MFM_CB_DBG") ? std::atoi(std::getenv("MFM_CB_DBG")) : 0;
              g.dbg = cb_dbg;
              // MFM_CB_PROF=n: phase sums of the hot walker and of range 0 (s_memrealtime, thread 0), printed every n launches
              static const int cb_prof = std::getenv("MFM_CB_PROF") ? std::atoi(std::getenv("MFM_CB_PROF")) : 0;
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
              const size_t lds_range = (size_t)CHAINB_MAXCOLS * (sizeof(double2) + 2 * sizeof(double)) +
                                       (size_t)CHAINB_MAXCOLS * (CHAINB_NT / WAVE) * sizeof(double2);
              hipLaunchKernelGGL((k_cb_persist<P>), dim3(CB_BUCKETS + 1), dim3(CHAINB_NT), std::max(lds_h, lds_range), s, a, g);
              if (cb_prof > 0 && ++prof_launches % cb_prof == 0) {
                unsigned long long h[16];
                MFM_HIP_CHECK(hipStreamSynchronize(s));
                MFM_HIP_CHECK(hipMemcpy(h, prof_buf.p, sizeof(h), hipMemcpyDeviceToHost));
                MFM_HIP_CHECK(hipMemset(prof_buf.p, 0, sizeof(h)));
                const double nb = (double)std::max<unsigned long long>(h[4], 1), nr = (double)std::max<unsigned long long>(h[11], 1);
                std::fprintf(stderr,
                             "[k_cb_persist] %ld launches, us per batch -- hot walker (%llu batches): wait %.2f, stage in %.2f, walk %.2f, "
                             "stage out + signal %.2f | range 0 (%llu): wait %.2f, near part %.2f, far part %.2f\n",
                             prof_launches, h[4], h[0] / nb / 100.0, h[1] / nb / 100.0, h[2] / nb / 100.0, h[3] / nb / 100.0, h[11],
                             h[8] / nr / 100.0, h[9] / nr / 100.0, h[10] / nr / 100.0);
              }
              continue;
            }
            const ChainBatch none{0, 0, 0, 0, 0, 0, 0, 0};
            double2 *pack[2] = {ls.cb_hot.p, ls.cb_hot2.p};
            for (int bi = 0; bi <= C.n_batches; bi++) {
              const ChainBatch &Bp = bi > 0 ? C.h_batches[bi - 1] : none;
              const ChainBatch &Bn = bi < C.n_batches ? C.h_batches[bi] : none;
              hipLaunchKernelGGL((k_cb_step<P>), dim3(CB_BUCKETS), dim3(CHAINB_NT), 0, s, a, Bp, bi - 1, Bn, bi, C.cols.p, C.bk_ptr.p,
                                 C.bk_row.p, C.bk_lcol.p, C.bk_x.p, C.hbk_ptr.p, C.hot_rows.p, ls.cb_oldnew.p, ls.cb_part.p,
                                 pack[(bi + 1) & 1], pack[bi & 1], C.col_group.p, ls.cb_colpack.p);
              if (bi < C.n_batches)
                hipLaunchKernelGGL((k_cb_hot<P>), dim3(1), dim3(CHAINB_NT), lds_h, s, a, Bn, C.cols.p, C.hot_ptr.p, C.hot_slot.p,
                                   C.hot_x.p, pack[bi & 1], C.col_group.p, std::max(C.max_hot, 1), mhe, ls.cb_part.p, CB_BUCKETS,
                                   ls.cb_oldnew.p, ls.cb_colpack.p);
            }
            continue;
          }
          for (int bi = 0; bi < C.n_batches; bi++) {
            const ChainBatch &B = C.h_batches[bi];
            const int ncold = C.h_cold_cnt[bi];
            const int g = std::max(1, std::min(GMAX, (ncold + CHAINB_NT * 4 - 1) / (CHAINB_NT * 4)));
            hipLaunchKernelGGL((k_cb_stats<P>), dim3(g), dim3(CHAINB_NT), 0, s, a, B, C.cols.p, C.cold_ptr.p, C.cold_row.p,
                               C.cold_lcol.p, C.cold_x.p, ls.cb_part.p, C.hot_rows.p, ls.cb_hot.p);
            hipLaunchKernelGGL((k_cb_hot<P>), dim3(1), dim3(CHAINB_NT), lds_h, s, a, B, C.cols.p, C.hot_ptr.p, C.hot_slot.p,
                               C.hot_x.p, ls.cb_hot.p, C.col_group.p, std::max(C.max_hot, 1), mhe, ls.cb_part.p, g, ls.cb_oldnew.p);
            if (ncold || B.n_hot)
              hipLaunchKernelGGL((k_cb_apply<P>), dim3(g), dim3(CHAINB_NT), 0, s, a, B, C.cold_ptr.p, C.cold_row.p,
                                 C.cold_lcol.p, C.cold_x.p, ls.cb_oldnew.p, C.hot_rows.p, ls.cb_hot.p);
          }
          continue;
        }
        hipLaunchKernelGGL((k_chain_batched<P>), dim3(1), dim3(CHAINB_NT), lds_b, s, a, C.batches.p, C.n_batches, C.cols.p,
                           C.cold_ptr.p, C.cold_row.p, C.cold_lcol.p, C.cold_x.p, C.hot_ptr.p, C.hot_slot.p, C.hot_x.p,
                           C.hot_rows.p, std::max(C.max_hot, 1), mhe);
      } else {
        hipLaunchKernelGGL((k_chain<P>), dim3(1), dim3(CHAIN_WG), 0, s, a, st.chain.desc.p, st.chain.n_cols);
      }
      continue;
    }
    const ParLevel &L = st.par;
    if (L.scattered && L.tiled) {
      if constexpr (P::REC_DOUBLES == 2 && !P::QFREE) {
        TimedLaunch t(tm, s, kc.scat, P::BYTES * L.n_ent);
        const int swz = xcd_swizzle_enabled();
        const size_t lds = sizeof(double2) << L.tile_bits;
        const int nt = tile_threads(L.tile_bits);
        if (lds > 64 * 1024) {  // beyond the default dynamic-LDS limit: opt in once per kernel
          static DeviceOnce raised;
          if (raised.need()) {
            MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_tile_stats<P, UNIT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                              (int)CHAIN_LDS_MAX));
            MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_tile_apply<P, UNIT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                              (int)CHAIN_LDS_MAX));
            raised.mark();
          }
        }
        hipLaunchKernelGGL(k_tile_old, dim3((L.n_cols + 255) / 256), dim3(256), 0, s, a.theta, L.scols.p, L.n_cols,
                           ls.vnext_col.p);
        hipLaunchKernelGGL((k_tile_stats<P, UNIT>), dim3(L.n_tiles), dim3(nt), lds, s, a, L.tent.p, L.ent_val.p, L.tile_ptr.p,
                           L.tile_row0.p, ls.vnext_col.p, L.run_base.p, L.slot_pos.p, L.slots.p, L.tile_bits, L.n_tiles, swz,
                           (const int32_t *)nullptr);
        hipLaunchKernelGGL((k_tile_draw<P>), dim3((L.n_cols + 3) / 4), dim3(WG), 0, s, a, L.scols.p, L.n_cols, L.slot_ptr.p,
                           L.slots.p, ls.oldnew_col.p);
        hipLaunchKernelGGL((k_tile_apply<P, UNIT>), dim3(L.n_tiles), dim3(nt), lds, s, a, L.tent.p, L.ent_val.p,
                           L.tile_ptr.p, L.tile_row0.p, ls.oldnew_col.p, L.tile_bits, L.n_tiles, swz);
      }
      continue;
    }
    if (L.scattered) {
      TimedLaunch t(tm, s, kc.scat, P::BYTES * L.n_ent);
      const int swz = xcd_swizzle_enabled();
      const int n_wg_s = (int)((L.n_ent + WG - 1) / WG);
      hipLaunchKernelGGL((k_scat_stats<P, UNIT>), dim3(n_wg_s), dim3(WG), 0, s, a, L.ent.p, L.ent_val.p, L.n_ent, L.run_base.p,
                         L.slots.p, n_wg_s, swz);
      hipLaunchKernelGGL((k_scat_draw<P>), dim3((L.n_cols + 3) / 4), dim3(WG), 0, s, a, L.scols.p, L.n_cols, L.slot_ptr.p,
                         L.slot_idx.p, L.slots.p, ls.oldnew_col.p);
      hipLaunchKernelGGL((k_scat_apply<PA, UNIT>), dim3(n_wg_s), dim3(WG), 0, s, a, L.ent.p, L.ent_val.p, L.n_ent,
                           ls.oldnew_col.p, n_wg_s, swz);
      continue;
    }
    if (first_builds_q && &st == &plan.steps.front())
      launch_binned_level<PMainVq<UNIT>, UNIT>(s, tm, L, a, ls, kc, plan.col_row0.p);  // (only instantiated use: P == PMainV)
    else
      launch_binned_level<P, UNIT>(s, tm, L, a, ls, kc, plan.col_row0.p);
    if (L.n_huge) {
      {
        TimedLaunch t(tm, s, kc.hstats, P::STAT_BYTES * L.nnz_huge);
        hipLaunchKernelGGL((k_long_stats<P>), dim3(L.n_hchunks), dim3(WG), 0, s, a, L.hchunks.p, L.cols_huge.p, ls.partial.p);
      }
      {
        TimedLaunch t(tm, s, kc.hdraw, 16.0 * L.n_hchunks);
        hipLaunchKernelGGL((k_long_draw<P>), dim3((L.n_huge + 63) / 64), dim3(64), 0, s, a, L.cols_huge.p, L.hchunk_ptr.p,
                           L.n_huge, ls.partial.p, ls.oldnew.p);
      }
      {
        TimedLaunch t(tm, s, kc.happly, P::BYTES * L.nnz_huge);
        hipLaunchKernelGGL((k_long_apply<P>), dim3(L.n_hchunks), dim3(WG), 0, s, a, L.hchunks.p, ls.oldnew.p);
      }
    }
  }
  MFM_HIP_CHECK(hipGetLastError());
}

template <class P>
static void run_plan(hipStream_t s, Timing &tm, const StepPlan &plan, const SweepArgs &a, LongScratch &ls,
                     const SweepClasses &kc, bool unit, bool first_builds_q = false) {
  static bool lds_attr_set = false;
  if (!lds_attr_set) {  // dynamic LDS above 64 KiB must be opted into
    MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_chain_lds<P>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)CHAIN_LDS_MAX));
    lds_attr_set = true;
  }
  if (unit)
    run_plan_t<P, true>(s, tm, plan, a, ls, kc, first_builds_q);
  else
    run_plan_t<P, false>(s, tm, plan, a, ls, kc, first_builds_q);
}

// In-place sum over the ranks of `count` doubles in device memory, enqueued in order on the ctx stream.
typedef int (*mfm_allreduce_fn)(void *user, void *dev_buf, int64_t count);
// Two providers: a caller-supplied callback (mfm_set_allreduce: e.g. torch.distributed on the ctx stream), or RCCL called
// from this library on the ctx stream (mfm_comm_init: ncclAllReduce over xGMI, no interpreter in the loop). librccl is
// bound at run time (dlopen) so that single-GPU use does not depend on it.
struct Rccl {
  void *lib = nullptr;
  std::string path;  // where the bound librccl lives (dladdr of ncclAllReduce)
  int (*GetUniqueId)(void *) = nullptr;
  int (*CommInitRank)(void **, int, mfm_nccl_id, int) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*CommCount)(void *, int *) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  // Resolution order: (1) a librccl the process has already mapped (a torch process: torch/lib/librccl.so -- two RCCL copies
  // in one process would each open their own xGMI rings), (2) the directory the HIP runtime in use was loaded from (a wheel
  // that bundles libamdhip64 bundles its RCCL next to it), (3) the dynamic loader's search path, (4) /opt/rocm/lib.
  static void *open_any() {
    static const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *n : names)
      if (void *h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD)) return h;
    Dl_info di;
    if (dladdr((const void *)&hipGetDeviceCount, &di) && di.dli_fname) {
      std::string dir(di.dli_fname);
      const size_t slash = dir.rfind('/');
      if (slash != std::string::npos) {
        dir.resize(slash + 1);
        for (const char *n : names)
          if (void *h = dlopen((dir + n).c_str(), RTLD_NOW | RTLD_GLOBAL)) return h;
      }
    }
    for (const char *n : names)
      if (void *h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) return h;
    for (const char *n : {"/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"})
      if (void *h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) return h;
    return nullptr;
  }
  static Rccl &get() {
    static Rccl r;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (!r.lib) {
      void *h = open_any();
      if (!h) throw Error(MFM_ERR_RUNTIME, std::string("cannot load librccl.so: ") + dlerror());
      r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
      r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
      r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
      r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
      r.CommCount = (decltype(r.CommCount))dlsym(h, "ncclCommCount");
      r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
      if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy)
        throw Error(MFM_ERR_RUNTIME, "librccl.so lacks the expected entry points");
      Dl_info di;
      if (dladdr((const void *)r.AllReduce, &di) && di.dli_fname) r.path = di.dli_fname;
      r.lib = h;
    }
    return r;
  }
  void check(int rc, const char *what) const {
    if (rc != 0)
      throw Error(MFM_ERR_RUNTIME, std::string(what) + " failed: " + (GetErrorString ? GetErrorString(rc) : "rccl error"));
  }
};

struct Comm {
  mfm_allreduce_fn fn = nullptr;
  void *user = nullptr;
  void *nccl = nullptr;        // ncclComm_t (mfm_comm_init)
  hipStream_t stream = nullptr;  // the ctx stream the native collective is enqueued on
  int rank = 0, world = 1;
  bool shard_set = false;  // mfm_set_shard / mfm_comm_init told us this rank's place (else: rank 0 <=> row offset 0)
  mutable int64_t calls = 0, doubles = 0;
  bool active() const { return fn != nullptr || nccl != nullptr; }
  void allreduce(void *buf, int64_t count) const {
    if (count <= 0 || !active()) return;
    calls++;
    doubles += count;
    if (nccl) {
      Rccl &r = Rccl::get();
      r.check(r.AllReduce(buf, buf, (size_t)count, /*ncclDouble*/ 8, /*ncclSum*/ 0, nccl, stream), "ncclAllReduce");
      return;
    }
    if (fn(user, buf, count) != 0) throw Error(MFM_ERR_RUNTIME, "all-reduce callback failed");
  }
  ~Comm() {
    if (nccl) (void)Rccl::get().CommDestroy(nccl);
  }
};

// Row-sharded sweep of one table: per level  statistics -> S -> all-reduce -> draw -> apply.
template <class P, bool UNIT>
static void run_plan_sharded_t(hipStream_t s, Timing &tm, const StepPlan &plan, const SweepArgs &a, LongScratch &ls,
                               const SweepClasses &kc, const Comm &comm) {
  for (const Step &st : plan.steps) {
    if (st.is_chain) throw Error(MFM_ERR_RUNTIME, "internal: chain step in a sharded plan");
    const ParLevel &L = st.par;
    double2 *S = ls.S_col.p;
    if (L.scattered) {
      TimedLaunch t(tm, s, kc.scat, P::STAT_BYTES * L.n_ent);
      const int n_wg_s = (int)((L.n_ent + WG - 1) / WG);
      hipLaunchKernelGGL((k_scat_stats<P, UNIT>), dim3(n_wg_s), dim3(WG), 0, s, a, L.ent.p, L.ent_val.p, L.n_ent, L.run_base.p,
                         L.slots.p, n_wg_s, xcd_swizzle_enabled());
      hipLaunchKernelGGL(k_scat_sum, dim3((L.n_cols + 3) / 4), dim3(WG), 0, s, L.scols.p, L.n_cols, L.slot_ptr.p, L.slot_idx.p,
                         L.slots.p, S);
    } else {
      const int grid = L.n_wg + (L.n_w16 + 3) / 4 + (L.n_w4 + 3) / 4 + (L.n_w1 + 3) / 4;
      if (grid) {
        TimedLaunch t(tm, s, kc.heavy, P::STAT_BYTES * (L.nnz_heavy + L.nnz_light));
        hipLaunchKernelGGL((k_level_split<P, UNIT, 1>), dim3(grid), dim3(WG), 0, s, a, L.cols_w1.p, L.n_w1, L.cols_w4.p, L.n_w4,
                           L.cols_w16.p, L.n_w16, L.cols_wg.p, L.n_wg, S, (const double2 *)nullptr);
      }
      if (L.n_huge) {
        TimedLaunch t(tm, s, kc.hstats, P::STAT_BYTES * L.nnz_huge);
        hipLaunchKernelGGL((k_long_stats<P>), dim3(L.n_hchunks), dim3(WG), 0, s, a, L.hchunks.p, L.cols_huge.p, ls.partial.p);
        hipLaunchKernelGGL(k_long_sum, dim3((L.n_huge + 63) / 64), dim3(64), 0, s, L.cols_huge.p, L.hchunk_ptr.p, L.n_huge,
                           ls.partial.p, S);
      }
    }
    MFM_HIP_CHECK(hipGetLastError());
    comm.allreduce(S + L.jmin, 2 * (int64_t)(L.jmax - L.jmin + 1));
    hipLaunchKernelGGL((k_col_draw<P>), dim3((L.n_all + 255) / 256), dim3(256), 0, s, a, L.cols_all.p, L.n_all, S,
                       ls.oldnew_col.p);
    if (L.scattered) {
      TimedLaunch t(tm, s, kc.scat, P::BYTES * L.n_ent);
      const int n_wg_s = (int)((L.n_ent + WG - 1) / WG);
      hipLaunchKernelGGL((k_scat_apply<P, UNIT>), dim3(n_wg_s), dim3(WG), 0, s, a, L.ent.p, L.ent_val.p, L.n_ent,
                         ls.oldnew_col.p, n_wg_s, xcd_swizzle_enabled());
    } else {
      const int grid = L.n_wg + (L.n_w16 + 3) / 4 + (L.n_w4 + 3) / 4 + (L.n_w1 + 3) / 4;
      if (grid) {
        TimedLaunch t(tm, s, kc.light, P::BYTES * (L.nnz_heavy + L.nnz_light));
        hipLaunchKernelGGL((k_level_split<P, UNIT, 2>), dim3(grid), dim3(WG), 0, s, a, L.cols_w1.p, L.n_w1, L.cols_w4.p, L.n_w4,
                           L.cols_w16.p, L.n_w16, L.cols_wg.p, L.n_wg, (double2 *)nullptr, ls.oldnew_col.p);
      }
      if (L.n_huge) {
        TimedLaunch t(tm, s, kc.happly, P::BYTES * L.nnz_huge);
        hipLaunchKernelGGL((k_long_apply_col<P>), dim3(L.n_hchunks), dim3(WG), 0, s, a, L.hchunks.p, L.cols_huge.p,
                           ls.oldnew_col.p);
      }
    }
    MFM_HIP_CHECK(hipGetLastError());
  }
}
template <class P>
static void run_plan_sharded(hipStream_t s, Timing &tm, const StepPlan &plan, const SweepArgs &a, LongScratch &ls,
                             const SweepClasses &kc, bool unit, const Comm &comm) {
  if (unit)
    run_plan_sharded_t<P, true>(s, tm, plan, a, ls, kc, comm);
  else
    run_plan_sharded_t<P, false>(s, tm, plan, a, ls, kc, comm);
}

// Can the latent sweep of this table run in the split layout (run_plan_soa)? PAR levels only, every
// scattered level on the row-tile path, and a first level that rebuilds q (so q never has to be packed).
static inline bool plan_supports_soa(const StepPlan &plan) {
  if (plan.steps.size() < 2) return false;
  for (const Step &st : plan.steps)
    if (st.is_chain || (st.par.scattered && !st.par.tiled)) return false;
  const ParLevel &first = plan.steps.front().par;
  return !first.scattered && first.first_and_once;
}

template <class P, bool UNIT>
static void launch_huge(hipStream_t s, Timing &tm, const ParLevel &L, const SweepArgs &a, LongScratch &ls,
                        const SweepClasses &kc) {
  if (!L.n_huge) return;
  {
    TimedLaunch t(tm, s, kc.hstats, P::STAT_BYTES * L.nnz_huge);
    hipLaunchKernelGGL((k_long_stats<P>), dim3(L.n_hchunks), dim3(WG), 0, s, a, L.hchunks.p, L.cols_huge.p, ls.partial.p);
  }
  {
    TimedLaunch t(tm, s, kc.hdraw, 16.0 * L.n_hchunks);
    hipLaunchKernelGGL((k_long_draw<P>), dim3((L.n_huge + 63) / 64), dim3(64), 0, s, a, L.cols_huge.p, L.hchunk_ptr.p,
                       L.n_huge, ls.partial.p, ls.oldnew.p);
  }
  {
    TimedLaunch t(tm, s, kc.happly, P::BYTES * L.nnz_huge);
    hipLaunchKernelGGL((k_long_apply<P>), dim3(L.n_hchunks), dim3(WG), 0, s, a, L.hchunks.p, ls.oldnew.p);
  }
}

// split-layout sweeps: the first factor's first level reads e from the interleaved array, the last factor's final
// apply pass writes it back there (SweepArgs::aos is set by the caller for the whole sweep)
static inline SweepArgs soa_first_args(const SweepArgs &a) {
  SweepArgs r = a;
  if (a.aos) {
    r.e_src = (const double *)a.aos;
    r.e_src_stride = 2;
  }
  return r;
}
static inline SweepArgs soa_final_args(const SweepArgs &a, bool is_final_factor) {
  SweepArgs r = a;
  if (!is_final_factor) r.aos = nullptr;
  return r;
}

// Is the last level's apply pass fusable with the next factor's first level (k_tile_apply_next)?
static inline bool plan_supports_fused_next(const StepPlan &plan) {
  if (!plan_supports_soa(plan) || !plan.aligned_tiles) return false;
  const ParLevel &first = plan.steps.front().par, &last = plan.steps.back().par;
  return first.contig && last.scattered && last.tiled;
}

// statistics (unless already produced by the previous factor's fused pass) and draw of a row-tile level
template <class P, bool UNIT>
static void launch_tile_head(hipStream_t s, const ParLevel &L, const SweepArgs &a, LongScratch &ls, int swz,
                             const double *theta_next = nullptr, bool have_stats = false) {
  const size_t lds = sizeof(double2) << L.tile_bits;
  const int nt = tile_threads(L.tile_bits);
  if (!have_stats) {
    hipLaunchKernelGGL(k_tile_old, dim3((L.n_cols + 255) / 256), dim3(256), 0, s, a.theta, L.scols.p, L.n_cols, ls.vnext_col.p);
    hipLaunchKernelGGL((k_tile_stats<P, UNIT, true>), dim3(L.n_tiles), dim3(nt), lds, s, a, L.tent.p, L.ent_val.p,
                       L.tile_ptr.p, L.tile_row0.p, ls.vnext_col.p, L.run_base.p, L.slot_pos.p, L.slots.p, L.tile_bits,
                       L.n_tiles, swz, (const int32_t *)nullptr);
  }
  hipLaunchKernelGGL((k_tile_draw<P>), dim3((L.n_cols + 3) / 4), dim3(WG), 0, s, a, L.scols.p, L.n_cols, L.slot_ptr.p,
                     L.slots.p, ls.oldnew_col.p, theta_next, ls.vnext_col.p);
}

// the fused pass keeps 256 bytes of reduction scratch behind its tile: above the default dynamic-LDS limit
template <bool UNIT>
static void raise_fused_lds_limit() {
  static DeviceOnce raised;
  if (!raised.need()) return;
  const int lim = (int)CHAIN_LDS_MAX;
  MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_tile_apply_next<UNIT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
  MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_tile_apply_next<UNIT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
  raised.mark();
}

// fused flow, first-level columns longer than a tile: k_tile_apply_next left their tiles' partial statistics
template <bool UNIT>
static void launch_long_finish(hipStream_t s, Timing &tm, const StepPlan &plan, const ParLevel &L, const SweepArgs &an,
                               LongScratch &ls, const SweepClasses &kc, int stats, size_t lds, int nt,
                               const double *vnext = nullptr) {
  if (!plan.n_long_cols) return;
  if (!vnext) vnext = ls.vnext_col.p;
  TimedLaunch t(tm, s, kc.coop, 52.0 * plan.n_long_tiles * (double)(1 << L.tile_bits));
  hipLaunchKernelGGL((k_long_tile_draw<PMainV>), dim3((plan.n_long_cols + 63) / 64), dim3(64), 0, s, an, plan.long_cols.p,
                     plan.long_tile_ptr.p, plan.long_tiles.p, plan.n_long_cols, plan.long_partial.p, plan.oldnew_long.p);
  hipLaunchKernelGGL((k_tile_long_finish<UNIT>), dim3(plan.n_long_tiles), dim3(nt), lds, s, an, L.tent.p, L.ent_val.p,
                     L.tile_ptr.p, L.tile_row0.p, L.tile_bits, plan.long_tiles.p, plan.tile_long_idx.p, plan.oldnew_long.p,
                     plan.long_cols.p, vnext, stats, L.run_base.p, L.slot_pos.p, L.slots.p);
}

// Split-layout sweep of a plan with MORE than two levels, all but the first on row tiles (several one-hot fields,
// table sorted by the first): per factor L - 1 passes over the tiles instead of 3 (L - 1) + 1 --
//   draw T1 | [apply T_l + statistics T_(l+1) | draw T_(l+1)] for l = 1 .. L-2 | apply T_(L-1) + q rebuild + the whole
//   first level of the next factor + statistics T1 of the next factor (k_tile_apply_next<.., SPLIT>).
static inline bool plan_supports_fused_multi(const StepPlan &plan) {
  if (!plan_supports_fused_next(plan) || plan.steps.size() < 3) return false;
  for (size_t i = 1; i < plan.steps.size(); i++)
    if (plan.steps[i].is_chain || !plan.steps[i].par.scattered || !plan.steps[i].par.tiled) return false;
  return true;
}

struct Comm;
template <class PS, class PA, bool UNIT>
static void run_level_sharded(hipStream_t s, Timing &tm, const ParLevel &L, const SweepArgs &a, LongScratch &ls,
                              const SweepClasses &kc, const Comm &comm, const int32_t *ccols, int n_c);

template <bool UNIT, class ArgsOf>
static void run_sweep_soa_multi(hipStream_t s, Timing &tm, const StepPlan &plan, ArgsOf args, int f_begin, int f_end,
                                LongScratch &ls, const SweepClasses &kc, const Comm *comm = nullptr) {
  // comm != null: row-sharded (plan.sharded_tiles): every tile level's slot sums are all-reduced before its draw, the
  // special first-level columns go through run_level_sharded, the locally complete ones stay local (see
  // run_sweep_soa_sharded)
  const int swz = xcd_swizzle_enabled();
  raise_fused_lds_limit<UNIT>();
  {
    static DeviceOnce raised;
    if (raised.need()) {
      MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_tile_apply_next<UNIT, false, true>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHAIN_LDS_MAX));
      raised.mark();
    }
  }
  const int nl = (int)plan.steps.size();
  const ParLevel &L1 = plan.steps.front().par, &T1 = plan.steps[1].par, &TL = plan.steps.back().par;
  const size_t lds = sizeof(double2) << T1.tile_bits;
  const int nt = tile_threads(T1.tile_bits);
  // every tile level a one-hot field (one entry per row): the next q comes from the entry streams (MULTIQ)
  bool multiq = nl - 1 <= 7 && !std::getenv("MFM_NO_FUSED_MULTIQ");
  for (int l = 1; l < nl; l++) multiq = multiq && plan.steps[l].par.covers_rows_once;
  if (ls.vnext_lvl.size() < (size_t)nl) ls.vnext_lvl.resize((size_t)nl);
  for (int l = 1; l < nl; l++)
    if (ls.vnext_lvl[l].n < (size_t)plan.steps[l].par.n_cols) ls.vnext_lvl[l].alloc((size_t)plan.steps[l].par.n_cols);
  {
    static DeviceOnce raised2;
    if (raised2.need()) {
      MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_tile_apply_next<UNIT, false, true, true>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHAIN_LDS_MAX));
      raised2.mark();
    }
  }
  auto draw = [&](int l, const SweepArgs &a, const double *theta_next) {
    const ParLevel &T = plan.steps[l].par;
    if (!comm) {
      hipLaunchKernelGGL((k_tile_draw<PMainV>), dim3((T.n_cols + 3) / 4), dim3(WG), 0, s, a, T.scols.p, T.n_cols, T.slot_ptr.p,
                         T.slots.p, ls.oldnew_col.p, theta_next, ls.vnext_lvl[l].p);
      return;
    }
    hipLaunchKernelGGL(k_tile_sum, dim3((T.n_cols + 3) / 4), dim3(WG), 0, s, T.n_cols, T.slot_ptr.p, T.slots.p, ls.S_col.p);
    MFM_HIP_CHECK(hipGetLastError());
    comm->allreduce(ls.S_col.p, 2 * (int64_t)T.n_cols);
    hipLaunchKernelGGL((k_tile_draw_S<PMainV>), dim3((T.n_cols + 255) / 256), dim3(256), 0, s, a, T.scols.p, T.n_cols, ls.S_col.p,
                       ls.oldnew_col.p, theta_next, ls.vnext_lvl[l].p);
  };
  auto first_level_rest = [&](const SweepArgs &ax) {  // (sharded) special first-level columns, then their tiles' statistics
    if (!comm) return;
    if (plan.n_special)
      run_level_sharded<PMainVsq<UNIT>, PMainVsqA<UNIT>, UNIT>(s, tm, plan.special_level, ax, ls, kc, *comm, plan.special_cols.p,
                                                               plan.n_special);
  };
  for (int f = f_begin; f < f_end; f++) {
    const SweepArgs a = args(f);
    const bool next = f + 1 < f_end;
    SweepArgs an = next ? args(f + 1) : a;
    an.row0 = plan.col_row0.p;
    if (f == f_begin) {
      const ParLevel &Lf = comm ? plan.local_level : L1;  // (sharded: the columns complete on this rank)
      const SweepArgs a0 = soa_first_args(a);
      launch_binned_level<PMainVsq<UNIT>, UNIT>(s, tm, Lf, a0, ls, kc, plan.col_row0.p);
      launch_huge<PMainVsq<UNIT>, UNIT>(s, tm, Lf, a0, ls, kc);
      first_level_rest(a0);
      TimedLaunch t(tm, s, kc.scat, 20.0 * T1.n_ent);
      hipLaunchKernelGGL(k_tile_old, dim3((T1.n_cols + 255) / 256), dim3(256), 0, s, a.theta, T1.scols.p, T1.n_cols,
                         ls.told_col.p);
      hipLaunchKernelGGL((k_tile_stats<PMainV, UNIT, true>), dim3(T1.n_tiles), dim3(nt), lds, s, a, T1.tent.p, T1.ent_val.p,
                         T1.tile_ptr.p, T1.tile_row0.p, ls.told_col.p, T1.run_base.p, T1.slot_pos.p, T1.slots.p, T1.tile_bits,
                         T1.n_tiles, swz, (const int32_t *)nullptr);
    }
    const double *tn = next ? (const double *)an.theta : (const double *)nullptr;
    {
      TimedLaunch t(tm, s, kc.scat, 16.0 * T1.n_runs);
      draw(1, a, tn);
    }
    for (int l = 1; l + 1 < nl; l++) {
      const ParLevel &A = plan.steps[l].par, &S = plan.steps[l + 1].par;
      TimedLaunch t(tm, s, kc.scat, 32.0 * plan.n_state_rows + (UNIT ? 4.0 : 12.0) * (A.n_ent + S.n_ent) + 32.0 * S.n_runs);
      hipLaunchKernelGGL(k_tile_old, dim3((S.n_cols + 255) / 256), dim3(256), 0, s, a.theta, S.scols.p, S.n_cols, ls.told_col.p);
      hipLaunchKernelGGL((k_tile_apply_stats<UNIT>), dim3(A.n_tiles), dim3(nt), lds, s, a, A.tent.p, A.ent_val.p, A.tile_ptr.p,
                         ls.oldnew_col.p, S.tent.p, S.ent_val.p, S.tile_ptr.p, ls.told_col.p, S.run_base.p, S.slot_pos.p,
                         S.slots.p, A.tile_row0.p, A.tile_bits, A.n_tiles, swz);
      draw(l + 1, a, tn);
    }
    if (!next) {
      TimedLaunch t(tm, s, kc.scat, 28.0 * TL.n_ent);
      hipLaunchKernelGGL((k_tile_apply<PMainV, UNIT, true, false>), dim3(TL.n_tiles), dim3(nt), lds, s, soa_final_args(a, true),
                         TL.tent.p, TL.ent_val.p, TL.tile_ptr.p, TL.tile_row0.p, ls.oldnew_col.p, TL.tile_bits, TL.n_tiles, swz);
      continue;
    }
    {
      TimedLaunch t(tm, s, KC_SWEEP_V_FUSED,
                    32.0 * plan.n_state_rows + (UNIT ? 4.0 : 12.0) * (TL.n_ent + T1.n_ent) + 16.0 * T1.n_runs);
      SweepArgs af = a;
      af.row0 = plan.col_row0.p;
      FuseArgs fa{an.theta,  an.z,          an.lambda,     an.mu,      plan.fuse_desc.p, plan.fuse_col_ptr.p, ls.vnext_lvl[1].p,
                  1,         T1.run_base.p, T1.slot_pos.p, T1.slots.p, plan.solo_col.p,  plan.long_partial.p,
                  T1.tent.p, T1.ent_val.p,  T1.tile_ptr.p};
      fa.vnextA = ls.vnext_lvl[nl - 1].p;
      fa.n_extra = 0;
      for (int l = 1; l + 1 < nl && multiq; l++) {
        const ParLevel &E = plan.steps[l].par;
        fa.ex_tent[fa.n_extra] = E.tent.p;
        fa.ex_tval[fa.n_extra] = E.ent_val.p;
        fa.ex_tile_ptr[fa.n_extra] = E.tile_ptr.p;
        fa.ex_vnext[fa.n_extra] = ls.vnext_lvl[l].p;
        fa.n_extra++;
      }
      if (multiq)
        hipLaunchKernelGGL((k_tile_apply_next<UNIT, false, true, true>), dim3(TL.n_tiles), dim3(nt), lds + 256, s, af, TL.tent.p,
                           TL.ent_val.p, TL.tile_ptr.p, TL.tile_row0.p, ls.oldnew_col.p, TL.tile_bits, TL.n_tiles, swz, fa);
      else
        hipLaunchKernelGGL((k_tile_apply_next<UNIT, false, true>), dim3(TL.n_tiles), dim3(nt), lds + 256, s, af, TL.tent.p,
                           TL.ent_val.p, TL.tile_ptr.p, TL.tile_row0.p, ls.oldnew_col.p, TL.tile_bits, TL.n_tiles, swz, fa);
    }
    launch_long_finish<UNIT>(s, tm, plan, T1, an, ls, kc, 1, lds, nt, ls.vnext_lvl[1].p);
    first_level_rest(an);
    if (comm && plan.n_solo_tiles) {
      TimedLaunch t(tm, s, kc.scat, 20.0 * plan.n_solo_tiles * (1 << T1.tile_bits));
      hipLaunchKernelGGL((k_tile_stats<PMainV, UNIT, true>), dim3(plan.n_solo_tiles), dim3(nt), lds, s, an, T1.tent.p,
                         T1.ent_val.p, T1.tile_ptr.p, T1.tile_row0.p, ls.vnext_lvl[1].p, T1.run_base.p, T1.slot_pos.p,
                         T1.slots.p, T1.tile_bits, T1.n_tiles, swz, plan.solo_tiles.p);
    }
  }
  MFM_HIP_CHECK(hipGetLastError());
}

// Latent sweep of factors [f_begin, f_end) in the split layout: args(f).state = e[N], .state2 = q[N].
// fuse: the last level's apply pass also runs the next factor's first level (short columns) on the tile.
template <bool UNIT, class ArgsOf>
static void run_sweep_soa(hipStream_t s, Timing &tm, const StepPlan &plan, ArgsOf args, int f_begin, int f_end,
                          LongScratch &ls, const SweepClasses &kc, bool fuse) {
  if (fuse && plan_supports_fused_multi(plan) && !std::getenv("MFM_NO_FUSED_MULTI")) {
    run_sweep_soa_multi<UNIT>(s, tm, plan, args, f_begin, f_end, ls, kc);
    return;
  }
  const int swz = xcd_swizzle_enabled();
  if (fuse) raise_fused_lds_limit<UNIT>();
  // two-level plan: the fused pass also produces the next factor's last-level statistics
  const int fuse_stats = fuse && plan.steps.size() == 2 && !std::getenv("MFM_NO_FUSED_STATS") ? 1 : 0;
  {
    static DeviceOnce raised;  // tiles beyond the default dynamic-LDS limit: opt in once
    if (raised.need() && plan.tile_bits > 12) {
      const int lim = (int)CHAIN_LDS_MAX;
      MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_tile_stats<PMainV, UNIT, true>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lim));
      MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_tile_apply<PMainV, UNIT, true, true>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lim));
      MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_tile_apply<PMainV, UNIT, true, false>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lim));
      raised.mark();
    }
  }
  for (int f = f_begin; f < f_end; f++) {
    const SweepArgs a = args(f);
    for (const Step &st : plan.steps) {
      const ParLevel &L = st.par;
      const bool first = &st == &plan.steps.front(), last = &st == &plan.steps.back();
      if (L.scattered) {  // (tiled: plan_supports_soa)
        const size_t lds = sizeof(double2) << L.tile_bits;
        const int nt = tile_threads(L.tile_bits);
        if (last && fuse && f + 1 < f_end) {
          SweepArgs an = args(f + 1);
          an.row0 = plan.col_row0.p;
          const bool two = plan.steps.size() == 2 && L.covers_rows_once && !std::getenv("MFM_NO_FUSED_TWO");
          {
            TimedLaunch t(tm, s, kc.scat, 20.0 * L.n_ent);
            launch_tile_head<PMainV, UNIT>(s, L, a, ls, swz, an.theta, fuse_stats && f > f_begin);
          }
          {
            // algorithmic bytes of the FUSED pass: e, q read + written once (32 B / row), the 4-byte entry stream
            // (+ 8-byte values when the table is not unit-valued), one 16-byte statistics slot per run
            TimedLaunch t(tm, s, KC_SWEEP_V_FUSED,
                          32.0 * plan.n_state_rows + (UNIT ? 4.0 : 12.0) * L.n_ent + (fuse_stats ? 16.0 * L.n_runs : 0.0));
            SweepArgs af = a;
            af.row0 = plan.col_row0.p;
            FuseArgs fa{an.theta,   an.z,         an.lambda,    an.mu,     plan.fuse_desc.p, plan.fuse_col_ptr.p, ls.vnext_col.p,
                        fuse_stats, L.run_base.p, L.slot_pos.p, L.slots.p, plan.solo_col.p,  plan.long_partial.p,
                        nullptr,    nullptr,      nullptr};
            if (two)
              hipLaunchKernelGGL((k_tile_apply_next<UNIT, true>), dim3(L.n_tiles), dim3(nt), lds + 256, s, af, L.tent.p,
                                 L.ent_val.p, L.tile_ptr.p, L.tile_row0.p, ls.oldnew_col.p, L.tile_bits, L.n_tiles, swz, fa);
            else
              hipLaunchKernelGGL((k_tile_apply_next<UNIT, false>), dim3(L.n_tiles), dim3(nt), lds + 256, s, af, L.tent.p,
                                 L.ent_val.p, L.tile_ptr.p, L.tile_row0.p, ls.oldnew_col.p, L.tile_bits, L.n_tiles, swz, fa);
          }
          // first-level columns longer than a tile: draw from their tiles' partial statistics, second pass
          launch_long_finish<UNIT>(s, tm, plan, L, an, ls, kc, fuse_stats, lds, nt);
          if (fuse_stats && plan.n_solo_tiles) {
            TimedLaunch t(tm, s, kc.scat, 20.0 * plan.n_solo_tiles * (1 << L.tile_bits));
            hipLaunchKernelGGL((k_tile_stats<PMainV, UNIT, true>), dim3(plan.n_solo_tiles), dim3(nt), lds, s, an, L.tent.p,
                               L.ent_val.p, L.tile_ptr.p, L.tile_row0.p, ls.vnext_col.p, L.run_base.p, L.slot_pos.p,
                               L.slots.p, L.tile_bits, L.n_tiles, swz, plan.solo_tiles.p);
          }
          continue;
        }
        TimedLaunch t(tm, s, kc.scat, (last ? 48.0 : 56.0) * L.n_ent);
        launch_tile_head<PMainV, UNIT>(s, L, a, ls, swz, nullptr, last && fuse && fuse_stats && f > f_begin);
        if (last)
          hipLaunchKernelGGL((k_tile_apply<PMainV, UNIT, true, false>), dim3(L.n_tiles), dim3(nt), lds, s,
                             soa_final_args(a, f + 1 == f_end), L.tent.p, L.ent_val.p, L.tile_ptr.p, L.tile_row0.p,
                             ls.oldnew_col.p, L.tile_bits, L.n_tiles, swz);
        else
          hipLaunchKernelGGL((k_tile_apply<PMainV, UNIT, true, true>), dim3(L.n_tiles), dim3(nt), lds, s, a, L.tent.p,
                             L.ent_val.p, L.tile_ptr.p, L.tile_row0.p, ls.oldnew_col.p, L.tile_bits, L.n_tiles, swz);
        continue;
      }
      if (first) {
        if (fuse && f > f_begin) continue;  // done by the previous factor's fused apply pass
        const SweepArgs a0 = f == f_begin ? soa_first_args(a) : a;
        launch_binned_level<PMainVsq<UNIT>, UNIT>(s, tm, L, a0, ls, kc, plan.col_row0.p);
        launch_huge<PMainVsq<UNIT>, UNIT>(s, tm, L, a0, ls, kc);
      } else if (last) {
        launch_binned_level<PMainVsl, UNIT>(s, tm, L, a, ls, kc, plan.col_row0.p);
        launch_huge<PMainVsl, UNIT>(s, tm, L, a, ls, kc);
      } else {
        launch_binned_level<PMainVs, UNIT>(s, tm, L, a, ls, kc, plan.col_row0.p);
        launch_huge<PMainVs, UNIT>(s, tm, L, a, ls, kc);
      }
    }
  }
  MFM_HIP_CHECK(hipGetLastError());
}

// ---- two-field pass (mfm_mf_kernels.hpp) --------------------------------------------------------------------
// Two levels, each covering every row once; the first contiguous (table sorted by it) with tiles aligned to it, the
// second on those row tiles.
static inline bool plan_supports_mf(const StepPlan &plan) {
  if (!plan_supports_fused_next(plan) || plan.steps.size() != 2 || !plan.mf_ready || plan.n_solo_tiles) return false;
  const ParLevel &first = plan.steps.front().par, &last = plan.steps.back().par;
  return first.first_and_once && last.covers_rows_once && last.tile_bits >= 9 && last.tile_bits <= 13;
}

static inline int mf_rows_per_thread(int64_t n_rows) {
  const char *e = std::getenv("MFM_MF_K");  // (read per sweep: the tests switch it inside one process)
  if (e) return std::atoi(e) == 4 ? 4 : 8;
  return n_rows < ((int64_t)1 << 20) ? 4 : 8;  // short tables: more, shorter-lived threads per tile
}

// Latent sweep of factors [f_begin, f_end): args(f).state = e[N] (split array), .aos = the interleaved {e, q} array the
// residual is read from at the start and written back to at the end. K + 1 passes over the residual for K factors.
template <bool UNIT, class ArgsOf>
static void run_sweep_mf(hipStream_t s, Timing &tm, const StepPlan &plan, ArgsOf args, int f_begin, int f_end, LongScratch &ls,
                         const SweepClasses &kc, const Comm *comm = nullptr) {
  // comm != null: row-sharded, every first-level column complete on one rank (shards cut between two of them): the user
  // level runs locally inside the tiles, per factor ONE all-reduce carries the item level's statistics (2 n_cols doubles)
  const ParLevel &L = plan.steps.back().par;
  const int swz = xcd_swizzle_enabled();
  const int KR = mf_rows_per_thread(plan.n_state_rows);
  const int nt = (1 << L.tile_bits) / KR;
  const size_t lds = mf_lds_bytes(L.tile_bits, UNIT);
  {
    static DeviceOnce raised;
    if (raised.need()) {
      const int lim = (int)CHAIN_LDS_MAX;
      MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_mf_pass<UNIT, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
      MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_mf_pass<UNIT, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
      MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_mf_long_finish<UNIT, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
      MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_mf_long_finish<UNIT, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
      raised.mark();
    }
  }
  if (ls.dv_col.n < (size_t)L.n_cols) ls.dv_col.alloc((size_t)std::max(L.n_cols, 1));
  const SweepArgs a0 = args(f_begin);
  MfArgs m;
  std::memset(&m, 0, sizeof(m));
  m.E = (double *)a0.state;
  m.tile_row0 = L.tile_row0.p;
  m.tile_ptr = L.tile_ptr.p;
  m.tent = L.tent.p;
  m.tval = L.ent_val.p;
  m.tile_bits = L.tile_bits;
  m.n_tiles = L.n_tiles;
  m.swz = swz;
  m.ucap = mf_user_cap(L.tile_bits);
  m.dv = ls.dv_col.p;
  m.udesc = plan.fuse_desc.p;
  m.ucol_ptr = plan.fuse_col_ptr.p;
  m.upart = plan.mf_upart.p;
  m.chunk = plan.mf_chunk.p;
  m.chunk_ptr = plan.mf_chunk_ptr.p;
  m.alpha = a0.alpha;
  m.dbg = std::getenv("MFM_MF_DBG") ? std::atoi(std::getenv("MFM_MF_DBG")) : 0;
  m.colptr = a0.colptr;
  m.cval = a0.val;
  m.col_row0 = plan.col_row0.p;
  m.run_base = L.run_base.p;
  m.slot_pos = L.slot_pos.p;
  m.slots = L.slots.p;
  m.solo_col = plan.n_long_cols ? plan.solo_col.p : nullptr;
  m.long_partial = plan.long_partial.p;
  // algorithmic bytes of one pass: e read + written once (16 B / row), the entry stream (4 B, + 8 B values and 8 B of
  // first-level values per row when the table is not unit-valued), one 16-byte statistics slot per run
  const double pass_bytes = 16.0 * plan.n_state_rows + (UNIT ? 4.0 : 20.0) * L.n_ent;
  auto pass = [&](const SweepArgs *cur, const SweepArgs *next, bool first, bool last) {
    m.e_in = first && a0.aos ? (const double *)a0.aos : m.E;
    m.e_in_stride = first && a0.aos ? 2 : 1;
    m.e_out = last && a0.aos ? (double *)a0.aos : m.E;
    m.e_out_stride = last && a0.aos ? 2 : 1;
    m.do_apply = cur ? 1 : 0;
    m.do_next = next ? 1 : 0;
    m.theta_cur = cur ? cur->theta : nullptr;
    m.theta_next = next ? next->theta : nullptr;
    m.z_next = next ? next->z : nullptr;
    m.lam_next = next ? next->lambda : nullptr;
    m.mu_next = next ? next->mu : nullptr;
    {
      TimedLaunch t(tm, s, KC_SWEEP_V_FUSED, pass_bytes + (next ? 16.0 * L.n_runs : 0.0));
      if (KR == 8)
        hipLaunchKernelGGL((k_mf_pass<UNIT, 8>), dim3(L.n_tiles), dim3(nt), lds, s, m);
      else
        hipLaunchKernelGGL((k_mf_pass<UNIT, 4>), dim3(L.n_tiles), dim3(nt), lds, s, m);
    }
    if (next && plan.n_long_cols) {
      // first-level columns longer than a tile: draw from their tiles' partial statistics, second pass over those tiles
      TimedLaunch t(tm, s, kc.coop, 36.0 * plan.n_long_tiles * (double)(1 << L.tile_bits));
      SweepArgs an = *next;
      hipLaunchKernelGGL((k_long_tile_draw<PMainV>), dim3((plan.n_long_cols + 63) / 64), dim3(64), 0, s, an, plan.long_cols.p,
                         plan.long_tile_ptr.p, plan.long_tiles.p, plan.n_long_cols, plan.long_partial.p, plan.oldnew_long.p);
      MfArgs mf = m;
      mf.e_out = m.E;  // (a pass with a next factor never is the sweep's last)
      mf.e_out_stride = 1;
      if (KR == 8)
        hipLaunchKernelGGL((k_mf_long_finish<UNIT, 8>), dim3(plan.n_long_tiles), dim3(nt), lds, s, mf, plan.long_tiles.p,
                           plan.tile_long_idx.p, plan.oldnew_long.p, plan.long_cols.p);
      else
        hipLaunchKernelGGL((k_mf_long_finish<UNIT, 4>), dim3(plan.n_long_tiles), dim3(nt), lds, s, mf, plan.long_tiles.p,
                           plan.tile_long_idx.p, plan.oldnew_long.p, plan.long_cols.p);
    }
  };
  {
    TimedLaunch t(tm, s, kc.scat, 24.0 * L.n_cols);
    hipLaunchKernelGGL(k_mf_gather, dim3((L.n_cols + 255) / 256), dim3(256), 0, s, a0.theta, L.scols.p, L.n_cols, ls.dv_col.p);
  }
  pass(nullptr, &a0, true, false);
  for (int f = f_begin; f < f_end; f++) {
    const SweepArgs a = args(f);
    const bool more = f + 1 < f_end;
    SweepArgs an;
    if (more) an = args(f + 1);
    {
      TimedLaunch t(tm, s, kc.scat, 16.0 * L.n_runs + 56.0 * L.n_cols);
      if (comm && comm->active()) {
        hipLaunchKernelGGL(k_tile_sum, dim3((L.n_cols + 3) / 4), dim3(WG), 0, s, L.n_cols, L.slot_ptr.p, L.slots.p, ls.S_col.p);
        comm->allreduce(ls.S_col.p, 2 * (int64_t)L.n_cols);
        hipLaunchKernelGGL(k_mf_draw_S, dim3((L.n_cols + 255) / 256), dim3(256), 0, s, a, L.scols.p, L.n_cols, ls.S_col.p,
                           more ? an.theta : (const double *)nullptr, ls.dv_col.p);
      } else {
        hipLaunchKernelGGL(k_mf_draw, dim3((L.n_cols + 3) / 4), dim3(WG), 0, s, a, L.scols.p, L.n_cols, L.slot_ptr.p, L.slots.p,
                           more ? an.theta : (const double *)nullptr, ls.dv_col.p);
      }
    }
    pass(&a, more ? &an : nullptr, false, !more);
  }
  MFM_HIP_CHECK(hipGetLastError());
}

// update_e of a two-field table on the row tiles of its latent sweep (k_mf_score); false: not applicable (rank)
template <bool UNIT>
static bool launch_mf_score(hipStream_t s, const StepPlan &plan, const SweepArgs &a, const double *Vt, const double *w, double w0,
                            int K, int KS, const double *y, double2 *eq) {
  if (K < 1 || KS > 128) return false;
  const ParLevel &L = plan.steps.back().par;
  MfScoreArgs m;
  std::memset(&m, 0, sizeof(m));
  m.tile_row0 = L.tile_row0.p;
  m.tile_ptr = L.tile_ptr.p;
  m.tent = L.tent.p;
  m.tval = L.ent_val.p;
  m.tile_bits = L.tile_bits;
  m.n_tiles = L.n_tiles;
  m.swz = xcd_swizzle_enabled();
  m.scols = L.scols.p;
  m.udesc = plan.fuse_desc.p;
  m.ucol_ptr = plan.fuse_col_ptr.p;
  m.chunk = plan.mf_chunk.p;
  m.chunk_ptr = plan.mf_chunk_ptr.p;
  m.solo_col = plan.n_long_cols ? plan.solo_col.p : nullptr;
  m.colptr = a.colptr;
  m.cval = a.val;
  m.col_row0 = plan.col_row0.p;
  m.Vt = Vt;
  m.w = w;
  m.w0 = w0;
  m.K = K;
  m.KS = KS;
  m.y = y;
  m.eq = eq;
  m.dbg = std::getenv("MFM_SCORE_DBG") ? std::atoi(std::getenv("MFM_SCORE_DBG")) : 0;
  size_t lds = ((size_t)8 << L.tile_bits) + (size_t)mf_user_cap(L.tile_bits) * 12 + 16;
  {
    // stage the tile's first-level Vt rows in LDS when they fit next to the score array (three workgroups per CU)
    const size_t rows = (size_t)std::max(plan.mf_max_users, 1);
    const size_t need = rows * (size_t)KS * 8;
    const size_t cap = std::getenv("MFM_SCORE_UCACHE_KB") ? (size_t)std::atoi(std::getenv("MFM_SCORE_UCACHE_KB")) * 1024 : 52 * 1024;
    if (std::getenv("MFM_SETUP_TIMING")) std::fprintf(stderr, "[k_mf_score] max users per tile %d, LDS %zu + %zu\n", plan.mf_max_users, lds, need);
    if (lds + need <= cap && !std::getenv("MFM_NO_SCORE_UCACHE")) {
      m.ucache = (int)rows;
      lds += need;
    }
  }
  const int kp = KS / 2;  // factor pairs
  const int nt = 512;
#define MFM_MFS(G, S)                                                                                                     \
  do {                                                                                                                    \
    static DeviceOnce raised;                                                                                           \
    if (raised.need()) {                                                                                                        \
      MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_mf_score<G, S, UNIT>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                        (int)CHAIN_LDS_MAX));                                                             \
      raised.mark();                                                                                                      \
    }                                                                                                                     \
    hipLaunchKernelGGL((k_mf_score<G, S, UNIT>), dim3(L.n_tiles), dim3(nt), lds, s, m);                                     \
  } while (0)
  // lanes per entry x pairs per lane >= kp
  if (kp <= 1)
    MFM_MFS(1, 1);
  else if (kp <= 2)
    MFM_MFS(2, 1);
  else if (kp <= 4)
    MFM_MFS(2, 2);
  else if (kp <= 8)
    MFM_MFS(4, 2);
  else if (kp <= 16)
    MFM_MFS(4, 4);
  else if (kp <= 32)
    MFM_MFS(8, 4);
  else
    MFM_MFS(16, 4);
#undef MFM_MFS
  MFM_HIP_CHECK(hipGetLastError());
  return true;
}

// ---- row-sharded fused path -------------------------------------------------------------------------------
// One non-scattered level, row-sharded: statistics -> S -> all-reduce -> draw -> apply. PS: statistics / draw
// policy, PA: apply policy (its load runs after the new coefficient was stored). ccols != null: only these
// n_c columns take part and their statistics travel in a compact buffer.
template <class PS, class PA, bool UNIT>
static void run_level_sharded(hipStream_t s, Timing &tm, const ParLevel &L, const SweepArgs &a, LongScratch &ls,
                              const SweepClasses &kc, const Comm &comm, const int32_t *ccols, int n_c) {
  double2 *S = ls.S_col.p;
  const int grid = L.n_wg + (L.n_w16 + 3) / 4 + (L.n_w4 + 3) / 4 + (L.n_w1 + 3) / 4;
  if (grid) {
    TimedLaunch t(tm, s, kc.heavy, PS::STAT_BYTES * (L.nnz_heavy + L.nnz_light));
    hipLaunchKernelGGL((k_level_split<PS, UNIT, 1>), dim3(grid), dim3(WG), 0, s, a, L.cols_w1.p, L.n_w1, L.cols_w4.p, L.n_w4,
                       L.cols_w16.p, L.n_w16, L.cols_wg.p, L.n_wg, S, (const double2 *)nullptr);
  }
  if (L.n_huge) {
    TimedLaunch t(tm, s, kc.hstats, PS::STAT_BYTES * L.nnz_huge);
    hipLaunchKernelGGL((k_long_stats<PS>), dim3(L.n_hchunks), dim3(WG), 0, s, a, L.hchunks.p, L.cols_huge.p, ls.partial.p);
    hipLaunchKernelGGL(k_long_sum, dim3((L.n_huge + 63) / 64), dim3(64), 0, s, L.cols_huge.p, L.hchunk_ptr.p, L.n_huge,
                       ls.partial.p, S);
  }
  MFM_HIP_CHECK(hipGetLastError());
  if (ccols) {
    if (ls.S_compact.n < (size_t)n_c) ls.S_compact.alloc((size_t)n_c);
    hipLaunchKernelGGL(k_gather_S, dim3((n_c + 255) / 256), dim3(256), 0, s, ccols, n_c, S, ls.S_compact.p);
    comm.allreduce(ls.S_compact.p, 2 * (int64_t)n_c);
    hipLaunchKernelGGL(k_scatter_S, dim3((n_c + 255) / 256), dim3(256), 0, s, ccols, n_c, ls.S_compact.p, S);
  } else {
    comm.allreduce(S + L.jmin, 2 * (int64_t)(L.jmax - L.jmin + 1));
  }
  hipLaunchKernelGGL((k_col_draw<PS>), dim3((L.n_all + 255) / 256), dim3(256), 0, s, a, L.cols_all.p, L.n_all, S,
                     ls.oldnew_col.p);
  if (grid) {
    TimedLaunch t(tm, s, kc.light, PA::BYTES * (L.nnz_heavy + L.nnz_light));
    hipLaunchKernelGGL((k_level_split<PA, UNIT, 2>), dim3(grid), dim3(WG), 0, s, a, L.cols_w1.p, L.n_w1, L.cols_w4.p, L.n_w4,
                       L.cols_w16.p, L.n_w16, L.cols_wg.p, L.n_wg, (double2 *)nullptr, ls.oldnew_col.p);
  }
  if (L.n_huge) {
    TimedLaunch t(tm, s, kc.happly, PA::BYTES * L.nnz_huge);
    hipLaunchKernelGGL((k_long_apply_col<PA>), dim3(L.n_hchunks), dim3(WG), 0, s, a, L.hchunks.p, L.cols_huge.p,
                       ls.oldnew_col.p);
  }
  MFM_HIP_CHECK(hipGetLastError());
}

// can the row-sharded latent sweep of this (local) table run the fused tile path? A contiguous first level covering
// every local row once, every other level on row tiles aligned to it (two levels: run_sweep_soa_sharded, more:
// run_sweep_soa_multi with the communicator).
static inline bool plan_supports_sharded_fused(const StepPlan &plan) {
  if (!plan.sharded_tiles || plan.steps.size() < 2 || !plan.aligned_tiles) return false;
  const Step &f = plan.steps.front();
  if (f.is_chain || f.par.scattered || !f.par.first_and_once || !f.par.contig) return false;
  for (size_t i = 1; i < plan.steps.size(); i++)
    if (plan.steps[i].is_chain || !plan.steps[i].par.scattered || !plan.steps[i].par.tiled) return false;
  return true;
}

// Row-sharded latent sweep, split e / q layout, fused tile pass (args(f).state = e, .state2 = q). The
// first-level columns that are complete on this rank go through k_tile_apply_next without any communication;
// per factor the ranks exchange the last level's statistics (2 n_cols doubles) and those of the few special
// first-level columns; the first-level coefficients are made identical on every rank once, after the sweep
// (sync_model in mfm_sweep_V).
template <bool UNIT, class ArgsOf>
static void run_sweep_soa_sharded(hipStream_t s, Timing &tm, const StepPlan &plan, ArgsOf args, int f_begin, int f_end,
                                  LongScratch &ls, const SweepClasses &kc, const Comm &comm) {
  const ParLevel &L1 = plan.steps.front().par, &L = plan.steps.back().par;
  const int swz = xcd_swizzle_enabled();
  const size_t lds = sizeof(double2) << L.tile_bits;
  const int nt = tile_threads(L.tile_bits);
  const bool two = L.covers_rows_once && !std::getenv("MFM_NO_FUSED_TWO");
  raise_fused_lds_limit<UNIT>();
  {
    // first factor's first level: the locally complete columns in one pass, the special ones all-reduced
    const SweepArgs a = soa_first_args(args(f_begin));
    launch_binned_level<PMainVsq<UNIT>, UNIT>(s, tm, plan.local_level, a, ls, kc, plan.col_row0.p);
    launch_huge<PMainVsq<UNIT>, UNIT>(s, tm, plan.local_level, a, ls, kc);
    if (plan.n_special)
      run_level_sharded<PMainVsq<UNIT>, PMainVsqA<UNIT>, UNIT>(s, tm, plan.special_level, a, ls, kc, comm,
                                                               plan.special_cols.p, plan.n_special);
  }
  for (int f = f_begin; f < f_end; f++) {
    const SweepArgs a = args(f);
    const bool next = f + 1 < f_end;
    {
      TimedLaunch t(tm, s, kc.scat, 20.0 * L.n_ent);
      if (f == f_begin) {
        hipLaunchKernelGGL(k_tile_old, dim3((L.n_cols + 255) / 256), dim3(256), 0, s, a.theta, L.scols.p, L.n_cols,
                           ls.vnext_col.p);
        hipLaunchKernelGGL((k_tile_stats<PMainV, UNIT, true>), dim3(L.n_tiles), dim3(nt), lds, s, a, L.tent.p, L.ent_val.p,
                           L.tile_ptr.p, L.tile_row0.p, ls.vnext_col.p, L.run_base.p, L.slot_pos.p, L.slots.p, L.tile_bits,
                           L.n_tiles, swz, (const int32_t *)nullptr);
      }
      hipLaunchKernelGGL(k_tile_sum, dim3((L.n_cols + 3) / 4), dim3(WG), 0, s, L.n_cols, L.slot_ptr.p, L.slots.p, ls.S_col.p);
    }
    MFM_HIP_CHECK(hipGetLastError());
    comm.allreduce(ls.S_col.p, 2 * (int64_t)L.n_cols);
    SweepArgs an = next ? args(f + 1) : a;
    hipLaunchKernelGGL((k_tile_draw_S<PMainV>), dim3((L.n_cols + 255) / 256), dim3(256), 0, s, a, L.scols.p, L.n_cols,
                       ls.S_col.p, ls.oldnew_col.p, next ? (const double *)an.theta : (const double *)nullptr, ls.vnext_col.p);
    if (!next) {
      TimedLaunch t(tm, s, kc.scat, 28.0 * L.n_ent);
      hipLaunchKernelGGL((k_tile_apply<PMainV, UNIT, true, false>), dim3(L.n_tiles), dim3(nt), lds, s, soa_final_args(a, true),
                         L.tent.p, L.ent_val.p, L.tile_ptr.p, L.tile_row0.p, ls.oldnew_col.p, L.tile_bits, L.n_tiles, swz);
      continue;
    }
    an.row0 = plan.col_row0.p;
    {
      TimedLaunch t(tm, s, KC_SWEEP_V_FUSED, 32.0 * plan.n_state_rows + (UNIT ? 4.0 : 12.0) * L.n_ent + 16.0 * L.n_runs);
      SweepArgs af = a;
      af.row0 = plan.col_row0.p;
      FuseArgs fa{an.theta, an.z,         an.lambda,    an.mu,     plan.fuse_desc.p, plan.fuse_col_ptr.p, ls.vnext_col.p,
                  1,        L.run_base.p, L.slot_pos.p, L.slots.p, plan.solo_col.p,  plan.long_partial.p,
                  nullptr,  nullptr,      nullptr};
      if (two)
        hipLaunchKernelGGL((k_tile_apply_next<UNIT, true>), dim3(L.n_tiles), dim3(nt), lds + 256, s, af, L.tent.p, L.ent_val.p,
                           L.tile_ptr.p, L.tile_row0.p, ls.oldnew_col.p, L.tile_bits, L.n_tiles, swz, fa);
      else
        hipLaunchKernelGGL((k_tile_apply_next<UNIT, false>), dim3(L.n_tiles), dim3(nt), lds + 256, s, af, L.tent.p,
                           L.ent_val.p, L.tile_ptr.p, L.tile_row0.p, ls.oldnew_col.p, L.tile_bits, L.n_tiles, swz, fa);
    }
    MFM_HIP_CHECK(hipGetLastError());
    // first-level columns of factor f + 1 longer than a tile (complete here: second pass over their tiles), then
    // the special ones (rows on several ranks: all-reduced statistics)
    launch_long_finish<UNIT>(s, tm, plan, L, an, ls, kc, 1, lds, nt);
    if (plan.n_special)
      run_level_sharded<PMainVsq<UNIT>, PMainVsqA<UNIT>, UNIT>(s, tm, plan.special_level, an, ls, kc, comm,
                                                               plan.special_cols.p, plan.n_special);
    if (plan.n_solo_tiles) {
      TimedLaunch t(tm, s, kc.scat, 20.0 * plan.n_solo_tiles * (1 << L.tile_bits));
      hipLaunchKernelGGL((k_tile_stats<PMainV, UNIT, true>), dim3(plan.n_solo_tiles), dim3(nt), lds, s, an, L.tent.p,
                         L.ent_val.p, L.tile_ptr.p, L.tile_row0.p, ls.vnext_col.p, L.run_base.p, L.slot_pos.p, L.slots.p,
                         L.tile_bits, L.n_tiles, swz, plan.solo_tiles.p);
    }
  }
  MFM_HIP_CHECK(hipGetLastError());
}

// latent sweep of the main table with the q-free policy (compact residual array in a.state)
static void run_plan_qfree(hipStream_t s, Timing &tm, const StepPlan &plan, const SweepArgs &a, LongScratch &ls,
                           const SweepClasses &kc, bool unit) {
  if (unit)
    run_plan_t<PMainVe<true, false>, true, PMainVe<true, true>>(s, tm, plan, a, ls, kc);
  else
    run_plan_t<PMainVe<false, false>, false, PMainVe<false, true>>(s, tm, plan, a, ls, kc);
}

// resident-workgroup capacity for the co-resident long-column kernel, with a safety margin
template <class P>
static int coop_capacity() {
  int dev = 0, cus = 0, per_cu_a = 0, per_cu_b = 0;
  MFM_HIP_CHECK(hipGetDevice(&dev));
  MFM_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  MFM_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_a, k_long_coop<P, false>, WG, 0));
  MFM_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_b, k_long_coop<P, true>, WG, 0));
  int per_cu = std::min(per_cu_a, per_cu_b);
  // The occupancy API over-reports by one block per CU only where the SGPR budget binds (7 vs 8 blocks,
  // MI355X_MICROARCH.md "Residency"); this kernel is VGPR-bound at 2-4 blocks. Other streams' kernels can
  // delay residency but never wait on anything, and the spin is bounded.
  if (per_cu >= 7) per_cu -= 1;
  per_cu = std::max(1, std::min(per_cu, 4));
  if (const char *e = std::getenv("MFM_COOP_PER_CU")) per_cu = std::max(1, std::atoi(e));
  return std::max(8, cus * per_cu);
}

}  // namespace mfm
