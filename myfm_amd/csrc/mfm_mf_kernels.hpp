// mfm_mf_kernels.hpp -- the latent sweep (update_V, FMTrainer.hpp:316-376) of a TWO-FIELD table: a main table whose
// columns form exactly two conflict-free levels, each covering every row once (user x item designs: configs[1] and
// configs[2] of BASELINE.json), sorted by the first field. One pass over HBM per factor, and no q-cache in HBM at all.
//
// With one entry of each level per row, q_t = a_t v_u(t) + b_t v_i(t) (FMTrainer.hpp:320; a, b the stored values), so
//   first level  (users):  h_t = a_t (q_t - a_t v_u) = a_t b_t v_i(t)          (:351-356)
//   second level (items):  h_t = b_t (q_t - b_t v_i) = b_t a_t v_u(t)  with the user's NEW coefficient
// and every statistic / update of :351-374 needs only e_t and the OTHER field's coefficient at the row. Per factor the
// residual crosses HBM once (read + write), together with the 4-byte entry stream of the second level (sorted by
// (row tile, item, row), mfm_plan.hpp build_tiled) and one 16-byte statistics slot per run of an item inside a tile.
//
// One workgroup owns a tile of <= 2^tile_bits rows aligned to the first level's columns (StepPlan::build_aligned_tiles).
// Pass "f -> f + 1" does, on the tile:
//   P0  loads e (row order, coalesced, K rows per thread), the tile's entries (item order) and per entry the pair
//       dv[item] = (v_i' - v_i of factor f, v_i of factor f + 1) the item draw left behind; lane u < n_users loads user
//       u's scalars (coefficients, variate, prior);
//   P1  scatters the pairs to their rows in LDS;
//   P2  row order: e1 = e + (a v_u^f)(b delta_i)  -- the item level's update of factor f (:371-375) --, the user
//       level's terms of factor f + 1, a wave-level segmented scan over the users' contiguous row ranges (DPP, no LDS),
//       one partial per (user, 64-row chunk) to LDS;
//   P3  thread u sums user u's partials in row order and draws v_u' (:357-369) -- every user of the tile in parallel;
//   P4  e2 = e1 + h (v_u' - v_u) (:373-374), stored to HBM from registers; (e2, a v_u') per row to LDS;
//   P5  item order: the item level's statistics of factor f + 1 from LDS, segmented scan over runs, one slot per run.
// All sums have a fixed association (scan tree, then row order): results are bit-reproducible.
#pragma once
#include "mfm_kernels.hpp"

namespace mfm {

// Workgroup barrier that orders LDS traffic only (__syncthreads() is a workgroup-scope release: it also waits for the
// wave's outstanding global stores). Global data written before such a barrier is never read by another thread of the launch.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct MfChunk {            // one per 64 consecutive rows of a tile
  unsigned long long heads; // bit l: row 64 c + l is the first row of a first-level column
  int32_t ubase;            // index (within the tile) of the column owning the chunk's first row
  int32_t pbase;            // index (within the tile) of the chunk's first partial
};

__host__ __device__ static inline int mf_user_cap(int tile_bits) {
  const int r = (1 << tile_bits) / 16;
  return r < 64 ? 64 : r;
}
// dynamic LDS of k_mf_pass / k_mf_long_finish
static inline size_t mf_lds_bytes(int tile_bits, bool unit) {
  const size_t R = (size_t)1 << tile_bits, ucap = (size_t)mf_user_cap(tile_bits);
  return 16 * R + 16 * ((R >> 6) + ucap + 2) + 16 * ucap + (unit ? 0 : 8 * ucap);
}

struct MfArgs {
  // residual: read at e_in[row * e_in_stride], written at e_out[row * e_out_stride]; E = the split array e[N]
  const double *e_in;
  double *e_out;
  double *E;
  int e_in_stride, e_out_stride;
  // row tiles and the second level's entries
  const int32_t *tile_row0, *tile_ptr;
  const uint32_t *tent;
  const double *tval;  // !UNIT
  int tile_bits, n_tiles, swz, ucap;
  const double2 *dv;   // per second-level column: (delta of factor f, coefficient of factor f + 1)
  // first-level columns per tile (row order, then the tile's share of never-occurring columns)
  const int4 *udesc;   // {column, length, first row - tile start, group}
  const int32_t *ucol_ptr;
  const int2 *upart;   // {first partial, number of partials}
  const MfChunk *chunk;
  const int32_t *chunk_ptr;
  const double *theta_cur;  // V[:, f]
  double *theta_next;       // V[:, f + 1]
  const double *z_next, *lam_next, *mu_next;
  double alpha;
  int do_apply, do_next;
  int dbg;  // timing experiments only (results are wrong when set): 1 skip the row-order scan, 2 skip P5, 4 skip P3-P4, 8 no slot stores, 16 no P5 scan, 32 slots tile-major
  // !UNIT: stored values of the first level (CSC, contiguous rows per column)
  const int64_t *colptr;
  const double *cval;
  const int32_t *col_row0;
  // statistics of the second level
  const int32_t *run_base, *slot_pos;
  double2 *slots;
  // tiles inside a first-level column longer than a tile
  const int32_t *solo_col;
  double2 *long_partial;
};

// run structure of this wave's wave tiles of entries: slot address per run tail, head / store flags
template <int K>
__device__ __forceinline__ void mf_run_structure(const uint32_t (&u)[K], const int (&rb)[K], int t0, int t1, int nw, int lane,
                                                 int tile_bits, const int32_t *__restrict__ slot_pos, int (&pos)[K],
                                                 unsigned (&flags)[K]) {
#pragma unroll
  for (int k = 0; k < K; k++) {
    pos[k] = 0;
    flags[k] = 0;
    if (t0 + k * nw >= t1) continue;  // wave-uniform
    const bool valid = u[k] != TILE_PAD;
    const int c = valid ? (int)(u[k] >> tile_bits) : -1 - lane;
    const int cp = dpp_i32<0x138, 0xf>(c, 0), cn = dpp_i32<0x130, 0xf>(c, 0);  // wave_shr:1 / wave_shl:1
    const bool head = lane == 0 || cp != c;
    const bool tail = lane == 63 || cn != c;
    const unsigned long long hb = __ballot(head);
    if (valid && tail) {
      const int run = rb[k] + __popcll(hb & ((2ull << lane) - 1ull)) - 1;
      pos[k] = slot_pos[run];
    }
    flags[k] = (head ? 1u : 0u) | (valid && tail ? 2u : 0u);
  }
}

template <bool UNIT, int K>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(1, 4))) void k_mf_pass(MfArgs a) {
  extern __shared__ double2 mf_lds[];
  const int R = 1 << a.tile_bits;
  d2_t *rowbuf = (d2_t *)mf_lds;                             // [R]
  d2_t *part = rowbuf + R;                                   // [R / 64 + ucap + 2]
  d2_t *uval = part + (R >> 6) + a.ucap + 2;                 // [ucap]
  long long *uvb = (long long *)(uval + a.ucap);             // [ucap]  (!UNIT)
  const int b = xcd_swizzle(blockIdx.x, a.n_tiles, a.swz);
  const int row0 = a.tile_row0[b];
  const int nr = a.tile_row0[b + 1] - row0;
  const int nt = blockDim.x, tid = threadIdx.x, lane = tid & 63, nw = nt >> 6;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t rmask = (uint32_t)R - 1u;
  const int do_apply = a.do_apply, do_next = a.do_next;

  // ---- P0: every global load of the tile
  double e[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int r = k * nt + tid;
    e[k] = r < nr ? __builtin_nontemporal_load(a.e_in + (int64_t)(row0 + r) * a.e_in_stride) : 0.0;
  }
  const int tp0 = a.tile_ptr[b], tp1 = a.tile_ptr[b + 1];
  const int64_t p0 = (int64_t)tp0 * WAVE + tid, p1 = (int64_t)tp1 * WAVE;
  uint32_t u[K];
  double xb[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int64_t p = p0 + (int64_t)k * nt;
    u[k] = TILE_PAD;
    xb[k] = 1.0;
    if (p < p1) {
      u[k] = __builtin_nontemporal_load(&a.tent[p]);
      if (!UNIT) xb[k] = __builtin_nontemporal_load(&a.tval[p]);
    }
  }
  const int ts0 = tp0 + wv;
  int rb[K];
#pragma unroll
  for (int k = 0; k < K; k++) rb[k] = do_next && ts0 + k * nw < tp1 ? a.run_base[ts0 + k * nw] : 0;
  d2_t dv[K];
#pragma unroll
  for (int k = 0; k < K; k++) dv[k] = u[k] != TILE_PAD ? ((const d2_t *)a.dv)[u[k] >> a.tile_bits] : d2_t{0.0, 0.0};
  const int solo_j = a.solo_col ? a.solo_col[b] : -1;
  const int c0 = a.ucol_ptr[b], nu = a.ucol_ptr[b + 1] - c0;
  // the row chunks' descriptors (wave-uniform) and the slot of every statistics run: every global load of the pass is
  // issued before the first barrier
  const int cbase = a.chunk_ptr[b];
  MfChunk CH[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    CH[k] = MfChunk{0ull, 0, 0};
    if ((k * nw + wv) * WAVE < nr) CH[k] = a.chunk[cbase + k * nw + wv];
  }
  int pos[K];
  unsigned flags[K];
  if (do_next) mf_run_structure<K>(u, rb, ts0, tp1, nw, lane, a.tile_bits, a.slot_pos, pos, flags);
  // this thread's first-level column (thread u <-> column u of the tile)
  int uj = 0, ulen = 0, upf = 0, upc = 0;
  double uold = 0.0, uz = 0.0, ulam = 0.0, umu = 0.0;
  if (solo_j < 0 && tid < nu) {
    const int4 d = a.udesc[c0 + tid];
    const int2 pp0 = a.upart[c0 + tid];
    upf = pp0.x;
    upc = pp0.y;
    uj = d.x;
    ulen = d.y;
    const double vcur = do_apply ? a.theta_cur[uj] : 0.0;
    if (do_next) {
      uold = a.theta_next[uj];
      uz = a.z_next[uj];
      ulam = a.lam_next[d.w];
      umu = a.mu_next[d.w];
    }
    if (ulen > 0) {  // (columns with rows come first and number <= ucap)
      uval[tid] = d2_t{vcur, uold};
      if (!UNIT) uvb[tid] = (long long)a.colptr[uj] - d.z;
    }
  }
  if (a.dbg & 64) {  // (timing experiment: the loads only)
#pragma unroll
    for (int k = 0; k < K; k++) asm volatile("" ::"v"(e[k]), "v"(dv[k][0]), "v"(dv[k][1]), "v"(rb[k]));
    asm volatile("" ::"v"(uold), "v"(uz), "v"(ulam), "v"(umu));
    return;
  }
  // ---- P1: (b delta_i, b v_i^{f+1}) to the entry's row
#pragma unroll
  for (int k = 0; k < K; k++)
    if (u[k] != TILE_PAD) rowbuf[u[k] & rmask] = UNIT ? dv[k] : d2_t{xb[k] * dv[k][0], xb[k] * dv[k][1]};
  __syncthreads();
  if (a.dbg & 128) {
#pragma unroll
    for (int k = 0; k < K; k++) asm volatile("" ::"v"(e[k]));
    return;
  }

  if (solo_j >= 0) {
    // A tile inside a first-level column longer than a tile: apply the second level's update, leave the tile's partial
    // statistics of the column (k_long_tile_draw sums them in tile order, k_mf_long_finish does P4 / P5).
    const double vcur = do_apply ? a.theta_cur[solo_j] : 0.0;
    const int64_t vbase = UNIT ? 0 : a.colptr[solo_j] + (row0 - a.col_row0[solo_j]);
    double S1 = 0.0, S2 = 0.0;
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int r = k * nt + tid;
      if (r < nr) {
        const d2_t dd = rowbuf[r];
        const double ar = UNIT ? 1.0 : a.cval[vbase + r];
        const double e1 = do_apply ? e[k] + (ar * vcur) * dd[0] : e[k];
        if (do_next) {
          const double h = ar * dd[1];
          S2 += h * h;
          S1 += (-e1) * h;
          __builtin_nontemporal_store(e1, &a.E[row0 + r]);
        } else {
          __builtin_nontemporal_store(e1, a.e_out + (int64_t)(row0 + r) * a.e_out_stride);
        }
      }
    }
    if (!do_next) return;
    S1 = wave_allreduce_sum(S1);
    S2 = wave_allreduce_sum(S2);
    if (lane == 0) part[wv] = d2_t{S1, S2};
    __syncthreads();
    if (tid == 0) {
      double T1 = 0.0, T2 = 0.0;
      for (int w = 0; w < nw; w++) {  // wave order: deterministic
        T1 += part[w][0];
        T2 += part[w][1];
      }
      a.long_partial[b] = make_double2(T1, T2);
    }
    return;
  }

  // ---- P2: row order
  int ul[K];
  double h[K], ar[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int ch = k * nw + wv, r = k * nt + tid;
    ul[k] = 0;
    h[k] = 0.0;
    ar[k] = 1.0;
    if (ch * WAVE >= nr) continue;  // wave-uniform
    const MfChunk C = CH[k];
    const int cnt = __popcll(C.heads & ((2ull << lane) - 2ull));  // column heads in rows (chunk start, this row]
    ul[k] = C.ubase + cnt;
    const bool valid = r < nr;
    const d2_t dd = valid ? rowbuf[r] : d2_t{0.0, 0.0};
    const d2_t uv = uval[ul[k]];
    if (!UNIT) ar[k] = valid ? a.cval[uvb[ul[k]] + r] : 0.0;
    const double e1 = do_apply ? e[k] + (ar[k] * uv[0]) * dd[0] : e[k];
    e[k] = e1;
    h[k] = ar[k] * dd[1];
    if (do_next) {
      double s1 = valid ? (-e1) * h[k] : 0.0, s2 = valid ? h[k] * h[k] : 0.0;
      int f = (lane == 0) | (int)((C.heads >> lane) & 1ull);
      if (!(a.dbg & 1)) wave_segscan2(s1, s2, f);
      const bool tail = lane == 63 || (((C.heads >> 1) >> lane) & 1ull);
      if (tail) part[C.pbase + cnt] = d2_t{s1, s2};
    } else if (valid) {
      __builtin_nontemporal_store(e1, a.e_out + (int64_t)(row0 + r) * a.e_out_stride);
    }
  }
  if (!do_next) return;
  lds_barrier();

  if (a.dbg & 4) return;
  // ---- P3: one thread per first-level column: partials in row order, draw
  for (int uu = tid; uu < nu; uu += nt) {
    if (uu != tid) {  // (more columns than threads: a tile that hosts many never-occurring columns)
      const int4 d = a.udesc[c0 + uu];
      const int2 pp = a.upart[c0 + uu];
      upf = pp.x;
      upc = pp.y;
      uj = d.x;
      ulen = d.y;
      uold = a.theta_next[uj];
      uz = a.z_next[uj];
      ulam = a.lam_next[d.w];
      umu = a.mu_next[d.w];
    }
    double S1 = 0.0, S2 = 0.0;
    for (int p = upf; p < upf + upc; p++) {
      const d2_t s = part[p];
      S1 += s[0];
      S2 += s[1];
    }
    const double fresh = PMainV::draw(S1, S2, uold, a.alpha, ulam, umu, uz);
    a.theta_next[uj] = fresh;
    if (ulen > 0) uval[uu] = d2_t{fresh, fresh - uold};
  }
  lds_barrier();

  // ---- P4: the user level's update; the row's state for the item level's statistics
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int ch = k * nw + wv, r = k * nt + tid;
    if (ch * WAVE >= nr) continue;
    const d2_t uv = uval[ul[k]];
    const double e2 = e[k] + h[k] * uv[1];
    if (r < nr) {
      __builtin_nontemporal_store(e2, a.e_out + (int64_t)(row0 + r) * a.e_out_stride);
      rowbuf[r] = d2_t{e2, UNIT ? uv[0] : ar[k] * uv[0]};
    }
  }
  lds_barrier();

  if (a.dbg & 2) return;
  // ---- P5: item order
#pragma unroll
  for (int k = 0; k < K; k++) {
    if (ts0 + k * nw >= tp1) continue;
    double s1 = 0.0, s2 = 0.0;
    if (u[k] != TILE_PAD) {
      const d2_t rr = rowbuf[u[k] & rmask];
      const double hh = UNIT ? rr[1] : xb[k] * rr[1];
      s2 = hh * hh;
      s1 = (-rr[0]) * hh;
    }
    int f = (int)(flags[k] & 1u);
    if (!(a.dbg & 16)) wave_segscan2(s1, s2, f);
    if (a.dbg & 8) {
      asm volatile("" ::"v"(s1), "v"(s2), "v"(pos[k]));
    } else if (a.dbg & 32) {  // the same stores, tile-major (contiguous per tile) instead of column-major
      if (flags[k] & 2u) a.slots[rb[k] + __popcll(__ballot(flags[k] & 1u) & ((2ull << lane) - 1ull)) - 1] = make_double2(s1, s2);
    } else if (flags[k] & 2u) {
      // (dbg 256: timing experiment -- all slot stores folded into a 4 MiB window, i.e. cache-resident targets)
      a.slots[(a.dbg & 256) ? (pos[k] & 0x3ffff) : pos[k]] = make_double2(s1, s2);
    }
  }
}

// Second pass over the tiles of the first-level columns longer than a tile, after k_long_tile_draw drew them: the
// column's update (P4) and the item level's statistics (P5) of factor f + 1; e1 was left in E by k_mf_pass.
template <bool UNIT, int K>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(1, 4))) void k_mf_long_finish(
    MfArgs a, const int32_t *__restrict__ long_tiles, const int32_t *__restrict__ tile_long_idx,
    const double2 *__restrict__ oldnew_long, const int32_t *__restrict__ long_cols) {
  extern __shared__ double2 mf_lds[];
  d2_t *rowbuf = (d2_t *)mf_lds;
  const int b = long_tiles[blockIdx.x];
  const int row0 = a.tile_row0[b];
  const int nr = a.tile_row0[b + 1] - row0;
  const int nt = blockDim.x, tid = threadIdx.x, lane = tid & 63, nw = nt >> 6;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t rmask = (1u << a.tile_bits) - 1u;
  const int l = tile_long_idx[b];
  const int j = long_cols[l];
  const d2_t on = ((const d2_t *)oldnew_long)[l];
  const int64_t vbase = UNIT ? 0 : a.colptr[j] + (row0 - a.col_row0[j]);
  double e[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int r = k * nt + tid;
    e[k] = r < nr ? __builtin_nontemporal_load(&a.E[row0 + r]) : 0.0;
  }
  const int tp0 = a.tile_ptr[b], tp1 = a.tile_ptr[b + 1];
  const int64_t p0 = (int64_t)tp0 * WAVE + tid, p1 = (int64_t)tp1 * WAVE;
  uint32_t u[K];
  double xb[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int64_t p = p0 + (int64_t)k * nt;
    u[k] = TILE_PAD;
    xb[k] = 1.0;
    if (p < p1) {
      u[k] = __builtin_nontemporal_load(&a.tent[p]);
      if (!UNIT) xb[k] = __builtin_nontemporal_load(&a.tval[p]);
    }
  }
  const int ts0 = tp0 + wv;
  int rb[K];
#pragma unroll
  for (int k = 0; k < K; k++) rb[k] = ts0 + k * nw < tp1 ? a.run_base[ts0 + k * nw] : 0;
#pragma unroll
  for (int k = 0; k < K; k++)
    if (u[k] != TILE_PAD) rowbuf[u[k] & rmask] = d2_t{0.0, xb[k] * a.dv[u[k] >> a.tile_bits].y};
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int r = k * nt + tid;
    if (r < nr) {
      const double ar = UNIT ? 1.0 : a.cval[vbase + r];
      const double hk = ar * rowbuf[r][1];
      const double e2 = e[k] + hk * (on[1] - on[0]);
      __builtin_nontemporal_store(e2, a.e_out + (int64_t)(row0 + r) * a.e_out_stride);
      rowbuf[r] = d2_t{e2, ar * on[1]};
    }
  }
  __syncthreads();
  int pos[K];
  unsigned flags[K];
  mf_run_structure<K>(u, rb, ts0, tp1, nw, lane, a.tile_bits, a.slot_pos, pos, flags);
#pragma unroll
  for (int k = 0; k < K; k++) {
    if (ts0 + k * nw >= tp1) continue;
    double s1 = 0.0, s2 = 0.0;
    if (u[k] != TILE_PAD) {
      const d2_t rr = rowbuf[u[k] & rmask];
      const double hh = UNIT ? rr[1] : xb[k] * rr[1];
      s2 = hh * hh;
      s1 = (-rr[0]) * hh;
    }
    int f = (int)(flags[k] & 1u);
    wave_segscan2(s1, s2, f);
    if (flags[k] & 2u) a.slots[pos[k]] = make_double2(s1, s2);
  }
}

// item level: a wavefront per column sums the column's slots (contiguous, fixed order), draws (FMTrainer.hpp:357-369) and
// leaves dv[c] = (v' - v, the column's coefficient of the NEXT factor) for the pass that applies the update.
__global__ __launch_bounds__(WG) void k_mf_draw(SweepArgs a, const int32_t *__restrict__ cols, int n_cols,
                                                const int32_t *__restrict__ slot_ptr, const double2 *__restrict__ slots,
                                                const double *__restrict__ theta_next, double2 *__restrict__ dv) {
  const int c = blockIdx.x * (WG / WAVE) + (threadIdx.x >> 6);
  if (c >= n_cols) return;
  const int lane = threadIdx.x & 63;
  const int j = cols[c];
  double vn = 0.0, old = 0.0, zj = 0.0;
  int g = 0;
  if (lane == 0) {
    vn = theta_next ? theta_next[j] : 0.0;
    old = a.theta[j];
    zj = a.z[j];
    g = a.group[j];
  }
  double S1 = 0.0, S2 = 0.0;
  {
    // a popular column has one slot per tile (thousands): 16 loads in flight per lane, so that its wavefront is not a chain
    // of dependent round trips (the kernel's duration is that of the longest column); fixed association
    const int k1 = slot_ptr[c + 1];
    int k = slot_ptr[c] + lane;
    for (; k + 15 * WAVE < k1; k += 16 * WAVE) {
      double2 t[16];
#pragma unroll
      for (int i = 0; i < 16; i++) t[i] = slots[k + i * WAVE];
#pragma unroll
      for (int i = 0; i < 16; i += 4) {
        S1 += (t[i].x + t[i + 1].x) + (t[i + 2].x + t[i + 3].x);
        S2 += (t[i].y + t[i + 1].y) + (t[i + 2].y + t[i + 3].y);
      }
    }
    for (; k + 3 * WAVE < k1; k += 4 * WAVE) {
      const double2 s0 = slots[k], s1 = slots[k + WAVE], s2 = slots[k + 2 * WAVE], s3 = slots[k + 3 * WAVE];
      S1 += (s0.x + s1.x) + (s2.x + s3.x);
      S2 += (s0.y + s1.y) + (s2.y + s3.y);
    }
    for (; k < k1; k += WAVE) {
      const double2 s = slots[k];
      S1 += s.x;
      S2 += s.y;
    }
  }
  S1 = wave_allreduce_sum(S1);
  S2 = wave_allreduce_sum(S2);
  if (lane == 0) {
    const double fresh = PMainV::draw(S1, S2, old, a.alpha, a.lambda[g], a.mu[g], zj);
    a.theta[j] = fresh;
    dv[c] = make_double2(fresh - old, vn);
  }
}

// update_e (FMTrainer.hpp:493-497 -> FM.hpp:54-136) of a two-field table, on the same row tiles: with one entry of each
// level per row the FM score is  w0 + a w_u + b w_i + a b <V_u, V_i>  (1/2 [(a v_u + b v_i)^2 - a^2 v_u^2 - b^2 v_i^2]
// summed over the factors). The entries are walked in ITEM order: GS lanes per entry (a lane owns a factor pair, 16-byte
// gathers from the row-major table Vt), adjacent lane groups take adjacent entries, so an item's row is fetched once per
// run of that item inside a tile instead of once per training row, and the tile's few users stay in L1. Scores go through
// LDS to be written back in row order (coalesced), minus y.
struct MfScoreArgs {
  const int32_t *tile_row0, *tile_ptr;
  const uint32_t *tent;
  const double *tval;
  int tile_bits, n_tiles, swz;
  const int32_t *scols;      // second-level column (position in the level) -> feature
  const int4 *udesc;
  const int32_t *ucol_ptr;
  const MfChunk *chunk;
  const int32_t *chunk_ptr;
  const int32_t *solo_col;
  const int64_t *colptr;
  const double *cval;
  const int32_t *col_row0;
  const double *Vt, *w;
  double w0;
  int K, KS;
  const double *y;  // may be null
  double2 *eq;
  int ucache;  // > 0: the Vt rows of the tile's first-level columns (at most this many) are staged in LDS
  int dbg;     // timing experiments (wrong results): 1 no item-row gather, 2 no user-row gather, 4 no final store
};

// sum over the GS lanes of a lane group (GS a power of two <= 64, groups aligned to GS), valid in the group's LAST lane:
// DPP row shifts (and row broadcasts above 16 lanes) in the VALU -- __shfl_xor would be ds_bpermute traffic on the LDS
// crossbar, 2 GS-step instructions per double and entry, which was this kernel's bottleneck
template <int GS>
__device__ __forceinline__ double group_sum_last(double v) {
  if (GS >= 2) v += dpp_f64<0x111, 0xf>(v);   // row_shr:1
  if (GS >= 4) v += dpp_f64<0x112, 0xf>(v);   // row_shr:2
  if (GS >= 8) v += dpp_f64<0x114, 0xf>(v);   // row_shr:4
  if (GS >= 16) v += dpp_f64<0x118, 0xf>(v);  // row_shr:8
  if (GS >= 32) v += dpp_f64<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  if (GS >= 64) v += dpp_f64<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}

template <int GS, int SPL, bool UNIT>
__global__ __launch_bounds__(512) void k_mf_score(MfScoreArgs a) {
  extern __shared__ double2 mf_lds[];
  const int R = 1 << a.tile_bits;
  double *sc = (double *)mf_lds;                  // [R]
  int *ucol = (int *)(sc + R);                    // [ucap] feature of the tile's u-th first-level column
  long long *uvb = (long long *)(ucol + mf_user_cap(a.tile_bits));  // [ucap] (!UNIT) value offset
  const int b = xcd_swizzle(blockIdx.x, a.n_tiles, a.swz);
  const int row0 = a.tile_row0[b];
  const int nr = a.tile_row0[b + 1] - row0;
  const int nt = blockDim.x, tid = threadIdx.x;
  const int lig = tid % GS, grp = tid / GS, ngrp = nt / GS;
  const int solo_j = a.solo_col ? a.solo_col[b] : -1;
  const int c0 = a.ucol_ptr[b], nu = a.ucol_ptr[b + 1] - c0;
  const int cbase = a.chunk_ptr[b];
  const uint32_t rmask = (uint32_t)R - 1u;
  const int ucap = mf_user_cap(a.tile_bits);
  for (int uu = tid; uu < min(nu, ucap); uu += nt) {  // (columns with rows come first and number <= ucap; the never-occurring
    const int4 d = a.udesc[c0 + uu];                  //  ones behind them get a valid feature too: nobody reads their row)
    ucol[uu] = d.x;
    if (!UNIT) uvb[uu] = (long long)a.colptr[d.x] - d.z;
  }
  __syncthreads();
  // The tile's few first-level columns are needed by every entry: their Vt rows go to LDS once (the stream of item rows
  // would keep evicting them from the 32 KiB L1, and every miss is another 256-byte L2 read)
  double2 *uV = (double2 *)(uvb + mf_user_cap(a.tile_bits));  // [ucache][KS / 2]
  const int KP = a.KS >> 1;
  const bool cached = a.ucache > 0;
  if (cached) {
    const int nrows = solo_j >= 0 ? 1 : min(nu, a.ucache);
    for (int i = tid; i < nrows * KP; i += nt) {
      const int ul = i / KP, pr = i - ul * KP;
      const int j = solo_j >= 0 ? solo_j : ucol[ul];  // (columns with rows come first: ul < number of them <= ucache)
      uV[i] = ((const double2 *)(a.Vt + (int64_t)j * a.KS))[pr];
    }
    __syncthreads();
  }
  const int64_t p0 = (int64_t)a.tile_ptr[b] * WAVE, p1 = (int64_t)a.tile_ptr[b + 1] * WAVE;
  // lane `lig` of an entry's GS lanes owns the factor pairs lig + s GS, s < SPL (pairs beyond KS / 2 do not exist; the pad
  // column of an odd K is zero in Vt). Few lanes per entry: the per-entry index work is not replicated 16 times.
  constexpr int RU = SPL >= 4 ? 2 : 4;
  // Software pipeline over the group's entries: while the Vt rows of batch i are gathered and reduced, the entries of
  // batch i + 1 and their features / chunk descriptors (two dependent round trips) are already in flight.
  uint32_t u[RU], un[RU];
  int ju[RU], ji[RU], jun[RU], jin[RU], lu[RU], lun[RU];
  double xa[RU], xb[RU], xan[RU], xbn[RU];
  auto load_entries = [&](int64_t pb, uint32_t (&uu)[RU], double (&xxb)[RU]) {
#pragma unroll
    for (int q = 0; q < RU; q++) {
      const int64_t p = pb + (int64_t)q * ngrp;
      uu[q] = p < p1 ? __builtin_nontemporal_load(&a.tent[p]) : TILE_PAD;  // (streams: keep the L2 for the Vt rows)
      xxb[q] = !UNIT && p < p1 ? __builtin_nontemporal_load(&a.tval[p]) : 1.0;
    }
  };
  auto resolve = [&](const uint32_t (&uu)[RU], int (&jju)[RU], int (&jji)[RU], double (&xxa)[RU], int (&uul)[RU]) {
#pragma unroll
    for (int q = 0; q < RU; q++) {
      jju[q] = 0;
      jji[q] = 0;
      uul[q] = 0;
      xxa[q] = 1.0;
      if (uu[q] == TILE_PAD) continue;
      const int r = (int)(uu[q] & rmask);
      jji[q] = a.scols[uu[q] >> a.tile_bits];
      if (solo_j >= 0) {
        jju[q] = solo_j;
        if (!UNIT) xxa[q] = a.cval[a.colptr[solo_j] + (row0 - a.col_row0[solo_j]) + r];
      } else {
        const MfChunk C = a.chunk[cbase + (r >> 6)];
        const int ul = C.ubase + __popcll(C.heads & ((2ull << (r & 63)) - 2ull));
        uul[q] = ul;
        jju[q] = ucol[ul];
        if (!UNIT) xxa[q] = a.cval[uvb[ul] + r];
      }
    }
  };
  const int64_t stride = (int64_t)ngrp * RU;
  int64_t pb = p0 + grp;
  load_entries(pb, u, xb);
  resolve(u, ju, ji, xa, lu);
  load_entries(pb + stride, un, xbn);
  for (; pb < p1; pb += stride) {
    double2 vi[RU][SPL];
    double dot[RU];
#pragma unroll
    for (int q = 0; q < RU; q++) {
#pragma unroll
      for (int sp = 0; sp < SPL; sp++) {
        vi[q][sp] = make_double2(0.0, 0.0);
        if (u[q] != TILE_PAD && lig + sp * GS < KP && !(a.dbg & 1))
          vi[q][sp] = ((const double2 *)(a.Vt + (int64_t)ji[q] * a.KS))[lig + sp * GS];
      }
    }
#pragma unroll
    for (int q = 0; q < RU; q++) {
      dot[q] = 0.0;
#pragma unroll
      for (int sp = 0; sp < SPL; sp++) {
        if (u[q] != TILE_PAD && lig + sp * GS < KP && !(a.dbg & 2)) {
          const double2 vu = cached ? uV[lu[q] * KP + lig + sp * GS] : ((const double2 *)(a.Vt + (int64_t)ju[q] * a.KS))[lig + sp * GS];
          dot[q] += vu.x * vi[q][sp].x + vu.y * vi[q][sp].y;
        }
      }
    }
    // next batch: features / descriptors of the entries loaded one iteration ago, entries of the batch after it
    resolve(un, jun, jin, xan, lun);
    uint32_t u2[RU];
    double xb2[RU];
    load_entries(pb + 2 * stride, u2, xb2);
#pragma unroll
    for (int q = 0; q < RU; q++) {
      const double part = group_sum_last<GS>(dot[q]);
      if (lig == GS - 1 && u[q] != TILE_PAD)
        sc[u[q] & rmask] = a.w0 + (xa[q] * a.w[ju[q]] + xb[q] * a.w[ji[q]]) + (xa[q] * xb[q]) * part;
    }
#pragma unroll
    for (int q = 0; q < RU; q++) {
      u[q] = un[q];
      lu[q] = lun[q];
      ju[q] = jun[q];
      ji[q] = jin[q];
      xa[q] = xan[q];
      xb[q] = xbn[q];
      un[q] = u2[q];
      xbn[q] = xb2[q];
    }
  }
  __syncthreads();
  if (a.dbg & 4) return;
  for (int i = tid; i < nr; i += nt) {
    const double yv = a.y ? __builtin_nontemporal_load(&a.y[row0 + i]) : 0.0;
    a.eq[row0 + i].x = sc[i] - yv;
  }
}

// row-sharded mode: the item draw from the all-reduced per-column statistics S (k_tile_sum), identical on every rank
__global__ void k_mf_draw_S(SweepArgs a, const int32_t *__restrict__ cols, int n_cols, const double2 *__restrict__ S,
                            const double *__restrict__ theta_next, double2 *__restrict__ dv) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cols) return;
  const int j = cols[c];
  const double vn = theta_next ? theta_next[j] : 0.0;
  const double2 s = S[c];
  const double old = a.theta[j];
  const int g = a.group[j];
  const double fresh = PMainV::draw(s.x, s.y, old, a.alpha, a.lambda[g], a.mu[g], a.z[j]);
  a.theta[j] = fresh;
  dv[c] = make_double2(fresh - old, vn);
}

// before the first pass of a sweep: dv[c] = (0, V[column c, first factor])
__global__ void k_mf_gather(const double *__restrict__ theta, const int32_t *__restrict__ cols, int n_cols,
                            double2 *__restrict__ dv) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n_cols) dv[c] = make_double2(0.0, theta[cols[c]]);
}

}  // namespace mfm
