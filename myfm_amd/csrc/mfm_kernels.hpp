// mfm_kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the Gibbs hot path.
//
// Data layout in HBM:
//   eq[N]      double2 {e_t, q_t}: the residual and the per-factor q-cache interleaved, so that
//              one 16-byte access serves both vectors of the latent sweep (FMTrainer.hpp:351-374
//              reads and writes e and q at the same rows).
//   rec[B][8]  one 64-byte record per relation-block row: {q, q_S, c, c_S, e, e_q, cardinality, -}
//              (RelationWiseCache, definitions.hpp:54-84) so a block-feature sweep gathers one
//              record per entry instead of seven scattered doubles.
//   CSC        colptr int64[D+1], rowidx int32[nnz], val f64[nnz]   (= X_t, BaseFMTrainer.hpp:61)
//   CSR        rowptr int32[N+1], colidx int32[nnz], val f64[nnz]
//   V          factor-major [K][D] (the reference's column-major (D,K)), w[D]
//   Vt         row-major gather table [D][KS] for the fused re-score, KS = K rounded up to even
//
// All arithmetic is fp64. The path is HBM/gather bound (< 0.1 flop/byte): no MFMA here by design.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mfm_policies.hpp"
#include "mfm_wave.hpp"

namespace mfm {

__global__ void k_e_pack(const double2 *__restrict__ eq, double *__restrict__ ec, int64_t N) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) ec[i] = eq[i].x;
}
__global__ void k_e_unpack(double2 *__restrict__ eq, const double *__restrict__ ec, int64_t N) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) eq[i].x = ec[i];
}

// One column handled by NT cooperating threads (NT = 64: a wavefront, NT = 256: the workgroup),
// at most NT * R entries. The column's (row, x, state) are staged in registers so that the CSC
// column and the gathered state are read from memory exactly once: pass 1 statistics, draw,
// pass 2 scatter (FMTrainer.hpp:351-375). UNIT: every stored value of the matrix is 1.0 (one-hot
// designs), the val array is not read at all.
template <class P, int R, int NT, bool UNIT>
__device__ __forceinline__ void column_update(const SweepArgs &a, int j, int tid, double *lds) {
  const int64_t begin = a.colptr[j];
  const int len = (int)(a.colptr[j + 1] - begin);
  const double old = a.theta[j];
  // everything the draw needs is requested now, not after the reduction
  const int g = a.group[j];
  const double zj = a.z[j];
  const int rbase = a.row0 ? a.row0[j] : 0;
  int32_t ri[R];
  double xv[R];
  typename P::St st[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int p = tid + r * NT;
    ri[r] = -1;
    xv[r] = UNIT ? 1.0 : 0.0;
    if (p < len) {
      ri[r] = a.row0 ? rbase + p : a.rowidx[begin + p];
      if (!UNIT) xv[r] = a.val[begin + p];
    }
  }
#pragma unroll
  for (int r = 0; r < R; r++)
    if (ri[r] >= 0) st[r] = P::load(a, ri[r]);
  const double lam = a.lambda[g], mu = a.mu[g];
  double S1 = 0.0, S2 = 0.0;
#pragma unroll
  for (int r = 0; r < R; r++)
    if (ri[r] >= 0) P::stats(xv[r], st[r], old, S1, S2);
  if (NT == WAVE) {
    S1 = wave_allreduce_sum(S1);
    S2 = wave_allreduce_sum(S2);
  } else {
    wg_allreduce2<NT / WAVE>(S1, S2, lds);
  }
  const double fresh = P::draw(S1, S2, old, a.alpha, lam, mu, zj);
#pragma unroll
  for (int r = 0; r < R; r++)
    if (ri[r] >= 0) P::apply(a, ri[r], xv[r], st[r], old, fresh);
  if (tid == 0) a.theta[j] = fresh;
}

// Columns of one level are binned by length (mfm_plan.hpp): W1 (<= 64 entries, wavefront, 1 entry
// per lane), W4 (<= 256), W16 (<= 1024, wavefront, 16 per lane), WG (<= 256 * R_WG, workgroup),
// LONG (cooperating chunks), HUGE (two-pass). One "light" and one "heavy" launch per level keep the
// register budget of the short columns small (more waves in flight) without a launch per bin.
// XCD-aware block index: workgroup b is observed to run on XCD b % 8 (MI355X_MICROARCH.md); map the
// blocks of one XCD to a CONTIGUOUS range of the work list so that columns with neighbouring rows
// share that XCD's L2. Bijective for any nb; a wrong placement guess only costs speed.
__device__ __forceinline__ int xcd_swizzle(int b, int nb, int on) {
  if (!on || nb < 16) return b;
  const int q = nb >> 3, r = nb & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

template <class P, bool UNIT>
__global__ __launch_bounds__(WG) void k_level_light(SweepArgs a, const int32_t *__restrict__ cols_w4, int n_w4,
                                                    const int32_t *__restrict__ cols_w1, int n_w1, int swz) {
  const int nb4 = (n_w4 + 3) >> 2;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if ((int)blockIdx.x < nb4) {
    const int w = xcd_swizzle(blockIdx.x, nb4, swz) * 4 + wv;
    if (w < n_w4) column_update<P, 4, WAVE, UNIT>(a, cols_w4[w], lane, nullptr);
  } else {
    const int w = xcd_swizzle(blockIdx.x - nb4, gridDim.x - nb4, swz) * 4 + wv;
    if (w < n_w1) column_update<P, 1, WAVE, UNIT>(a, cols_w1[w], lane, nullptr);
  }
}

template <class P, bool UNIT>
__global__ __launch_bounds__(WG) void k_level_heavy(SweepArgs a, const int32_t *__restrict__ cols_wg, int n_wg,
                                                    const int32_t *__restrict__ cols_w16, int n_w16, int swz) {
  __shared__ double lds[2 * WG / WAVE];
  if ((int)blockIdx.x < n_wg) {
    column_update<P, P::R_WG, WG, UNIT>(a, cols_wg[xcd_swizzle(blockIdx.x, n_wg, swz)], threadIdx.x, lds);
  } else if (P::R_W16 > 0) {
    const int w = xcd_swizzle(blockIdx.x - n_wg, gridDim.x - n_wg, swz) * 4 + (threadIdx.x >> 6);
    if (w < n_w16) column_update<P, (P::R_W16 > 0 ? P::R_W16 : 1), WAVE, UNIT>(a, cols_w16[w], threadIdx.x & 63, nullptr);
  }
}

// ---- LONG columns, single pass: one workgroup per chunk of <= 256 * R_WG entries, all chunks of a
// column co-resident (the launch never exceeds the device's resident workgroup capacity). Every
// chunk stages its entries in registers, publishes its partial statistics, waits until all chunks
// of its column have arrived, sums the partials in chunk order (deterministic, identical in every
// chunk), draws, and scatters from registers. The exchange uses 8-byte agent-scope atomics on both
// sides (cdna_hip_programming.md G16: valid across XCDs, L1 bypassed), one arrival counter per column,
// monotone across launches (target = epoch * n_chunks), bounded spin.
struct CoopArgs {
  const ChunkDesc *chunks;
  const int32_t *lcols;
  const int32_t *chunk_ptr;     // [n_long + 1]
  double *partial;              // [2 * n_chunks]
  unsigned long long *arrive;   // [n_long]
  unsigned long long epoch;     // >= 1
  int *error;                   // set to 1 on spin timeout
};

template <class P, bool UNIT>
__global__ __launch_bounds__(WG) void k_long_coop(SweepArgs a, CoopArgs ca, int chunk_base) {
  constexpr int R = P::R_WG;
  __shared__ double lds[2 * WG / WAVE + 2];
  const int c = chunk_base + blockIdx.x;
  const ChunkDesc d = ca.chunks[c];
  const int l = d.lcol;
  const int j = ca.lcols[l];
  const int tid = threadIdx.x;
  const double old = a.theta[j];
  const int rbase = a.row0 ? a.row0[j] + (int)(d.begin - a.colptr[j]) : 0;
  int32_t ri[R];
  double xv[R];
  typename P::St st[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int p = tid + r * WG;
    ri[r] = -1;
    xv[r] = UNIT ? 1.0 : 0.0;
    if (p < d.len) {
      ri[r] = a.row0 ? rbase + p : a.rowidx[d.begin + p];
      if (!UNIT) xv[r] = a.val[d.begin + p];
    }
  }
#pragma unroll
  for (int r = 0; r < R; r++)
    if (ri[r] >= 0) st[r] = P::load(a, ri[r]);
  double S1 = 0.0, S2 = 0.0;
#pragma unroll
  for (int r = 0; r < R; r++)
    if (ri[r] >= 0) P::stats(xv[r], st[r], old, S1, S2);
  wg_allreduce2<WG / WAVE>(S1, S2, lds);
  if (tid == 0) {
    __hip_atomic_store(ca.partial + 2 * c, S1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(ca.partial + 2 * c + 1, S2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(ca.arrive + l, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int c0 = ca.chunk_ptr[l], c1 = ca.chunk_ptr[l + 1];
    const unsigned long long target = ca.epoch * (unsigned long long)(c1 - c0);
    unsigned spins = 0;
    while (__hip_atomic_load(ca.arrive + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1u << 22)) {
        *ca.error = 1;
        break;
      }
    }
    double T1 = 0.0, T2 = 0.0;
    for (int cc = c0; cc < c1; cc++) {
      T1 += __hip_atomic_load(ca.partial + 2 * cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      T2 += __hip_atomic_load(ca.partial + 2 * cc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    lds[2 * WG / WAVE] = T1;
    lds[2 * WG / WAVE + 1] = T2;
  }
  __syncthreads();
  S1 = lds[2 * WG / WAVE];
  S2 = lds[2 * WG / WAVE + 1];
  const int g = a.group[j];
  const double fresh = P::draw(S1, S2, old, a.alpha, a.lambda[g], a.mu[g], a.z[j]);
#pragma unroll
  for (int r = 0; r < R; r++)
    if (ri[r] >= 0) P::apply(a, ri[r], xv[r], st[r], old, fresh);
  if (tid == 0 && c == ca.chunk_ptr[l]) a.theta[j] = fresh;
}

// ---- HUGE columns (more chunks than can be co-resident): statistics per chunk, draw, apply ---------
template <class P>
__global__ __launch_bounds__(WG) void k_long_stats(SweepArgs a, const ChunkDesc *__restrict__ chunks,
                                                   const int32_t *__restrict__ lcols, double2 *__restrict__ partial) {
  __shared__ double lds[2 * WG / WAVE];
  const ChunkDesc c = chunks[blockIdx.x];
  const double old = a.theta[lcols[c.lcol]];
  double S1 = 0.0, S2 = 0.0;
  for (int p = threadIdx.x; p < c.len; p += WG) {
    const int32_t row = a.rowidx[c.begin + p];
    const double x = a.val[c.begin + p];
    const typename P::St s = P::load(a, row);
    P::stats(x, s, old, S1, S2);
  }
  wg_allreduce2<WG / WAVE>(S1, S2, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = make_double2(S1, S2);
}

template <class P>
__global__ void k_long_draw(SweepArgs a, const int32_t *__restrict__ lcols, const int32_t *__restrict__ chunk_ptr,
                            int n_long, const double2 *__restrict__ partial, double2 *__restrict__ oldnew) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n_long) return;
  const int j = lcols[l];
  double S1 = 0.0, S2 = 0.0;
  for (int c = chunk_ptr[l]; c < chunk_ptr[l + 1]; c++) {  // fixed order: deterministic
    S1 += partial[c].x;
    S2 += partial[c].y;
  }
  const double old = a.theta[j];
  const int g = a.group[j];
  const double fresh = P::draw(S1, S2, old, a.alpha, a.lambda[g], a.mu[g], a.z[j]);
  a.theta[j] = fresh;
  oldnew[l] = make_double2(old, fresh);
}

template <class P>
__global__ __launch_bounds__(WG) void k_long_apply(SweepArgs a, const ChunkDesc *__restrict__ chunks,
                                                   const double2 *__restrict__ oldnew) {
  const ChunkDesc c = chunks[blockIdx.x];
  const double2 on = oldnew[c.lcol];
  for (int p = threadIdx.x; p < c.len; p += WG) {
    const int32_t row = a.rowidx[c.begin + p];
    const double x = a.val[c.begin + p];
    const typename P::St s = P::load(a, row);
    P::apply(a, row, x, s, on.x, on.y);
  }
}

// ---- scattered levels: row-blocked two-pass path ------------------------------------------------------
// A level whose columns touch rows far apart (the second field of a bipartite one-hot design) pays a
// 64-byte fetch + 32-byte write-back for every 16-byte e/q access in the column-major kernels above
// (measured: ~1.3 TB/s algorithmic vs 4-6 TB/s for contiguous columns). For such levels the entries are
// re-sorted at setup by (row block, column, row) with row blocks of SCAT_RB rows (1 MiB of e/q), and the
// sweep runs entry-parallel in that order so that concurrently resident workgroups gather from an
// L2-resident window:
//   k_scat_stats : thread per entry; wave-level segmented reduction over runs of equal column; each run
//                  total goes to its own precomputed slot (deterministic, no atomics)
//   k_scat_draw  : wavefront per column: sums the column's slots in fixed order, draws
//   k_scat_apply : thread per entry: scatter update of e/q inside the same window
constexpr int SCAT_RB = 65536;

template <class P, bool UNIT>
__global__ __launch_bounds__(WG) void k_scat_stats(SweepArgs a, const int2 *__restrict__ ent, const double *__restrict__ eval,
                                                   int64_t n_ent, const int32_t *__restrict__ run_base,
                                                   double2 *__restrict__ slots, int n_wg, int swz) {
  const int wgi = xcd_swizzle(blockIdx.x, n_wg, swz);
  const int lane = threadIdx.x & 63;
  const int64_t tile = (int64_t)wgi * (WG / WAVE) + (threadIdx.x >> 6);
  const int64_t e = tile * WAVE + lane;
  const bool valid = e < n_ent;
  int j = -1 - lane;  // invalid lanes: distinct keys, never stored
  double s1 = 0.0, s2 = 0.0;
  if (valid) {
    const int2 rc = ent[e];
    j = rc.y;
    const double x = UNIT ? 1.0 : eval[e];
    const typename P::St st = P::load(a, rc.x);
    P::stats(x, st, a.theta[j], s1, s2);
  }
  const int jp = __shfl_up(j, 1, WAVE), jn = __shfl_down(j, 1, WAVE);
  const bool head = lane == 0 || jp != j;
  const bool tail = lane == 63 || jn != j;
  int f = head ? 1 : 0;
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    const double u1 = __shfl_up(s1, d, WAVE), u2 = __shfl_up(s2, d, WAVE);
    const int fu = __shfl_up(f, d, WAVE);
    if (lane >= d && !f) {
      s1 += u1;
      s2 += u2;
      f |= fu;
    }
  }
  const unsigned long long hb = __ballot(head);
  if (valid && tail) {
    const int run = __popcll(hb & ((2ull << lane) - 1ull)) - 1;
    slots[run_base[tile] + run] = make_double2(s1, s2);
  }
}

template <class P>
__global__ __launch_bounds__(WG) void k_scat_draw(SweepArgs a, const int32_t *__restrict__ cols, int n_cols,
                                                  const int32_t *__restrict__ slot_ptr, const int32_t *__restrict__ slot_idx,
                                                  const double2 *__restrict__ slots, double2 *__restrict__ oldnew) {
  const int c = blockIdx.x * (WG / WAVE) + (threadIdx.x >> 6);
  if (c >= n_cols) return;
  const int lane = threadIdx.x & 63;
  double S1 = 0.0, S2 = 0.0;
  for (int k = slot_ptr[c] + lane; k < slot_ptr[c + 1]; k += WAVE) {
    const double2 s = slots[slot_idx[k]];
    S1 += s.x;
    S2 += s.y;
  }
  S1 = wave_allreduce_sum(S1);
  S2 = wave_allreduce_sum(S2);
  if (lane == 0) {
    const int j = cols[c];
    const double old = a.theta[j];
    const int g = a.group[j];
    const double fresh = P::draw(S1, S2, old, a.alpha, a.lambda[g], a.mu[g], a.z[j]);
    a.theta[j] = fresh;
    oldnew[j] = make_double2(old, fresh);
  }
}

template <class P, bool UNIT>
__global__ __launch_bounds__(WG) void k_scat_apply(SweepArgs a, const int2 *__restrict__ ent, const double *__restrict__ eval,
                                                   int64_t n_ent, const double2 *__restrict__ oldnew, int n_wg, int swz) {
  const int64_t e = (int64_t)xcd_swizzle(blockIdx.x, n_wg, swz) * WG + threadIdx.x;
  if (e >= n_ent) return;
  const int2 rc = ent[e];
  const double2 on = oldnew[rc.y];
  const double x = UNIT ? 1.0 : eval[e];
  const typename P::St st = P::load(a, rc.x);
  P::apply(a, rc.x, x, st, on.x, on.y);
}

// ---- scattered levels, row-tile path --------------------------------------------------------------------
// The L2-window kernels above still issue one divergent 16-byte global access per entry and pass (a full
// cache-line transaction in the CU's vector memory pipeline each), which is what bounds them (~2 TB/s
// algorithmic). Here a workgroup owns a tile of 2^tile_bits rows whose {e, q} records fit in LDS: it loads
// the tile with coalesced 16-byte accesses, gathers / updates the records in LDS, and (apply pass) writes
// the tile back coalesced -- global memory only sees streams. The entries of the level are sorted by
// (tile, column, row) and packed to 4 bytes, (index of the column in the level) << tile_bits | (row - tile
// start); every tile's entry range is padded to whole 64-entry wave tiles with 0xffffffff.
//   k_tile_stats : tile -> LDS; per wave tile: segmented reduction over runs of equal column -> slots
//   k_tile_draw  : wavefront per column: sums its slots in fixed order, draws, writes (old, new)[column]
//   k_tile_apply : tile -> LDS; updates in LDS (rows of one level are disjoint); LDS -> tile
constexpr uint32_t TILE_PAD = 0xffffffffu;
constexpr int TILE_K = 8;  // rows (and at most entries: a level has <= 1 entry per row) per thread of a tile


// before the statistics pass: the current coefficient per column of the level as a compact array, so that the
// entry loop needs one gather (in the fused flow k_tile_draw of the previous factor fills it instead)
__global__ void k_tile_old(const double *__restrict__ theta, const int32_t *__restrict__ cols, int n_cols,
                           double *__restrict__ told) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n_cols) told[c] = theta[cols[c]];
}

// statistics of the <= TILE_K wave tiles of entries a wave holds in registers, from the tile in LDS:
// segmented reduction over runs of equal column, one slot per run (column-major slot order: the draw streams)
template <class P>
__device__ __forceinline__ void tile_entry_stats(const double2 *lds_rec, const uint32_t (&u)[TILE_K], const double (&x)[TILE_K],
                                                 const double (&old)[TILE_K], const int (&rb)[TILE_K], int t0, int t1, int nw,
                                                 int lane, int tile_bits, const int32_t *__restrict__ slot_pos,
                                                 double2 *__restrict__ slots) {
  const uint32_t rmask = (1u << tile_bits) - 1u;
  // pass 1: run structure and slot addresses of all wave tiles (the slot-position loads are in flight together)
  int pos[TILE_K];
  unsigned flags[TILE_K];  // bit 0 head, bit 1 store (valid tail)
#pragma unroll
  for (int k = 0; k < TILE_K; k++) {
    const int t = t0 + k * nw;
    pos[k] = 0;
    flags[k] = 0;
    if (t >= t1) continue;  // wave-uniform
    const bool valid = u[k] != TILE_PAD;
    const int c = valid ? (int)(u[k] >> tile_bits) : -1 - lane;  // padding lanes: distinct keys, never stored
    const int cp = dpp_i32<0x138, 0xf>(c, 0), cn = dpp_i32<0x130, 0xf>(c, 0);  // wave_shr:1 / wave_shl:1
    const bool head = lane == 0 || cp != c;
    const bool tail = lane == 63 || cn != c;
    const unsigned long long hb = __ballot(head);
    if (valid && tail) pos[k] = slot_pos[rb[k] + __popcll(hb & ((2ull << lane) - 1ull)) - 1];
    flags[k] = (head ? 1u : 0u) | (valid && tail ? 2u : 0u);
  }
  // pass 2: statistics from LDS, segmented scan, one store per run
#pragma unroll
  for (int k = 0; k < TILE_K; k++) {
    const int t = t0 + k * nw;
    if (t >= t1) continue;
    double s1 = 0.0, s2 = 0.0;
    if (u[k] != TILE_PAD) P::stats(x[k], P::from_rec(lds_rec[u[k] & rmask]), old[k], s1, s2);
    int f = (int)(flags[k] & 1u);
    wave_segscan2(s1, s2, f);
    if (flags[k] & 2u) slots[pos[k]] = make_double2(s1, s2);
  }
}

// blockDim.x = 2^tile_bits / TILE_K. All global loads of the workgroup (its tile of records, its <= TILE_K
// wave tiles of entries per wave, the gathered old coefficients) are issued before the barrier.
template <class P, bool UNIT, bool SOA = false>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(1, 4))) void k_tile_stats(
    SweepArgs a, const uint32_t *__restrict__ tent, const double *__restrict__ tval, const int32_t *__restrict__ tile_ptr,
    const int32_t *__restrict__ tile_row0, const double *__restrict__ told, const int32_t *__restrict__ run_base,
    const int32_t *__restrict__ slot_pos, double2 *__restrict__ slots, int tile_bits, int n_tiles, int swz,
    const int32_t *__restrict__ tile_list) {
  extern __shared__ double2 lds_rec[];
  const int b = tile_list ? tile_list[blockIdx.x] : xcd_swizzle(blockIdx.x, n_tiles, swz);
  const int64_t row0 = tile_row0[b];
  const int nr = tile_row0[b + 1] - (int)row0;
  const int nt = blockDim.x, tid = threadIdx.x, lane = tid & 63, nw = nt >> 6;
  d2_t rec[TILE_K];
#pragma unroll
  for (int k = 0; k < TILE_K; k++) {
    rec[k] = d2_t{0.0, 0.0};
    if (tid + k * nt < nr) {
      if (SOA)
        rec[k] = d2_t{((const double *)a.state)[row0 + tid + k * nt], a.state2[row0 + tid + k * nt]};
      else
        rec[k] = ((const d2_t *)a.state)[row0 + tid + k * nt];
    }
  }
  const int t0 = tile_ptr[b] + (tid >> 6), t1 = tile_ptr[b + 1];
  uint32_t u[TILE_K];
  int rb[TILE_K];
  double x[TILE_K], old[TILE_K];
#pragma unroll
  for (int k = 0; k < TILE_K; k++) {
    const int t = t0 + k * nw;
    u[k] = TILE_PAD;
    x[k] = 1.0;
    rb[k] = 0;
    if (t < t1) {
      rb[k] = run_base[t];
      u[k] = tent[(int64_t)t * WAVE + lane];
      if (!UNIT) x[k] = tval[(int64_t)t * WAVE + lane];
    }
  }
#pragma unroll
  for (int k = 0; k < TILE_K; k++) old[k] = u[k] != TILE_PAD ? told[u[k] >> tile_bits] : 0.0;
#pragma unroll
  for (int k = 0; k < TILE_K; k++)
    if (tid + k * nt < nr) ((d2_t *)lds_rec)[tid + k * nt] = rec[k];
  __syncthreads();
  tile_entry_stats<P>(lds_rec, u, x, old, rb, t0, t1, nw, lane, tile_bits, slot_pos, slots);
}

// (old, new) indexed by the column's position in the level; a column's slots are contiguous
template <class P>
__global__ __launch_bounds__(WG) void k_tile_draw(SweepArgs a, const int32_t *__restrict__ cols, int n_cols,
                                                  const int32_t *__restrict__ slot_ptr, const double2 *__restrict__ slots,
                                                  double2 *__restrict__ oldnew, const double *__restrict__ theta_next = nullptr,
                                                  double *__restrict__ vnext_col = nullptr) {
  const int c = blockIdx.x * (WG / WAVE) + (threadIdx.x >> 6);
  if (c >= n_cols) return;
  const int lane = threadIdx.x & 63;
  if (theta_next && lane == 1) vnext_col[c] = theta_next[cols[c]];  // for k_tile_apply_next<.., TWO>
  double S1 = 0.0, S2 = 0.0;
  {
    // four loads in flight per lane (a popular column has one slot per tile: thousands); fixed order
    const int k1 = slot_ptr[c + 1];
    int k = slot_ptr[c] + lane;
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
    const int j = cols[c];
    const double old = a.theta[j];
    const int g = a.group[j];
    const double fresh = P::draw(S1, S2, old, a.alpha, a.lambda[g], a.mu[g], a.z[j]);
    a.theta[j] = fresh;
    oldnew[c] = make_double2(old, fresh);
  }
}

template <class P, bool UNIT, bool SOA = false, bool WRITE_Q = true>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(1, 4))) void k_tile_apply(SweepArgs a, const uint32_t *__restrict__ tent,
                                                     const double *__restrict__ tval, const int32_t *__restrict__ tile_ptr,
                                                     const int32_t *__restrict__ tile_row0,
                                                     const double2 *__restrict__ oldnew, int tile_bits, int n_tiles, int swz) {
  extern __shared__ double2 lds_rec[];
  const int b = xcd_swizzle(blockIdx.x, n_tiles, swz);
  const int64_t row0 = tile_row0[b];
  const int nr = tile_row0[b + 1] - (int)row0;
  const int nt = blockDim.x, tid = threadIdx.x;
  d2_t rec[TILE_K];
#pragma unroll
  for (int k = 0; k < TILE_K; k++) {
    rec[k] = d2_t{0.0, 0.0};
    if (tid + k * nt < nr) {
      if (SOA)
        rec[k] = d2_t{((const double *)a.state)[row0 + tid + k * nt], a.state2[row0 + tid + k * nt]};
      else
        rec[k] = ((const d2_t *)a.state)[row0 + tid + k * nt];
    }
  }
  const int64_t p0 = (int64_t)tile_ptr[b] * WAVE + tid, p1 = (int64_t)tile_ptr[b + 1] * WAVE;
  uint32_t u[TILE_K];
  double x[TILE_K];
  d2_t on[TILE_K];
#pragma unroll
  for (int k = 0; k < TILE_K; k++) {
    const int64_t p = p0 + (int64_t)k * nt;
    u[k] = TILE_PAD;
    x[k] = 1.0;
    if (p < p1) {
      u[k] = tent[p];
      if (!UNIT) x[k] = tval[p];
    }
  }
#pragma unroll
  for (int k = 0; k < TILE_K; k++) {
    on[k] = d2_t{0.0, 0.0};
    if (u[k] != TILE_PAD) on[k] = ((const d2_t *)oldnew)[u[k] >> tile_bits];
  }
#pragma unroll
  for (int k = 0; k < TILE_K; k++)
    if (tid + k * nt < nr) ((d2_t *)lds_rec)[tid + k * nt] = rec[k];
  __syncthreads();
  const uint32_t rmask = (1u << tile_bits) - 1u;
#pragma unroll
  for (int k = 0; k < TILE_K; k++) {
    if (u[k] == TILE_PAD) continue;
    const uint32_t r = u[k] & rmask;
    const double2 old_rec = lds_rec[r];
    lds_rec[r] = P::to_rec(old_rec, P::updated(x[k], P::from_rec(old_rec), on[k][0], on[k][1]));
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < TILE_K; k++)
    if (tid + k * nt < nr) {
      const d2_t r = ((const d2_t *)lds_rec)[tid + k * nt];
      if (SOA) {
        if (!WRITE_Q && a.aos)  // the sweep's last pass: e goes back to the interleaved array
          a.aos[row0 + tid + k * nt].x = r[0];
        else
          ((double *)a.state)[row0 + tid + k * nt] = r[0];
        if (WRITE_Q) a.state2[row0 + tid + k * nt] = r[1];
      } else {
        ((d2_t *)a.state)[row0 + tid + k * nt] = r;
      }
    }
}

// row-sharded mode: the slot sums of a tile level per column (position in the level), to be all-reduced ...
__global__ __launch_bounds__(WG) void k_tile_sum(int n_cols, const int32_t *__restrict__ slot_ptr,
                                                 const double2 *__restrict__ slots, double2 *__restrict__ S) {
  const int c = blockIdx.x * (WG / WAVE) + (threadIdx.x >> 6);
  if (c >= n_cols) return;
  const int lane = threadIdx.x & 63;
  double S1 = 0.0, S2 = 0.0;
  const int k1 = slot_ptr[c + 1];
  int k = slot_ptr[c] + lane;
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
  S1 = wave_allreduce_sum(S1);
  S2 = wave_allreduce_sum(S2);
  if (lane == 0) S[c] = make_double2(S1, S2);
}
// ... and the draw from the reduced statistics (same outputs as k_tile_draw)
template <class P>
__global__ void k_tile_draw_S(SweepArgs a, const int32_t *__restrict__ cols, int n_cols, const double2 *__restrict__ S,
                              double2 *__restrict__ oldnew, const double *__restrict__ theta_next,
                              double *__restrict__ vnext_col) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cols) return;
  const int j = cols[c];
  if (theta_next) vnext_col[c] = theta_next[j];
  const double2 s = S[c];
  const double old = a.theta[j];
  const int g = a.group[j];
  const double fresh = P::draw(s.x, s.y, old, a.alpha, a.lambda[g], a.mu[g], a.z[j]);
  a.theta[j] = fresh;
  oldnew[c] = make_double2(old, fresh);
}
// statistics of a few columns <-> a compact buffer (the all-reduce of the special first-level columns)
__global__ void k_gather_S(const int32_t *__restrict__ cols, int n, const double2 *__restrict__ S, double2 *__restrict__ Sc) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n) Sc[c] = S[cols[c]];
}
__global__ void k_scatter_S(const int32_t *__restrict__ cols, int n, const double2 *__restrict__ Sc, double2 *__restrict__ S) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n) S[cols[c]] = Sc[c];
}
// model synchronisation after a row-sharded latent sweep: every rank keeps the columns it contributes
// (mask 1) and zeroes the rest; the all-reduce that follows leaves the same V on every rank
__global__ void k_mask_rows(double *__restrict__ V, const double *__restrict__ mask, int64_t D, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) V[i] *= mask[i % D];
}

// Apply pass of a factor's LAST level fused with the FIRST level of the next factor (split e / q layout,
// tiles aligned to the first level's contiguous columns, StepPlan::build_aligned_tiles). While the tile is in
// LDS with the final e of factor f, the workgroup
//   1. rebuilds q for factor f + 1 from the rows' CSR entries and V[:, f + 1] (FMTrainer.hpp:320),
//   2. runs the conditional of every first-level column lying inside the tile: statistics over the column's
//      (contiguous) rows from LDS, draw, update of e and q in LDS (FMTrainer.hpp:343-376) -- a wavefront per
//      column, the columns' scalars prefetched 64 at a time,
//   3. writes e and the new q back.
// The tile's rows cross HBM once for both levels, and the latency-bound column kernels of the first level
// are left with only the columns longer than a tile (k_long_coop, launched afterwards).
struct FuseArgs {
  double *theta_next;       // V[:, f + 1]
  const double *z_next;
  const double *lam_next;   // [G]
  const double *mu_next;
  const int4 *fuse_desc;        // first-level columns inside the tiles, row order: {column, length, row offset, group}
  const int32_t *fuse_col_ptr;  // [n_tiles + 1]
  const double *vnext_col;      // V[col, f + 1] per column of the last level (k_tile_draw)
  // stats != 0 (two-level plans): the pass also computes the last level's statistics of factor f + 1
  int stats;
  const int32_t *run_base, *slot_pos;
  double2 *slots;
  // tiles inside a first-level column longer than a tile (complete on this rank): owner column per tile (-1: none)
  // and the tile's partial statistics of that column for factor f + 1 (k_long_tile_draw, k_tile_long_finish)
  const int32_t *solo_col;
  double2 *long_partial;
  // SPLIT (plans with more than two levels): the statistics are those of ANOTHER tile level (the second level of
  // the next factor) than the one applied (this factor's last level): its entry stream
  const uint32_t *tentS;
  const double *tvalS;
  const int32_t *tile_ptrS;
  // MULTIQ (every tile level covers every row once: one-hot fields): the next factor's q from the levels' entry
  // streams -- x * V[col, f + 1] per entry, gathered from a compact per-level array the level's draw leaves behind --
  // instead of from the CSR rows: vnextA for the applied level, ex_* for the other tile levels
  const double *vnextA;
  int n_extra;
  const uint32_t *ex_tent[6];
  const double *ex_tval[6];
  const int32_t *ex_tile_ptr[6];
  const double *ex_vnext[6];
};

// TWO: the plan has exactly these two levels and the last one covers every row once (a two-field one-hot
// table): q of the next factor is x_last * V[last col, f + 1] (gathered per ENTRY from the compact vnext_col,
// entries are sorted by column) + x_first * V[first col, f + 1] (uniform per first-level column) -- no CSR
// read and no per-row gather; with two terms the sum is the same in either order.
template <bool UNIT, bool TWO, bool SPLIT = false, bool MULTIQ = false>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(1, 4))) void k_tile_apply_next(
    SweepArgs a, const uint32_t *__restrict__ tent, const double *__restrict__ tval, const int32_t *__restrict__ tile_ptr,
    const int32_t *__restrict__ tile_row0, const double2 *__restrict__ oldnew, int tile_bits, int n_tiles, int swz,
    FuseArgs fa) {
  extern __shared__ double2 lds_rec[];
  const int b = xcd_swizzle(blockIdx.x, n_tiles, swz);
  const int64_t row0 = tile_row0[b];
  const int nr = tile_row0[b + 1] - (int)row0;
  const int nt = blockDim.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = nt >> 6;
  double *E = (double *)a.state, *Q = a.state2;
  // the tile and entry streams are non-temporal: they pass through L2 once, and keeping them from displacing
  // the scattered 16-byte slot stores lets those merge into whole lines before they leave L2 (measured:
  // 185 -> 148 us per launch at config 3)
  d2_t rec[TILE_K];
#pragma unroll
  for (int k = 0; k < TILE_K; k++) {
    rec[k] = d2_t{0.0, 0.0};
    if (tid + k * nt < nr)
      rec[k] = d2_t{__builtin_nontemporal_load(&E[row0 + tid + k * nt]), __builtin_nontemporal_load(&Q[row0 + tid + k * nt])};
  }
  const int64_t p0 = (int64_t)tile_ptr[b] * WAVE + tid, p1 = (int64_t)tile_ptr[b + 1] * WAVE;
  uint32_t u[TILE_K];
  double x[TILE_K];
  d2_t on[TILE_K];
#pragma unroll
  for (int k = 0; k < TILE_K; k++) {
    const int64_t p = p0 + (int64_t)k * nt;
    u[k] = TILE_PAD;
    x[k] = 1.0;
    if (p < p1) {
      u[k] = __builtin_nontemporal_load(&tent[p]);
      if (!UNIT) x[k] = __builtin_nontemporal_load(&tval[p]);
    }
  }
  // first run of each of this wave's wave tiles (wave-uniform: scalar loads, issued here so that the statistics
  // phase starts with its slot-position gathers)
  const int ts0 = __builtin_amdgcn_readfirstlane(tile_ptr[b] + wv), ts1 = tile_ptr[b + 1];
  int rb[TILE_K];
#pragma unroll
  for (int k = 0; k < TILE_K; k++) rb[k] = !SPLIT && fa.stats && ts0 + k * nw < ts1 ? fa.run_base[ts0 + k * nw] : 0;
  // next factor's coefficient of each entry's column (TWO: its term of the next q; stats: the "old" value)
  double vn[TILE_K];
#pragma unroll
  for (int k = 0; k < TILE_K; k++)
    vn[k] = u[k] == TILE_PAD ? 0.0
            : MULTIQ ? fa.vnextA[u[k] >> tile_bits]
            : (TWO || (fa.stats && !SPLIT)) ? fa.vnext_col[u[k] >> tile_bits] : 0.0;
  // q of the next factor for this thread's rows (TWO: the last level's term, per entry)
  double qn[TILE_K];
#pragma unroll
  for (int k = 0; k < TILE_K; k++) {
    qn[k] = 0.0;
    if (TWO || MULTIQ) {
      qn[k] = x[k] * vn[k];
    } else if (tid + k * nt < nr) {
      const int64_t row = row0 + tid + k * nt;
      if (a.r_ell == 2) {
        const int2 ci = *(const int2 *)(a.r_colidx + row * 2);
        const double v0 = fa.theta_next[ci.x], v1 = fa.theta_next[ci.y];
        qn[k] = (UNIT ? 1.0 : a.r_val[row * 2]) * v0;
        qn[k] += (UNIT ? 1.0 : a.r_val[row * 2 + 1]) * v1;
      } else {
        int64_t pb, pe;
        if (a.r_ell >= 0) {
          pb = row * a.r_ell;
          pe = pb + a.r_ell;
        } else {
          pb = a.r_rowptr[row];
          pe = a.r_rowptr[row + 1];
        }
        double q = 0.0;
        for (int64_t p = pb; p < pe; p++) q += (UNIT ? 1.0 : a.r_val[p]) * fa.theta_next[a.r_colidx[p]];
        qn[k] = q;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < TILE_K; k++) {
    on[k] = d2_t{0.0, 0.0};
    if (u[k] != TILE_PAD) on[k] = ((const d2_t *)oldnew)[u[k] >> tile_bits];
  }
#pragma unroll
  for (int k = 0; k < TILE_K; k++)
    if (tid + k * nt < nr) ((d2_t *)lds_rec)[tid + k * nt] = rec[k];
  __syncthreads();
  // last level of factor f: only e matters from here on
  const uint32_t rmask = (1u << tile_bits) - 1u;
#pragma unroll
  for (int k = 0; k < TILE_K; k++) {
    if (u[k] == TILE_PAD) continue;
    const uint32_t r = u[k] & rmask;
    const double en = PMainV::updated(x[k], lds_rec[r], on[k][0], on[k][1]).x;
    if (TWO || MULTIQ)
      lds_rec[r] = make_double2(en, qn[k]);  // q_f of this row is dead now
    else
      lds_rec[r].x = en;
  }
  __syncthreads();
  if (MULTIQ) {  // the other tile levels' terms of the next q (a level has one entry per row: no two threads meet)
    for (int e = 0; e < fa.n_extra; e++) {
      const int64_t q0 = (int64_t)fa.ex_tile_ptr[e][b] * WAVE + tid, q1 = (int64_t)fa.ex_tile_ptr[e][b + 1] * WAVE;
      uint32_t ue[TILE_K];
      double xe[TILE_K], ve[TILE_K];
#pragma unroll
      for (int k = 0; k < TILE_K; k++) {
        const int64_t p = q0 + (int64_t)k * nt;
        ue[k] = TILE_PAD;
        xe[k] = 1.0;
        if (p < q1) {
          ue[k] = __builtin_nontemporal_load(&fa.ex_tent[e][p]);
          if (!UNIT) xe[k] = __builtin_nontemporal_load(&fa.ex_tval[e][p]);
        }
      }
#pragma unroll
      for (int k = 0; k < TILE_K; k++) ve[k] = ue[k] != TILE_PAD ? fa.ex_vnext[e][ue[k] >> tile_bits] : 0.0;
#pragma unroll
      for (int k = 0; k < TILE_K; k++)
        if (ue[k] != TILE_PAD) lds_rec[ue[k] & rmask].y += xe[k] * ve[k];
      __syncthreads();
    }
  }
  if (!TWO && !MULTIQ) {
#pragma unroll
    for (int k = 0; k < TILE_K; k++)
      if (tid + k * nt < nr) lds_rec[tid + k * nt].y = qn[k];
    __syncthreads();
  }
  // A tile inside a first-level column longer than a tile: the column's statistics for factor f + 1 are summed
  // over its tiles (this tile's part goes to long_partial), the draw follows in k_long_tile_draw and the column's
  // apply pass + the last level's statistics in k_tile_long_finish. The tile leaves with e and the FULL new q.
  const int solo_j = fa.solo_col ? fa.solo_col[b] : -1;
  if (solo_j >= 0) {  // (workgroup-uniform)
    const double old = fa.theta_next[solo_j];
    const int64_t vbase = a.colptr[solo_j] + (row0 - a.row0[solo_j]);
    double S1 = 0.0, S2 = 0.0;
#pragma unroll
    for (int k = 0; k < TILE_K; k++) {
      const int i = tid + k * nt;
      if (i < nr) {
        double2 st = lds_rec[i];
        const double xv = UNIT ? 1.0 : a.val[vbase + i];
        if (TWO || MULTIQ) st.y += xv * old;
        PMainV::stats(xv, st, old, S1, S2);
        __builtin_nontemporal_store(st.x, &E[row0 + i]);
        __builtin_nontemporal_store(st.y, &Q[row0 + i]);
      }
    }
    S1 = wave_allreduce_sum(S1);
    S2 = wave_allreduce_sum(S2);
    double *scr = (double *)(lds_rec + ((size_t)1 << tile_bits));
    if (lane == 0) {
      scr[2 * wv] = S1;
      scr[2 * wv + 1] = S2;
    }
    __syncthreads();
    if (tid == 0) {
      double T1 = 0.0, T2 = 0.0;
      for (int w = 0; w < nw; w++) {  // wave order: deterministic
        T1 += scr[2 * w];
        T2 += scr[2 * w + 1];
      }
      fa.long_partial[b] = make_double2(T1, T2);
    }
    return;
  }
  // first level of factor f + 1: one wavefront per column, lane m of a batch prefetches column m's scalars
  const int c0 = fa.fuse_col_ptr[b], c1 = fa.fuse_col_ptr[b + 1];
  for (int cb = c0 + wv; cb < c1; cb += nw * WAVE) {
    const int cm = cb + lane * nw;
    int pj = 0, plen = 0, plr0 = 0;
    int64_t pbeg = 0;
    double pold = 0.0, pz = 0.0, plam = 0.0, pmu = 0.0;
    if (cm < c1) {
      const int4 d = fa.fuse_desc[cm];  // one coalesced load, then four independent gathers
      pj = d.x;
      plen = d.y;
      plr0 = d.z;
      pold = fa.theta_next[pj];
      pz = fa.z_next[pj];
      plam = fa.lam_next[d.w];
      pmu = fa.mu_next[d.w];
      if (!UNIT) pbeg = a.colptr[pj];
    }
    const int nb_cols = min(WAVE, (c1 - cb + nw - 1) / nw);
    for (int m = 0; m < nb_cols; m++) {
      const int j = __builtin_amdgcn_readlane(pj, m), len = __builtin_amdgcn_readlane(plen, m);
      const int lr0 = __builtin_amdgcn_readlane(plr0, m);
      const int64_t beg = readlane_i64(pbeg, m);
      const double old = readlane_f64(pold, m), zj = readlane_f64(pz, m);
      const double lam = readlane_f64(plam, m), mu = readlane_f64(pmu, m);
      double S1 = 0.0, S2 = 0.0;
      for (int i = lane; i < len; i += WAVE) {
        const double xv = UNIT ? 1.0 : a.val[beg + i];
        double2 st = lds_rec[lr0 + i];
        if (TWO || MULTIQ) st.y += xv * old;
        PMainV::stats(xv, st, old, S1, S2);
      }
      S1 = wave_allreduce_sum(S1);
      S2 = wave_allreduce_sum(S2);
      const double fresh = PMainV::draw(S1, S2, old, a.alpha, lam, mu, zj);
      for (int i = lane; i < len; i += WAVE) {
        const double xv = UNIT ? 1.0 : a.val[beg + i];
        double2 st = lds_rec[lr0 + i];
        if (TWO || MULTIQ) st.y += xv * old;
        lds_rec[lr0 + i] = PMainV::updated(xv, st, old, fresh);
      }
      if (lane == 0) fa.theta_next[j] = fresh;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < TILE_K; k++)
    if (tid + k * nt < nr) {
      const d2_t r = ((const d2_t *)lds_rec)[tid + k * nt];
      __builtin_nontemporal_store(r[0], &E[row0 + tid + k * nt]);
      __builtin_nontemporal_store(r[1], &Q[row0 + tid + k * nt]);
    }
  // The tile now holds the state the last level of factor f + 1 starts from (two-level plan): its statistics
  // from LDS, same entries. Tiles of a first-level column longer than a tile (no columns of their own here)
  // are finished by k_long_coop and get their statistics from k_tile_stats afterwards.
  if (fa.stats && c1 > c0) {
    if (!SPLIT) {
      tile_entry_stats<PMainV>(lds_rec, u, x, vn, rb, ts0, ts1, nw, lane, tile_bits, fa.slot_pos, fa.slots);
    } else {  // the statistics level's own entries (loaded here: the registers of the applied level's are free again)
      const int s0 = __builtin_amdgcn_readfirstlane(fa.tile_ptrS[b] + wv), s1 = fa.tile_ptrS[b + 1];
      uint32_t u2[TILE_K];
      int rb2[TILE_K];
      double x2[TILE_K], v2[TILE_K];
#pragma unroll
      for (int k = 0; k < TILE_K; k++) {
        const int t = s0 + k * nw;
        u2[k] = TILE_PAD;
        x2[k] = 1.0;
        rb2[k] = 0;
        if (t < s1) {
          rb2[k] = fa.run_base[t];
          u2[k] = __builtin_nontemporal_load(&fa.tentS[(int64_t)t * WAVE + lane]);
          if (!UNIT) x2[k] = __builtin_nontemporal_load(&fa.tvalS[(int64_t)t * WAVE + lane]);
        }
      }
#pragma unroll
      for (int k = 0; k < TILE_K; k++) v2[k] = u2[k] != TILE_PAD ? fa.vnext_col[u2[k] >> tile_bits] : 0.0;
      tile_entry_stats<PMainV>(lds_rec, u2, x2, v2, rb2, s0, s1, nw, lane, tile_bits, fa.slot_pos, fa.slots);
    }
  }
}

// Plans with more than two levels, between two tile levels of one factor: ONE pass applies the level just drawn
// (entries A, (old, new) per column) and takes the statistics of the next level (entries S, current coefficient per
// column in told) on the tile in LDS.
template <bool UNIT>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(1, 4))) void k_tile_apply_stats(
    SweepArgs a, const uint32_t *__restrict__ tentA, const double *__restrict__ tvalA, const int32_t *__restrict__ tile_ptrA,
    const double2 *__restrict__ oldnewA, const uint32_t *__restrict__ tentS, const double *__restrict__ tvalS,
    const int32_t *__restrict__ tile_ptrS, const double *__restrict__ told, const int32_t *__restrict__ run_base,
    const int32_t *__restrict__ slot_pos, double2 *__restrict__ slots, const int32_t *__restrict__ tile_row0, int tile_bits,
    int n_tiles, int swz) {
  extern __shared__ double2 lds_rec[];
  const int b = xcd_swizzle(blockIdx.x, n_tiles, swz);
  const int64_t row0 = tile_row0[b];
  const int nr = tile_row0[b + 1] - (int)row0;
  const int nt = blockDim.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = nt >> 6;
  double *E = (double *)a.state, *Q = a.state2;
  d2_t rec[TILE_K];
#pragma unroll
  for (int k = 0; k < TILE_K; k++) {
    rec[k] = d2_t{0.0, 0.0};
    if (tid + k * nt < nr)
      rec[k] = d2_t{__builtin_nontemporal_load(&E[row0 + tid + k * nt]), __builtin_nontemporal_load(&Q[row0 + tid + k * nt])};
  }
  const int64_t p0 = (int64_t)tile_ptrA[b] * WAVE + tid, p1 = (int64_t)tile_ptrA[b + 1] * WAVE;
  uint32_t u[TILE_K];
  double x[TILE_K];
  d2_t on[TILE_K];
#pragma unroll
  for (int k = 0; k < TILE_K; k++) {
    const int64_t p = p0 + (int64_t)k * nt;
    u[k] = TILE_PAD;
    x[k] = 1.0;
    if (p < p1) {
      u[k] = __builtin_nontemporal_load(&tentA[p]);
      if (!UNIT) x[k] = __builtin_nontemporal_load(&tvalA[p]);
    }
  }
#pragma unroll
  for (int k = 0; k < TILE_K; k++) {
    on[k] = d2_t{0.0, 0.0};
    if (u[k] != TILE_PAD) on[k] = ((const d2_t *)oldnewA)[u[k] >> tile_bits];
  }
#pragma unroll
  for (int k = 0; k < TILE_K; k++)
    if (tid + k * nt < nr) ((d2_t *)lds_rec)[tid + k * nt] = rec[k];
  __syncthreads();
  const uint32_t rmask = (1u << tile_bits) - 1u;
#pragma unroll
  for (int k = 0; k < TILE_K; k++) {
    if (u[k] == TILE_PAD) continue;
    const uint32_t r = u[k] & rmask;
    lds_rec[r] = PMainV::updated(x[k], lds_rec[r], on[k][0], on[k][1]);
  }
  // the next level's entries
  const int s0 = __builtin_amdgcn_readfirstlane(tile_ptrS[b] + wv), s1 = tile_ptrS[b + 1];
  uint32_t u2[TILE_K];
  int rb2[TILE_K];
  double x2[TILE_K], v2[TILE_K];
#pragma unroll
  for (int k = 0; k < TILE_K; k++) {
    const int t = s0 + k * nw;
    u2[k] = TILE_PAD;
    x2[k] = 1.0;
    rb2[k] = 0;
    if (t < s1) {
      rb2[k] = run_base[t];
      u2[k] = __builtin_nontemporal_load(&tentS[(int64_t)t * WAVE + lane]);
      if (!UNIT) x2[k] = __builtin_nontemporal_load(&tvalS[(int64_t)t * WAVE + lane]);
    }
  }
#pragma unroll
  for (int k = 0; k < TILE_K; k++) v2[k] = u2[k] != TILE_PAD ? told[u2[k] >> tile_bits] : 0.0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < TILE_K; k++)
    if (tid + k * nt < nr) {
      const d2_t r = ((const d2_t *)lds_rec)[tid + k * nt];
      __builtin_nontemporal_store(r[0], &E[row0 + tid + k * nt]);
      __builtin_nontemporal_store(r[1], &Q[row0 + tid + k * nt]);
    }
  tile_entry_stats<PMainV>(lds_rec, u2, x2, v2, rb2, s0, s1, nw, lane, tile_bits, slot_pos, slots);
}

// long first-level columns of the fused flow: sum the tiles' partial statistics (tile order), draw
template <class P>
__global__ void k_long_tile_draw(SweepArgs an, const int32_t *__restrict__ long_cols, const int32_t *__restrict__ long_tile_ptr,
                                 const int32_t *__restrict__ long_tiles, int n_long, const double2 *__restrict__ partial,
                                 double2 *__restrict__ oldnew_long) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n_long) return;
  double S1 = 0.0, S2 = 0.0;
  for (int t = long_tile_ptr[l]; t < long_tile_ptr[l + 1]; t++) {
    const double2 p = partial[long_tiles[t]];
    S1 += p.x;
    S2 += p.y;
  }
  const int j = long_cols[l];
  const double old = an.theta[j];
  const int g = an.group[j];
  const double fresh = P::draw(S1, S2, old, an.alpha, an.lambda[g], an.mu[g], an.z[j]);
  an.theta[j] = fresh;
  oldnew_long[l] = make_double2(old, fresh);
}

// ... and their second pass over the column's tiles: the column's apply (FMTrainer.hpp:371-375) on the tile in LDS,
// then the last level's statistics of the same factor on the tile's entries (two-level plans), write-back.
template <bool UNIT>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(1, 4))) void k_tile_long_finish(
    SweepArgs an, const uint32_t *__restrict__ tent, const double *__restrict__ tval, const int32_t *__restrict__ tile_ptr,
    const int32_t *__restrict__ tile_row0, int tile_bits, const int32_t *__restrict__ long_tiles,
    const int32_t *__restrict__ tile_long_idx, const double2 *__restrict__ oldnew_long, const int32_t *__restrict__ long_cols,
    const double *__restrict__ vnext_col, int stats, const int32_t *__restrict__ run_base, const int32_t *__restrict__ slot_pos,
    double2 *__restrict__ slots) {
  extern __shared__ double2 lds_rec[];
  const int b = long_tiles[blockIdx.x];
  const int64_t row0 = tile_row0[b];
  const int nr = tile_row0[b + 1] - (int)row0;
  const int nt = blockDim.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = nt >> 6;
  double *E = (double *)an.state, *Q = an.state2;
  const int l = tile_long_idx[b];
  const int j = long_cols[l];
  const d2_t on = ((const d2_t *)oldnew_long)[l];
  const int64_t vbase = an.colptr[j] + (row0 - an.row0[j]);
  const int t0 = tile_ptr[b] + wv, t1 = tile_ptr[b + 1];
  uint32_t u[TILE_K];
  int rb[TILE_K];
  double x[TILE_K], vn[TILE_K];
#pragma unroll
  for (int k = 0; k < TILE_K; k++) {
    const int t = t0 + k * nw;
    u[k] = TILE_PAD;
    x[k] = 1.0;
    rb[k] = 0;
    if (stats && t < t1) {
      rb[k] = run_base[t];
      u[k] = __builtin_nontemporal_load(&tent[(int64_t)t * WAVE + lane]);
      if (!UNIT) x[k] = __builtin_nontemporal_load(&tval[(int64_t)t * WAVE + lane]);
    }
  }
#pragma unroll
  for (int k = 0; k < TILE_K; k++) vn[k] = u[k] != TILE_PAD ? vnext_col[u[k] >> tile_bits] : 0.0;
#pragma unroll
  for (int k = 0; k < TILE_K; k++) {
    const int i = tid + k * nt;
    if (i < nr) {
      const double xv = UNIT ? 1.0 : an.val[vbase + i];
      const double2 st = make_double2(__builtin_nontemporal_load(&E[row0 + i]), __builtin_nontemporal_load(&Q[row0 + i]));
      const double2 nw2 = PMainV::updated(xv, st, on[0], on[1]);
      lds_rec[i] = nw2;
      __builtin_nontemporal_store(nw2.x, &E[row0 + i]);
      __builtin_nontemporal_store(nw2.y, &Q[row0 + i]);
    }
  }
  if (!stats) return;
  __syncthreads();
  tile_entry_stats<PMainV>(lds_rec, u, x, vn, rb, t0, t1, nw, lane, tile_bits, slot_pos, slots);
}

// ---- row-sharded (multi-GPU) mode: statistics -> all-reduce -> draw -> apply -------------------------------
// With the training rows partitioned over GPUs a column's sufficient statistics are partial
// on every rank: each level runs as local statistics into S[col], ONE all-reduce over the level's column
// range, an identical draw on every rank (replicated model state and variates), and an apply pass.
template <class P, int R, int NT, bool UNIT>
__device__ __forceinline__ void column_stats(const SweepArgs &a, int j, int tid, double *lds, double2 *__restrict__ S) {
  const int64_t begin = a.colptr[j];
  const int len = (int)(a.colptr[j + 1] - begin);
  const double old = a.theta[j];
  double S1 = 0.0, S2 = 0.0;
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int p = tid + r * NT;
    if (p < len) {
      const int32_t row = a.rowidx[begin + p];
      const double x = UNIT ? 1.0 : a.val[begin + p];
      const typename P::St st = P::load(a, row);
      P::stats(x, st, old, S1, S2);
    }
  }
  if (NT == WAVE) {
    S1 = wave_allreduce_sum(S1);
    S2 = wave_allreduce_sum(S2);
  } else {
    wg_allreduce2<NT / WAVE>(S1, S2, lds);
  }
  if (tid == 0) S[j] = make_double2(S1, S2);
}
template <class P, int R, int NT, bool UNIT>
__device__ __forceinline__ void column_apply(const SweepArgs &a, int j, int tid, const double2 *__restrict__ oldnew) {
  const int64_t begin = a.colptr[j];
  const int len = (int)(a.colptr[j + 1] - begin);
  const double2 on = oldnew[j];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int p = tid + r * NT;
    if (p < len) {
      const int32_t row = a.rowidx[begin + p];
      const double x = UNIT ? 1.0 : a.val[begin + p];
      const typename P::St st = P::load(a, row);
      P::apply(a, row, x, st, on.x, on.y);
    }
  }
}
// PHASE 1: statistics, PHASE 2: apply. Same binning / grid as k_level_light + k_level_heavy.
template <class P, bool UNIT, int PHASE>
__global__ __launch_bounds__(WG) void k_level_split(SweepArgs a, const int32_t *__restrict__ cols_w1, int n_w1,
                                                    const int32_t *__restrict__ cols_w4, int n_w4,
                                                    const int32_t *__restrict__ cols_w16, int n_w16,
                                                    const int32_t *__restrict__ cols_wg, int n_wg, double2 *__restrict__ S,
                                                    const double2 *__restrict__ oldnew) {
  __shared__ double lds[2 * WG / WAVE];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int b = blockIdx.x;
  if (b < n_wg) {
    if (PHASE == 1) column_stats<P, P::R_WG, WG, UNIT>(a, cols_wg[b], threadIdx.x, lds, S);
    else column_apply<P, P::R_WG, WG, UNIT>(a, cols_wg[b], threadIdx.x, oldnew);
    return;
  }
  b -= n_wg;
  const int nb16 = (n_w16 + 3) >> 2, nb4 = (n_w4 + 3) >> 2;
  if (b < nb16) {
    const int w = b * 4 + wv;
    if (w < n_w16) {
      constexpr int R16 = P::R_W16 > 0 ? P::R_W16 : 1;
      if (PHASE == 1) column_stats<P, R16, WAVE, UNIT>(a, cols_w16[w], lane, nullptr, S);
      else column_apply<P, R16, WAVE, UNIT>(a, cols_w16[w], lane, oldnew);
    }
    return;
  }
  b -= nb16;
  if (b < nb4) {
    const int w = b * 4 + wv;
    if (w < n_w4) {
      if (PHASE == 1) column_stats<P, 4, WAVE, UNIT>(a, cols_w4[w], lane, nullptr, S);
      else column_apply<P, 4, WAVE, UNIT>(a, cols_w4[w], lane, oldnew);
    }
    return;
  }
  b -= nb4;
  const int w = b * 4 + wv;
  if (w < n_w1) {
    if (PHASE == 1) column_stats<P, 1, WAVE, UNIT>(a, cols_w1[w], lane, nullptr, S);
    else column_apply<P, 1, WAVE, UNIT>(a, cols_w1[w], lane, oldnew);
  }
}
// chunk partials -> S[col] (fixed chunk order)
__global__ void k_long_sum(const int32_t *__restrict__ lcols, const int32_t *__restrict__ chunk_ptr, int n_long,
                           const double2 *__restrict__ partial, double2 *__restrict__ S) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n_long) return;
  double S1 = 0.0, S2 = 0.0;
  for (int c = chunk_ptr[l]; c < chunk_ptr[l + 1]; c++) {
    S1 += partial[c].x;
    S2 += partial[c].y;
  }
  S[lcols[l]] = make_double2(S1, S2);
}
// run totals of a scattered level -> S[col] (wavefront per column, stream order)
__global__ __launch_bounds__(WG) void k_scat_sum(const int32_t *__restrict__ cols, int n_cols,
                                                 const int32_t *__restrict__ slot_ptr, const int32_t *__restrict__ slot_idx,
                                                 const double2 *__restrict__ slots, double2 *__restrict__ S) {
  const int c = blockIdx.x * (WG / WAVE) + (threadIdx.x >> 6);
  if (c >= n_cols) return;
  const int lane = threadIdx.x & 63;
  double S1 = 0.0, S2 = 0.0;
  for (int k = slot_ptr[c] + lane; k < slot_ptr[c + 1]; k += WAVE) {
    const double2 s = slots[slot_idx[k]];
    S1 += s.x;
    S2 += s.y;
  }
  S1 = wave_allreduce_sum(S1);
  S2 = wave_allreduce_sum(S2);
  if (lane == 0) S[cols[c]] = make_double2(S1, S2);
}
// the draw of every column of a level from the (all-reduced) statistics
template <class P>
__global__ void k_col_draw(SweepArgs a, const int32_t *__restrict__ cols, int n_cols, const double2 *__restrict__ S,
                           double2 *__restrict__ oldnew) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cols) return;
  const int j = cols[c];
  const double2 s = S[j];
  const double old = a.theta[j];
  const int g = a.group[j];
  const double fresh = P::draw(s.x, s.y, old, a.alpha, a.lambda[g], a.mu[g], a.z[j]);
  a.theta[j] = fresh;
  oldnew[j] = make_double2(old, fresh);
}
// apply for chunked long columns with (old, new) indexed by column
template <class P>
__global__ __launch_bounds__(WG) void k_long_apply_col(SweepArgs a, const ChunkDesc *__restrict__ chunks,
                                                       const int32_t *__restrict__ lcols, const double2 *__restrict__ oldnew) {
  const ChunkDesc c = chunks[blockIdx.x];
  const double2 on = oldnew[lcols[c.lcol]];
  for (int p = threadIdx.x; p < c.len; p += WG) {
    const int32_t row = a.rowidx[c.begin + p];
    const double x = a.val[c.begin + p];
    const typename P::St s = P::load(a, row);
    P::apply(a, row, x, s, on.x, on.y);
  }
}

// ---- sequential chain: a run of tiny levels handled by ONE workgroup, column after column -------
// Used where the conflict graph leaves no parallelism across columns (dense / multi-hot columns,
// relation blocks): a launch per level would be launch-latency bound. This is the variant for tables
// whose state does not fit in LDS (k_chain_lds otherwise). Software pipeline, one column ahead: while
// column c is processed, the descriptor of c+2 and the entries / old coefficient / variate / group
// hyper-parameters of c+1 are already in flight; the first CHAIN_WG * CHAIN_R entries of a column are
// staged in registers together with their gathered state (read once, like column_update).
struct ChainDesc {
  int64_t begin;
  int32_t len;
  int32_t col;
};
constexpr int CHAIN_R = 4;

template <class P>
__global__ __launch_bounds__(CHAIN_WG) void k_chain(SweepArgs a, const ChainDesc *__restrict__ desc, int n_cols) {
  __shared__ double lds[2 * CHAIN_WG / WAVE];
  constexpr int R = CHAIN_R, NT = CHAIN_WG;
  const int tid = threadIdx.x;
  ChainDesc d_next = {0, 0, 0}, d_next2 = {0, 0, 0};
  int32_t nidx[R];
  double nval[R];
  double n_old = 0, n_z = 0, n_lam = 0, n_mu = 0;
  auto issue = [&](const ChainDesc &d) {  // stage B for the column described by d
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int p = tid + r * NT;
      nidx[r] = -1;
      nval[r] = 0.0;
      if (p < d.len) {
        nidx[r] = a.rowidx[d.begin + p];
        nval[r] = a.val[d.begin + p];
      }
    }
    n_old = a.theta[d.col];
    n_z = a.z[d.col];
    const int g = a.group[d.col];
    n_lam = a.lambda[g];
    n_mu = a.mu[g];
  };
  if (n_cols > 0) d_next = desc[0];
  if (n_cols > 1) d_next2 = desc[1];
  if (n_cols > 0) issue(d_next);
  for (int c = 0; c < n_cols; c++) {
    const ChainDesc d = d_next;
    int32_t cidx[R];
    double cval[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      cidx[r] = nidx[r];
      cval[r] = nval[r];
    }
    const double old = n_old, zc = n_z, lam = n_lam, mu = n_mu;
    d_next = d_next2;
    if (c + 2 < n_cols) d_next2 = desc[c + 2];
    if (c + 1 < n_cols) issue(d_next);
    // ---- column c ----
    typename P::St st[R];
    double S1 = 0.0, S2 = 0.0;
#pragma unroll
    for (int r = 0; r < R; r++)
      if (cidx[r] >= 0) {
        st[r] = P::load(a, cidx[r]);
        P::stats(cval[r], st[r], old, S1, S2);
      }
    for (int p = tid + R * NT; p < d.len; p += NT) {
      const int32_t row = a.rowidx[d.begin + p];
      const double x = a.val[d.begin + p];
      const typename P::St s = P::load(a, row);
      P::stats(x, s, old, S1, S2);
    }
    wg_allreduce2<NT / WAVE>(S1, S2, lds);
    const double fresh = P::template draw<true>(S1, S2, old, a.alpha, lam, mu, zc);
#pragma unroll
    for (int r = 0; r < R; r++)
      if (cidx[r] >= 0) P::apply(a, cidx[r], cval[r], st[r], old, fresh);
    for (int p = tid + R * NT; p < d.len; p += NT) {
      const int32_t row = a.rowidx[d.begin + p];
      const double x = a.val[d.begin + p];
      const typename P::St s = P::load(a, row);
      P::apply(a, row, x, s, old, fresh);
    }
    if (tid == 0) a.theta[d.col] = fresh;
    __syncthreads();  // this column's stores are visible to the workgroup before the next column's gathers
  }
}

// ---- sequential chain, state resident in LDS ---------------------------------------------------------
// When the table's per-row state fits in LDS (small relation blocks: B x 64 bytes, toy main tables),
// ONE wavefront walks the chain with the state staged in LDS for the whole launch: a column then costs
// LDS round trips instead of L2/HBM latency. Columns are processed in batches of CHAIN_CB; everything
// a column needs from global memory (descriptor, old coefficient, variate, group hyper-parameters and
// its first 64 entries) is prefetched into registers one batch ahead, descriptors two batches ahead.
constexpr int CHAIN_CB = 16;

template <class P>
__global__ __launch_bounds__(WAVE) void k_chain_lds(SweepArgs a, const ChainDesc *__restrict__ desc, int n_cols,
                                                    int64_t n_rows, int rec_doubles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CB = CHAIN_CB;
  double2 *lst = (double2 *)smem;
  const int lane = threadIdx.x;
  const int w16 = rec_doubles / 2;              // 16-byte words per record in global memory
  const int l16 = w16 > 1 ? w16 + 1 : w16;      // ... in LDS: 64-byte records are padded to 80 bytes (bank spread)
  const int64_t n16 = n_rows * w16;
  double2 *gst = (double2 *)a.state;
  // staging: eight independent 16-byte loads in flight per lane (a plain copy loop of one wavefront pays a memory round
  // trip per KB: ~100 us for the 105 KB of an ML-100k-sized block, a tenth of the launch)
  for (int64_t i0 = lane; i0 < n16; i0 += 8 * WAVE) {
    double2 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (i0 + u * WAVE < n16) v[u] = gst[i0 + u * WAVE];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int64_t i = i0 + u * WAVE;
      if (i < n16) lst[(i / w16) * l16 + (i % w16)] = v[u];
    }
  }
  SweepArgs al = a;
  al.state = lst;
  al.rec2 = l16;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

  // per-lane (lane < CB) column scalars and per-column entry registers: cur = batch being processed, nxt = batch in
  // flight. No load inside the loop waits on another load of the same step (the wavefront issues in order: a dependent
  // pair would stall the chain for a memory round trip per batch): descriptors are fetched three batches ahead (dA), the
  // group index of a column two batches ahead (gB, from the descriptors dB that have landed), and a batch's coefficients,
  // variates, hyper-parameters and entries one batch ahead. The drawn coefficients are collected per lane and stored by
  // ONE instruction after the next batch's registers have been taken over (vmcnt counts stores as well: a store issued
  // just before that hand-over would make it wait for the store's acknowledgement).
  ChainDesc dA = {0, 0, 0}, dB = {0, 0, 0};
  int gB = 0;
  int32_t c_col = 0, n_col = 0, c_len = 0, n_len = 0, p_col = -1;
  int64_t c_begin = 0, n_begin = 0;
  double c_old = 0, n_old = 0, c_z = 0, n_z = 0, c_lam = 0, n_lam = 0, c_mu = 0, n_mu = 0, c_new = 0, p_new = 0;
  // two entries per lane and column are staged ahead (columns of up to 128 entries never touch global memory inside
  // the chain: a load there costs the chain a full memory round trip per column)
  int32_t cidx[CB], nidx[CB], cidx2[CB], nidx2[CB];
  double cval[CB], nval[CB], cval2[CB], nval2[CB];
  auto load_desc = [&](int base) {
    ChainDesc d = {0, 0, 0};
    if (lane < CB && base + lane < n_cols) d = desc[base + lane];
    return d;
  };
  auto load_group = [&](const ChainDesc &d, int base) { return (lane < CB && base + lane < n_cols) ? a.group[d.col] : 0; };
  auto load_batch = [&](const ChainDesc &dsc, int g) {  // into nxt
    n_col = dsc.col;
    n_len = dsc.len;
    n_begin = dsc.begin;
    if (lane < CB) {
      n_old = a.theta[n_col];
      n_z = a.z[n_col];
      n_lam = a.lambda[g];
      n_mu = a.mu[g];
    }
#pragma unroll
    for (int k = 0; k < CB; k++) {
      const int64_t b = readlane_i64(n_begin, k);
      const int l = __builtin_amdgcn_readlane(n_len, k);
      nidx[k] = -1;
      nval[k] = 0.0;
      nidx2[k] = -1;
      nval2[k] = 0.0;
      if (lane < l) {
        nidx[k] = a.rowidx[b + lane];
        nval[k] = a.val[b + lane];
      }
      if (lane + WAVE < l) {
        nidx2[k] = a.rowidx[b + lane + WAVE];
        nval2[k] = a.val[b + lane + WAVE];
      }
    }
  };
  dB = load_desc(0);
  gB = load_group(dB, 0);
  load_batch(dB, gB);
  dB = load_desc(CB);
  gB = load_group(dB, CB);
  dA = load_desc(2 * CB);
  for (int base = 0; base < n_cols; base += CB) {
    // rotate: nxt -> cur, start the loads of the following batch
    c_col = n_col;
    c_len = n_len;
    c_begin = n_begin;
    c_old = n_old;
    c_z = n_z;
    c_lam = n_lam;
    c_mu = n_mu;
#pragma unroll
    for (int k = 0; k < CB; k++) {
      cidx[k] = nidx[k];
      cval[k] = nval[k];
      cidx2[k] = nidx2[k];
      cval2[k] = nval2[k];
    }
    if (lane < CB && p_col >= 0) a.theta[p_col] = p_new;  // the batch before this one
    if (base + CB < n_cols) {
      load_batch(dB, gB);
      dB = dA;
      gB = load_group(dB, base + 2 * CB);
      dA = load_desc(base + 3 * CB);
    }
#pragma unroll
    for (int k = 0; k < CB; k++) {
      if (base + k < n_cols) {
        const int clen = __builtin_amdgcn_readlane(c_len, k);
        const int64_t cbegin = readlane_i64(c_begin, k);
        const double old = readlane_f64(c_old, k);
        double S1 = 0.0, S2 = 0.0;
        typename P::St st0, st1;
        // lanes past the column's end read row 0 with x = 0 (a zero contribution) instead of branching on the execution
        // mask: both halves' loads are issued back to back and their statistics interleave
        const int r0 = max(cidx[k], 0), r1 = max(cidx2[k], 0);
        const bool two = clen > WAVE;  // (wave-uniform)
        st0 = P::load(al, r0);
        if (two) {
          double T1 = 0.0, T2 = 0.0;
          st1 = P::load(al, r1);
          ChainOps<P>::stats(cval[k], st0, old, S1, S2);
          ChainOps<P>::stats(cval2[k], st1, old, T1, T2);
          S1 += T1;
          S2 += T2;
        } else {
          ChainOps<P>::stats(cval[k], st0, old, S1, S2);
        }
        for (int p = lane + 2 * WAVE; p < clen; p += WAVE) {
          const int32_t row = a.rowidx[cbegin + p];
          const double x = a.val[cbegin + p];
          const typename P::St st = P::load(al, row);
          ChainOps<P>::stats(x, st, old, S1, S2);
        }
        wave_allreduce_sum2(S1, S2);
        const double fresh =
            P::template draw<true>(S1, S2, old, a.alpha, readlane_f64(c_lam, k), readlane_f64(c_mu, k), readlane_f64(c_z, k));
        if (lane < clen) ChainOps<P>::apply(al, r0, cval[k], st0, old, fresh);
        if (two && lane + WAVE < clen) ChainOps<P>::apply(al, r1, cval2[k], st1, old, fresh);
        for (int p = lane + 2 * WAVE; p < clen; p += WAVE) {
          const int32_t row = a.rowidx[cbegin + p];
          const double x = a.val[cbegin + p];
          const typename P::St st = P::load(al, row);
          ChainOps<P>::apply(al, row, x, st, old, fresh);
        }
        if (lane == k) c_new = fresh;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
    }
    p_col = base + lane < n_cols ? c_col : -1;
    p_new = c_new;
  }
  if (lane < CB && p_col >= 0) a.theta[p_col] = p_new;
  for (int64_t i = lane; i < n16; i += WAVE) gst[i] = lst[(i / w16) * l16 + (i % w16)];
}

// ---- sequential chains, conflict-batched ------------------------------------------------------------------
// A chain walks columns that pairwise share rows, one after the other; over state in global memory every column
// costs the full dependent round trip gather -> reduce -> draw -> scatter -> visible (~1.9 us). Consecutive
// columns of a multi-hot table share only FEW rows, though. For a batch of B consecutive columns call a row HOT
// if more than one column of the batch touches it, COLD otherwise. A cold row's state is read and written by
// exactly one column of the batch, so, exactly as in the sequential order,
//   A  (all threads) every column's statistics over its COLD entries are taken from the state at batch start,
//   B  (one wavefront) the columns are walked in order over their HOT entries only, on the hot rows' records
//      staged in LDS: S = S_cold + S_hot -> draw -> update of the hot records -- a ~0.25 us step,
//   C  (all threads) the cold entries are updated with their column's (old, new); the hot records go back.
// The draws are those of the sequential sweep (only the order of the floating-point sums differs).
struct ChainBatch {
  int32_t col0, ncols;      // columns [col0, col0 + ncols) of the run
  int32_t hot_row0, n_hot;  // hot rows: hot_rows[hot_row0 ..)
  int32_t cold_b, cold_e;   // cold entries [cold_ptr[col0], cold_ptr[col0 + ncols])  (copies: one dependent load less per launch)
  int32_t hot_b, hot_e;     // hot entries  [hot_ptr[col0], hot_ptr[col0 + ncols])
};
constexpr int CHAINB_NT = 512;
constexpr int CHAINB_MAXCOLS = 64;

// One column of a hot walk (k_chain_batched phase B, k_cb_hot), by ONE wavefront over records staged in LDS: statistics of
// the column's hot entries [hb, he), draw, update. Up to 4 x 64 entries are handled in a single round -- their slots and
// records are loaded back to back (one LDS round trip instead of one per 64 entries), the four partial statistics
// interleave, and the records stay in registers for the update. Lanes past the end read slot 0 and contribute nothing.
template <class P>
__device__ __forceinline__ double hot_column(const SweepArgs &a, const SweepArgs &al, const double *h_x, const int *h_slot, int hb,
                                             int he, int lane, double S1c, double S2c, double old, double lam, double mu, double z) {
  constexpr int U = 4;
  if (he - hb <= U * WAVE) {
    int sl[U];
    double hx[U], t1[U], t2[U];
    typename P::St st[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int h = hb + u * WAVE + lane;
      const bool ok = h < he;
      sl[u] = ok ? h_slot[h] : 0;
      hx[u] = ok ? h_x[h] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; u++) st[u] = P::load(al, sl[u]);
#pragma unroll
    for (int u = 0; u < U; u++) {
      t1[u] = 0.0;
      t2[u] = 0.0;
      ChainOps<P>::stats(hx[u], st[u], old, t1[u], t2[u]);
      // (a batch without hot rows has staged no record at all: slot 0 is then whatever the LDS held, and 0 * NaN is NaN)
      if (!(hb + u * WAVE + lane < he)) t1[u] = t2[u] = 0.0;
    }
    double h1 = (t1[0] + t1[1]) + (t1[2] + t1[3]), h2 = (t2[0] + t2[1]) + (t2[2] + t2[3]);
    wave_allreduce_sum2(h1, h2);
    const double fresh = P::template draw<true>(S1c + h1, S2c + h2, old, a.alpha, lam, mu, z);
#pragma unroll
    for (int u = 0; u < U; u++)
      if (hb + u * WAVE + lane < he) ChainOps<P>::apply(al, sl[u], hx[u], st[u], old, fresh);
    return fresh;
  }
  double h1 = 0.0, h2 = 0.0;
  for (int h = hb + lane; h < he; h += WAVE) ChainOps<P>::stats(h_x[h], P::load(al, h_slot[h]), old, h1, h2);
  wave_allreduce_sum2(h1, h2);
  const double fresh = P::template draw<true>(S1c + h1, S2c + h2, old, a.alpha, lam, mu, z);
  for (int h = hb + lane; h < he; h += WAVE) {
    const int slot = h_slot[h];
    ChainOps<P>::apply(al, slot, h_x[h], P::load(al, slot), old, fresh);
  }
  return fresh;
}

template <class P>
__global__ __launch_bounds__(CHAINB_NT) void k_chain_batched(SweepArgs a, const ChainBatch *__restrict__ batches, int n_batches,
                                                             const int32_t *__restrict__ cols,
                                                             const int32_t *__restrict__ cold_ptr,
                                                             const int32_t *__restrict__ cold_row,
                                                             const int32_t *__restrict__ cold_lcol,
                                                             const double *__restrict__ cold_x,
                                                             const int32_t *__restrict__ hot_ptr,
                                                             const int32_t *__restrict__ hot_slot,
                                                             const double *__restrict__ hot_x,
                                                             const int32_t *__restrict__ hot_rows, int max_hot,
                                                             int max_hot_ent) {
  extern __shared__ double2 lds_hot[];  // [max_hot * rec2_lds] records, then the per-batch arrays
  constexpr int NT = CHAINB_NT, NW = NT / WAVE, MC = CHAINB_MAXCOLS;
  constexpr int U = 4;  // wave tiles of cold entries in flight per wave
  constexpr int rec2_g = P::REC_DOUBLES / 2;                          // 16-byte words of a record in global memory
  constexpr int rec2_l = P::REC_DOUBLES > 2 ? rec2_g + 1 : rec2_g;    // ... in LDS (bank spread for 64-byte records)
  double *c_old = (double *)(lds_hot + (size_t)max_hot * rec2_l);
  double *c_z = c_old + MC, *c_lam = c_z + MC, *c_mu = c_lam + MC, *c_new = c_mu + MC;
  double2 *part = (double2 *)(c_new + MC);       // [MC][NW]
  double *h_x = (double *)(part + MC * NW);      // [max_hot_ent] the batch's hot entries: value ...
  int *h_slot = (int *)(h_x + max_hot_ent);      // ... and slot
  int *h_ptr = h_slot + max_hot_ent;             // [MC + 1] per column, relative to the batch
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  SweepArgs al = a;  // the hot records in LDS, addressed by slot
  al.state = lds_hot;
  al.rec2 = rec2_l;
  const int rec2_global = P::REC_DOUBLES > 2 ? a.rec2 : 1;
  ChainBatch B_next = batches[0];
  for (int bi = 0; bi < n_batches; bi++) {
    const ChainBatch B = B_next;
    if (bi + 1 < n_batches) B_next = batches[bi + 1];  // (in flight during this batch)
    // ---- stage the hot records, the hot entries and the columns' scalars ----------------------------------------
    for (int i = tid; i < B.n_hot * rec2_g; i += NT) {
      const int slot = i / rec2_g, w = i - slot * rec2_g;
      lds_hot[(size_t)slot * rec2_l + w] = ((const double2 *)a.state)[(int64_t)hot_rows[B.hot_row0 + slot] * rec2_global + w];
    }
    const int hb0 = hot_ptr[B.col0], hb1 = hot_ptr[B.col0 + B.ncols];
    for (int i = tid; i < hb1 - hb0; i += NT) {
      h_x[i] = hot_x[hb0 + i];
      h_slot[i] = hot_slot[hb0 + i];
    }
    if (tid <= B.ncols) h_ptr[tid] = hot_ptr[B.col0 + tid] - hb0;
    if (tid < B.ncols) {
      const int j = cols[B.col0 + tid];
      c_old[tid] = a.theta[j];
      c_z[tid] = a.z[j];
      const int g = a.group[j];
      c_lam[tid] = a.lambda[g];
      c_mu[tid] = a.mu[g];
    }
    for (int i = tid; i < MC * NW; i += NT) part[i] = make_double2(0.0, 0.0);
    __syncthreads();
    // ---- A: statistics of the cold entries: wave tiles of 64 consecutive entries (sorted by column), U in flight ----
    const int cb = cold_ptr[B.col0], ce = cold_ptr[B.col0 + B.ncols];
    for (int base = cb + wv * WAVE * U; base < ce; base += NW * WAVE * U) {
      int lc[U], row[U];
      double xv[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int p = base + u * WAVE + lane;
        lc[u] = -1 - lane;
        row[u] = -1;
        xv[u] = 0.0;
        if (p < ce) {
          lc[u] = cold_lcol[p];
          row[u] = cold_row[p];
          xv[u] = cold_x[p];
        }
      }
      typename P::St st[U];
#pragma unroll
      for (int u = 0; u < U; u++)
        if (row[u] >= 0) st[u] = P::load(a, row[u]);
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (base + u * WAVE >= ce) break;  // wave-uniform
        double s1 = 0.0, s2 = 0.0;
        if (row[u] >= 0) P::stats(xv[u], st[u], c_old[lc[u]], s1, s2);
        const int lp = dpp_i32<0x138, 0xf>(lc[u], 0), ln = dpp_i32<0x130, 0xf>(lc[u], 0);
        const bool head = lane == 0 || lp != lc[u], tail = lane == 63 || ln != lc[u];
        int f = head ? 1 : 0;
        wave_segscan2(s1, s2, f);
        if (row[u] >= 0 && tail) {  // one lane per column of this tile; a wave adds its tiles in program order
          double2 &q = part[lc[u] * NW + wv];
          q.x += s1;
          q.y += s2;
        }
      }
    }
    __syncthreads();
    if (tid < B.ncols) {  // a column's cold statistics: its waves' partials in wave order
      double S1 = 0.0, S2 = 0.0;
      for (int w = 0; w < NW; w++) {
        S1 += part[tid * NW + w].x;
        S2 += part[tid * NW + w].y;
      }
      part[tid * NW] = make_double2(S1, S2);
    }
    __syncthreads();
    // ---- B: the columns in order, hot entries only (one wavefront, LDS only) -----------------------------------------
    if (wv == 0) {
      for (int c = 0; c < B.ncols; c++) {
        const double S1 = part[c * NW].x, S2 = part[c * NW].y;
        const double old = c_old[c];
        const int hb = h_ptr[c], he = h_ptr[c + 1];
        const double fresh = hot_column<P>(a, al, h_x, h_slot, hb, he, lane, S1, S2, old, c_lam[c], c_mu[c], c_z[c]);
        if (lane == 0) c_new[c] = fresh;
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this column's LDS updates before the next column's loads
      }
    }
    __syncthreads();
    // ---- C: cold entries take their column's update; the hot records and the coefficients go back -------------------
    for (int base = cb + tid; base < ce; base += NT * U) {
      int lc[U], row[U];
      double xv[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int p = base + u * NT;
        row[u] = -1;
        lc[u] = 0;
        xv[u] = 0.0;
        if (p < ce) {
          lc[u] = cold_lcol[p];
          row[u] = cold_row[p];
          xv[u] = cold_x[p];
        }
      }
      typename P::St st[U];
#pragma unroll
      for (int u = 0; u < U; u++)
        if (row[u] >= 0) st[u] = P::load(a, row[u]);
#pragma unroll
      for (int u = 0; u < U; u++)
        if (row[u] >= 0) P::apply(a, row[u], xv[u], st[u], c_old[lc[u]], c_new[lc[u]]);
    }
    for (int i = tid; i < B.n_hot * rec2_g; i += NT) {
      const int slot = i / rec2_g, w = i - slot * rec2_g;
      ((double2 *)a.state)[(int64_t)hot_rows[B.hot_row0 + slot] * rec2_global + w] = lds_hot[(size_t)slot * rec2_l + w];
    }
    if (tid < B.ncols) a.theta[cols[B.col0 + tid]] = c_new[tid];
    __syncthreads();  // visible to the workgroup before the next batch's gathers
  }
}

__global__ void k_gather_i32(const int32_t *__restrict__ src, const int32_t *__restrict__ idx, int n, int32_t *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = src[idx[i]];
}

// ---- the same batches, cold parts on the WHOLE GPU ------------------------------------------------------------------
// When a batch's cold entries number in the tens of thousands (relation blocks with 10^5..10^6 rows: config 5) one
// workgroup streaming them through a single CU is the bottleneck (~25 GB/s). Phases A and C touch every row at most once
// per batch, so they run as grid launches; only the walk over the hot rows (B) stays on one workgroup:
//   k_cb_stats  (grid)  per-column partial statistics of the cold entries, one partial row per workgroup
//   k_cb_hot    (1 WG)  partials summed in workgroup order, hot rows in LDS, the columns in order: draw, (old, new) out
//   k_cb_apply  (grid)  cold entries updated with their column's (old, new)
template <class P>
__global__ __launch_bounds__(CHAINB_NT) void k_cb_stats(SweepArgs a, ChainBatch B, const int32_t *__restrict__ cols,
                                                        const int32_t *__restrict__ cold_ptr, const int32_t *__restrict__ cold_row,
                                                        const int32_t *__restrict__ cold_lcol, const double *__restrict__ cold_x,
                                                        double2 *__restrict__ part_g /* [gridDim.x][CHAINB_MAXCOLS] */,
                                                        const int32_t *__restrict__ hot_rows, double2 *__restrict__ hot_pack) {
  constexpr int NT = CHAINB_NT, NW = NT / WAVE, MC = CHAINB_MAXCOLS, U = 4;
  __shared__ double c_old[MC];
  __shared__ double2 part[MC * NW];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  {
    // the batch's hot records, gathered by the whole grid into a packed buffer: k_cb_hot then stages them with ONE coalesced
    // round trip instead of a dependent pair (row list -> records) on a single workgroup
    constexpr int rec2_g = P::REC_DOUBLES / 2;
    const int rec2_global = P::REC_DOUBLES > 2 ? a.rec2 : 1;
    for (int i = (int)blockIdx.x * NT + tid; i < B.n_hot * rec2_g; i += (int)gridDim.x * NT) {
      const int slot = i / rec2_g, w = i - slot * rec2_g;
      hot_pack[i] = ((const double2 *)a.state)[(int64_t)hot_rows[B.hot_row0 + slot] * rec2_global + w];
    }
  }
  if (tid < B.ncols) c_old[tid] = a.theta[cols[B.col0 + tid]];
  for (int i = tid; i < MC * NW; i += NT) part[i] = make_double2(0.0, 0.0);
  __syncthreads();
  const int cb = B.cold_b, ce = B.cold_e;
  for (int base = cb + ((int)blockIdx.x * NW + wv) * WAVE * U; base < ce; base += (int)gridDim.x * NW * WAVE * U) {
    int lc[U], row[U];
    double xv[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int p = base + u * WAVE + lane;
      lc[u] = -1 - lane;
      row[u] = -1;
      xv[u] = 0.0;
      if (p < ce) {
        lc[u] = cold_lcol[p];
        row[u] = cold_row[p];
        xv[u] = cold_x[p];
      }
    }
    typename P::St st[U];
#pragma unroll
    for (int u = 0; u < U; u++)
      if (row[u] >= 0) st[u] = P::load(a, row[u]);
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (base + u * WAVE >= ce) break;  // wave-uniform
      double s1 = 0.0, s2 = 0.0;
      if (row[u] >= 0) P::stats(xv[u], st[u], c_old[lc[u]], s1, s2);
      const int lp = dpp_i32<0x138, 0xf>(lc[u], 0), ln = dpp_i32<0x130, 0xf>(lc[u], 0);
      const bool head = lane == 0 || lp != lc[u], tail = lane == 63 || ln != lc[u];
      int f = head ? 1 : 0;
      wave_segscan2(s1, s2, f);
      if (row[u] >= 0 && tail) {  // one lane per column of this tile; a wave adds its tiles in program order
        double2 &q = part[lc[u] * NW + wv];
        q.x += s1;
        q.y += s2;
      }
    }
  }
  __syncthreads();
  if (tid < MC) {
    double S1 = 0.0, S2 = 0.0;
    if (tid < B.ncols)
      for (int w = 0; w < NW; w++) {  // wave order: deterministic
        S1 += part[tid * NW + w].x;
        S2 += part[tid * NW + w].y;
      }
    part_g[(size_t)blockIdx.x * MC + tid] = make_double2(S1, S2);
  }
}

template <class P>
__global__ __launch_bounds__(CHAINB_NT) void k_cb_hot(SweepArgs a, ChainBatch B, const int32_t *__restrict__ cols,
                                                      const int32_t *__restrict__ hot_ptr, const int32_t *__restrict__ hot_slot,
                                                      const double *__restrict__ hot_x, double2 *__restrict__ hot_pack,
                                                      const int32_t *__restrict__ col_group, int max_hot, int max_hot_ent,
                                                      const double2 *__restrict__ part_g, int n_part,
                                                      double2 *__restrict__ oldnew_g /* [CHAINB_MAXCOLS] */,
                                                      const double *__restrict__ colpack = nullptr) {
  extern __shared__ double2 lds_hot[];
  constexpr int NT = CHAINB_NT, MC = CHAINB_MAXCOLS;
  constexpr int rec2_g = P::REC_DOUBLES / 2;
  constexpr int rec2_l = P::REC_DOUBLES > 2 ? rec2_g + 1 : rec2_g;
  double *c_old = (double *)(lds_hot + (size_t)max_hot * rec2_l);
  double *c_z = c_old + MC, *c_lam = c_z + MC, *c_mu = c_lam + MC, *c_new = c_mu + MC;
  double2 *csum = (double2 *)(c_new + MC);    // [MC]
  double *h_x = (double *)(csum + MC);        // [max_hot_ent]
  int *h_slot = (int *)(h_x + max_hot_ent);
  int *h_ptr = h_slot + max_hot_ent;          // [MC + 1]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  SweepArgs al = a;
  al.state = lds_hot;
  al.rec2 = rec2_l;
  if (tid < B.ncols) {  // (first: the longest dependent chain of the prologue, column -> coefficient / variate, group -> lambda / mu)
    if (colpack) {      //  ... or one load each when k_cb_step packed them)
      c_old[tid] = colpack[tid];
      c_z[tid] = colpack[MC + tid];
      c_lam[tid] = colpack[2 * MC + tid];
      c_mu[tid] = colpack[3 * MC + tid];
    } else {
      const int j = cols[B.col0 + tid];
      const int g = col_group[B.col0 + tid];
      c_old[tid] = a.theta[j];
      c_z[tid] = a.z[j];
      c_lam[tid] = a.lambda[g];
      c_mu[tid] = a.mu[g];
    }
  }
  for (int i = tid; i < B.n_hot * rec2_g; i += NT) {
    const int slot = i / rec2_g, w = i - slot * rec2_g;
    lds_hot[(size_t)slot * rec2_l + w] = hot_pack[i];  // (packed by k_cb_stats)
  }
  const int hb0 = B.hot_b, hb1 = B.hot_e;
  for (int i = tid; i < hb1 - hb0; i += NT) {
    h_x[i] = hot_x[hb0 + i];
    h_slot[i] = hot_slot[hb0 + i];
  }
  if (tid <= B.ncols) h_ptr[tid] = hot_ptr[B.col0 + tid] - hb0;
  if (tid < B.ncols) {
    double S1 = 0.0, S2 = 0.0;
    for (int w0 = 0; w0 < n_part; w0 += 8) {  // workgroup order: deterministic (eight loads in flight)
      double2 q[8];
#pragma unroll
      for (int u = 0; u < 8; u++) q[u] = w0 + u < n_part ? part_g[(size_t)(w0 + u) * MC + tid] : make_double2(0.0, 0.0);
#pragma unroll
      for (int u = 0; u < 8; u++) {
        S1 += q[u].x;
        S2 += q[u].y;
      }
    }
    csum[tid] = make_double2(S1, S2);
  }
  __syncthreads();
  if (wv == 0) {
    for (int c = 0; c < B.ncols; c++) {
      const double S1 = csum[c].x, S2 = csum[c].y;
      const double old = c_old[c];
      const int hb = h_ptr[c], he = h_ptr[c + 1];
      const double fresh = hot_column<P>(a, al, h_x, h_slot, hb, he, lane, S1, S2, old, c_lam[c], c_mu[c], c_z[c]);
      if (lane == 0) c_new[c] = fresh;
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    }
  }
  __syncthreads();
  for (int i = tid; i < B.n_hot * rec2_g; i += NT) {
    const int slot = i / rec2_g, w = i - slot * rec2_g;
    hot_pack[i] = lds_hot[(size_t)slot * rec2_l + w];  // (k_cb_apply scatters them to their rows)
  }
  if (tid < B.ncols) {
    a.theta[cols[B.col0 + tid]] = c_new[tid];
    oldnew_g[tid] = make_double2(c_old[tid], c_new[tid]);
  }
}

template <class P>
__global__ __launch_bounds__(CHAINB_NT) void k_cb_apply(SweepArgs a, ChainBatch B, const int32_t *__restrict__ cold_ptr,
                                                        const int32_t *__restrict__ cold_row, const int32_t *__restrict__ cold_lcol,
                                                        const double *__restrict__ cold_x, const double2 *__restrict__ oldnew_g,
                                                        const int32_t *__restrict__ hot_rows, const double2 *__restrict__ hot_pack) {
  constexpr int NT = CHAINB_NT, MC = CHAINB_MAXCOLS, U = 4;
  __shared__ double2 on[MC];
  const int tid = threadIdx.x;
  {
    constexpr int rec2_g = P::REC_DOUBLES / 2;
    const int rec2_global = P::REC_DOUBLES > 2 ? a.rec2 : 1;
    for (int i = (int)blockIdx.x * NT + tid; i < B.n_hot * rec2_g; i += (int)gridDim.x * NT) {
      const int slot = i / rec2_g, w = i - slot * rec2_g;
      ((double2 *)a.state)[(int64_t)hot_rows[B.hot_row0 + slot] * rec2_global + w] = hot_pack[i];
    }
  }
  if (tid < B.ncols) on[tid] = oldnew_g[tid];
  __syncthreads();
  const int cb = B.cold_b, ce = B.cold_e;
  for (int base = cb + (int)blockIdx.x * NT * U + tid; base < ce; base += (int)gridDim.x * NT * U) {
    int lc[U], row[U];
    double xv[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int p = base + u * NT;
      row[u] = -1;
      lc[u] = 0;
      xv[u] = 0.0;
      if (p < ce) {
        lc[u] = cold_lcol[p];
        row[u] = cold_row[p];
        xv[u] = cold_x[p];
      }
    }
    typename P::St st[U];
#pragma unroll
    for (int u = 0; u < U; u++)
      if (row[u] >= 0) st[u] = P::load(a, row[u]);
#pragma unroll
    for (int u = 0; u < U; u++)
      if (row[u] >= 0) P::apply(a, row[u], xv[u], st[u], on[lc[u]].x, on[lc[u]].y);
  }
}

// ---- cold update of batch b and cold statistics of batch b + 1 in ONE launch ----------------------------------------------
// Both touch block-row records only, so the rows are split into CB_BUCKETS contiguous ranges and workgroup w owns range w in
// both halves (the batches' cold entries and hot rows are bucketed by row range on the host): the statistics of batch b + 1
// depend on the update of batch b only through rows of the same range, i.e. through this workgroup -- a workgroup barrier
// replaces the kernel boundary (two launches per batch instead of three). Bp.ncols == 0: nothing to update (first batch);
// Bn.ncols == 0: no statistics to take (after the last batch).
constexpr int CB_BUCKETS = 16;
template <class P>
__global__ __launch_bounds__(CHAINB_NT) void k_cb_step(SweepArgs a, ChainBatch Bp, int ip, ChainBatch Bn, int in_, const int32_t *__restrict__ cols,
                                                       const int32_t *__restrict__ bk_ptr, const int32_t *__restrict__ bk_row,
                                                       const int32_t *__restrict__ bk_lcol, const double *__restrict__ bk_x,
                                                       const int32_t *__restrict__ hbk_ptr, const int32_t *__restrict__ hot_rows,
                                                       const double2 *__restrict__ oldnew_g, double2 *__restrict__ part_g,
                                                       const double2 *__restrict__ hot_pack_p, double2 *__restrict__ hot_pack_n,
                                                       const int32_t *__restrict__ col_group, double *__restrict__ colpack) {
  constexpr int NT = CHAINB_NT, NW = NT / WAVE, MC = CHAINB_MAXCOLS, U = 4, NB = CB_BUCKETS;
  constexpr int rec2_g = P::REC_DOUBLES / 2;
  __shared__ double2 on[MC];
  __shared__ double c_old[MC];
  __shared__ double2 part[MC * NW];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, w = blockIdx.x;
  const int rec2_global = P::REC_DOUBLES > 2 ? a.rec2 : 1;
  if (tid < Bp.ncols) on[tid] = oldnew_g[tid];
  if (tid < Bn.ncols) {
    const int j = cols[Bn.col0 + tid];
    const double th = a.theta[j];
    c_old[tid] = th;
    if (w == 0) {  // the hot walk's per-column scalars, packed: k_cb_hot reads them with one round trip
      const int g = col_group[Bn.col0 + tid];
      colpack[tid] = th;
      colpack[MC + tid] = a.z[j];
      colpack[2 * MC + tid] = a.lambda[g];
      colpack[3 * MC + tid] = a.mu[g];
    }
  }
  for (int i = tid; i < MC * NW; i += NT) part[i] = make_double2(0.0, 0.0);
  // the first round of the statistics half's entries is requested now: it does not depend on the update half, and after the
  // barrier only the record gather is left of that half's dependent loads
  const int cbn = Bn.ncols > 0 ? bk_ptr[in_ * (NB + 1) + w] : 0, cen = Bn.ncols > 0 ? bk_ptr[in_ * (NB + 1) + w + 1] : 0;
  int lc0[U], row0[U];
  double xv0[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int p = cbn + wv * WAVE * U + u * WAVE + lane;
    lc0[u] = -1 - lane;
    row0[u] = -1;
    xv0[u] = 0.0;
    if (p < cen) {
      lc0[u] = bk_lcol[p];
      row0[u] = bk_row[p];
      xv0[u] = bk_x[p];
    }
  }
  __syncthreads();
  if (Bp.ncols > 0) {
    const int hb = hbk_ptr[ip * (NB + 1) + w], he = hbk_ptr[ip * (NB + 1) + w + 1];
    for (int i = hb * rec2_g + tid; i < he * rec2_g; i += NT) {
      const int slot = i / rec2_g, q = i - slot * rec2_g;
      ((double2 *)a.state)[(int64_t)hot_rows[Bp.hot_row0 + slot] * rec2_global + q] = hot_pack_p[i];
    }
    const int cb = bk_ptr[ip * (NB + 1) + w], ce = bk_ptr[ip * (NB + 1) + w + 1];
    for (int base = cb + tid; base < ce; base += NT * U) {
      int lc[U], row[U];
      double xv[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int p = base + u * NT;
        row[u] = -1;
        lc[u] = 0;
        xv[u] = 0.0;
        if (p < ce) {
          lc[u] = bk_lcol[p];
          row[u] = bk_row[p];
          xv[u] = bk_x[p];
        }
      }
      typename P::St st[U];
#pragma unroll
      for (int u = 0; u < U; u++)
        if (row[u] >= 0) st[u] = P::load(a, row[u]);
#pragma unroll
      for (int u = 0; u < U; u++)
        if (row[u] >= 0) P::apply(a, row[u], xv[u], st[u], on[lc[u]].x, on[lc[u]].y);
    }
  }
  __threadfence_block();
  __syncthreads();  // this range's records are up to date for everything below (the only reader of them is this workgroup)
  if (Bn.ncols > 0) {
    const int hb = hbk_ptr[in_ * (NB + 1) + w], he = hbk_ptr[in_ * (NB + 1) + w + 1];
    for (int i = hb * rec2_g + tid; i < he * rec2_g; i += NT) {
      const int slot = i / rec2_g, q = i - slot * rec2_g;
      hot_pack_n[i] = ((const double2 *)a.state)[(int64_t)hot_rows[Bn.hot_row0 + slot] * rec2_global + q];
    }
    const int cb = cbn, ce = cen;
    for (int base = cb + wv * WAVE * U; base < ce; base += NW * WAVE * U) {
      int lc[U], row[U];
      double xv[U];
      const bool first = base == cb + wv * WAVE * U;
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int p = base + u * WAVE + lane;
        lc[u] = first ? lc0[u] : -1 - lane;
        row[u] = first ? row0[u] : -1;
        xv[u] = first ? xv0[u] : 0.0;
        if (!first && p < ce) {
          lc[u] = bk_lcol[p];
          row[u] = bk_row[p];
          xv[u] = bk_x[p];
        }
      }
      typename P::St st[U];
#pragma unroll
      for (int u = 0; u < U; u++)
        if (row[u] >= 0) st[u] = P::load(a, row[u]);
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (base + u * WAVE >= ce) break;  // wave-uniform
        double s1 = 0.0, s2 = 0.0;
        if (row[u] >= 0) P::stats(xv[u], st[u], c_old[lc[u]], s1, s2);
        const int lp = dpp_i32<0x138, 0xf>(lc[u], 0), ln = dpp_i32<0x130, 0xf>(lc[u], 0);
        const bool head = lane == 0 || lp != lc[u], tail = lane == 63 || ln != lc[u];
        int f = head ? 1 : 0;
        wave_segscan2(s1, s2, f);
        if (row[u] >= 0 && tail) {  // one lane per column of this tile; a wave adds its tiles in program order
          double2 &q = part[lc[u] * NW + wv];
          q.x += s1;
          q.y += s2;
        }
      }
    }
    __syncthreads();
    if (tid < MC) {
      double S1 = 0.0, S2 = 0.0;
      if (tid < Bn.ncols)
        for (int k = 0; k < NW; k++) {  // wave order: deterministic
          S1 += part[tid * NW + k].x;
          S2 += part[tid * NW + k].y;
        }
      part_g[(size_t)w * MC + tid] = make_double2(S1, S2);
    }
  }
}

// ---- the whole batch sequence of a chain run as ONE launch (k_cb_persist) ------------------------------------------------------
// k_cb_step and k_cb_hot alternate strictly (step b -> hot b -> step b + 1): 2 launches per batch, and at ~11 columns per batch a
// kernel's start, its prologue round trips and the boundary's cache write-back / invalidate are most of its 16-20 us. Here
// workgroup 0 is the hot walker and workgroups 1 .. CB_BUCKETS own the row ranges for the whole run; what used to be a kernel
// boundary is a counter in device memory. Everything the two sides exchange (the batch's packed hot records, the per-range
// partial statistics, the packed column scalars, the (old, new) pairs) is written with agent-scope stores and read with
// agent-scope loads (write-through / L2-bypassing on gfx950: visible across XCDs without a fence), so no L2 is written back
// or invalidated between batches and a range's block-row records stay in its workgroup's L2. The block-row records themselves
// are only ever touched by their range's workgroup. Same arithmetic, same summation order as the two-launch form.
struct CbSync {
  unsigned long long step_done;  // += 1 per range workgroup and batch step
  unsigned long long pad0[15];
  unsigned long long hot_done;  // batches walked
  unsigned long long pad1[15];
};
__device__ __forceinline__ double2 cb_ld2(const double2 *p) {
  return make_double2(__hip_atomic_load(&p->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                      __hip_atomic_load(&p->y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void cb_st2(double2 *p, double2 v) {
  __hip_atomic_store(&p->x, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&p->y, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double cb_ld(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cb_st(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// thread 0 of a workgroup: wait until *w >= target (bounded: a lost partner sets *error and every later wait falls through)
__device__ __forceinline__ void cb_wait(const unsigned long long *w, unsigned long long target, int *error, bool &dead) {
  if (dead) return;
  unsigned spins = 0;
  while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 1023u) == 0u) {
      if (spins > (1u << 23) || __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        __hip_atomic_store(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dead = true;
        return;
      }
    }
  }
}

struct CbPersistArgs {
  const ChainBatch *batches;
  int n_batches;
  const int32_t *cols, *col_group;
  const int32_t *bk_ptr, *bk_row, *bk_lcol, *hbk_ptr, *hot_rows;
  const int32_t *bk_cls;  // [n_batches][CB_BUCKETS][3]: class boundaries c1, c2, c3 inside a (batch, range)
  const double *bk_x;
  const int32_t *hot_ptr, *hot_slot;
  const double *hot_x;
  int max_hot, max_hot_ent;
  double2 *oldnew_g;  // [MC]
  double2 *part_g;    // [CB_BUCKETS][MC]
  double2 *pack[2];   // packed hot records, by batch parity
  double *colpack;    // [2][4 MC], by batch parity
  CbSync *sync;       // zeroed before the launch
  int *error;
  unsigned long long *prof;  // MFM_CB_PROF: [16] sums of s_memrealtime ticks (100 MHz) of thread 0 of the hot walker (0..4: wait,
                             // stage in, walk, stage out, batches) and of range 0 (8..11: wait, near part, far part, batches)
  int dbg;            // timing experiments only (MFM_CB_DBG; results are wrong when set): 1 no hot walk, 2 no hot-record staging,
                      // 4 no cold statistics, 8 no cold update
};

template <class P>
__global__ __launch_bounds__(CHAINB_NT) void k_cb_persist(SweepArgs a, CbPersistArgs g) {
  extern __shared__ double2 lds_hot[];
  constexpr int NT = CHAINB_NT, NW = NT / WAVE, MC = CHAINB_MAXCOLS, U = 4, NB = CB_BUCKETS;
  constexpr int rec2_g = P::REC_DOUBLES / 2;
  constexpr int rec2_l = P::REC_DOUBLES > 2 ? rec2_g + 1 : rec2_g;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int rec2_global = P::REC_DOUBLES > 2 ? a.rec2 : 1;
  bool dead = false;
  const ChainBatch none{0, 0, 0, 0, 0, 0, 0, 0};
  if (blockIdx.x == 0) {
    // ---- the hot walker -----------------------------------------------------------------------------------------------
    double *c_old = (double *)(lds_hot + (size_t)g.max_hot * rec2_l);
    double *c_z = c_old + MC, *c_lam = c_z + MC, *c_mu = c_lam + MC, *c_new = c_mu + MC;
    double2 *csum = (double2 *)(c_new + MC);  // [MC]
    double *h_x = (double *)(csum + MC);      // [max_hot_ent]
    int *h_slot = (int *)(h_x + g.max_hot_ent);
    int *h_ptr = h_slot + g.max_hot_ent;  // [MC + 1]
    SweepArgs al = a;
    al.state = lds_hot;
    al.rec2 = rec2_l;
    for (int bi = 0; bi < g.n_batches; bi++) {
      const ChainBatch B = g.batches[bi];
      double2 *hot_pack = g.pack[bi & 1];
      const double *colpack = g.colpack + (size_t)(bi & 1) * 4 * MC;
      // static parts of the batch while the range workgroups are still at work
      const int hb0 = B.hot_b, hb1 = B.hot_e;
      for (int i = tid; i < hb1 - hb0; i += NT) {
        h_x[i] = g.hot_x[hb0 + i];
        h_slot[i] = g.hot_slot[hb0 + i];
      }
      if (tid <= B.ncols) h_ptr[tid] = g.hot_ptr[B.col0 + tid] - hb0;
      const bool pf = g.prof && tid == 0;
      unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
      if (pf) t0 = __builtin_amdgcn_s_memrealtime();
      if (tid == 0) cb_wait(&g.sync->step_done, (unsigned long long)NB * (bi + 1), g.error, dead);
      if (pf) t1 = __builtin_amdgcn_s_memrealtime();
      __syncthreads();
      if (tid < B.ncols) {
        c_old[tid] = cb_ld(colpack + tid);
        c_z[tid] = cb_ld(colpack + MC + tid);
        c_lam[tid] = cb_ld(colpack + 2 * MC + tid);
        c_mu[tid] = cb_ld(colpack + 3 * MC + tid);
      }
      for (int i = tid; i < ((g.dbg & 2) ? 0 : B.n_hot * rec2_g); i += NT) {
        const int slot = i / rec2_g, w = i - slot * rec2_g;
        lds_hot[(size_t)slot * rec2_l + w] = cb_ld2(hot_pack + i);
      }
      if (tid < B.ncols) {
        double S1 = 0.0, S2 = 0.0;
        for (int w0 = 0; w0 < NB; w0 += 8) {  // range order: deterministic (eight loads in flight)
          double2 q[8];
#pragma unroll
          for (int u = 0; u < 8; u++) q[u] = cb_ld2(g.part_g + (size_t)(w0 + u) * MC + tid);
#pragma unroll
          for (int u = 0; u < 8; u++) {
            S1 += q[u].x;
            S2 += q[u].y;
          }
        }
        csum[tid] = make_double2(S1, S2);
      }
      __syncthreads();
      if (pf) t2 = __builtin_amdgcn_s_memrealtime();
      if (wv == 0 && !(g.dbg & 1)) {
        for (int c = 0; c < B.ncols; c++) {
          const double S1 = csum[c].x, S2 = csum[c].y;
          const double old = c_old[c];
          const int hb = h_ptr[c], he = h_ptr[c + 1];
          const double fresh = hot_column<P>(a, al, h_x, h_slot, hb, he, lane, S1, S2, old, c_lam[c], c_mu[c], c_z[c]);
          if (lane == 0) c_new[c] = fresh;
          __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        }
      }
      if ((g.dbg & 1) && tid < B.ncols) c_new[tid] = c_old[tid];
      __syncthreads();
      if (pf) t3 = __builtin_amdgcn_s_memrealtime();
      for (int i = tid; i < ((g.dbg & 2) ? 0 : B.n_hot * rec2_g); i += NT) {
        const int slot = i / rec2_g, w = i - slot * rec2_g;
        cb_st2(hot_pack + i, lds_hot[(size_t)slot * rec2_l + w]);
      }
      if (tid < B.ncols) {
        a.theta[g.cols[B.col0 + tid]] = c_new[tid];
        cb_st2(g.oldnew_g + tid, make_double2(c_old[tid], c_new[tid]));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(&g.sync->hot_done, (unsigned long long)(bi + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (pf) {
        t4 = __builtin_amdgcn_s_memrealtime();
        g.prof[0] += t1 - t0;
        g.prof[1] += t2 - t1;
        g.prof[2] += t3 - t2;
        g.prof[3] += t4 - t3;
        g.prof[4] += 1;
      }
    }
    return;
  }
  // ---- a row range -----------------------------------------------------------------------------------------------------------
  // Per batch b the range's cold entries come in four classes (ChainRun::build_batched): the row is / is not touched by batch
  // b - 1 ("statistics near / far") and is / is not touched by batch b + 1 ("update near / far"), stored in the order
  // [far,near | near,near | near,far | far,far] (statistics,update), boundaries c1 c2 c3 in bk_cls. Only the near parts sit
  // between the hot walker's two hand-overs:
  //   wait hot(b - 1) -> update near(b - 1) + hot records back -> statistics near(b) + hot records out -> signal
  //   and, while the hot walker is busy with batch b:  update far(b - 1) -> statistics far(b + 1)
  // (a far row of batch b + 1 is not touched by batch b at all, so its record is final once batch b - 1 is applied).
  const int w = (int)blockIdx.x - 1;
  double2 *on = lds_hot;                         // [MC]
  double *c_oldb = (double *)(on + MC);          // [2][MC], by batch parity
  double2 *part = (double2 *)(c_oldb + 2 * MC);  // [MC * NW]
  for (int i = tid; i < MC * NW; i += NT) part[i] = make_double2(0.0, 0.0);
  if (g.n_batches > 0) {
    const ChainBatch B0 = g.batches[0];
    if (tid < B0.ncols) c_oldb[tid] = a.theta[g.cols[B0.col0 + tid]];
  }
  __syncthreads();
  // statistics of the entries [lo, hi) into part (pre: the first tile of this wave, already in registers)
  auto stats_range = [&](int lo, int hi, const double *c_old, bool pre, const int *lc0, const int *row0, const double *xv0) {
    for (int base = lo + wv * WAVE * U; base < hi; base += NW * WAVE * U) {
      int lc[U], row[U];
      double xv[U];
      const bool first = pre && base == lo + wv * WAVE * U;
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int p = base + u * WAVE + lane;
        lc[u] = first ? lc0[u] : -1 - lane;
        row[u] = first ? row0[u] : -1;
        xv[u] = first ? xv0[u] : 0.0;
        if (!first && p < hi) {
          lc[u] = g.bk_lcol[p];
          row[u] = g.bk_row[p];
          xv[u] = g.bk_x[p];
        }
      }
      typename P::St st[U];
#pragma unroll
      for (int u = 0; u < U; u++)
        if (row[u] >= 0) st[u] = P::load(a, row[u]);
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (base + u * WAVE >= hi) break;  // wave-uniform
        double s1 = 0.0, s2 = 0.0;
        if (row[u] >= 0) P::stats(xv[u], st[u], c_old[lc[u]], s1, s2);
        const int lp = dpp_i32<0x138, 0xf>(lc[u], 0), ln = dpp_i32<0x130, 0xf>(lc[u], 0);
        const bool head = lane == 0 || lp != lc[u], tail = lane == 63 || ln != lc[u];
        int f = head ? 1 : 0;
        wave_segscan2(s1, s2, f);
        if (row[u] >= 0 && tail) {  // one lane per column of this tile; a wave adds its tiles in program order
          double2 &q = part[lc[u] * NW + wv];
          q.x += s1;
          q.y += s2;
        }
      }
    }
  };
  auto update_range = [&](int lo, int hi) {
    for (int base = lo + tid; base < hi; base += NT * U) {
      int lc[U], row[U];
      double xv[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int p = base + u * NT;
        row[u] = -1;
        lc[u] = 0;
        xv[u] = 0.0;
        if (p < hi) {
          lc[u] = g.bk_lcol[p];
          row[u] = g.bk_row[p];
          xv[u] = g.bk_x[p];
        }
      }
      typename P::St st[U];
#pragma unroll
      for (int u = 0; u < U; u++)
        if (row[u] >= 0) st[u] = P::load(a, row[u]);
#pragma unroll
      for (int u = 0; u < U; u++)
        if (row[u] >= 0) P::apply(a, row[u], xv[u], st[u], on[lc[u]].x, on[lc[u]].y);
    }
  };
  for (int bi = 0; bi <= g.n_batches; bi++) {
    const int ip = bi - 1, in_ = bi, iq = bi + 1;
    const ChainBatch Bp = bi > 0 ? g.batches[bi - 1] : none;
    const ChainBatch Bn = bi < g.n_batches ? g.batches[bi] : none;
    const ChainBatch Bq = bi + 1 < g.n_batches ? g.batches[bi + 1] : none;
    const double2 *hot_pack_p = g.pack[(bi + 1) & 1];
    double2 *hot_pack_n = g.pack[bi & 1];
    const double *c_old = c_oldb + (size_t)(bi & 1) * MC;
    if (w == 0 && tid < Bn.ncols) {  // the hot walk's per-column scalars, packed
      const int j = g.cols[Bn.col0 + tid];
      double *colpack = g.colpack + (size_t)(bi & 1) * 4 * MC;
      const int gr = g.col_group[Bn.col0 + tid];
      cb_st(colpack + tid, c_old[tid]);
      cb_st(colpack + MC + tid, a.z[j]);
      cb_st(colpack + 2 * MC + tid, a.lambda[gr]);
      cb_st(colpack + 3 * MC + tid, a.mu[gr]);
    }
    // batch bi: statistics near = [n1, n2) and [n2, n3); the first tile is requested before the wait
    // (two calls: each class is ordered by column, and a wave tile must not hold two runs of one column -- its tail lanes
    // add to the column's LDS partial in one instruction)
    int n1 = 0, n2 = 0, n3 = 0;
    if (Bn.ncols > 0) {
      n1 = g.bk_cls[(in_ * NB + w) * 3 + 0];
      n2 = g.bk_cls[(in_ * NB + w) * 3 + 1];
      n3 = g.bk_cls[(in_ * NB + w) * 3 + 2];
    }
    if (g.dbg & 4) n2 = n3 = n1;
    int lc0[U], row0[U];
    double xv0[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int p = n1 + wv * WAVE * U + u * WAVE + lane;
      lc0[u] = -1 - lane;
      row0[u] = -1;
      xv0[u] = 0.0;
      if (p < n2) {
        lc0[u] = g.bk_lcol[p];
        row0[u] = g.bk_row[p];
        xv0[u] = g.bk_x[p];
      }
    }
    int hbp = 0, hep = 0, p0 = 0, p2 = 0, p4 = 0;
    const bool pfr = g.prof && tid == 0 && w == 0;
    unsigned long long r0 = 0, r1 = 0, r2 = 0;
    if (Bp.ncols > 0) {
      hbp = g.hbk_ptr[ip * (NB + 1) + w];
      hep = g.hbk_ptr[ip * (NB + 1) + w + 1];
      p0 = g.bk_ptr[ip * (NB + 1) + w];
      p4 = g.bk_ptr[ip * (NB + 1) + w + 1];
      p2 = g.bk_cls[(ip * NB + w) * 3 + 1];
      if (g.dbg & 8) p0 = p2 = p4 = 0;
      if (pfr) r0 = __builtin_amdgcn_s_memrealtime();
      if (tid == 0) cb_wait(&g.sync->hot_done, (unsigned long long)bi, g.error, dead);
      if (pfr) r1 = __builtin_amdgcn_s_memrealtime();
    }
    __syncthreads();
    if (tid < Bp.ncols) on[tid] = cb_ld2(g.oldnew_g + tid);
    __syncthreads();
    if (Bp.ncols > 0) {  // update near: the hot records back to their rows, the entries whose row batch bi touches
      for (int i = hbp * rec2_g + tid; i < hep * rec2_g; i += NT) {
        const int slot = i / rec2_g, q = i - slot * rec2_g;
        ((double2 *)a.state)[(int64_t)g.hot_rows[Bp.hot_row0 + slot] * rec2_global + q] = cb_ld2(hot_pack_p + i);
      }
      update_range(p0, p2);
      if (g.dbg & 16) update_range(p2, p4);
    }
    __threadfence_block();
    __syncthreads();  // this range's records are up to date for everything below (the only reader of them is this workgroup)
    if (Bn.ncols > 0) {
      const int hb = g.hbk_ptr[in_ * (NB + 1) + w], he = g.hbk_ptr[in_ * (NB + 1) + w + 1];
      for (int i = hb * rec2_g + tid; i < he * rec2_g; i += NT) {
        const int slot = i / rec2_g, q = i - slot * rec2_g;
        cb_st2(hot_pack_n + i, ((const double2 *)a.state)[(int64_t)g.hot_rows[Bn.hot_row0 + slot] * rec2_global + q]);
      }
      stats_range(n1, n2, c_old, true, lc0, row0, xv0);
      stats_range(n2, n3, c_old, false, nullptr, nullptr, nullptr);
      __syncthreads();
      if (tid < MC) {
        double S1 = 0.0, S2 = 0.0;
        if (tid < Bn.ncols)
          for (int k = 0; k < NW; k++) {  // wave order: deterministic
            S1 += part[tid * NW + k].x;
            S2 += part[tid * NW + k].y;
            part[tid * NW + k] = make_double2(0.0, 0.0);
          }
        cb_st2(g.part_g + (size_t)w * MC + tid, make_double2(S1, S2));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(&g.sync->step_done, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (pfr) r2 = __builtin_amdgcn_s_memrealtime();
    // ---- while the hot walker is busy with batch bi ----
    if (Bp.ncols > 0 && !(g.dbg & 16)) update_range(p2, p4);  // update far: rows batch bi does not touch
    __threadfence_block();
    __syncthreads();
    if (Bq.ncols > 0) {  // statistics far of batch bi + 1: rows batch bi does not touch
      double *c_oldq = c_oldb + (size_t)(iq & 1) * MC;
      if (tid < Bq.ncols) c_oldq[tid] = a.theta[g.cols[Bq.col0 + tid]];
      const int q0 = g.bk_ptr[iq * (NB + 1) + w], q4 = g.bk_ptr[iq * (NB + 1) + w + 1];
      const int q1 = g.bk_cls[(iq * NB + w) * 3 + 0], q3 = g.bk_cls[(iq * NB + w) * 3 + 2];
      __syncthreads();
      if (!(g.dbg & 4)) {
        stats_range(q3, q4, c_oldq, false, nullptr, nullptr, nullptr);
        stats_range(q0, q1, c_oldq, false, nullptr, nullptr, nullptr);
      }
    }
    __syncthreads();
    if (pfr && Bp.ncols > 0 && Bn.ncols > 0) {
      g.prof[8] += r1 - r0;
      g.prof[9] += r2 - r1;
      g.prof[10] += __builtin_amdgcn_s_memrealtime() - r2;
      g.prof[11] += 1;
    }
  }
}

// ---- q-cache build: q = X v_f (+ block contributions)  (FMTrainer.hpp:320-340), CSR SpMV --------
// The first MAX_BLOCKS relation blocks travel in the kernel arguments; a design with more (BaseFMTrainer.hpp:58-68 takes any
// vector of blocks) passes the rest through device arrays (x*: index b - MAX_BLOCKS).
constexpr int MAX_BLOCKS = 16;
struct BlockGatherArgs {
  int n_blocks;
  const int32_t *map[MAX_BLOCKS];
  const double *rec[MAX_BLOCKS];
  int stride[MAX_BLOCKS];  // doubles between two block rows' q_B in rec[b]: BLOCK_REC (the 64-byte records) or 1 (a compact copy)
  const int32_t *const *xmap;
  const double *const *xrec;
  const int *xstride;
  __device__ __forceinline__ double gather(int bi, int64_t i) const {
    if (bi < MAX_BLOCKS) return rec[bi][(int64_t)map[bi][i] * stride[bi]];
    const int x = bi - MAX_BLOCKS;
    return xrec[x][(int64_t)xmap[x][i] * xstride[x]];
  }
};

// thread per row: the right shape for one-hot / few-nnz rows (coalesced over consecutive rows).
// UNIT: all stored values are 1.0 (val not read); ELL >= 0: every row has exactly ELL entries (rowptr
// not read) -- one-hot designs with a fixed number of fields stream 4 bytes per entry.
template <bool UNIT, bool ELL>
__global__ __launch_bounds__(WG) void k_qbuild_rows(const int32_t *__restrict__ rowptr,
                                                    const int32_t *__restrict__ colidx,
                                                    const double *__restrict__ val, const double *__restrict__ vf,
                                                    double2 *__restrict__ eq, int64_t N, int ell, BlockGatherArgs blk,
                                                    const int32_t *__restrict__ map_prev = nullptr,
                                                    const double2 *__restrict__ q_prev = nullptr) {
  const int64_t i = (int64_t)blockIdx.x * WG + threadIdx.x;
  if (i >= N) return;
  int64_t b, e;
  if (ELL) {
    b = i * ell;
    e = b + ell;
  } else {
    b = rowptr[i];
    e = rowptr[i + 1];
  }
  // Every index this row needs is requested first, then every gather, then the sums in the reference's order (main entries,
  // then the blocks in order: bit-identical to the plain loops). With the loops as written -- dynamic trip counts, pointers
  // picked out of the kernel arguments per iteration -- a thread had ONE dependent pair (index -> gather) in flight at a time,
  // six pairs in a row at config 5, and the pass ran at the latency of those round trips (1.43 ms for 2.8 GB at N = 50 M).
  constexpr int NB4 = 4;
  const int nb = blk.n_blocks;
  int mi[NB4];
#pragma unroll
  for (int bi = 0; bi < NB4; bi++) mi[bi] = bi < nb ? blk.map[bi][i] : 0;
  const int mp = map_prev ? map_prev[i] : 0;
  double2 v = make_double2(0.0, 0.0);
  if (map_prev) v = eq[i];
  double s = 0.0;
  if (ELL && ell == 2) {
    const int2 ci = *(const int2 *)(colidx + b);
    double x0 = 1.0, x1 = 1.0;
    if (!UNIT) {
      x0 = val[b];
      x1 = val[b + 1];
    }
    const double v0 = vf[ci.x], v1 = vf[ci.y];
    s += x0 * v0;
    s += x1 * v1;
  } else {
    for (int64_t p = b; p < e; p++) s += (UNIT ? 1.0 : val[p]) * vf[colidx[p]];
  }
  double gq[NB4];
#pragma unroll
  for (int bi = 0; bi < NB4; bi++) gq[bi] = bi < nb ? blk.rec[bi][(int64_t)mi[bi] * blk.stride[bi]] : 0.0;
  double2 qp = make_double2(0.0, 0.0);
  if (map_prev) qp = q_prev[mp];
#pragma unroll
  for (int bi = 0; bi < NB4; bi++)
    if (bi < nb) s += gq[bi];  // :335-337
  for (int bi = NB4; bi < nb; bi++) s += blk.gather(bi, i);
  if (map_prev) {
    // the re-sync the last block of the PREVIOUS factor still owes (FMTrainer.hpp:473-480; its (q_B, q_S) of that factor were
    // saved before the row caches were rebuilt): the residual term uses the old q, which this pass then overwrites
    v.x += (v.y * qp.x + 0.5 * qp.x * qp.x - 0.5 * qp.y);
    v.y = s;
    eq[i] = v;
  } else {
    eq[i].y = s;
  }
}
// wavefront per row: long rows (dense main tables)
__global__ __launch_bounds__(WG) void k_qbuild_wave(const int32_t *__restrict__ rowptr,
                                                    const int32_t *__restrict__ colidx,
                                                    const double *__restrict__ val, const double *__restrict__ vf,
                                                    double2 *__restrict__ eq, int64_t N, BlockGatherArgs blk) {
  const int64_t i = (int64_t)blockIdx.x * (WG / WAVE) + (threadIdx.x >> 6);
  if (i >= N) return;
  const int lane = threadIdx.x & 63;
  const int32_t b = rowptr[i], e = rowptr[i + 1];
  double s = 0.0;
  for (int32_t p = b + lane; p < e; p += WAVE) s += val[p] * vf[colidx[p]];
  s = wave_allreduce_sum(s);
  if (lane == 0) {
    for (int bi = 0; bi < blk.n_blocks; bi++) s += blk.gather(bi, i);
    eq[i].y = s;
  }
}

// ---- fused re-score (update_e, FMTrainer.hpp:494 -> FM.hpp:78-135) ------------------------------
// score_t = w0 + sum_j x w_j + 1/2 sum_f [ (sum_j x v_jf)^2 - sum_j x^2 v_jf^2 ]  (+ block terms)
// One CSR pass instead of the reference's 2K+1 SpMVs: GS lanes cooperate on one row, lane s of the
// group owns factor slots s, s+GS, ...; V rows are gathered from the row-major table Vt.
// Block contributions come pre-reduced per block row (bq: [B][KS] q_B per factor, bl: [B] linear,
// bs: [B] sum_f q_S) and are gathered through original_to_block.
struct BlockScoreArgs {
  int n_blocks;
  const int32_t *map[MAX_BLOCKS];
  const double *bq[MAX_BLOCKS];  // [B][KS]
  const double *bl[MAX_BLOCKS];  // [B]
  const double *bs[MAX_BLOCKS];  // [B]
  // blocks MAX_BLOCKS ..: device arrays, index b - MAX_BLOCKS
  const int32_t *const *xmap;
  const double *const *xbq, *const *xbl, *const *xbs;
};

// OUT_MODE 0: eq[t].x = score - y[t] (y may be null => score)   1: out[t] = score
// Lane `lig` of a GS-lane group owns factor PAIRS (2 lig, 2 lig + 1), (2 (lig + GS), ...): V rows are
// gathered with 16-byte loads (KS is even, rows are 16-byte aligned). Each group works on RU rows at
// a time so that several independent gathers are in flight per lane (the pass is L2-gather bound).
// UNIT: all values are 1.0 (val not read); ELL: every row has `ell` entries (rowptr not read).
constexpr int SCORE_RU = 4;
template <int GS, int SPL, int OUT_MODE, bool UNIT, bool ELL>
__global__ __launch_bounds__(WG) void k_score(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                                              const double *__restrict__ val, const double *__restrict__ Vt,
                                              const double *__restrict__ w, double w0, int K, int KS, int ell,
                                              const double *__restrict__ y, double2 *__restrict__ eq,
                                              double *__restrict__ out, int64_t N, BlockScoreArgs blk) {
  constexpr int RU = SCORE_RU;
  const int lig = threadIdx.x % GS;
  const int64_t t0 = (((int64_t)blockIdx.x * WG + threadIdx.x) / GS) * RU;
  double2 a[RU][SPL];
  double b[RU], lin[RU];
  int64_t pb[RU];
  int len[RU];
  int maxlen = 0;
#pragma unroll
  for (int u = 0; u < RU; u++) {
    b[u] = 0.0;
    lin[u] = 0.0;
#pragma unroll
    for (int k = 0; k < SPL; k++) a[u][k] = make_double2(0.0, 0.0);
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
            const double2 v = row[pr];  // pad column of an odd K is zero
            a[u][s].x += x * v.x;
            a[u][s].y += x * v.y;
            b[u] += x2 * (v.x * v.x);
            b[u] += x2 * (v.y * v.y);
          }
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < RU; u++) {
    const int64_t t = t0 + u;
    if (t < N) {
      for (int bi = 0; bi < blk.n_blocks; bi++) {
        const bool in_args = bi < MAX_BLOCKS;
        const int bx = in_args ? 0 : bi - MAX_BLOCKS;
        const int64_t i = in_args ? blk.map[bi][t] : blk.xmap[bx][t];
        const double *bqp = in_args ? blk.bq[bi] : blk.xbq[bx];
        if (lig == 0) lin[u] += in_args ? blk.bl[bi][i] : blk.xbl[bx][i];
        const double2 *row = (const double2 *)(bqp + i * KS);
#pragma unroll
        for (int s = 0; s < SPL; s++) {
          const int pr = lig + s * GS;
          if (2 * pr < K) {
            const double2 v = row[pr];
            a[u][s].x += v.x;
            a[u][s].y += v.y;
          }
        }
        if (lig == 0) b[u] += in_args ? blk.bs[bi][i] : blk.xbs[bx][i];
      }
    }
  }
#pragma unroll
  for (int u = 0; u < RU; u++) {
    double part = 0.0;
#pragma unroll
    for (int s = 0; s < SPL; s++) part += a[u][s].x * a[u][s].x + a[u][s].y * a[u][s].y;
    part = 0.5 * (part - b[u]) + lin[u];
#pragma unroll
    for (int m = GS / 2; m >= 1; m >>= 1) part += __shfl_xor(part, m, WAVE);
    const int64_t t = t0 + u;
    if (t < N && lig == 0) {
      const double score = w0 + part;
      if (OUT_MODE == 0) {
        eq[t].x = y ? (score - y[t]) : score;
      } else {
        out[t] = score;
      }
    }
  }
}

// Vt[j][s] = V[s][j]  (s < K), row stride KS
__global__ __launch_bounds__(WG) void k_build_vt(const double *__restrict__ V, double *__restrict__ Vt, int64_t D, int K,
                                                 int KS) {
  __shared__ double tile[32][33];
  // 32 x 32 tiles: read coalesced along j, write coalesced along s
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

// ---- reductions over the residual ------------------------------------------------------------------
constexpr int REDUCE_BLOCKS = 1024;
__global__ __launch_bounds__(WG) void k_reduce_e_partial(const double2 *__restrict__ eq, int64_t N,
                                                         double2 *__restrict__ partial) {
  __shared__ double lds[2 * WG / WAVE];
  double s = 0.0, s2 = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * WG + threadIdx.x; i < N; i += (int64_t)gridDim.x * WG) {
    const double e = eq[i].x;
    s += e;
    s2 += e * e;
  }
  wg_allreduce2<WG / WAVE>(s, s2, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = make_double2(s, s2);
}
__global__ __launch_bounds__(WG) void k_reduce_final(const double2 *__restrict__ partial, int n,
                                                     double2 *__restrict__ out) {
  __shared__ double lds[2 * WG / WAVE];
  double s = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < n; i += WG) {
    s += partial[i].x;
    s2 += partial[i].y;
  }
  wg_allreduce2<WG / WAVE>(s, s2, lds);
  if (threadIdx.x == 0) out[0] = make_double2(s, s2);
}

__global__ __launch_bounds__(WG) void k_shift_e(double2 *__restrict__ eq, int64_t N, double delta) {
  const int64_t i = (int64_t)blockIdx.x * WG + threadIdx.x;
  if (i < N) eq[i].x += delta;
}

// group statistics: block (g, f, chunk) reduces theta over a chunk of GS_CHUNK features of group g (sorted
// list); k_group_stats_final adds a (g, f)'s chunk partials in chunk order (deterministic).
// out[(f * G + g)] = { sum theta, sum (theta - mu)^2 }
constexpr int GS_CHUNK = 4096;
__global__ __launch_bounds__(WG) void k_group_stats(const double *__restrict__ theta, int64_t D,
                                                    const int32_t *__restrict__ feat_sorted,
                                                    const int64_t *__restrict__ group_ptr,
                                                    const double *__restrict__ mu, int G, double2 *__restrict__ partial) {
  __shared__ double lds[2 * WG / WAVE];
  const int g = blockIdx.x, f = blockIdx.y, ch = blockIdx.z;
  const double m = mu[f * G + g];
  const double *th = theta + (int64_t)f * D;
  const int64_t b = group_ptr[g] + (int64_t)ch * GS_CHUNK, e = min(group_ptr[g + 1], b + GS_CHUNK);
  double s = 0.0, ss = 0.0;
  constexpr int U = GS_CHUNK / WG;
  double v[U];
#pragma unroll
  for (int r = 0; r < U; r++) {
    const int64_t p = b + threadIdx.x + r * WG;
    v[r] = p < e ? th[feat_sorted[p]] : m;  // (m: contributes 0 to ss; s is masked below)
  }
#pragma unroll
  for (int r = 0; r < U; r++) {
    const int64_t p = b + threadIdx.x + r * WG;
    if (p < e) s += v[r];
    const double d = v[r] - m;
    ss += d * d;
  }
  wg_allreduce2<WG / WAVE>(s, ss, lds);
  if (threadIdx.x == 0) partial[((int64_t)f * G + g) * gridDim.z + ch] = make_double2(s, ss);
}
__global__ void k_group_stats_final(const double2 *__restrict__ partial, int n, int n_chunks, double2 *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0, ss = 0.0;
  for (int c = 0; c < n_chunks; c++) {
    const double2 p = partial[(int64_t)i * n_chunks + c];
    s += p.x;
    ss += p.y;
  }
  out[i] = make_double2(s, ss);
}

// The hyper-parameter conditionals of one regression iteration on the device (mfm_regression_iteration): update_alpha
// (FMTrainer.hpp:127-145), update_w0 (:218-229), update_lambda_w / update_mu_w (:150-200), update_lambda_V / update_mu_V (:202-216)
// in the order of BaseFMTrainer.hpp:137-148, from the statistics the reduction kernels left in `st` and the iteration's unit
// variates `hv` (draw program of the trainer: [gamma: alpha][normal: w0][G gammas: lambda_w][G normals: mu_w][K G gammas:
// lambda_V, factor outer][K G normals: mu_V]). The arithmetic is the host trainer's, operation for operation (a gamma draw is
// the unit variate times the scale, a normal draw `first / quad + z / sqrt(quad)`), so a chain does not depend on where its
// hyper-parameters are drawn. out: [alpha, w0', w0' - w0, -][lambda_w][mu_w][lambda_V][mu_V].
struct HyperPrior {
  double alpha_0, beta_0, gamma_0, mu_0, reg_0, n_total;
  int fit_w0, G, K, pad;
};
__global__ void k_hyper_regression(HyperPrior P, const double2 *__restrict__ st, const double *__restrict__ hv,
                                   const double *__restrict__ n_in_group, const double *__restrict__ w0_old,
                                   double *__restrict__ out) {
  const int G = P.G, GK = P.G * P.K, t = threadIdx.x;
  const int o_w0 = 1, o_lw = o_w0 + (P.fit_w0 ? 1 : 0), o_mw = o_lw + G, o_lv = o_mw + G, o_mv = o_lv + GK;
  double *lam_w = out + 4, *mu_w = lam_w + G, *lam_V = mu_w + G, *mu_V = lam_V + GK;
  __shared__ double alpha_s;
  if (t == 0) {
    const double variance = (P.beta_0 + st[0].y) / 2;  // (exponent (alpha_0 + N) / 2 is the unit variate's shape)
    alpha_s = hv[0] * (1 / variance);
    out[0] = alpha_s;
  }
  for (int i = t; i < G + GK; i += blockDim.x) {  // lambda: gamma((alpha_0 + n_g) / 2, 2 / (beta_0 + ssd))
    const bool isw = i < G;
    const int k = isw ? i : i - G;
    const double beta = P.beta_0 + st[1 + i].y;
    const double lam = hv[(isw ? o_lw : o_lv) + k] * (2 / beta);
    (isw ? lam_w : lam_V)[k] = lam;
  }
  __syncthreads();
  if (t == 0) {
    const double w0 = w0_old[0];
    double w0_new = 0.0, shift = 0.0;
    if (P.fit_w0) {
      const double lin = alpha_s * (P.n_total * w0 - st[0].x);
      const double quad = alpha_s * P.n_total + P.reg_0;
      w0_new = (lin / quad) + hv[o_w0] / sqrt(quad);
      shift = w0_new - w0;
    }
    out[1] = w0_new;
    out[2] = shift;
    out[3] = 0.0;
  }
  for (int i = t; i < G + GK; i += blockDim.x) {  // mu: normal(lam (gamma_0 + n_g), lam (gamma_0 mu_0 + sum))
    const bool isw = i < G;
    const int k = isw ? i : i - G, g = isw ? i : (i - G) % G;
    const double lam = (isw ? lam_w : lam_V)[k];
    const double square = lam * (P.gamma_0 + n_in_group[g]);
    double linear = P.gamma_0 * P.mu_0 + st[1 + i].x;
    linear *= lam;
    (isw ? mu_w : mu_V)[k] = (linear / square) + hv[(isw ? o_mw : o_mv) + k] / sqrt(square);
  }
}

__global__ void k_set_eq(double2 *__restrict__ eq, const double *__restrict__ src, int64_t N, int which) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) {
    if (which == 0)
      eq[i].x = src[i];
    else
      eq[i].y = src[i];
  }
}
__global__ void k_get_eq(const double2 *__restrict__ eq, double *__restrict__ dst, int64_t N, int which) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) dst[i] = which == 0 ? eq[i].x : eq[i].y;
}

}  // namespace mfm
