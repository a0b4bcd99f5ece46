// mfm_res.hpp -- update_V (FMTrainer.hpp:316-376) of a two-field one-hot table as ONE persistent launch with the residual
// resident on chip.
//
// The per-factor passes of mfm_mf_kernels.hpp read and write e[N] once per factor (K + 1 passes per update_V) and hand the
// second level's statistics over through one 16-byte slot per (4096-row tile, item) run. Here a workgroup per CU owns a
// fixed, contiguous range of first-level columns ("users") -- N / #CUs rows -- and keeps e_t of its rows in REGISTERS
// for all K factors: e crosses HBM once per update_V instead of K + 1 times, and the second level ("items") is reduced over
// the workgroup's whole row range before it leaves the CU: one partial per (workgroup, item) instead of one per (tile, item).
//
// Layout. A workgroup's rows are held in ITEM order (sorted by second-level column, then row); thread t owns the R
// consecutive slots [t R, (t + 1) R). Per slot a static 32-bit word (item | user-in-workgroup << item_bits), stored
// [workgroup][r][thread] so that the R loads of a thread are coalesced across the wave.
//   item level (:343-376 for the second field): h_t = v_u'(t) (unit values), a thread sums (-e h, h^2) over its consecutive
//       slots run by run (no cross-lane traffic), runs that continue in the next thread are stitched by one wave-level
//       segmented scan (DPP) + a carry across the waves; one partial per (workgroup, item), stored item-major;
//   user level (first field): h_t = v_i(t); the sums of a user's rows -- scattered over the threads in item order -- go
//       through LDS: every wave owns a private accumulator array (ds_add_f64; lanes of one instruction that hit the same
//       user are serialised by the LDS unit in lane order, a wave's instructions run in program order) and the arrays are
//       added in wave order: every sum has a fixed association, results are bit-reproducible;
//   between the two levels of a factor nothing leaves the CU; between the item statistics and the item draw, and between
//       the draw and the next factor, the workgroups meet at a grid barrier (all payloads are write-through agent-scope
//       stores, one agent-scope acquire per workgroup after the barrier: cdna_hip_programming.md Guideline 16).
// Per factor:  sweep A (apply the previous factor's item update, user statistics) -> user draw (thread per user) ->
//   sweep B (apply the user update, item statistics) -> barrier -> item draw (each workgroup a slice of the items, wave per
//   item over its contiguous partials) -> barrier.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mfm_common.hpp"
#include "mfm_mf_kernels.hpp"

namespace mfm {

struct ResArgs {
  double2 *eq;               // residual: read at [row].x at the start, written back at the end
  const int32_t *perm;       // [G][R][NT] row of the slot, -1: pad
  const uint32_t *uidw;      // [G][R / 2][NT] the slots' users (index within the workgroup), 16 bits each; pads: umax - 1
  const uint32_t *headw;     // [G][R / 16][NT] bit r % 16: slot r starts a new item run (slot 0 of thread 0 always)
  const int32_t *run_item;   // [runs + 1] item of every (workgroup, item) run; pad runs: the pad item n_items
  const int32_t *first_run;  // [G][NT]   run containing the thread's first slot
  const int32_t *wg_user_ptr;  // [G + 1]
  const int2 *user_desc;     // {feature, group}
  const int32_t *wg_item_ptr;  // [G + 1] items (positions in the level) drawn by the workgroup
  const int32_t *ent_ptr;    // [G + 1] the workgroup's slice of `entries`, in entries (a multiple of 64)
  const int2 *entries;       // item-major: {run whose partial belongs to the item, row of the item draw's LDS table}
  const int2 *item_rid;      // per item: {first row of that table, number of rows}
  const int32_t *scols;      // item -> feature
  double *partials;          // [runs + 1][2], workgroup-major: a thread's runs are consecutive; the last stays (0, 0)
  double *dv;                // [items][2]  (delta of this factor, coefficient of the next)
  double *V;                 // factor-major [K][D]
  int64_t D;
  int f_begin, f_end;
  const double *z;           // variates of factor f at z + (f - f_begin) D
  const double *lam, *mu;    // [K][n_groups]
  const int32_t *group;      // per feature
  int n_groups;
  double alpha;
  int item_bits, umax;       // umax: LDS stride of the per-wave user arrays (>= users of any workgroup)
  int rid_max;               // rows of the item draw's LDS table
  unsigned long long *bar;   // monotone arrival counter
  unsigned long long bar_base;  // its value before this launch
  int n_wg;
  int *error;                // set on a spin timeout
  int dbg;                   // timing experiments only (MFM_RES_DBG; results are wrong when set): 4 no grid barriers, 32 no item
                             // draw, 64 no sweep A, 128 no sweep B (whole phases only: a switch inside a sweep's batches breaks
                             // them into basic blocks and serialises their loads)
};

__device__ __forceinline__ void res_store2(double *p, double a, double b) {
  __hip_atomic_store(p, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through (sc1): visible to every XCD
  __hip_atomic_store(p + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Grid barrier #k of the launch (k counted from 1): every workgroup arrives once; payloads were stored write-through and are
// drained by every wave before the arrival; one acquire per workgroup afterwards, then plain loads.
__device__ __forceinline__ void res_grid_barrier(const ResArgs &a, unsigned long long k, int tid, bool &dead) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0 && !dead && !(a.dbg & 4)) {
    // (the partials are plain 16-byte stores -- a thread's runs are adjacent, whole lines form in L2 --: one agent-scope
    //  release writes them back; the asm wait restates the wait the compiler may drop, Guideline 16 pitfall 12)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(a.bar, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long target = a.bar_base + k * (unsigned long long)a.n_wg;
    unsigned spins = 0;
    while (__hip_atomic_load(a.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 1023u) == 0u) {
        if (spins > (1u << 22) || __hip_atomic_load(a.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
          __hip_atomic_store(a.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          dead = true;  // (every later barrier of this workgroup falls through: the launch ends, the host reports)
          break;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// Optimisation fence of the unrolled sweeps: the lane offset and the base pointers of the next batch of loads pass through an
// empty volatile asm, so that the compiler can neither precompute the addresses of all R / B batches (80 SGPR pairs + the
// VGPRs of every hoisted load: the residual spilled) nor move a batch's loads above the previous batch.
__device__ __forceinline__ int res_fence_lane(int t) {
  asm volatile("" : "+v"(t)::"memory");
  return t;
}
typedef double res_d16_t __attribute__((ext_vector_type(16)));
typedef unsigned res_u8_t __attribute__((ext_vector_type(8)));

// NT threads; a thread's slots come in groups of 16: NGV groups live in registers (one 16-double vector each), NGL groups in
// LDS ([16 NGL][NT] doubles: lane-consecutive, conflict-free). Everything static about a group stays in registers for the whole
// launch: 16 head bits (a new item run starts at the slot) and the slots' users, 16 bits each (8 words). The sweeps are REAL
// loops over batches of 4 slots -- the loop counter is wave-uniform, so the batch's residuals and words are picked out of the
// register vectors with s_set_gpr_idx (no scratch, no waterfall) -- and only the group loop is unrolled. (A fully unrolled
// sweep leaves the compiler free to hoist the loads of all 20 batches -- every address is a function of static registers --
// and it then spills the residual; pinning each batch behind the previous one with empty asm statements was tried and lost.)
// The item of a run comes from run_item[run] (4 bytes per (workgroup, item) run, walked sequentially by every thread: cache
// resident); the head bits give the run index of every slot without a memory access, so the run_item loads of batch b + 1 are
// in flight while batch b computes and the only dependent global access of a batch is its dv gather.
template <int NT, int NGV, int NGL>
__global__ __launch_bounds__(NT) void k_mf_resident(ResArgs a) {
  extern __shared__ __attribute__((aligned(16))) char res_smem[];
  constexpr int NW = NT / WAVE, NG = NGV + NGL, R = 16 * NG, RL = 16 * NGL;
  constexpr int B = 4;  // slots per batch
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int U = a.umax;
  double *elds = (double *)res_smem;      // [RL][NT] residual of the LDS-resident slots
  double *acc1 = elds + (size_t)RL * NT;  // [NW][U]  sum of -e h per (wave, user)
  double *acc2 = acc1 + NW * U;           // [NW][U]  sum of h^2
  d2_t *utab = (d2_t *)(acc2 + NW * U);   // [U] {new coefficient, new - old}
  d2_t *wcarry = utab + U;                // [NW]
  d2_t *cpart = wcarry + NW;              // [rid_max] item draw: sums per (64-entry chunk, item)
  int *wflag = (int *)(cpart + a.rid_max);  // [NW]
  const int32_t *perm_g = a.perm + (int64_t)g * R * NT;
  const d2_t *dv2 = (const d2_t *)a.dv;
  const int u0 = a.wg_user_ptr[g], nu = a.wg_user_ptr[g + 1] - u0;
  bool dead = false;
  unsigned long long nbar = 0;

  for (int i = tid; i < 2 * NW * U; i += NT) acc1[i] = 0.0;
  for (int i = tid; i < U; i += NT) utab[i] = d2_t{0.0, 0.0};

  // static per-thread words
  res_u8_t uwv[NG];
  unsigned hbv[NG];
#pragma unroll
  for (int j = 0; j < NG; j++) {
#pragma unroll
    for (int w = 0; w < 8; w++) uwv[j][w] = a.uidw[((int64_t)g * (8 * NG) + 8 * j + w) * NT + tid];
    hbv[j] = a.headw[((int64_t)g * NG + j) * NT + tid];
  }
  const int run0 = a.first_run[g * NT + tid];  // the run containing this thread's first slot
  const bool head0 = (hbv[0] & 1u) != 0u;

  // Pad slots carry (pad item, pad user): an item whose dv entry stays (0, 0) and a user slot nobody draws, so that every slot
  // runs the same straight-line code: a pad's statistics add 0 whatever its residual holds.
  res_d16_t ev[NGV > 0 ? NGV : 1];
#pragma unroll
  for (int j = 0; j < NG; j++) {
#pragma unroll 1
    for (int bb = 0; bb < 4; bb++) {
      int row[B];
#pragma unroll
      for (int k = 0; k < B; k++) row[k] = perm_g[(16 * j + 4 * bb + k) * NT + tid];
#pragma unroll
      for (int k = 0; k < B; k++) {
        const double x = a.eq[row[k] < 0 ? 0 : row[k]].x;
        if (j < NGV)
          ev[j < NGV ? j : 0][4 * bb + k] = x;
        else
          elds[(16 * (j - NGV) + 4 * bb + k) * NT + tid] = x;
      }
    }
  }
  __syncthreads();

  for (int f = a.f_begin; f < a.f_end; f++) {
    double *Vf = a.V + (int64_t)f * a.D;
    const double *zf = a.z + (int64_t)(f - a.f_begin) * a.D;
    // this thread's user: everything its draw needs is requested now
    int uj = 0;
    double uold = 0.0, uz = 0.0, ulam = 0.0, umu = 0.0;
    if (tid < nu) {
      const int2 d = a.user_desc[u0 + tid];
      uj = d.x;
      uold = Vf[uj];
      uz = zf[uj];
      ulam = a.lam[(int64_t)f * a.n_groups + d.y];
      umu = a.mu[(int64_t)f * a.n_groups + d.y];
    }
    // ---- sweep A: the item update of the previous factor (:371-375; dv.x = 0 before the first), the user level's
    //      statistics (:351-356)
    if (!(a.dbg & 64)) {
      // run indices of the first batch (slot 0 belongs to run0 whether or not it is a head) and their items
      int itn[B];
      int run_last = run0;
      {
        const unsigned h4 = hbv[0] & 15u;
#pragma unroll
        for (int k = 0; k < B; k++) {
          if (k > 0) run_last += (int)((h4 >> k) & 1u);
          itn[k] = a.run_item[run_last];
        }
      }
#pragma unroll
      for (int j = 0; j < NG; j++) {
#pragma unroll 1
        for (int bb = 0; bb < 4; bb++) {
          int it[B];
#pragma unroll
          for (int k = 0; k < B; k++) it[k] = itn[k];
          {  // the next batch's items: in flight while this batch computes (past the last batch: the same runs again)
            const unsigned hn = bb < 3 ? (hbv[j] >> (4 * (bb + 1))) & 15u : (j + 1 < NG ? hbv[j + 1 < NG ? j + 1 : j] & 15u : 0u);
#pragma unroll
            for (int k = 0; k < B; k++) {
              run_last += (int)((hn >> k) & 1u);
              itn[k] = a.run_item[run_last];
            }
          }
          d2_t dd[B];
#pragma unroll
          for (int k = 0; k < B; k++) dd[k] = dv2[it[k]];
          const unsigned w0 = uwv[j][2 * bb], w1 = uwv[j][2 * bb + 1];
          const int uid[B] = {(int)(w0 & 0xffffu), (int)(w0 >> 16), (int)(w1 & 0xffffu), (int)(w1 >> 16)};
          double up[B], ex[B];
#pragma unroll
          for (int k = 0; k < B; k++) {
            up[k] = utab[uid[k]][0];
            ex[k] = j < NGV ? ev[j < NGV ? j : 0][4 * bb + k] : elds[(16 * (j - NGV) + 4 * bb + k) * NT + tid];
          }
#pragma unroll
          for (int k = 0; k < B; k++) {
            const double er = ex[k] + up[k] * dd[k][0];
            if (j < NGV)
              ev[j < NGV ? j : 0][4 * bb + k] = er;
            else
              elds[(16 * (j - NGV) + 4 * bb + k) * NT + tid] = er;
            const double c = dd[k][1];
            __hip_atomic_fetch_add(&acc1[wv * U + uid[k]], (-er) * c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&acc2[wv * U + uid[k]], c * c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    }
    __syncthreads();
    // ---- user draw (:357-369): thread u sums user u's wave accumulators in wave order
    if (tid < nu) {
      double S1 = 0.0, S2 = 0.0;
#pragma unroll
      for (int w = 0; w < NW; w++) {
        S1 += acc1[w * U + tid];
        S2 += acc2[w * U + tid];
        acc1[w * U + tid] = 0.0;
        acc2[w * U + tid] = 0.0;
      }
      const double fresh = PMainV::draw(S1, S2, uold, a.alpha, ulam, umu, uz);
      Vf[uj] = fresh;
      utab[tid] = d2_t{fresh, fresh - uold};
    }
    __syncthreads();
    // ---- sweep B: the user update (:371-375), the item level's statistics run by run
    {
      bool have_head = false;
      double f1 = 0.0, f2 = 0.0, s1 = 0.0, s2 = 0.0;
      d2_t *part2 = (d2_t *)a.partials;
      if (!(a.dbg & 128)) {
        int itn[B], rkn[B];
        int run_last = run0;
        {
          const unsigned h4 = hbv[0] & 15u;
#pragma unroll
          for (int k = 0; k < B; k++) {
            if (k > 0) run_last += (int)((h4 >> k) & 1u);
            rkn[k] = run_last;
            itn[k] = a.run_item[run_last];
          }
        }
#pragma unroll
        for (int j = 0; j < NG; j++) {
#pragma unroll 1
          for (int bb = 0; bb < 4; bb++) {
            int it[B], rk[B];
#pragma unroll
            for (int k = 0; k < B; k++) {
              it[k] = itn[k];
              rk[k] = rkn[k];
            }
            const unsigned h4 = (hbv[j] >> (4 * bb)) & 15u;
            {
              const unsigned hn = bb < 3 ? (hbv[j] >> (4 * (bb + 1))) & 15u : (j + 1 < NG ? hbv[j + 1 < NG ? j + 1 : j] & 15u : 0u);
#pragma unroll
              for (int k = 0; k < B; k++) {
                run_last += (int)((hn >> k) & 1u);
                rkn[k] = run_last;
                itn[k] = a.run_item[run_last];
              }
            }
            double cc[B];
#pragma unroll
            for (int k = 0; k < B; k++) cc[k] = a.dv[2 * (int64_t)it[k] + 1];
            const unsigned w0 = uwv[j][2 * bb], w1 = uwv[j][2 * bb + 1];
            const int uid[B] = {(int)(w0 & 0xffffu), (int)(w0 >> 16), (int)(w1 & 0xffffu), (int)(w1 >> 16)};
            d2_t ut[B];
            double ex[B];
#pragma unroll
            for (int k = 0; k < B; k++) {
              ut[k] = utab[uid[k]];
              ex[k] = j < NGV ? ev[j < NGV ? j : 0][4 * bb + k] : elds[(16 * (j - NGV) + 4 * bb + k) * NT + tid];
            }
#pragma unroll
            for (int k = 0; k < B; k++) {
              const bool head = ((h4 >> k) & 1u) != 0u;
              // a run that began and ended in this thread: the slot before this head closed run rk[k] - 1
              if (head && have_head) part2[rk[k] - 1] = d2_t{s1, s2};
              f1 = head && !have_head ? s1 : f1;
              f2 = head && !have_head ? s2 : f2;
              have_head = have_head || head;
              const double er = ex[k] + cc[k] * ut[k][1];
              if (j < NGV)
                ev[j < NGV ? j : 0][4 * bb + k] = er;
              else
                elds[(16 * (j - NGV) + 4 * bb + k) * NT + tid] = er;
              s1 = (head ? 0.0 : s1) + (-er) * ut[k][0];
              s2 = (head ? 0.0 : s2) + ut[k][0] * ut[k][0];
            }
          }
        }
      }
      // stitch the runs that cross thread boundaries: a thread with a head restarts the running sum with its open tail,
      // a thread without one passes its whole sum on
      double v1 = s1, v2 = s2;
      int fl = have_head ? 1 : 0;
      wave_segscan2(v1, v2, fl);
      if (lane == 63) {
        wcarry[wv] = d2_t{v1, v2};
        wflag[wv] = fl;
      }
      // exclusive: what the lanes before this one carry
      double x1 = dpp_f64<0x138, 0xf>(v1), x2 = dpp_f64<0x138, 0xf>(v2);  // wave_shr:1
      int xf = dpp_i32<0x138, 0xf>(fl, 0);
      if (lane == 0) {
        x1 = 0.0;
        x2 = 0.0;
        xf = 0;
      }
      lds_barrier();
      if (have_head && tid > 0) {
        if (!xf) {  // no head in the earlier lanes of this wave: the carry of the waves before it
          double c1 = 0.0, c2 = 0.0;
          for (int w = 0; w < wv; w++) {  // wave order: deterministic
            const d2_t cw = wcarry[w];
            const bool fw = wflag[w] != 0;
            c1 = (fw ? 0.0 : c1) + cw[0];
            c2 = (fw ? 0.0 : c2) + cw[1];
          }
          x1 = c1 + x1;
          x2 = c2 + x2;
        }
        // the run that ends at this thread's first head: run0 when the first slot is not a head, else the one before
        const int closing = head0 ? run0 - 1 : run0;
        part2[closing] = d2_t{x1 + f1, x2 + f2};
      }
    }
    res_grid_barrier(a, ++nbar, tid, dead);
    // ---- item draw (:357-369). The workgroup's items own a contiguous, item-major slice of `entries`; a wave takes 64-entry
    //      chunks of it (the partials are gathered, 4 chunks in flight), a DPP segmented scan sums each chunk's stretch of
    //      one item, the stretch's last lane leaves the sum in the LDS table at a precomputed row; then a thread per item adds
    //      the item's rows in order (fixed association) and draws -- every item of the slice in parallel.
    if (!(a.dbg & 32)) {
      constexpr int CH = 4;
      const int c0 = a.ent_ptr[g] >> 6, c1 = a.ent_ptr[g + 1] >> 6;
      const d2_t *part2 = (const d2_t *)a.partials;
      for (int cb = c0 + wv * CH; cb < c1; cb += NW * CH) {
        int2 en[CH];
#pragma unroll
        for (int k = 0; k < CH; k++) en[k] = cb + k < c1 ? a.entries[(int64_t)(cb + k) * WAVE + lane] : make_int2(0, -1 - lane);
        d2_t sv[CH];
#pragma unroll
        for (int k = 0; k < CH; k++) sv[k] = cb + k < c1 ? part2[en[k].x] : d2_t{0.0, 0.0};
#pragma unroll
        for (int k = 0; k < CH; k++) {
          if (cb + k >= c1) break;  // wave-uniform
          const int rid = en[k].y;
          const int rp = dpp_i32<0x138, 0xf>(rid, 0), rn = dpp_i32<0x130, 0xf>(rid, 0);  // wave_shr:1 / wave_shl:1
          int hd = (lane == 0 || rp != rid) ? 1 : 0;
          const bool tail = lane == 63 || rn != rid;
          double s1 = sv[k][0], s2 = sv[k][1];
          wave_segscan2(s1, s2, hd);
          if (tail) cpart[rid] = d2_t{s1, s2};
        }
      }
      lds_barrier();
      const int i0 = a.wg_item_ptr[g], i1 = a.wg_item_ptr[g + 1];
      const bool more = f + 1 < a.f_end;
      for (int i = i0 + tid; i < i1; i += NT) {
        const int2 rr = a.item_rid[i];
        const int j = a.scols[i];
        const int gj = a.group[j];
        const double old = Vf[j], zj = zf[j], vn = more ? a.V[(int64_t)(f + 1) * a.D + j] : 0.0;
        const double lj = a.lam[(int64_t)f * a.n_groups + gj], mj = a.mu[(int64_t)f * a.n_groups + gj];
        double S1 = 0.0, S2 = 0.0;
        for (int r = rr.x; r < rr.x + rr.y; r++) {
          const d2_t cp = cpart[r];
          S1 += cp[0];
          S2 += cp[1];
        }
        const double fresh = PMainV::draw(S1, S2, old, a.alpha, lj, mj, zj);
        Vf[j] = fresh;
        res_store2(a.dv + 2 * (int64_t)i, fresh - old, vn);
      }
    }
    res_grid_barrier(a, ++nbar, tid, dead);
  }
  // the last factor's item update, then the residual goes back
  {
    int run_last = run0;
#pragma unroll
    for (int j = 0; j < NG; j++) {
#pragma unroll 1
      for (int bb = 0; bb < 4; bb++) {
        const unsigned h4 = (hbv[j] >> (4 * bb)) & 15u;
        int row[B], it[B];
#pragma unroll
        for (int k = 0; k < B; k++) {
          if (16 * j + 4 * bb + k > 0) run_last += (int)((h4 >> k) & 1u);
          it[k] = a.run_item[run_last];
          row[k] = perm_g[(16 * j + 4 * bb + k) * NT + tid];
        }
        const unsigned w0 = uwv[j][2 * bb], w1 = uwv[j][2 * bb + 1];
        const int uid[B] = {(int)(w0 & 0xffffu), (int)(w0 >> 16), (int)(w1 & 0xffffu), (int)(w1 >> 16)};
#pragma unroll
        for (int k = 0; k < B; k++) {
          const double dl = a.dv[2 * (int64_t)it[k]];
          const double ex = j < NGV ? ev[j < NGV ? j : 0][4 * bb + k] : elds[(16 * (j - NGV) + 4 * bb + k) * NT + tid];
          if (row[k] >= 0) a.eq[row[k]].x = ex + utab[uid[k]][0] * dl;
        }
      }
    }
  }
}

// ---- host side: the resident layout of a two-field table and the launch ---------------------------------------------
__global__ void k_res_init_dv(const double *__restrict__ theta, const int32_t *__restrict__ scols, int n_items,
                              double *__restrict__ dv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_items) {
    dv[2 * i] = 0.0;
    dv[2 * i + 1] = theta[scols[i]];
  } else if (i == n_items) {  // the pad item: (0, 0) for ever
    dv[2 * i] = 0.0;
    dv[2 * i + 1] = 0.0;
  }
}

struct ResPlan {
  bool ready = false;
  int G = 0, NT = 512, RV = 0, RL = 0, umax = 0, item_bits = 0, n_items = 0;
  int64_t n_rows = 0, n_runs = 0;  // (workgroup, item) pairs
  size_t lds_bytes = 0;
  DevBuf<int32_t> perm, first_run, wg_user_ptr, wg_item_ptr, ent_ptr, scols;
  DevBuf<int2> entries, item_rid;
  int rid_max = 0;
  DevBuf<uint32_t> uidw, headw;
  DevBuf<int32_t> run_item;
  DevBuf<int2> user_desc;
  DevBuf<double> partials, dv;
  DevBuf<unsigned long long> bar;
  unsigned long long bar_count = 0;
  std::string why;  // why the layout was not built (diagnostics)

  static int bits_for(int64_t n) {  // bits that hold the values 0 .. n - 1
    int b = 1;
    while (((int64_t)1 << b) < n) b++;
    return b;
  }
  // (NT, RV, RL) variants compiled below, by rows per workgroup
  struct Variant {
    int rv, rl;
  };
  static const Variant *variants(int &n) {
    static const Variant v[] = {{16, 0}, {32, 0}, {64, 16}};
    n = 3;
    return v;
  }

  // csc = X_t of the table (column j: ascending rows), level[j] in {0, 1}: 0 = first field (contiguous row ranges in
  // ascending order covering every row once), 1 = second field (every row once). Unit values.
  bool build(const HostCsr &csc, const std::vector<int32_t> &level, const std::vector<int32_t> *group_of, int n_cu) {
    ready = false;
    const int64_t N = csc.cols, D0 = csc.rows;
    n_rows = N;
    if (N < 1 || n_cu < 1 || (int64_t)level.size() != D0) return fail("shape");
    std::vector<int32_t> users, items, empties;
    for (int64_t j = 0; j < D0; j++) {
      if (level[j] == 0) {
        (csc.ptr[j + 1] > csc.ptr[j] ? users : empties).push_back((int32_t)j);
      } else if (level[j] == 1) {
        items.push_back((int32_t)j);
      } else {
        return fail("more than two levels");
      }
    }
    n_items = (int)items.size();
    if (users.empty() || items.empty()) return fail("empty level");
    std::sort(users.begin(), users.end(), [&](int32_t x, int32_t y) { return csc.idx[csc.ptr[x]] < csc.idx[csc.ptr[y]]; });
    // user boundaries (rows), checked contiguous and covering
    std::vector<int64_t> ustart(users.size() + 1);
    int64_t next_row = 0, max_user = 0;
    for (size_t u = 0; u < users.size(); u++) {
      const int32_t j = users[u];
      const int64_t b = csc.ptr[j], len = csc.ptr[j + 1] - b;
      if (csc.idx[b] != next_row || csc.idx[b + len - 1] != next_row + len - 1) return fail("first level not contiguous");
      ustart[u] = next_row;
      next_row += len;
      max_user = std::max(max_user, len);
    }
    ustart[users.size()] = next_row;
    if (next_row != N) return fail("first level does not cover the rows");
    std::vector<int32_t> item_of_row((size_t)N, -1);
    {
      int64_t cnt = 0;
      for (int c = 0; c < n_items; c++)
        for (int64_t p = csc.ptr[items[c]]; p < csc.ptr[items[c] + 1]; p++) {
          if (item_of_row[csc.idx[p]] >= 0) return fail("second level touches a row twice");
          item_of_row[csc.idx[p]] = c;
          cnt++;
        }
      if (cnt != N) return fail("second level does not cover the rows");
    }
    // workgroups: contiguous user ranges of at most cap rows; the smallest variant that fits the device
    int nv = 0;
    const Variant *vs = variants(nv);
    std::vector<int64_t> ucut;  // user ordinal boundaries of the workgroups
    bool found = false;
    for (int vi = 0; vi < nv && !found; vi++) {
      const int64_t cap = (int64_t)NT * (vs[vi].rv + vs[vi].rl) - 1;  // (at least one pad slot closes the last run)
      if (max_user > cap) continue;
      const int64_t slack = std::min<int64_t>(cap / 8, max_user);
      int64_t Gw = (N + (cap - slack) - 1) / (cap - slack);
      if (Gw > n_cu && N <= (int64_t)n_cu * cap) Gw = n_cu;  // (tight: the cuts below decide whether it fits)
      if (Gw > n_cu) continue;
      if (const char *e = std::getenv("MFM_RES_WGS")) Gw = std::min<int64_t>(n_cu, std::max<int64_t>(Gw, std::atoll(e)));
      Gw = std::min<int64_t>(Gw, (int64_t)users.size());
      // cut g at the user boundary nearest to g N / G, never beyond cap rows
      ucut.assign(1, 0);
      bool ok = true;
      for (int64_t g = 1; g <= Gw && ok; g++) {
        const int64_t lo_u = ucut.back();
        int64_t hi_u;
        if (g == Gw) {
          hi_u = (int64_t)users.size();
        } else {
          const int64_t want = (N * g) / Gw;
          hi_u = std::upper_bound(ustart.begin(), ustart.end(), want) - ustart.begin() - 1;  // last boundary <= want
          if (hi_u + 1 <= (int64_t)users.size() && ustart[hi_u + 1] - want < want - ustart[hi_u]) hi_u++;
          hi_u = std::max(hi_u, lo_u + 1);
          // leave at least one user for each of the remaining workgroups
          hi_u = std::min<int64_t>(hi_u, (int64_t)users.size() - (Gw - g));
          while (hi_u > lo_u + 1 && ustart[hi_u] - ustart[lo_u] > cap) hi_u--;
        }
        if (hi_u <= lo_u || ustart[hi_u] - ustart[lo_u] > cap) ok = false;
        ucut.push_back(hi_u);
      }
      if (!ok) continue;
      G = (int)Gw;
      RV = vs[vi].rv;
      RL = vs[vi].rl;
      found = true;
    }
    if (!found) return fail("no variant fits (rows per CU, or a first-level column longer than a workgroup's capacity)");
    const int R = RV + RL;
    const int64_t cap_slots = (int64_t)NT * R;
    // users per workgroup (+ the never-occurring ones, dealt round-robin), the pad user
    std::vector<std::vector<int32_t>> wg_users((size_t)G);
    for (int g = 0; g < G; g++)
      for (int64_t u = ucut[g]; u < ucut[g + 1]; u++) wg_users[g].push_back(users[u]);
    for (size_t k = 0; k < empties.size(); k++) wg_users[k % G].push_back(empties[k]);
    int maxu = 0;
    for (int g = 0; g < G; g++) maxu = std::max(maxu, (int)wg_users[g].size());
    if (maxu > NT) return fail("more first-level columns in a workgroup than threads");
    umax = maxu + 1;
    item_bits = bits_for((int64_t)n_items + 1);
    // slots: per workgroup the rows in (item, row) order
    const int UW = R / 2, HW = R / 16;  // users 16 bits each, head bits 16 per word (one word per group of 16 slots)
    std::vector<uint32_t> h_uidw((size_t)G * UW * NT, (uint32_t)(umax - 1) * 0x10001u);  // every slot starts as a pad
    std::vector<uint32_t> h_headw((size_t)G * HW * NT, 0);
    std::vector<int32_t> h_perm((size_t)G * cap_slots, -1), h_first((size_t)G * NT, 0);
    std::vector<std::vector<int32_t>> wg_run_item((size_t)G);
    std::vector<int32_t> uord_of_row((size_t)N), wg_of_user(users.size());
    for (int g = 0; g < G; g++)
      for (int64_t u = ucut[g]; u < ucut[g + 1]; u++) wg_of_user[u] = g;
    for (size_t u = 0; u < users.size(); u++)
      for (int64_t r = ustart[u]; r < ustart[u + 1]; r++) uord_of_row[r] = (int32_t)u;
    std::vector<int64_t> fill((size_t)G, 0);
    std::vector<int32_t> last_item((size_t)G, -1), nruns((size_t)G, 0);
    std::vector<int32_t> pair_g, pair_local;  // the (workgroup, item) runs in item-major order
    std::vector<int32_t> h_slot_ptr((size_t)n_items + 1, 0);
    int64_t counter = 0;
    for (int c = 0; c < n_items; c++) {
      h_slot_ptr[c] = (int32_t)counter;
      for (int64_t p = csc.ptr[items[c]]; p < csc.ptr[items[c] + 1]; p++) {
        const int32_t row = csc.idx[p];
        const int32_t uo = uord_of_row[row];
        const int g = wg_of_user[uo];
        const int64_t sidx = fill[g]++;
        if (last_item[g] != c) {
          last_item[g] = c;
          pair_g.push_back(g);
          pair_local.push_back(nruns[g]);
          wg_run_item[g].push_back(c);
          counter++;
          nruns[g]++;
          const int th = (int)(sidx / R), rh = (int)(sidx % R);
          h_headw[((size_t)g * HW + rh / 16) * NT + th] |= 1u << (rh % 16);
        }
        const int t = (int)(sidx / R), r = (int)(sidx % R);
        h_perm[(size_t)g * cap_slots + (size_t)r * NT + t] = row;
        uint32_t &uw = h_uidw[((size_t)g * UW + r / 2) * NT + t];
        const int sh = 16 * (r % 2);
        uw = (uw & ~(0xffffu << sh)) | ((uint32_t)(uo - (int32_t)ucut[g]) << sh);
        if (r == 0) h_first[(size_t)g * NT + t] = nruns[g] - 1;
      }
    }
    h_slot_ptr[n_items] = (int32_t)counter;
    n_runs = counter;
    if (counter >= ((int64_t)1 << 31) - 2) return fail("too many runs");
    // runs, workgroup-major: global run index = run_base[g] + local; every workgroup ends with its pad run (never stored)
    std::vector<int32_t> run_base((size_t)G + 1, 0);
    for (int g = 0; g < G; g++) run_base[g + 1] = run_base[g] + nruns[g] + 1;
    const int32_t zero_run = run_base[G];  // a partial that stays (0, 0): what the padding entries of the item draw gather
    std::vector<int32_t> h_run_item((size_t)zero_run + 1, n_items);
    for (int g = 0; g < G; g++) {
      std::copy(wg_run_item[g].begin(), wg_run_item[g].end(), h_run_item.begin() + run_base[g]);
      // the first pad slot opens the pad run (item n_items); thread 0's first slot is a head whatever it holds
      if (fill[g] < cap_slots) {
        const int th = (int)(fill[g] / R), rh = (int)(fill[g] % R);
        h_headw[((size_t)g * HW + rh / 16) * NT + th] |= 1u << (rh % 16);
      }
      h_headw[((size_t)g * HW) * NT] |= 1u;
      // threads whose first slot is a pad: the pad run
      for (int t = 0; t < NT; t++) {
        const int64_t s0 = (int64_t)t * R;
        int32_t &fr = h_first[(size_t)g * NT + t];
        fr = (s0 >= fill[g] ? nruns[g] : fr) + run_base[g];
      }
    }
    // users
    std::vector<int32_t> h_uptr((size_t)G + 1, 0);
    std::vector<int2> h_udesc;
    for (int g = 0; g < G; g++) {
      for (int32_t j : wg_users[g]) h_udesc.push_back(make_int2(j, group_of && (size_t)j < group_of->size() ? (*group_of)[j] : 0));
      h_uptr[g + 1] = (int32_t)h_udesc.size();
    }
    // item draw: contiguous item ranges of about equal cost (partials + a constant per item)
    std::vector<int32_t> h_iptr((size_t)G + 1, 0);
    {
      const double total = (double)counter + 16.0 * n_items;
      double acc = 0.0;
      int g = 1;
      for (int c = 0; c < n_items && g < G; c++) {
        acc += (double)(h_slot_ptr[c + 1] - h_slot_ptr[c]) + 16.0;
        while (g < G && acc >= total * g / G) h_iptr[g++] = c + 1;
      }
      for (; g <= G; g++) h_iptr[g] = n_items;
      h_iptr[G] = n_items;
    }
    // item draw tables: per workgroup the item-major entries of its items, padded to whole 64-entry chunks; the (chunk, item)
    // stretches of a slice are numbered chunk + item (both only grow along the slice: distinct stretches, distinct rows)
    std::vector<int2> h_entries, h_item_rid((size_t)n_items, make_int2(0, 0));
    std::vector<int32_t> h_eptr((size_t)G + 1, 0);
    rid_max = 1;
    for (int g = 0; g < G; g++) {
      const size_t e_begin = h_entries.size();
      int last_rid = 0;
      for (int c = h_iptr[g]; c < h_iptr[g + 1]; c++) {
        const int li = c - h_iptr[g];
        int first = -1, last = -1;
        for (int32_t q = h_slot_ptr[c]; q < h_slot_ptr[c + 1]; q++) {
          const int k = (int)(h_entries.size() - e_begin);
          const int rid = (k >> 6) + li;
          if (first < 0) first = rid;
          last = rid;
          h_entries.push_back(make_int2(run_base[pair_g[q]] + pair_local[q], rid));
        }
        if (first >= 0) {
          h_item_rid[c] = make_int2(first, last - first + 1);
          last_rid = last;
        }
      }
      while ((h_entries.size() - e_begin) % WAVE) h_entries.push_back(make_int2(zero_run, last_rid));
      h_eptr[g + 1] = (int32_t)h_entries.size();
      rid_max = std::max(rid_max, last_rid + 1);
    }
    if (h_entries.size() >= ((size_t)1 << 31)) return fail("too many entries");
    lds_bytes = (size_t)RL * NT * 8 + (size_t)2 * (NT / WAVE) * umax * 8 + (size_t)umax * 16 + (size_t)(NT / WAVE) * 16 + (size_t)rid_max * 16 +
                (NT / WAVE) * 4 + 64;
    if (lds_bytes > 160 * 1024 - 512) return fail("LDS");
    uidw.upload(h_uidw);
    headw.upload(h_headw);
    run_item.upload(h_run_item);
    perm.upload(h_perm);
    first_run.upload(h_first);
    wg_user_ptr.upload(h_uptr);
    user_desc.upload(h_udesc.data(), h_udesc.size());
    wg_item_ptr.upload(h_iptr);
    ent_ptr.upload(h_eptr);
    entries.upload(h_entries.data(), h_entries.size());
    item_rid.upload(h_item_rid.data(), h_item_rid.size());
    scols.upload(items);
    partials.alloc((size_t)2 * ((size_t)zero_run + 1));
    MFM_HIP_CHECK(hipMemset(partials.p, 0, (size_t)16 * ((size_t)zero_run + 1)));
    dv.alloc((size_t)2 * (n_items + 1));
    bar.alloc(1);
    MFM_HIP_CHECK(hipMemset(bar.p, 0, sizeof(unsigned long long)));
    bar_count = 0;
    ready = true;
    why.clear();
    return true;
  }
  bool fail(const char *w) {
    why = w;
    ready = false;
    return false;
  }
};

// update_V of factors [f_begin, f_end) in one launch. zbase: variates of factor f_begin (factor f at + (f - f_begin) D).
static inline void run_sweep_resident(hipStream_t s, Timing &tm, ResPlan &rp, int kernel_class, double2 *eq, double *V, int64_t D,
                                      int f_begin, int f_end, const double *zbase, const double *lam, const double *mu,
                                      const int32_t *group, int n_groups, double alpha, int *error) {
  ResArgs a;
  std::memset(&a, 0, sizeof(a));
  a.eq = eq;
  a.perm = rp.perm.p;
  a.uidw = rp.uidw.p;
  a.headw = rp.headw.p;
  a.run_item = rp.run_item.p;
  a.first_run = rp.first_run.p;
  a.wg_user_ptr = rp.wg_user_ptr.p;
  a.user_desc = rp.user_desc.p;
  a.wg_item_ptr = rp.wg_item_ptr.p;
  a.ent_ptr = rp.ent_ptr.p;
  a.entries = rp.entries.p;
  a.item_rid = rp.item_rid.p;
  a.rid_max = rp.rid_max;
  a.scols = rp.scols.p;
  a.partials = rp.partials.p;
  a.dv = rp.dv.p;
  a.V = V;
  a.D = D;
  a.f_begin = f_begin;
  a.f_end = f_end;
  a.z = zbase;
  a.lam = lam;
  a.mu = mu;
  a.group = group;
  a.n_groups = n_groups;
  a.alpha = alpha;
  a.item_bits = rp.item_bits;
  a.umax = rp.umax;
  a.bar = rp.bar.p;
  a.bar_base = rp.bar_count;
  a.n_wg = rp.G;
  a.error = error;
  a.dbg = std::getenv("MFM_RES_DBG") ? std::atoi(std::getenv("MFM_RES_DBG")) : 0;
  rp.bar_count += 2ull * (unsigned long long)(f_end - f_begin) * (unsigned long long)rp.G;
  const int K = f_end - f_begin;
  // algorithmic bytes of the launch: e read + written once, the per-slot words twice per factor (+ once at either end),
  // one 16-byte partial per (workgroup, item) written and read per factor
  const double bytes = 16.0 * rp.n_rows + 8.0 * rp.n_rows + K * (8.0 * rp.n_rows + 32.0 * rp.n_runs);
  hipLaunchKernelGGL(k_res_init_dv, dim3((rp.n_items + 256) / 256), dim3(256), 0, s, V + (int64_t)f_begin * D, rp.scols.p, rp.n_items,
                     rp.dv.p);
  TimedLaunch t(tm, s, kernel_class, bytes);
#define MFM_RES_LAUNCH(RV_, RL_)                                                                                              \
  do {                                                                                                                        \
    static DeviceOnce raised;                                                                                                 \
    if (raised.need()) {                                                                                                      \
      MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_mf_resident<512, RV_ / 16, RL_ / 16>,                                  \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                             \
      raised.mark();                                                                                                          \
    }                                                                                                                         \
    hipLaunchKernelGGL((k_mf_resident<512, RV_ / 16, RL_ / 16>), dim3(rp.G), dim3(512), rp.lds_bytes, s, a);                    \
  } while (0)
  if (rp.RV == 16 && rp.RL == 0)
    MFM_RES_LAUNCH(16, 0);
  else if (rp.RV == 32 && rp.RL == 0)
    MFM_RES_LAUNCH(32, 0);
  else if (rp.RV == 64 && rp.RL == 16)
    MFM_RES_LAUNCH(64, 16);
  else
    throw Error(MFM_ERR_RUNTIME, "internal: no resident kernel variant for this plan");
#undef MFM_RES_LAUNCH
  MFM_HIP_CHECK(hipGetLastError());
}

}  // namespace mfm
