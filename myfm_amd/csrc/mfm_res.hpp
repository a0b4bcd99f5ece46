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
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "mfm_common.hpp"
#include "mfm_mf_kernels.hpp"

namespace mfm {

#ifndef MFM_RES_CH
#define MFM_RES_CH 4
#endif
constexpr int RES_MAX_PEERS = 8;

struct ResArgs {
  double2 *eq;               // residual: read at [row].x at the start, written back at the end ...
  const double *e_in;        // set: the residual is read from this slot-ordered copy (what the previous launch or the slot-order
                             // scorer k_res_score left) instead of eq
  double *e_slots;           // ... unless this is set: [G][R][NT], the residual in slot order (coalesced; k_res_unpermute
                             // moves it to eq when somebody needs it there -- update_e, which follows in the Gibbs loop,
                             // recomputes the residual and does not)
  const int32_t *perm;       // [G][R][NT] row of the slot, -1: pad
  const uint32_t *uidw;      // [G][5 R / 16][NT] the slots' users (index within the workgroup), 10 bits each: per group of
                             // 16 slots 4 words (3 users + 2 bits of the 4th of a batch) + 1 word (its other 8 bits);
                             // pads: umax - 1
  const uint32_t *headw;     // [G][R / 16][NT] bit r % 16: slot r starts a new item run (slot 0 of thread 0 always)
  const int32_t *run_item;   // [runs + 1] item of every (workgroup, item) run; pad runs: the pad item n_items
  const int32_t *first_run;  // [G][NT]   run containing the thread's first slot
  const int32_t *wg_run_ptr; // [G + 1] first run of the workgroup
  const int32_t *wg_nruns;   // [G] its runs; run wg_run_ptr[g] + wg_nruns[g] is its pad run (never gathered)
  const int32_t *wg_user_ptr;  // [G + 1]
  const int2 *user_desc;     // {feature, group}
  const int32_t *wg_item_ptr;  // [G + 1] items (positions in the level) drawn by the workgroup: at most umax - 1
  const int32_t *ent_ptr;    // [G + 1] the workgroup's slice of `entries`, in entries (a multiple of 64)
  const int2 *entries;       // per slice, source-workgroup-major: {run whose partial belongs to the item, item - first item
                             // of the slice}; pads: {the zero run, 0}
  const int32_t *scols;      // item -> feature
  const int2 *item_desc;     // item -> {feature, group}
  double *partials;          // [runs + 1][2], workgroup-major: a thread's runs are consecutive; the last stays (0, 0)
  double *dv;                // [items + 1][2]  (delta of this factor, coefficient of the next); the pad item stays (0, 0)
  double *V;                 // factor-major [K][D]
  int64_t D;
  int f_begin, f_end;
  // update_w (FMTrainer.hpp:231-254) as one more sweep in front of the factors: for a one-hot table it IS the latent sweep
  // with h = 1 for both levels (the other field's "coefficient" is 1: S2 = the column's entry count, S1 = -sum e, and
  // lin = alpha (S1 + S2 w_old) + lambda mu is :241-249's -alpha sum(e - w_old) + lambda mu)
  int linear;                // 1: sweep w first
  int n_sw;                  // sweeps of the launch: f_end - f_begin + linear
  double *w;                 // [D]
  const double *zw;          // [D] variates of the linear sweep
  const double *lam_w, *mu_w;  // [n_groups]
  double e_shift;            // added to every residual as it is loaded (update_w0's e += w0' - w0, :226)
  const double *z;           // variates of factor f at z + (f - f_begin) D
  const double *lam, *mu;    // [K][n_groups]
  const int32_t *group;      // per feature
  int n_groups;
  double alpha;
  const double *scal;        // non-null: alpha = scal[0] and e_shift = scal[2] are read from device memory (mfm_regression_iteration: the
                             // hyper-parameter draws were made on the device, the host has not seen them when the launch is enqueued)
  int n_items, umax;         // umax: LDS stride of the per-wave accumulator arrays (> users of any workgroup, > items of any slice)
  unsigned long long *bar;   // barrier state (RES_BAR_*), zeroed before the launch
  int n_wg;
  int *error;                // set on a spin timeout
  unsigned long long *prof;  // MFM_RES_PROF: [G][factors][8] s_memrealtime stamps of thread 0 (100 MHz), else null
  // row-sharded, one process per GPU (k_mf_resident<.., XCH = true>): the ranks' item sums meet INSIDE the launch. Every rank
  // owns an exchange array xsum[r] = [source rank][parity of the sweep][n_items + 1][2] and a flag word per source rank; a
  // workgroup writes its slice's sums into EVERY rank's array (its own rank's slot), the last workgroup of a rank to do so
  // raises the rank's flag on every peer, and whoever has seen all flags of the sweep adds the ranks' sums in rank order.
  int xworld, xrank;
  unsigned long long xepoch0;                 // sweeps of the launches before this one (the flags and the counter never go back)
  double *xsum[RES_MAX_PEERS];
  unsigned long long *xflag[RES_MAX_PEERS];   // [source rank] 16 words apart
  unsigned long long *xarrive;                // this rank's workgroups that have published the sweep
  // ... and the first-level coefficients: a user is drawn where its rows live and written into every rank's w / V there and then
  // (xmodel; the sweep's flag covers them: it is raised after every workgroup's stores), so the replicas need no
  // synchronisation after the launch
  int xmodel;
  double *xw[RES_MAX_PEERS], *xV[RES_MAX_PEERS];
  int ngx;                   // OVF: overflow groups of 16 slots per thread (their residual lives in e_slots, which must be set)
  int no_store;              // the residual is not written back at the end (the caller recomputes it: update_e follows, FMTrainer.hpp:494)
  int rot;                   // workgroup g runs as block (g - rot) mod G (MFM_RES_ROT: placement experiments)
  int dbg;                   // timing experiments only (MFM_RES_DBG; results are wrong when set): 4 no grid barriers, 32 no item
                             // draw, 64 no sweep A, 128 no sweep B, 4096 no partial stores inside sweep B
};

__device__ __forceinline__ void res_store2(double *p, double a, double b) {
  __hip_atomic_store(p, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through (sc1): visible to every XCD
  __hip_atomic_store(p + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Grid barrier, hierarchical by XCD (MI355X_MICROARCH.md "barrier-xcd"). The workgroups of one XCD share its L2: each of
// them drains its own stores into that L2 and arrives at the XCD's counter; only the LAST arriver of the XCD runs the
// agent-scope release (buffer_wbl2: the whole L2's dirty lines -- 128 KB of partials per workgroup) and then meets the other
// XCDs' leaders at the top counter; it releases its XCD through a generation word. (With a release per workgroup every
// early arriver started a write-back of the L2 the late ones were still storing into: the slowest workgroups of sweep B ran
// 20 us longer than the rest.) Every workgroup acquires afterwards. State: 128-byte spaced words, zeroed before each launch.
// The XCD a workgroup runs on is read from the hardware (HW_REG_XCC_ID), the workgroups per XCD are counted at the start of
// the launch: nothing depends on how the dispatcher places blocks.
enum { RES_BAR_START = 0, RES_BAR_TOP = 16, RES_BAR_XCNT = 32, RES_BAR_GEN = 160, RES_BAR_CENSUS = 288, RES_BAR_WORDS = 416 };
struct ResBar {
  int xcc;                     // this workgroup's XCD
  unsigned long long n_x, nx;  // workgroups on it, XCDs in use
};

__device__ __forceinline__ bool res_spin(const ResArgs &a, const unsigned long long *w, unsigned long long target, bool &dead) {
  unsigned spins = 0;
  while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
#ifndef MFM_RES_NO_SLEEP
    __builtin_amdgcn_s_sleep(1);
#endif
    if ((++spins & 1023u) == 0u) {
      if (spins > (1u << 22) || __hip_atomic_load(a.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        __hip_atomic_store(a.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dead = true;  // (every later barrier of this workgroup falls through: the launch ends, the host reports)
        return false;
      }
    }
  }
  return true;
}

// the same wait on a word another GPU writes (system scope)
__device__ __forceinline__ bool res_spin_sys(const ResArgs &a, const unsigned long long *w, unsigned long long target, bool &dead) {
  unsigned spins = 0;
  while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < target) {
    __builtin_amdgcn_s_sleep(2);
    if ((++spins & 1023u) == 0u) {
      const int seen = __hip_atomic_load(a.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (spins > (1u << 23) || seen != 0) {
        if (seen == 0) __hip_atomic_store(a.error, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (2: a peer rank did not show up)
        dead = true;
        return false;
      }
    }
  }
  return true;
}

// start of the launch (thread 0 only): census of the workgroups per XCD, one flat barrier
__device__ __forceinline__ void res_bar_init(const ResArgs &a, ResBar &rb, bool &dead) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
  rb.xcc = (int)(x & 7u);
  __hip_atomic_fetch_add(a.bar + RES_BAR_CENSUS + 16 * rb.xcc, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __hip_atomic_fetch_add(a.bar + RES_BAR_START, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  rb.n_x = 1;
  rb.nx = 1;
  if (a.dbg & 4) return;
  if (!res_spin(a, a.bar + RES_BAR_START, (unsigned long long)a.n_wg, dead)) return;
  rb.nx = 0;
  for (int i = 0; i < 8; i++) {
    const unsigned long long c = __hip_atomic_load(a.bar + RES_BAR_CENSUS + 16 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    rb.nx += c > 0 ? 1 : 0;
    if (i == rb.xcc) rb.n_x = c;
  }
}

// barrier #k of the launch (k counted from 1)
__device__ __forceinline__ void res_grid_barrier(const ResArgs &a, const ResBar &rb, unsigned long long k, int tid, bool &dead) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its stores have reached the XCD's L2
  __syncthreads();
  if (tid == 0 && !dead && !(a.dbg & 4)) {
    const unsigned long long old =
        __hip_atomic_fetch_add(a.bar + RES_BAR_XCNT + 16 * rb.xcc, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == k * rb.n_x) {
      // (the asm wait restates the wait the compiler may drop after the fence, Guideline 16 pitfall 12)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(a.bar + RES_BAR_TOP, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (res_spin(a, a.bar + RES_BAR_TOP, k * rb.nx, dead))
        __hip_atomic_store(a.bar + RES_BAR_GEN + 16 * rb.xcc, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      res_spin(a, a.bar + RES_BAR_GEN + 16 * rb.xcc, k, dead);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

typedef double res_d16_t __attribute__((ext_vector_type(16)));
typedef unsigned res_u4_t __attribute__((ext_vector_type(4)));

// NT threads; a thread's slots come in groups of 16: NGV groups live in registers (one 16-double vector each), NGL groups in
// LDS ([16 NGL][NT] doubles: lane-consecutive, conflict-free). Everything static about a group stays in registers for the whole
// launch: 16 head bits (a new item run starts at the slot) and the slots' users, 10 bits each (5 words). The sweeps are REAL
// loops over batches of 4 slots -- the loop counter is wave-uniform, so the batch's residuals are picked out of the register
// vectors with s_set_gpr_idx (no scratch, no waterfall) -- and only the group loop is unrolled.
//
// What a slot needs from outside the CU is its item's pair dv[item]; the item changes only where a run starts (every 4.9th
// slot at config 3). A 64-lane gather costs the CU's address path one cycle per distinct line, so the sweeps gather ONLY at
// run starts: every other lane of the instruction reads one wave-uniform dummy element (no branch, no execution mask) and
// carries the previous slot's value. The item of a run comes from run_item[run] -- the head bits give the run index of
// every slot without a memory access -- gathered the same way. Both chains are software-pipelined by hand: while batch b
// computes, the dv gathers of batch b + 1 and the run_item gathers of batch b + 2 are in flight.
// OVF (tables beyond the on-chip capacity, 40 960 rows per CU): a thread owns a.ngx MORE groups of 16 slots whose residual stays in
// the slot-ordered buffer in global memory (e_slots: 8 bytes read + 8 written per slot and sweep, coalesced) and whose static
// words are read again every sweep; they run through the same batch steps behind the on-chip groups (the run / dv gather
// pipelines carry straight on), so the sums, partials and draws are those of the all-on-chip form.
template <int NT, int NGV, int NGL, bool XCH = false, bool OVF = false>
__global__ __launch_bounds__(NT) void k_mf_resident(ResArgs a) {
  extern __shared__ __attribute__((aligned(16))) char res_smem[];
  constexpr int NW = NT / WAVE, NG = NGV + NGL, R = 16 * NG, RL = 16 * NGL;
  constexpr int B = 4;  // slots per batch
  const int g = (int)((blockIdx.x + (unsigned)a.rot) % gridDim.x), tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int U = a.umax;
  const double alpha_k = a.scal ? a.scal[0] : a.alpha, e_shift_k = a.scal ? a.scal[2] : a.e_shift;
  const int ngx = OVF ? a.ngx : 0, NGt = NG + ngx, Rt = 16 * NGt;  // groups / slots of a thread in all (the layout arrays' strides)
  double *elds = (double *)res_smem;      // [RL][NT] residual of the LDS-resident slots
  double *acc1 = elds + (size_t)RL * NT;  // [NW][U]  sum of -e h per (wave, user) / (wave, item of the slice)
  double *acc2 = acc1 + NW * U;           // [NW][U]  sum of h^2
  d2_t *utab = (d2_t *)(acc2 + NW * U);   // [U] {new coefficient, new - old}
  d2_t *wcarry = utab + U;                // [NW]
  int *wflag = (int *)(wcarry + NW);      // [NW]
  const int32_t *perm_g = a.perm + (int64_t)g * Rt * NT;
  const d2_t *dv2 = (const d2_t *)a.dv;
  const int u0 = a.wg_user_ptr[g], nu = a.wg_user_ptr[g + 1] - u0;
  const int rb0 = a.wg_run_ptr[g];
  const int pad_item = a.n_items;
  bool dead = false;
  unsigned long long nbar = 0;
  ResBar rbar;
  rbar.xcc = 0;
  rbar.n_x = rbar.nx = 1;
#define RES_STAMP0(k) \
  if (a.prof && tid == 0) a.prof[(int64_t)g * a.n_sw * 64 + 56 + (k)] = __builtin_amdgcn_s_memrealtime()
  RES_STAMP0(0);
  if (tid == 0) res_bar_init(a, rbar, dead);
  RES_STAMP0(1);
  const bool nost = (a.dbg & 4096) != 0;  // (4096: no partial stores inside sweep B)
  // The two wavefronts of a SIMD (w and w + 4) take turns at the higher issue priority, group by group: the arbiter otherwise
  // prefers the older one throughout, which then finishes a sweep 5 us ahead and leaves the younger to run alone (MFM_RES_PROF:
  // waves 0-3 at 20 us, waves 4-7 at 25 us of sweep B; with the turns 19.0 and 19.7, config 3 333.4 -> 338-341 it/s in one box;
  // changing turns every half group gains nothing more). a.dbg & 262144 switches it off (A/B).
  const bool prio_swap = !(a.dbg & 262144);
  const int wv_hi = wv >> 2;
#define RES_PRIO(j)                              \
  if (prio_swap) {                               \
    if ((((j) & 1) ^ wv_hi) != 0)                \
      __builtin_amdgcn_s_setprio(1);             \
    else                                         \
      __builtin_amdgcn_s_setprio(0);             \
  }
  const int trash_run = a.wg_run_ptr[g] + a.wg_nruns[g];
#define RES_STAMP(k)                                                                                      \
  if (a.prof && tid == 0) a.prof[((int64_t)g * a.n_sw + (f - f_first)) * 64 + (k)] = __builtin_amdgcn_s_memrealtime()

  for (int i = tid; i < 2 * NW * U; i += NT) acc1[i] = 0.0;
  for (int i = tid; i < U; i += NT) utab[i] = d2_t{0.0, 0.0};

  // static per-thread words
  res_u4_t uw[NG];            // word bb of a group: users 0..2 of batch bb and the low 2 bits of user 3
  unsigned ux[NG];            // byte bb: the high 8 bits of user 3 of batch bb
  unsigned hbv[NG], nbv[NG];  // head bits; "gather here" bits = the head bits + the thread's first slot
#pragma unroll
  for (int j = 0; j < NG; j++) {
#pragma unroll
    for (int w = 0; w < 4; w++) uw[j][w] = a.uidw[((int64_t)g * (5 * NGt) + 5 * j + w) * NT + tid];
    ux[j] = a.uidw[((int64_t)g * (5 * NGt) + 5 * j + 4) * NT + tid];
    hbv[j] = a.headw[((int64_t)g * NGt + j) * NT + tid];
    nbv[j] = hbv[j] | (j == 0 ? 1u : 0u);
  }
  // OVF: the head bits of the first overflow group (the last on-chip group's gathers look two batches ahead), the overflow
  // groups' residuals in the slot-ordered buffer, their static words
  const unsigned nbx0 = OVF && ngx > 0 ? a.headw[((int64_t)g * NGt + NG) * NT + tid] : 0u;
  double *eog = OVF ? a.e_slots + ((int64_t)g * Rt + 16 * NG) * NT + tid : nullptr;
  const uint32_t *uidw_x = a.uidw + ((int64_t)g * (5 * NGt) + 5 * NG) * NT + tid;  // word w of overflow group jx: [(5 jx + w) NT]
  const uint32_t *headw_x = a.headw + ((int64_t)g * NGt + NG) * NT + tid;          // [jx NT]
  const int run0 = a.first_run[g * NT + tid];  // the run containing this thread's first slot
  const bool head0 = (hbv[0] & 1u) != 0u;
  // gather bits of batch q of group j; q may run past the group (0 .. 5)
#define RES_NIB(j, q) ((((nbv[j]) | ((j) + 1 < NG ? nbv[(j) + 1 < NG ? (j) + 1 : (j)] << 16 : nbx0 << 16)) >> (4 * (q))) & 15u)
  // the four users of batch bb of group j
#define RES_UIDS(j, bb, uid)                                                     \
  {                                                                              \
    const unsigned lo_ = uw[j][bb];                                              \
    uid[0] = (int)(lo_ & 0x3ffu);                                                \
    uid[1] = (int)((lo_ >> 10) & 0x3ffu);                                        \
    uid[2] = (int)((lo_ >> 20) & 0x3ffu);                                        \
    uid[3] = (int)((lo_ >> 30) | (((ux[j] >> (8 * (bb))) & 0xffu) << 2));        \
  }

#define RES_UIDSX(bb, uid)                                                       \
  {                                                                              \
    const unsigned lo_ = uwx[bb];                                                \
    uid[0] = (int)(lo_ & 0x3ffu);                                                \
    uid[1] = (int)((lo_ >> 10) & 0x3ffu);                                        \
    uid[2] = (int)((lo_ >> 20) & 0x3ffu);                                        \
    uid[3] = (int)((lo_ >> 30) | (((uxx >> (8 * (bb))) & 0xffu) << 2));          \
  }

  // Pad slots carry (pad item, pad user): an item whose dv entry stays (0, 0) and a user slot nobody draws, so that every slot
  // runs the same straight-line code: a pad's statistics add 0 whatever its residual holds.
  res_d16_t ev[NGV > 0 ? NGV : 1];
#pragma unroll
  for (int j = 0; j < NG; j++) {
#pragma unroll 1
    for (int bb = 0; bb < 4; bb++) {
      int row[B];
#pragma unroll
      for (int k = 0; k < B; k++) row[k] = a.e_in ? 0 : perm_g[(16 * j + 4 * bb + k) * NT + tid];
#pragma unroll
      for (int k = 0; k < B; k++) {
        const double x = (a.e_in ? a.e_in[((int64_t)g * Rt + 16 * j + 4 * bb + k) * NT + tid] : a.eq[row[k] < 0 ? 0 : row[k]].x) + e_shift_k;
        if (j < NGV)
          ev[j < NGV ? j : 0][4 * bb + k] = x;
        else
          elds[(16 * (j - NGV) + 4 * bb + k) * NT + tid] = x;
      }
    }
  }
  if (OVF) {  // the overflow slots: into (or shifted inside) the slot-ordered buffer they live in for the launch
    for (int jx = 0; jx < ngx; jx++) {
#pragma unroll 4
      for (int i = 0; i < 16; i++) {
        const int64_t sl = 16 * (NG + jx) + i;
        const int row = a.e_in ? 0 : perm_g[sl * NT + tid];
        const double x = (a.e_in ? a.e_in[((int64_t)g * Rt + sl) * NT + tid] : a.eq[row < 0 ? 0 : row].x) + e_shift_k;
        eog[(16 * jx + i) * NT] = x;
      }
    }
  }
  __syncthreads();

  RES_STAMP0(2);
  const int f_first = a.f_begin - a.linear;
  for (int f = f_first; f < a.f_end; f++) {
    const bool lin = f < a.f_begin;  // the linear sweep
    double *Vf = lin ? a.w : a.V + (int64_t)f * a.D;
    const double *zf = lin ? a.zw : a.z + (int64_t)(f - a.f_begin) * a.D;
    const double *lamf = lin ? a.lam_w : a.lam + (int64_t)f * a.n_groups;
    const double *muf = lin ? a.mu_w : a.mu + (int64_t)f * a.n_groups;
    // this thread's user: everything its draw needs is requested now
    int uj = 0;
    double uold = 0.0, uz = 0.0, ulam = 0.0, umu = 0.0;
    if (tid < nu) {
      const int2 d = a.user_desc[u0 + tid];
      uj = d.x;
      uold = Vf[uj];
      uz = zf[uj];
      ulam = lamf[d.y];
      umu = muf[d.y];
    }
    RES_STAMP(0);
    // ---- sweep A: the item update of the previous factor (:371-375; dv.x = 0 before the first), the user level's
    //      statistics (:351-356)
    if (!(a.dbg & 64)) {
      int rc = run0 - 1;  // run counter of the run_item stage
      int itA[B];         // items of the next batch (in flight)
      d2_t ddA[B];        // dv pairs of this batch (in flight)
      {
        int it0[B];
        const unsigned n0 = RES_NIB(0, 0), n1 = RES_NIB(0, 1);
#pragma unroll
        for (int k = 0; k < B; k++) {
          rc += (int)((n0 >> k) & 1u);
          it0[k] = a.run_item[(n0 >> k) & 1u ? rc : rb0];
        }
#pragma unroll
        for (int k = 0; k < B; k++) {
          rc += (int)((n1 >> k) & 1u);
          itA[k] = a.run_item[(n1 >> k) & 1u ? rc : rb0];
        }
#pragma unroll
        for (int k = 0; k < B; k++) ddA[k] = dv2[(n0 >> k) & 1u ? it0[k] : pad_item];
      }
      d2_t ddc = d2_t{0.0, 0.0};
      int itB[B];
      d2_t ddB[B];
      // one batch: the run_item gathers of batch b + 2 (-> ito), the dv gathers of batch b + 1 (items iti -> ddo), then batch
      // b from ddi. Two steps with the buffers swapped make one iteration of the rolled loop: no register is copied while
      // its load is in flight.
#define RES_STEP_A(j, bb, iti, ito, ddi, ddo)                                                                              \
  {                                                                                                                        \
    const unsigned n4 = RES_NIB(j, bb), n4a = RES_NIB(j, (bb) + 1), n4b = RES_NIB(j, (bb) + 2);                            \
    _Pragma("unroll") for (int k = 0; k < B; k++) {                                                                        \
      rc += (int)((n4b >> k) & 1u);                                                                                        \
      ito[k] = a.run_item[(n4b >> k) & 1u ? rc : rb0];                                                                     \
    }                                                                                                                      \
    _Pragma("unroll") for (int k = 0; k < B; k++) ddo[k] = dv2[(n4a >> k) & 1u ? iti[k] : pad_item];                       \
    int uid[B];                                                                                                            \
    RES_UIDS(j, bb, uid);                                                                                                  \
    double up[B], ex[B];                                                                                                   \
    _Pragma("unroll") for (int k = 0; k < B; k++) {                                                                        \
      up[k] = utab[uid[k]][0];                                                                                             \
      ex[k] = j < NGV ? ev[j < NGV ? j : 0][4 * (bb) + k] : elds[(16 * (j - NGV) + 4 * (bb) + k) * NT + tid];              \
    }                                                                                                                      \
    _Pragma("unroll") for (int k = 0; k < B; k++) {                                                                        \
      const bool need = ((n4 >> k) & 1u) != 0u;                                                                            \
      ddc[0] = need ? ddi[k][0] : ddc[0];                                                                                  \
      ddc[1] = need ? ddi[k][1] : ddc[1];                                                                                  \
      const double er = ex[k] + up[k] * ddc[0];                                                                            \
      if (j < NGV)                                                                                                         \
        ev[j < NGV ? j : 0][4 * (bb) + k] = er;                                                                            \
      else                                                                                                                 \
        elds[(16 * (j - NGV) + 4 * (bb) + k) * NT + tid] = er;                                                             \
      const double c = ddc[1];                                                                                             \
      __hip_atomic_fetch_add(&acc1[wv * U + uid[k]], (-er) * c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);           \
      __hip_atomic_fetch_add(&acc2[wv * U + uid[k]], c * c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);               \
    }                                                                                                                      \
  }
#define RES_STEP_AX(bb, iti, ito, ddi, ddo)                                                                                 \
  {                                                                                                                        \
    const unsigned n4 = ((nbw >> (4 * (bb))) & 15u), n4a = ((nbw >> (4 * ((bb) + 1))) & 15u), n4b = ((nbw >> (4 * ((bb) + 2))) & 15u);                            \
    _Pragma("unroll") for (int k = 0; k < B; k++) {                                                                        \
      rc += (int)((n4b >> k) & 1u);                                                                                        \
      ito[k] = a.run_item[(n4b >> k) & 1u ? rc : rb0];                                                                     \
    }                                                                                                                      \
    _Pragma("unroll") for (int k = 0; k < B; k++) ddo[k] = dv2[(n4a >> k) & 1u ? iti[k] : pad_item];                       \
    int uid[B];                                                                                                            \
    RES_UIDSX(bb, uid);                                                                                                      \
    double up[B], ex[B];                                                                                                   \
    _Pragma("unroll") for (int k = 0; k < B; k++) {                                                                        \
      up[k] = utab[uid[k]][0];                                                                                             \
      ex[k] = eo[4 * (bb) + k];              \
    }                                                                                                                      \
    _Pragma("unroll") for (int k = 0; k < B; k++) {                                                                        \
      const bool need = ((n4 >> k) & 1u) != 0u;                                                                            \
      ddc[0] = need ? ddi[k][0] : ddc[0];                                                                                  \
      ddc[1] = need ? ddi[k][1] : ddc[1];                                                                                  \
      const double er = ex[k] + up[k] * ddc[0];                                                                            \
      eo[4 * (bb) + k] = er;                                                             \
      const double c = ddc[1];                                                                                             \
      __hip_atomic_fetch_add(&acc1[wv * U + uid[k]], (-er) * c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);           \
      __hip_atomic_fetch_add(&acc2[wv * U + uid[k]], c * c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);               \
    }                                                                                                                      \
  }
#pragma unroll
      for (int j = 0; j < NG; j++) {
        RES_PRIO(j);
#pragma unroll 1
        for (int bp = 0; bp < 2; bp++) {
          RES_STEP_A(j, 2 * bp, itA, itB, ddA, ddB);
          RES_STEP_A(j, 2 * bp + 1, itB, itA, ddB, ddA);
        }
      }
      if (OVF) {  // the overflow groups: static words and residuals from global memory, group by group
        unsigned hb_c = nbx0;
        for (int jx = 0; jx < ngx; jx++) {
          res_u4_t uwx;
#pragma unroll
          for (int w = 0; w < 4; w++) uwx[w] = uidw_x[(5 * jx + w) * NT];
          const unsigned uxx = uidw_x[(5 * jx + 4) * NT];
          const unsigned hb_n = jx + 1 < ngx ? headw_x[(jx + 1) * NT] : 0u;
          const unsigned nbw = hb_c | (hb_n << 16);
          res_d16_t eo;
#pragma unroll
          for (int i = 0; i < 16; i++) eo[i] = eog[(16 * jx + i) * NT];
#pragma unroll 1
          for (int bp = 0; bp < 2; bp++) {
            RES_STEP_AX(2 * bp, itA, itB, ddA, ddB);
            RES_STEP_AX(2 * bp + 1, itB, itA, ddB, ddA);
          }
#pragma unroll
          for (int i = 0; i < 16; i++) eog[(16 * jx + i) * NT] = eo[i];
          hb_c = hb_n;
        }
      }
#undef RES_STEP_A
#undef RES_STEP_AX
    }
    __syncthreads();
    RES_STAMP(1);
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
      const double fresh = PMainV::draw(S1, S2, uold, alpha_k, ulam, umu, uz);
      Vf[uj] = fresh;
      if (XCH && a.xmodel) {  // the other ranks' replicas of this coefficient
        for (int r = 0; r < a.xworld; r++)
          if (r != a.xrank)
            __hip_atomic_store((lin ? a.xw[r] : a.xV[r] + (int64_t)f * a.D) + uj, fresh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      utab[tid] = d2_t{lin ? 1.0 : fresh, fresh - uold};  // {h of the item level, delta}
    }
    __syncthreads();
    RES_STAMP(2);
    // ---- sweep B: the user update (:371-375), the item level's statistics run by run
    {
      bool have_head = false;
      double f1 = 0.0, f2 = 0.0, s1 = 0.0, s2 = 0.0;
      d2_t *part2 = (d2_t *)a.partials;
      if (!(a.dbg & 128)) {
        int rc = run0 - 1, rcC = run0 - 1;  // run counters of the run_item stage and of the compute stage
        int itA[B];
        double ccA[B];
        {
          int it0[B];
          const unsigned n0 = RES_NIB(0, 0), n1 = RES_NIB(0, 1);
#pragma unroll
          for (int k = 0; k < B; k++) {
            rc += (int)((n0 >> k) & 1u);
            it0[k] = a.run_item[(n0 >> k) & 1u ? rc : rb0];
          }
#pragma unroll
          for (int k = 0; k < B; k++) {
            rc += (int)((n1 >> k) & 1u);
            itA[k] = a.run_item[(n1 >> k) & 1u ? rc : rb0];
          }
#pragma unroll
          for (int k = 0; k < B; k++) ccA[k] = a.dv[2 * (int64_t)((n0 >> k) & 1u ? it0[k] : pad_item) + 1];
        }
        double ccc = 0.0;
        int itB[B];
        double ccB[B];
#ifdef MFM_RES_UCST  // (experiment: a fixed number of stores per batch, the slots that close nothing store to the pad run)
#define RES_PARTIAL_STORE() part2[head && have_head ? rcC - 1 : trash_run] = d2_t{s1, s2}
#else
#define RES_PARTIAL_STORE() \
  if (head && have_head && !nost) part2[rcC - 1] = d2_t{s1, s2}
#endif
#define RES_STEP_B(j, bb, iti, ito, cci, cco)                                                                              \
  {                                                                                                                        \
    const unsigned n4 = RES_NIB(j, bb), n4a = RES_NIB(j, (bb) + 1), n4b = RES_NIB(j, (bb) + 2);                            \
    const unsigned h4 = (hbv[j] >> (4 * (bb))) & 15u;                                                                      \
    _Pragma("unroll") for (int k = 0; k < B; k++) {                                                                        \
      rc += (int)((n4b >> k) & 1u);                                                                                        \
      ito[k] = a.run_item[(n4b >> k) & 1u ? rc : rb0];                                                                     \
    }                                                                                                                      \
    _Pragma("unroll") for (int k = 0; k < B; k++) cco[k] = a.dv[2 * (int64_t)((n4a >> k) & 1u ? iti[k] : pad_item) + 1];   \
    int uid[B];                                                                                                            \
    RES_UIDS(j, bb, uid);                                                                                                  \
    d2_t ut[B];                                                                                                            \
    double ex[B];                                                                                                          \
    _Pragma("unroll") for (int k = 0; k < B; k++) {                                                                        \
      ut[k] = utab[uid[k]];                                                                                                \
      ex[k] = j < NGV ? ev[j < NGV ? j : 0][4 * (bb) + k] : elds[(16 * (j - NGV) + 4 * (bb) + k) * NT + tid];              \
    }                                                                                                                      \
    _Pragma("unroll") for (int k = 0; k < B; k++) {                                                                        \
      const bool need = ((n4 >> k) & 1u) != 0u;                                                                            \
      const bool head = ((h4 >> k) & 1u) != 0u;                                                                            \
      ccc = need ? cci[k] : ccc;                                                                                           \
      rcC += need ? 1 : 0; /* the run of this slot */                                                                      \
      /* a run that began and ended in this thread: the slot before this head closed run rcC - 1 */                        \
      RES_PARTIAL_STORE();                                                                                                 \
      f1 = head && !have_head ? s1 : f1;                                                                                   \
      f2 = head && !have_head ? s2 : f2;                                                                                   \
      have_head = have_head || head;                                                                                       \
      const double er = ex[k] + ccc * ut[k][1];                                                                            \
      if (j < NGV)                                                                                                         \
        ev[j < NGV ? j : 0][4 * (bb) + k] = er;                                                                            \
      else                                                                                                                 \
        elds[(16 * (j - NGV) + 4 * (bb) + k) * NT + tid] = er;                                                             \
      s1 = (head ? 0.0 : s1) + (-er) * ut[k][0];                                                                           \
      s2 = (head ? 0.0 : s2) + ut[k][0] * ut[k][0];                                                                        \
    }                                                                                                                      \
  }
#define RES_STEP_BX(bb, iti, ito, cci, cco)                                                                                 \
  {                                                                                                                        \
    const unsigned n4 = ((nbw >> (4 * (bb))) & 15u), n4a = ((nbw >> (4 * ((bb) + 1))) & 15u), n4b = ((nbw >> (4 * ((bb) + 2))) & 15u);                            \
    const unsigned h4 = (hb_c >> (4 * (bb))) & 15u;                                                                       \
    _Pragma("unroll") for (int k = 0; k < B; k++) {                                                                        \
      rc += (int)((n4b >> k) & 1u);                                                                                        \
      ito[k] = a.run_item[(n4b >> k) & 1u ? rc : rb0];                                                                     \
    }                                                                                                                      \
    _Pragma("unroll") for (int k = 0; k < B; k++) cco[k] = a.dv[2 * (int64_t)((n4a >> k) & 1u ? iti[k] : pad_item) + 1];   \
    int uid[B];                                                                                                            \
    RES_UIDSX(bb, uid);                                                                                                      \
    d2_t ut[B];                                                                                                            \
    double ex[B];                                                                                                          \
    _Pragma("unroll") for (int k = 0; k < B; k++) {                                                                        \
      ut[k] = utab[uid[k]];                                                                                                \
      ex[k] = eo[4 * (bb) + k];              \
    }                                                                                                                      \
    _Pragma("unroll") for (int k = 0; k < B; k++) {                                                                        \
      const bool need = ((n4 >> k) & 1u) != 0u;                                                                            \
      const bool head = ((h4 >> k) & 1u) != 0u;                                                                            \
      ccc = need ? cci[k] : ccc;                                                                                           \
      rcC += need ? 1 : 0; /* the run of this slot */                                                                      \
      /* a run that began and ended in this thread: the slot before this head closed run rcC - 1 */                        \
      RES_PARTIAL_STORE();                                                                                                 \
      f1 = head && !have_head ? s1 : f1;                                                                                   \
      f2 = head && !have_head ? s2 : f2;                                                                                   \
      have_head = have_head || head;                                                                                       \
      const double er = ex[k] + ccc * ut[k][1];                                                                            \
      eo[4 * (bb) + k] = er;                                                             \
      s1 = (head ? 0.0 : s1) + (-er) * ut[k][0];                                                                           \
      s2 = (head ? 0.0 : s2) + ut[k][0] * ut[k][0];                                                                        \
    }                                                                                                                      \
  }
#pragma unroll
        for (int j = 0; j < NG; j++) {
          RES_PRIO(j);
#pragma unroll 1
          for (int bp = 0; bp < 2; bp++) {
              RES_STEP_B(j, 2 * bp, itA, itB, ccA, ccB);
            if (a.prof && lane == 0 && (wv == 4 || wv == 7))  // (waves 4 and 7: every batch)
              a.prof[((int64_t)g * a.n_sw + (f - f_first)) * 64 + 16 + (wv == 4 ? 0 : 20) + 4 * j + 2 * bp] =
                  __builtin_amdgcn_s_memrealtime();
            RES_STEP_B(j, 2 * bp + 1, itB, itA, ccB, ccA);
            if (a.prof && lane == 0 && (wv == 4 || wv == 7))
              a.prof[((int64_t)g * a.n_sw + (f - f_first)) * 64 + 16 + (wv == 4 ? 0 : 20) + 4 * j + 2 * bp + 1] =
                  __builtin_amdgcn_s_memrealtime();
          }
        }
        if (OVF) {
          unsigned hb_c = nbx0;
          for (int jx = 0; jx < ngx; jx++) {
            res_u4_t uwx;
#pragma unroll
            for (int w = 0; w < 4; w++) uwx[w] = uidw_x[(5 * jx + w) * NT];
            const unsigned uxx = uidw_x[(5 * jx + 4) * NT];
            const unsigned hb_n = jx + 1 < ngx ? headw_x[(jx + 1) * NT] : 0u;
            const unsigned nbw = hb_c | (hb_n << 16);
            res_d16_t eo;
#pragma unroll
            for (int i = 0; i < 16; i++) eo[i] = eog[(16 * jx + i) * NT];
#pragma unroll 1
            for (int bp = 0; bp < 2; bp++) {
              RES_STEP_BX(2 * bp, itA, itB, ccA, ccB);
              RES_STEP_BX(2 * bp + 1, itB, itA, ccB, ccA);
            }
#pragma unroll
            for (int i = 0; i < 16; i++) eog[(16 * jx + i) * NT] = eo[i];
            hb_c = hb_n;
          }
        }
#undef RES_STEP_B
#undef RES_STEP_BX
#undef RES_PARTIAL_STORE
      }
      if (a.prof && lane == 0)  // (per wave: the end of its own sweep B)
        a.prof[((int64_t)g * a.n_sw + (f - f_first)) * 64 + 8 + wv] = __builtin_amdgcn_s_memrealtime();
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
    RES_STAMP(3);
    // everything of the item draw that does not depend on the partials is requested before the barrier: this thread's item
    // (coefficient, variate, hyper-parameters, the next factor's coefficient) and the wave's first list entries
    constexpr int CH = MFM_RES_CH;
    const int c0 = a.ent_ptr[g] >> 6, c1 = a.ent_ptr[g + 1] >> 6;
    const int per = (c1 - c0 + NW - 1) / NW;
    const int wb = c0 + wv * per, we = wb + per < c1 ? wb + per : c1;
    const int i0 = a.wg_item_ptr[g], ni = a.wg_item_ptr[g + 1] - i0;
    const bool more = f + 1 < a.f_end;
    int ij = 0;
    double iold = 0.0, iz = 0.0, ivn = 0.0, ilam = 0.0, imu = 0.0;
    if (tid < ni) {
      const int2 d = a.item_desc[i0 + tid];
      ij = d.x;
      iold = Vf[ij];
      iz = zf[ij];
      ivn = more ? a.V[(int64_t)(f + 1) * a.D + ij] : 0.0;
      ilam = lamf[d.y];
      imu = muf[d.y];
    }
    int2 en[CH];
#pragma unroll
    for (int k = 0; k < CH; k++) en[k] = make_int2(0, 0);
    if (wb < we) {  // (wave-uniform)
#pragma unroll
      for (int k = 0; k < CH; k++) en[k] = a.entries[(int64_t)(wb + k < we ? wb + k : wb) * WAVE + lane];
    }
    res_grid_barrier(a, rbar, ++nbar, tid, dead);
    RES_STAMP(4);
    // ---- item draw (:357-369). The workgroup draws a contiguous slice of the items. The slice's partials are listed source
    //      workgroup by source workgroup (a source's runs of consecutive items are adjacent in `partials`: the gathers read
    //      whole lines); a wave takes a contiguous stretch of the list and adds into its own accumulator array (ds_add_f64:
    //      the lanes of one instruction that hit the same item are serialised in lane order, a wave's instructions run in
    //      program order); then a thread per item adds the wave arrays in wave order (fixed association) and draws.
    if (!(a.dbg & 32)) {
      const d2_t *part2 = (const d2_t *)a.partials;
      for (int cb = wb; cb < we; cb += CH) {
        d2_t sv[CH];
#pragma unroll
        for (int k = 0; k < CH; k++) sv[k] = part2[en[k].x];
        int il[CH];
#pragma unroll
        for (int k = 0; k < CH; k++) il[k] = en[k].y;
        if (cb + CH < we) {  // the next entries while the partials are in flight (wave-uniform)
#pragma unroll
          for (int k = 0; k < CH; k++) en[k] = a.entries[(int64_t)(cb + CH + k < we ? cb + CH + k : cb + CH) * WAVE + lane];
        }
#pragma unroll
        for (int k = 0; k < CH; k++) {
          if (cb + k >= we) break;  // wave-uniform
          __hip_atomic_fetch_add(&acc1[wv * U + il[k]], sv[k][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(&acc2[wv * U + il[k]], sv[k][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      __syncthreads();
      double S1 = 0.0, S2 = 0.0;
      if (tid < ni) {
#pragma unroll
        for (int w = 0; w < NW; w++) {
          S1 += acc1[w * U + tid];
          S2 += acc2[w * U + tid];
          acc1[w * U + tid] = 0.0;
          acc2[w * U + tid] = 0.0;
        }
      }
      if (XCH) {
        // the other ranks' rows of these items: publish this rank's sums to every rank, wait for everybody's, add in rank order
        const unsigned long long E = a.xepoch0 + (unsigned long long)(f - f_first) + 1ull;
        const int64_t stride = 2 * ((int64_t)a.n_items + 1), par = (int64_t)(E & 1ull);
        if (tid < ni) {
          const int64_t off = ((int64_t)a.xrank * 2 + par) * stride + 2 * (int64_t)(i0 + tid);
          for (int r = 0; r < a.xworld; r++) {
            __hip_atomic_store(a.xsum[r] + off, S1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(a.xsum[r] + off + 1, S2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
        // Everything a peer reads was written with system-scope (write-through) stores into uncached memory, so "published" means
        // "the wave's stores are acknowledged" (vmcnt = 0) -- no release fence: a system-scope fence writes the whole L2 back
        // (the partials of the sweep) and, run by every workgroup, made this phase 51 us instead of 11 (MFM_RES_PROF).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0 && !dead) {
          const unsigned long long old = __hip_atomic_fetch_add(a.xarrive, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (old + 1 == (unsigned long long)a.n_wg * E) {  // every workgroup of this rank has published: tell the peers
            // Ordering. Every workgroup's sums were written with system-scope (sc0 sc1: write-through, no line kept in any L2)
            // stores into uncached memory and acknowledged (s_waitcnt vmcnt(0)) before its arrival above; this thread has seen all
            // arrivals. The flag itself is a system-scope RELEASE store (one per rank and sweep: buffer_wbl2 sc0 sc1 + the store),
            // so that the hand-over is a release / acquire pair in the memory model as well; the readers poll relaxed and read
            // the sums with system-scope loads, which bypass the non-coherent caches (no acquire fence: it would drop the XCD's L2).
            for (int r = 0; r < a.xworld && !(a.dbg & 8192); r++)  // (8192: a test of the time-out path -- the flags stay down)
              __hip_atomic_store(a.xflag[r] + 16 * a.xrank, E, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
          }
          for (int r = 0; r < a.xworld && !dead; r++) res_spin_sys(a, a.xflag[a.xrank] + 16 * r, E, dead);
        }
        __syncthreads();
        if (tid < ni) {
          S1 = S2 = 0.0;
          for (int r = 0; r < a.xworld; r++) {
            const int64_t off = ((int64_t)r * 2 + par) * stride + 2 * (int64_t)(i0 + tid);
            S1 += __hip_atomic_load(a.xsum[a.xrank] + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            S2 += __hip_atomic_load(a.xsum[a.xrank] + off + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
      }
      if (tid < ni) {
        const double fresh = PMainV::draw(S1, S2, iold, alpha_k, ilam, imu, iz);
        Vf[ij] = fresh;
        res_store2(a.dv + 2 * (int64_t)(i0 + tid), fresh - iold, ivn);
      }
    }
    RES_STAMP(5);
    res_grid_barrier(a, rbar, ++nbar, tid, dead);
    RES_STAMP(6);
  }
#undef RES_STAMP
  RES_STAMP0(3);
  // the last factor's item update, then the residual goes back -- unless nobody will read it (no_store)
  if (!a.no_store) {
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
        int uid[B];
        RES_UIDS(j, bb, uid);
#pragma unroll
        for (int k = 0; k < B; k++) {
          const double dl = a.dv[2 * (int64_t)it[k]];
          const double ex = j < NGV ? ev[j < NGV ? j : 0][4 * bb + k] : elds[(16 * (j - NGV) + 4 * bb + k) * NT + tid];
          const double ef = ex + utab[uid[k]][0] * dl;
          if (a.e_slots)
            a.e_slots[((int64_t)g * Rt + 16 * j + 4 * bb + k) * NT + tid] = ef;
          else if (row[k] >= 0)
            a.eq[row[k]].x = ef;
        }
      }
    }
    if (OVF) {
      for (int jx = 0; jx < ngx; jx++) {
        res_u4_t uwx;
#pragma unroll
        for (int w = 0; w < 4; w++) uwx[w] = uidw_x[(5 * jx + w) * NT];
        const unsigned uxx = uidw_x[(5 * jx + 4) * NT];
        const unsigned hbx = headw_x[jx * NT];
#pragma unroll 1
        for (int bb = 0; bb < 4; bb++) {
          const unsigned h4 = (hbx >> (4 * bb)) & 15u;
          int it[B], uid[B];
#pragma unroll
          for (int k = 0; k < B; k++) {
            run_last += (int)((h4 >> k) & 1u);
            it[k] = a.run_item[run_last];
          }
          RES_UIDSX(bb, uid);
#pragma unroll
          for (int k = 0; k < B; k++) {
            const double dl = a.dv[2 * (int64_t)it[k]];
            double *ep = eog + (16 * jx + 4 * bb + k) * NT;
            *ep = *ep + utab[uid[k]][0] * dl;
          }
        }
      }
    }
  }
#undef RES_PRIO
#undef RES_NIB
#undef RES_UIDS
#undef RES_UIDSX
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  RES_STAMP0(4);
#undef RES_STAMP0
}

// ---- host side: the resident layout of a two-field table and the launch ---------------------------------------------
__global__ void k_res_init_dv(const double *__restrict__ theta, const int32_t *__restrict__ scols, int n_items,
                              double *__restrict__ dv, unsigned long long *__restrict__ bar) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0)  // the launch's barrier words start from zero (one dispatch less than a memset of their own)
    for (int k = threadIdx.x; k < RES_BAR_WORDS; k += blockDim.x) bar[k] = 0ull;
  if (i < n_items) {
    dv[2 * i] = 0.0;
    dv[2 * i + 1] = theta ? theta[scols[i]] : 1.0;  // (theta == null: the linear sweep, h = 1)
  } else if (i == n_items) {  // the pad item: (0, 0) for ever
    dv[2 * i] = 0.0;
    dv[2 * i + 1] = 0.0;
  }
}

// eq[row].x of every slot's row from the slot-ordered copy the resident sweep left
__global__ void k_res_unpermute(const double *__restrict__ e_slots, const int32_t *__restrict__ perm, int64_t n_slots,
                                double2 *__restrict__ eq) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_slots) return;
  const int32_t row = perm[i];
  if (row >= 0) eq[row].x = e_slots[i];
}

// ---- update_e (FMTrainer.hpp:493-497 -> FM.hpp:54-136) in the resident layout ----------------------------------------------
// e = w0 + w_u + w_i + <V_u, V_i> - y for every slot, written in slot order: the next persistent launch reads it back coalesced
// (no slot -> row gather), nothing touches eq inside the Gibbs loop (materialize_e moves it there when somebody asks). A
// workgroup's users' rows of Vt sit in LDS for the launch; the slots are in item order, so a thread walks its R slots with the
// current item's row in registers and fetches a new one at every run head (the run's feature comes from run_feat, one round
// trip). sum e and sum e^2 (update_alpha, update_w0: FMTrainer.hpp:127-145, 218-229) are taken on the way: one partial per
// workgroup, thread order inside the waves and wave order inside the workgroup fixed.
struct ResScoreArgs {
  const uint32_t *uidw, *headw;
  const int32_t *first_run, *run_feat, *wg_user_ptr, *wg_fill;
  const int2 *user_desc;
  int umax, KS;
  const double *Vt, *w;
  double w0;
  const double *w0p;  // non-null: the intercept is read from device memory (see ResArgs::scal)
  const double *y_slots;
  double *e_slots;
  double2 *sums;  // [G] {sum e, sum e^2}
  int dbg;        // timing experiments (wrong results): 1 no item-row fetch, 2 no user-row reads, 4 one user row for all
  unsigned long long *prof;  // (MFM_RES_SCORE_PROF) [G][4] s_memrealtime at start / users staged / slots done / end
};

template <int NT, int NG, int KPT>
__global__ __launch_bounds__(NT) void k_res_score(ResScoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) char res_smem[];
  constexpr int R = 16 * NG, NW = NT / WAVE;
  constexpr int US = 2 * KPT + 2;  // doubles per user row in LDS (16-byte aligned, rows shifted by 4 banks)
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int KP = a.KS >> 1;
  const double w0_k = a.w0p ? *a.w0p : a.w0;
  double *uV = (double *)res_smem;         // [umax][US]
  double *uW = uV + (size_t)a.umax * US;   // [umax]
  double2 *wsum = (double2 *)(uW + ((a.umax + 1) & ~1));  // [NW]
  if (a.prof && tid == 0) a.prof[4 * g] = __builtin_amdgcn_s_memrealtime();
  int *ucol = (int *)(wsum + NW);  // [umax] feature of the workgroup's u-th user
  const int u0 = a.wg_user_ptr[g], nu = a.wg_user_ptr[g + 1] - u0;
  for (int ul = tid; ul < a.umax; ul += NT) {  // (first the features: the row copy below then depends on LDS only)
    const int ju = ul < nu ? a.user_desc[u0 + ul].x : -1;
    ucol[ul] = ju;
    uW[ul] = ju >= 0 ? a.w[ju] : 0.0;
  }
  __syncthreads();
#pragma unroll 4
  for (int i = tid; i < a.umax * KPT; i += NT) {
    const int ul = i / KPT, pr = i - ul * KPT;
    const int ju = ucol[ul];
    double2 v = make_double2(0.0, 0.0);
    if (ju >= 0 && pr < KP) v = ((const double2 *)(a.Vt + (int64_t)ju * a.KS))[pr];
    ((double2 *)(uV + (size_t)ul * US))[pr] = v;
  }
  // static per-thread words (as in k_mf_resident)
  res_u4_t uw[NG];
  unsigned ux[NG], hbv[NG];
#pragma unroll
  for (int j = 0; j < NG; j++) {
#pragma unroll
    for (int w = 0; w < 4; w++) uw[j][w] = a.uidw[((int64_t)g * (5 * NG) + 5 * j + w) * NT + tid];
    ux[j] = a.uidw[((int64_t)g * (5 * NG) + 5 * j + 4) * NT + tid];
    hbv[j] = a.headw[((int64_t)g * NG + j) * NT + tid];
  }
  int run = a.first_run[g * NT + tid];
  const int64_t fill = a.wg_fill[g];
  __syncthreads();
  if (a.prof && tid == 0) a.prof[4 * g + 1] = __builtin_amdgcn_s_memrealtime();
  // (measured: reading the item's row again for every slot by straight-line code, two slots in flight, is twice as slow as
  //  this divergent fetch at the run heads -- the texture path pays per cache line touched, and 64 lanes on 64 different
  //  rows touch 64 lines per instruction)
  double2 vi[KPT];
  double wi = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int p = 0; p < KPT; p++) vi[p] = make_double2(0.0, 0.0);
  // the features of a batch's runs are requested one batch ahead by straight-line code (the run of every slot follows from
  // the head bits alone): the divergent fetch at a run head is then ONE round trip (the row), not two
  int jc[4], jn[4];
  double yn[4];
#pragma unroll
  for (int k = 0; k < 4; k++) yn[k] = a.y_slots[((int64_t)g * R + k) * NT + tid];
  {
    const unsigned h0 = hbv[0] & 15u;
    int r = run;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (k > 0) r += (int)((h0 >> k) & 1u);
      jc[k] = a.run_feat[r];
    }
    run = r;  // (the run of the last slot whose feature has been requested)
  }
#pragma unroll
  for (int j = 0; j < NG; j++) {
    const unsigned hb2 = hbv[j] | (j + 1 < NG ? hbv[j + 1 < NG ? j + 1 : j] << 16 : 0u);
#pragma unroll 1
    for (int bb = 0; bb < 4; bb++) {
      {
        const unsigned hn = (hb2 >> (4 * (bb + 1))) & 15u;
        int r = run;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          r += (int)((hn >> k) & 1u);
          jn[k] = a.run_feat[r];
        }
        run = r;
      }
      const unsigned lo_ = uw[j][bb];
      int uid[4];
      uid[0] = (int)(lo_ & 0x3ffu);
      uid[1] = (int)((lo_ >> 10) & 0x3ffu);
      uid[2] = (int)((lo_ >> 20) & 0x3ffu);
      uid[3] = (int)((lo_ >> 30) | (((ux[j] >> (8 * bb)) & 0xffu) << 2));
      const unsigned h4 = (hbv[j] >> (4 * bb)) & 15u;
      // the residual's other operand is requested first: it does not depend on anything computed here
      double yv[4], eb[4];
#pragma unroll
      for (int k = 0; k < 4; k++) yv[k] = yn[k];
      {  // (the next batch's targets, one batch ahead like its features; past the last slot: the first ones again)
        const int nb = 16 * j + 4 * bb + 4 < R ? 16 * j + 4 * bb + 4 : 0;
#pragma unroll
        for (int k = 0; k < 4; k++) yn[k] = a.y_slots[((int64_t)g * R + nb + k) * NT + tid];
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int slot = 16 * j + 4 * bb + k;
        const bool head = slot == 0 || ((h4 >> k) & 1u);
        if (head) {
          const int ji = jc[k];  // (-1: the pad run)
          wi = 0.0;
          if (ji >= 0 && !(a.dbg & 1)) {
            const double2 *src = (const double2 *)(a.Vt + (int64_t)ji * a.KS);
#pragma unroll
            for (int p = 0; p < KPT; p++) vi[p] = p < KP ? src[p] : make_double2(0.0, 0.0);
            wi = a.w[ji];
          } else {
#pragma unroll
            for (int p = 0; p < KPT; p++) vi[p] = make_double2(0.0, 0.0);
          }
        }
        const double2 *vu = (const double2 *)(uV + (size_t)((a.dbg & 4) ? 0 : uid[k]) * US);
        double d0 = 0.0, d1 = 0.0;  // two chains: even / odd pairs
#pragma unroll
        for (int p = 0; p < KPT; p += 2) {
          if (a.dbg & 2) break;
          const double2 x0 = vu[p], x1 = vu[p + 1];
          d0 += x0.x * vi[p].x + x0.y * vi[p].y;
          d1 += x1.x * vi[p + 1].x + x1.y * vi[p + 1].y;
        }
        const double pred = w0_k + (uW[uid[k]] + wi) + (d0 + d1);
        const bool real = (int64_t)tid * R + slot < fill;
        const double e = real ? pred - yv[k] : 0.0;  // (pads stay finite: 0 * NaN would poison the sweeps' statistics)
        eb[k] = e;
        s1 += e;
        s2 += e * e;
      }
      // (the batch's stores together, behind its last fetch: the wave has ONE memory counter, and a wait for a row that was
      //  requested after a store also waits for that store's acknowledgement)
#pragma unroll
      for (int k = 0; k < 4; k++) a.e_slots[((int64_t)g * R + 16 * j + 4 * bb + k) * NT + tid] = eb[k];
#pragma unroll
      for (int k = 0; k < 4; k++) jc[k] = jn[k];
    }
  }
  if (a.prof && lane == 0) a.prof[4 * g + 2] = __builtin_amdgcn_s_memrealtime();  // (the last wave to finish wins)
  wave_allreduce_sum2(s1, s2);
  if (lane == 0) wsum[wv] = make_double2(s1, s2);
  __syncthreads();
  if (a.prof && tid == 0) a.prof[4 * g + 3] = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) {
    double t1 = 0.0, t2 = 0.0;
    for (int w = 0; w < NW; w++) {
      t1 += wsum[w].x;
      t2 += wsum[w].y;
    }
    a.sums[g] = make_double2(t1, t2);
  }
}

// y in slot order (pads: 0), once per plan
__global__ void k_res_permute_y(const double *__restrict__ y, const int32_t *__restrict__ perm, int64_t n_slots,
                                double *__restrict__ y_slots) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_slots) return;
  const int32_t row = perm[i];
  y_slots[i] = row >= 0 ? y[row] : 0.0;
}

// host threads for the layout's per-workgroup work
template <class F>
static inline void res_parallel_for(int n, F f) {
  const int hw = (int)std::thread::hardware_concurrency();
  const int T = std::max(1, std::min({n, 16, hw > 0 ? hw : 1}));
  if (T == 1) {
    for (int i = 0; i < n; i++) f(i);
    return;
  }
  std::atomic<int> next(0);
  auto work = [&]() {
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n) return;
      f(i);
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < T; t++) pool.emplace_back(work);
  work();
  for (auto &t : pool) t.join();
}

struct ResPlan {
  bool ready = false;
  int G = 0, NT = 512, RV = 0, RL = 0, umax = 0, item_bits = 0, n_items = 0;
  int RX = 0;  // slots per thread beyond the on-chip ones (a multiple of 16): their residual stays in e_slots (k_mf_resident<.., OVF>)
  int R() const { return RV + RL + RX; }
  bool allow_overflow = true;  // (false: row-sharded -- the exchange variant of the kernel has no overflow form)
  int64_t n_rows = 0, n_runs = 0;  // (workgroup, item) pairs
  size_t lds_bytes = 0;
  DevBuf<int32_t> perm, first_run, wg_run_ptr, wg_nruns, wg_user_ptr, wg_item_ptr, ent_ptr, scols;
  DevBuf<int2> entries;
  DevBuf<uint32_t> uidw, headw;
  DevBuf<int32_t> run_item;
  DevBuf<int2> user_desc, item_desc;
  DevBuf<double> partials, dv, e_slots;
  DevBuf<unsigned long long> bar;
  // the slot-order scorer (k_res_score)
  DevBuf<int32_t> run_feat, wg_fill;  // feature of every run (-1: pad run); rows of every workgroup
  DevBuf<double> y_slots;             // y in slot order, built at the first scoring
  DevBuf<double2> sums;               // [G] {sum e, sum e^2} of the last scoring
  int maxu_rows = 0;                  // users of the largest workgroup (+ the pad user)
  // row-sharded: this rank's exchange array and flags (allocated with the layout), every rank's (mfm_peer_set), sweeps so far
  int xworld = 1, xrank = 0;
  bool peers_set = false;
  DevBuf<double> xsum;
  DevBuf<unsigned long long> xflag, xarrive;
  double *peer_sum[RES_MAX_PEERS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  unsigned long long *peer_flag[RES_MAX_PEERS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  unsigned long long xepoch = 0;
  bool peers_model = false;  // the peers' w / V are known too: first-level coefficients are written to every replica in the launch
  double *peer_w[RES_MAX_PEERS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  double *peer_V[RES_MAX_PEERS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  std::vector<void *> peer_mapped;  // IPC mappings to close (mfm_peer_import)
  size_t xsum_bytes(int world) const { return (size_t)world * 2 * 2 * ((size_t)n_items + 1) * sizeof(double); }
  void alloc_exchange(int world, int rank, hipStream_t s) {
    xworld = world;
    xrank = rank;
    peers_set = false;
    peers_model = false;
    xepoch = 0;
    // (what other GPUs write and this one reads inside a running kernel: uncached device memory, coherent for every agent. No
    //  fall-back to ordinary cached memory: a stale line in this GPU's L2 would be read as a peer's sum. A refusal throws, the
    //  layout is not taken and every rank runs the per-factor passes.)
    auto alloc_shared = [&](auto &buf, size_t count) {
      buf.release();
      void *p = nullptr;
      const size_t bytes = count * sizeof(*buf.p);
      if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        throw Error(MFM_ERR_DEVICE, "no uncached device memory for the ranks' exchange buffers (hipExtMallocWithFlags(hipDeviceMallocUncached))");
      }
      buf.p = (decltype(buf.p))p;
      buf.n = count;
      buf.owned = true;
      MFM_HIP_CHECK(hipMemsetAsync(p, 0, bytes, s));
    };
    alloc_shared(xsum, xsum_bytes(world) / sizeof(double));
    alloc_shared(xflag, (size_t)16 * RES_MAX_PEERS);
    xarrive.alloc_zero(16, s);
  }
  std::string why;  // why the layout was not built (diagnostics)
  std::vector<int32_t> h_nruns;  // runs per workgroup (diagnostics)
  std::vector<std::string> h_diag;  // per workgroup (MFM_RES_PROF only)

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

  // workgroups: contiguous user ranges (ustart: first row of every user in row order, ustart[n_users] = N) of at most cap rows;
  // the smallest variant that fits the device. Sets G, RV, RL and the user ordinal boundaries ucut[0 .. G].
  bool choose_layout(int64_t N, const std::vector<int64_t> &ustart, int64_t max_user, int n_cu, std::vector<int64_t> &ucut) {
    const int64_t n_users = (int64_t)ustart.size() - 1;
    int nv = 0;
    const Variant *vs = variants(nv);
    bool found = false;
    for (int vi = 0; vi < nv && !found; vi++) {
      const int64_t cap = (int64_t)NT * (vs[vi].rv + vs[vi].rl) - 1;  // (at least one pad slot closes the last run)
      if (max_user > cap) continue;
      const int64_t slack = std::min<int64_t>(cap / 8, max_user);
      int64_t Gw = (N + (cap - slack) - 1) / (cap - slack);
      if (Gw > n_cu && N <= (int64_t)n_cu * cap) Gw = n_cu;  // (tight: the cuts below decide whether it fits)
      if (Gw > n_cu) continue;
      if (const char *e = std::getenv("MFM_RES_WGS")) Gw = std::min<int64_t>(n_cu, std::max<int64_t>(Gw, std::atoll(e)));
      Gw = std::min<int64_t>(Gw, n_users);
      // cut g at the user boundary nearest to g N / G, never beyond cap rows
      ucut.assign(1, 0);
      bool ok = true;
      for (int64_t g = 1; g <= Gw && ok; g++) {
        const int64_t lo_u = ucut.back();
        int64_t hi_u;
        if (g == Gw) {
          hi_u = n_users;
        } else {
          const int64_t want = (N * g) / Gw;
          hi_u = std::upper_bound(ustart.begin(), ustart.end(), want) - ustart.begin() - 1;  // last boundary <= want
          if (hi_u + 1 <= n_users && ustart[hi_u + 1] - want < want - ustart[hi_u]) hi_u++;
          hi_u = std::max(hi_u, lo_u + 1);
          // leave at least one user for each of the remaining workgroups
          hi_u = std::min<int64_t>(hi_u, n_users - (Gw - g));
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
    // Tables beyond the on-chip capacity (512 x 80 slots per CU): the largest variant plus RX slots per thread whose residual is
    // streamed from / to the slot-ordered buffer every sweep (k_mf_resident<.., OVF>), up to twice the capacity -- beyond that the
    // per-factor passes. Not row-sharded (the exchange variant has no overflow form), not with MFM_RES_EAGER_STORE.
    RX = 0;
    if (!found && allow_overflow && !std::getenv("MFM_RES_EAGER_STORE") && !std::getenv("MFM_RES_NO_OVERFLOW")) {
      const Variant big = vs[nv - 1];
      for (int rx = 16; rx <= 80 && !found; rx += 16) {
        const int64_t cap = (int64_t)NT * (big.rv + big.rl + rx) - 1;
        if (max_user > cap || N > (int64_t)n_cu * cap) continue;
        int64_t Gw = std::min<int64_t>(n_cu, n_users);
        ucut.assign(1, 0);
        bool ok = true;
        for (int64_t g = 1; g <= Gw && ok; g++) {
          const int64_t lo_u = ucut.back();
          int64_t hi_u;
          if (g == Gw) {
            hi_u = n_users;
          } else {
            const int64_t want = (N * g) / Gw;
            hi_u = std::upper_bound(ustart.begin(), ustart.end(), want) - ustart.begin() - 1;
            if (hi_u + 1 <= n_users && ustart[hi_u + 1] - want < want - ustart[hi_u]) hi_u++;
            hi_u = std::max(hi_u, lo_u + 1);
            hi_u = std::min<int64_t>(hi_u, n_users - (Gw - g));
            while (hi_u > lo_u + 1 && ustart[hi_u] - ustart[lo_u] > cap) hi_u--;
          }
          if (hi_u <= lo_u || ustart[hi_u] - ustart[lo_u] > cap) ok = false;
          ucut.push_back(hi_u);
        }
        if (!ok) continue;
        G = (int)Gw;
        RV = big.rv;
        RL = big.rl;
        RX = rx;
        found = true;
      }
    }
    if (!found) return fail("no variant fits (rows per CU, or a first-level column longer than a workgroup's capacity)");
    return true;
  }
  // item draw: contiguous item ranges of about equal cost (partials + a constant per item), at most NT items each (a thread per
  // item). slot_ptr: prefix sum of the (workgroup, item) runs per item
  bool choose_item_slices(const std::vector<int32_t> &slot_ptr, int64_t counter, std::vector<int32_t> &iptr) {
    iptr.assign((size_t)G + 1, 0);
    bool ok = false;
    for (double scale = 1.0; scale < 3.0 && !ok; scale *= 1.1) {
      const double target = scale * ((double)counter + 16.0 * n_items) / G;
      double acc = 0.0;
      int gq = 0, cnt = 0;
      for (int c = 0; c < n_items; c++) {
        acc += (double)(slot_ptr[c + 1] - slot_ptr[c]) + 16.0;
        cnt++;
        if ((acc >= target || cnt == NT) && gq + 1 < G) {
          iptr[++gq] = c + 1;
          acc = 0.0;
          cnt = 0;
        }
      }
      ok = cnt <= NT;
      for (gq++; gq <= G; gq++) iptr[gq] = n_items;
    }
    if (!ok) return fail("more second-level columns than the workgroups can draw");
    return true;
  }
  // the workgroups' run ranges: global run index = run_base[g] + local; every workgroup ends with its pad run (never stored)
  static std::vector<int32_t> run_bases(const std::vector<int32_t> &nruns) {
    const int G = (int)nruns.size();
    std::vector<int32_t> run_base((size_t)G + 1, 0);
    const int run_align = std::getenv("MFM_RES_RUN_ALIGN") ? std::max(1, std::atoi(std::getenv("MFM_RES_RUN_ALIGN"))) : 1;
    const int run_skew = std::getenv("MFM_RES_RUN_SKEW") ? std::atoi(std::getenv("MFM_RES_RUN_SKEW")) : 0;
    for (int g = 0; g < G; g++) {
      int64_t nb = run_base[g] + nruns[g] + 1;
      nb = (nb + run_align - 1) / run_align * run_align + (run_align > 1 ? (int64_t)run_skew * ((g + 1) % 8) : 0);
      run_base[g + 1] = (int32_t)nb;
    }
    return run_base;
  }

  // csc = X_t of the table (column j: ascending rows), level[j] in {0, 1}: 0 = first field (contiguous row ranges in
  // ascending order covering every row once), 1 = second field (every row once). Unit values.
  // draw_empty (row-sharded): level-0 columns without rows HERE are drawn by this rank only when flagged (= without rows on
  // every rank: all ranks draw them alike); the others belong to the rank that holds their rows
  bool build(const HostCsr &csc, const std::vector<int32_t> &level, const std::vector<int32_t> *group_of, int n_cu,
             const std::vector<char> *draw_empty = nullptr) {
    ready = false;
    const int64_t N = csc.cols, D0 = csc.rows;
    n_rows = N;
    if (N < 1 || n_cu < 1 || (int64_t)level.size() != D0) return fail("shape");
    std::vector<int32_t> users, items, empties;
    for (int64_t j = 0; j < D0; j++) {
      if (level[j] == 0) {
        if (csc.ptr[j + 1] > csc.ptr[j])
          users.push_back((int32_t)j);
        else if (!draw_empty || (*draw_empty)[(size_t)j])
          empties.push_back((int32_t)j);
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
      for (int c = 0; c < n_items; c++) cnt += csc.ptr[items[c] + 1] - csc.ptr[items[c]];
      if (cnt != N) return fail("second level does not cover the rows");
      const int chunks = std::max(1, std::min(n_items, 64));
      res_parallel_for(chunks, [&](int k) {
        for (int c = (int)((int64_t)n_items * k / chunks); c < (int)((int64_t)n_items * (k + 1) / chunks); c++)
          for (int64_t p = csc.ptr[items[c]]; p < csc.ptr[items[c] + 1]; p++) item_of_row[csc.idx[p]] = c;
      });
      // (N entries and no row left out: every row exactly once)
      std::atomic<bool> bad(false);
      res_parallel_for(chunks, [&](int k) {
        for (int64_t r = N * k / chunks; r < N * (k + 1) / chunks; r++)
          if (item_of_row[r] < 0) bad = true;
      });
      if (bad) return fail("second level touches a row twice");
    }
    std::vector<int64_t> ucut;  // user ordinal boundaries of the workgroups
    if (!choose_layout(N, ustart, max_user, n_cu, ucut)) return false;
    const int R = RV + RL + RX;
    const int64_t cap_slots = (int64_t)NT * R;
    // users per workgroup (+ the never-occurring ones, dealt round-robin), the pad user
    std::vector<std::vector<int32_t>> wg_users((size_t)G);
    for (int g = 0; g < G; g++)
      for (int64_t u = ucut[g]; u < ucut[g + 1]; u++) wg_users[g].push_back(users[u]);
    for (size_t k = 0; k < empties.size(); k++) wg_users[k % G].push_back(empties[k]);
    int maxu = 0;
    for (int g = 0; g < G; g++) maxu = std::max(maxu, (int)wg_users[g].size());
    if (maxu > NT) return fail("more first-level columns in a workgroup than threads");
    item_bits = bits_for((int64_t)n_items + 1);
    // slots: per workgroup the rows in (item, row) order
    // users 10 bits each (16 slots in 5 words), head bits 16 per word (one word per group of 16 slots); every slot starts
    // as a pad (user field 0: rewritten below once umax is known)
    const int UW = 5 * (R / 16), HW = R / 16;
    std::vector<uint16_t> h_uid((size_t)G * cap_slots, 0xffffu);  // [g][slot]: user within the workgroup, 0xffff: pad
    std::vector<uint32_t> h_headw((size_t)G * HW * NT, 0);
    std::vector<int32_t> h_perm((size_t)G * cap_slots, -1), h_first((size_t)G * NT, 0);
    std::vector<std::vector<int32_t>> wg_run_item((size_t)G);
    std::vector<int64_t> fill((size_t)G, 0);
    std::vector<int32_t> nruns((size_t)G, 0);
    // per workgroup (its rows are one contiguous range): the rows in (item, row) order by a counting sort, run starts,
    // slot -> row, slot -> user, the first run of every thread. Workgroups are independent: host threads.
    res_parallel_for(G, [&](int g) {
      const int64_t row0 = ustart[ucut[g]], row1 = ustart[ucut[g + 1]];
      const int64_t n = row1 - row0;
      std::vector<int32_t> cnt((size_t)n_items + 1, 0), uloc((size_t)n);
      for (int64_t u = ucut[g]; u < ucut[g + 1]; u++)
        for (int64_t r = ustart[u]; r < ustart[u + 1]; r++) uloc[(size_t)(r - row0)] = (int32_t)(u - ucut[g]);
      for (int64_t r = row0; r < row1; r++) cnt[(size_t)item_of_row[r] + 1]++;
      std::vector<int32_t> &runs = wg_run_item[g];
      for (int c = 0; c < n_items; c++) {
        if (cnt[c + 1]) {  // a run of the workgroup starts at slot cnt[c] (after the prefix sum)
          const int64_t sidx = cnt[c];
          const int th = (int)(sidx / R), rh = (int)(sidx % R);
          h_headw[((size_t)g * HW + rh / 16) * NT + th] |= 1u << (rh % 16);
          runs.push_back(c);
        }
        cnt[c + 1] += cnt[c];
      }
      // the run containing every thread's first slot: the last run that starts at or before it
      {
        size_t ri = 0;
        std::vector<int32_t> run_start(runs.size());
        for (size_t k = 0; k < runs.size(); k++) run_start[k] = cnt[runs[k]];
        for (int t = 0; t < NT; t++) {
          const int64_t s0 = (int64_t)t * R;
          if (s0 >= n) break;
          while (ri + 1 < runs.size() && run_start[ri + 1] <= s0) ri++;
          h_first[(size_t)g * NT + t] = (int32_t)ri;
        }
      }
      for (int64_t r = row0; r < row1; r++) {  // rows ascending: within an item the slots keep the row order
        const int64_t sidx = cnt[(size_t)item_of_row[r]]++;
        const int t = (int)(sidx / R), rr = (int)(sidx % R);
        h_perm[(size_t)g * cap_slots + (size_t)rr * NT + t] = (int32_t)r;
        h_uid[(size_t)g * cap_slots + (size_t)sidx] = (uint16_t)uloc[(size_t)(r - row0)];
      }
      fill[g] = n;
      nruns[g] = (int32_t)runs.size();
    });
    std::vector<int32_t> h_slot_ptr((size_t)n_items + 1, 0);  // (workgroup, item) runs per item, as a prefix sum
    int64_t counter = 0;
    for (int g = 0; g < G; g++) {
      counter += nruns[g];
      for (int32_t c : wg_run_item[g]) h_slot_ptr[(size_t)c + 1]++;
    }
    for (int c = 0; c < n_items; c++) h_slot_ptr[(size_t)c + 1] += h_slot_ptr[c];
    h_slot_ptr[n_items] = (int32_t)counter;
    n_runs = counter;
    if (counter >= ((int64_t)1 << 31) - 2) return fail("too many runs");
    // runs, workgroup-major
    std::vector<int32_t> run_base = run_bases(nruns);
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
    std::vector<int32_t> h_iptr;
    if (!choose_item_slices(h_slot_ptr, counter, h_iptr)) return false;
    int imax = 0;
    for (int g = 0; g < G; g++) imax = std::max(imax, h_iptr[g + 1] - h_iptr[g]);
    umax = std::max(maxu, imax) + 1;  // stride of the per-wave accumulator arrays; the pad user is umax - 1
    if (umax > 1024) return fail("internal: user field");
    std::vector<uint32_t> h_uidw((size_t)G * UW * NT, 0);
    res_parallel_for(G, [&](int g) {
      for (int64_t sl = 0; sl < cap_slots; sl++) {
        const uint16_t u16 = h_uid[(size_t)g * cap_slots + (size_t)sl];
        const uint32_t u = u16 == 0xffffu ? (uint32_t)(umax - 1) : u16;
        const int t = (int)(sl / R), r = (int)(sl % R);
        // word bb of a group: users 0 .. 2 of batch bb and the low 2 bits of user 3; word 4: byte bb = its high 8 bits
        const int w0 = 5 * (r / 16), bb = (r % 16) / 4, k = r % 4;
        if (k < 3) {
          h_uidw[((size_t)g * UW + w0 + bb) * NT + t] |= u << (10 * k);
        } else {
          h_uidw[((size_t)g * UW + w0 + bb) * NT + t] |= (u & 3u) << 30;
          h_uidw[((size_t)g * UW + w0 + 4) * NT + t] |= (u >> 2) << (8 * bb);
        }
      }
    });
    h_uid = std::vector<uint16_t>();
    // item draw tables: per slice its (run, item) pairs source workgroup by source workgroup (a source's runs are in item
    // order: the slice's share of them is a contiguous stretch of its partials), padded to whole 64-entry chunks
    std::vector<int32_t> slice_of((size_t)n_items);
    for (int g = 0; g < G; g++)
      for (int c = h_iptr[g]; c < h_iptr[g + 1]; c++) slice_of[c] = g;
    std::vector<int32_t> h_eptr((size_t)G + 1, 0);
    {
      std::vector<int64_t> cnt((size_t)G, 0);
      for (int g = 0; g < G; g++)
        for (int32_t c : wg_run_item[g]) cnt[slice_of[c]]++;
      for (int g = 0; g < G; g++) {
        const int64_t e = h_eptr[g] + ((cnt[g] + WAVE - 1) / WAVE) * WAVE;
        if (e >= ((int64_t)1 << 31)) return fail("too many entries");
        h_eptr[g + 1] = (int32_t)e;
      }
    }
    std::vector<int2> h_entries((size_t)h_eptr[G], make_int2(zero_run, 0));
    {
      std::vector<int64_t> pos((size_t)G);
      for (int g = 0; g < G; g++) pos[g] = h_eptr[g];
      for (int g = 0; g < G; g++)
        for (size_t l = 0; l < wg_run_item[g].size(); l++) {
          const int32_t c = wg_run_item[g][l];
          const int sgl = slice_of[c];
          h_entries[(size_t)pos[sgl]++] = make_int2(run_base[g] + (int32_t)l, c - h_iptr[sgl]);
        }
    }
    if (h_entries.size() >= ((size_t)1 << 31)) return fail("too many entries");
    lds_bytes = (size_t)RL * NT * 8 + (size_t)2 * (NT / WAVE) * umax * 8 + (size_t)umax * 16 + (size_t)(NT / WAVE) * 16 + (NT / WAVE) * 4 + 64;
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
    wg_run_ptr.upload(run_base);
    wg_nruns.upload(nruns);
    e_slots.alloc((size_t)G * cap_slots);
    {
      std::vector<int32_t> h_run_feat(h_run_item.size() + 8, -1);  // (the scorer's look-ahead reads a few runs past the last)
      for (size_t r = 0; r < h_run_item.size(); r++) h_run_feat[r] = h_run_item[r] < n_items ? items[(size_t)h_run_item[r]] : -1;
      run_feat.upload(h_run_feat);
      std::vector<int32_t> h_fill((size_t)G);
      for (int g = 0; g < G; g++) h_fill[g] = (int32_t)fill[g];
      wg_fill.upload(h_fill);
      sums.alloc((size_t)G);
      y_slots = DevBuf<double>();
    }
    h_nruns = nruns;
    if (std::getenv("MFM_RES_PROF")) {
      h_diag.assign((size_t)G, "");
      for (int g = 0; g < G; g++) {
        int64_t maxlen = 0;
        for (int64_t u = ucut[g]; u < ucut[g + 1]; u++) maxlen = std::max(maxlen, ustart[u + 1] - ustart[u]);
        // store instructions of sweep B: (wave, slot) pairs in which some lane closes a run; lanes that store per thread
        int64_t sinstr = 0, maxst = 0, nthr_head = 0;
        for (int w = 0; w < NT / WAVE; w++) {
          std::vector<char> hh(WAVE, 0);
          std::vector<int> st(WAVE, 0);
          for (int r = 0; r < R; r++) {
            bool any = false;
            for (int l = 0; l < WAVE; l++) {
              const int t = w * WAVE + l;
              const bool head = (h_headw[((size_t)g * HW + r / 16) * NT + t] >> (r % 16)) & 1u;
              if (head && hh[l]) {
                any = true;
                st[l]++;
              }
              hh[l] = hh[l] || head;
            }
            sinstr += any;
          }
          for (int l = 0; l < WAVE; l++) {
            maxst = std::max<int64_t>(maxst, st[l]);
            nthr_head += hh[l];
          }
        }
        char buf[160];
        std::snprintf(buf, sizeof buf, "users %d maxlen %lld rows %lld store-instr %lld max-stores/thread %lld threads-with-head %lld",
                      (int)(ucut[g + 1] - ucut[g]), (long long)maxlen, (long long)fill[g], (long long)sinstr, (long long)maxst,
                      (long long)nthr_head);
        h_diag[g] = buf;
      }
    }
    scols.upload(items);
    {
      std::vector<int2> h_idesc(items.size());
      for (size_t i = 0; i < items.size(); i++) {
        const int32_t j = items[i];
        h_idesc[i] = make_int2(j, group_of && (size_t)j < group_of->size() ? (*group_of)[j] : 0);
      }
      item_desc.upload(h_idesc.data(), h_idesc.size());
    }
    partials.alloc((size_t)2 * ((size_t)zero_run + 1));
    MFM_HIP_CHECK(hipMemset(partials.p, 0, (size_t)16 * ((size_t)zero_run + 1)));
    dv.alloc((size_t)2 * (n_items + 1));
    bar.alloc(RES_BAR_WORDS);
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

// resident workgroups of the plan's kernel variant per CU (0: it does not fit): the grid barrier needs all G at once
static inline hipError_t res_occupancy(const ResPlan &rp, int *per_cu) {
  *per_cu = 0;
  const void *fn = nullptr;
  if (rp.RV == 16 && rp.RL == 0)
    fn = (const void *)k_mf_resident<512, 1, 0>;
  else if (rp.RV == 32 && rp.RL == 0)
    fn = (const void *)k_mf_resident<512, 2, 0>;
  else if (rp.RV == 64 && rp.RL == 16)
    fn = rp.RX ? (const void *)k_mf_resident<512, 4, 1, false, true> : (const void *)k_mf_resident<512, 4, 1>;
  else
    return hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return e;
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, fn, 512, rp.lds_bytes);
}

// update_V of factors [f_begin, f_end) in one launch. zbase: variates of factor f_begin (factor f at + (f - f_begin) D).
static inline void run_sweep_resident(hipStream_t s, Timing &tm, ResPlan &rp, int kernel_class, double2 *eq, double *V, int64_t D,
                                      int f_begin, int f_end, const double *zbase, const double *lam, const double *mu,
                                      const int32_t *group, int n_groups, double alpha, int *error, bool lazy_store,
                                      double *w = nullptr, const double *zw = nullptr, const double *lam_w = nullptr,
                                      const double *mu_w = nullptr, double e_shift = 0.0, bool load_slots = false,
                                      bool no_store = false, const double *scal = nullptr) {
  ResArgs a;
  std::memset(&a, 0, sizeof(a));
  a.scal = scal;
  a.eq = eq;
  a.e_slots = (lazy_store || rp.RX) ? rp.e_slots.p : nullptr;  // (overflow slots live there)
  a.ngx = rp.RX / 16;
  a.no_store = no_store ? 1 : 0;
  a.e_in = load_slots ? rp.e_slots.p : nullptr;
  a.linear = w ? 1 : 0;
  a.n_sw = f_end - f_begin + a.linear;
  a.w = w;
  a.zw = zw;
  a.lam_w = lam_w;
  a.mu_w = mu_w;
  a.e_shift = e_shift;
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
  a.wg_run_ptr = rp.wg_run_ptr.p;
  a.wg_nruns = rp.wg_nruns.p;
  a.scols = rp.scols.p;
  a.item_desc = rp.item_desc.p;
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
  a.n_items = rp.n_items;
  a.umax = rp.umax;
  a.bar = rp.bar.p;
  a.n_wg = rp.G;
  a.error = error;
  a.dbg = std::getenv("MFM_RES_DBG") ? std::atoi(std::getenv("MFM_RES_DBG")) : 0;
  a.rot = std::getenv("MFM_RES_ROT") ? std::atoi(std::getenv("MFM_RES_ROT")) : 0;
  const bool xch = rp.xworld > 1;
  if (xch && rp.xrank == 1 && std::getenv("MFM_RES_XCH_BREAK")) a.dbg |= 8192;  // tests: rank 1 never raises its flags
  a.xworld = rp.xworld;
  a.xrank = rp.xrank;
  a.xepoch0 = rp.xepoch;
  a.xarrive = rp.xarrive.p;
  a.xmodel = xch && rp.peers_model ? 1 : 0;
  for (int r = 0; r < RES_MAX_PEERS; r++) {
    a.xsum[r] = rp.peer_sum[r];
    a.xflag[r] = rp.peer_flag[r];
    a.xw[r] = rp.peer_w[r];
    a.xV[r] = rp.peer_V[r];
  }
  if (xch && !rp.peers_set) throw Error(MFM_ERR_RUNTIME, "row-sharded persistent sweep: the peers' exchange buffers are not set (mfm_peer_set)");
  // MFM_RES_PROF=n: the n-th launch of the process records the phase stamps of every workgroup and prints a summary
  static int prof_launch = 0;
  const bool prof = std::getenv("MFM_RES_PROF") && ++prof_launch == std::atoi(std::getenv("MFM_RES_PROF"));
  DevBuf<unsigned long long> prof_buf;
  if (prof) {
    prof_buf.alloc((size_t)rp.G * a.n_sw * 64);
    MFM_HIP_CHECK(hipMemsetAsync(prof_buf.p, 0, (size_t)rp.G * a.n_sw * 512, s));
    a.prof = prof_buf.p;
  }
  const int K = a.n_sw;
  // algorithmic bytes of the launch: e read once (8 B) with its slot map (4 B) and the static slot words (11 bits), written
  // once (8 B: slot order, or scattered to eq); per sweep one 16-byte partial per (workgroup, item) run written and read,
  // its 8-byte list entry, its item read by both sweeps (2 x 4 B)
  double bytes = (8.0 + (load_slots ? 0.0 : 4.0) + 1.4 + (no_store ? 0.0 : 8.0)) * rp.n_rows + K * 48.0 * rp.n_runs;  // (slot order: no map; no_store: not written back)
  if (rp.RX) bytes += (double)K * (16.0 + 1.75) * (double)rp.G * rp.NT * rp.RX;  // overflow slots: residual read + written, static words, per sweep
  (void)lazy_store;
  hipLaunchKernelGGL(k_res_init_dv, dim3((rp.n_items + 256) / 256), dim3(256), 0, s,
                     w ? (const double *)nullptr : V + (int64_t)f_begin * D, rp.scols.p, rp.n_items,
                     rp.dv.p, rp.bar.p);
  TimedLaunch t(tm, s, kernel_class, bytes);
#define MFM_RES_LAUNCH(RV_, RL_)                                                                                              \
  do {                                                                                                                        \
    static DeviceOnce raised;                                                                                                 \
    if (raised.need()) {                                                                                                      \
      MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_mf_resident<512, RV_ / 16, RL_ / 16>,                                  \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                             \
      raised.mark();                                                                                                          \
    }                                                                                                                         \
    if (xch) {                                                                                                                \
      static DeviceOnce raised_x;                                                                                             \
      if (raised_x.need()) {                                                                                                  \
        MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_mf_resident<512, RV_ / 16, RL_ / 16, true>,                          \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                           \
        raised_x.mark();                                                                                                      \
      }                                                                                                                       \
      hipLaunchKernelGGL((k_mf_resident<512, RV_ / 16, RL_ / 16, true>), dim3(rp.G), dim3(512), rp.lds_bytes, s, a);            \
    } else {                                                                                                                  \
      hipLaunchKernelGGL((k_mf_resident<512, RV_ / 16, RL_ / 16>), dim3(rp.G), dim3(512), rp.lds_bytes, s, a);                  \
    }                                                                                                                         \
  } while (0)
  if (rp.RV == 16 && rp.RL == 0)
    MFM_RES_LAUNCH(16, 0);
  else if (rp.RV == 32 && rp.RL == 0)
    MFM_RES_LAUNCH(32, 0);
  else if (rp.RV == 64 && rp.RL == 16 && rp.RX > 0) {
    if (xch) throw Error(MFM_ERR_RUNTIME, "internal: no row-sharded resident kernel with overflow slots");
    static DeviceOnce raised_o;
    if (raised_o.need()) {
      MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_mf_resident<512, 4, 1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      raised_o.mark();
    }
    hipLaunchKernelGGL((k_mf_resident<512, 4, 1, false, true>), dim3(rp.G), dim3(512), rp.lds_bytes, s, a);
  } else if (rp.RV == 64 && rp.RL == 16)
    MFM_RES_LAUNCH(64, 16);
  else
    throw Error(MFM_ERR_RUNTIME, "internal: no resident kernel variant for this plan");
#undef MFM_RES_LAUNCH
  MFM_HIP_CHECK(hipGetLastError());
  if (xch) rp.xepoch += (unsigned long long)a.n_sw;  // (every rank runs the same launches: the epochs agree)
  if (prof) {
    MFM_HIP_CHECK(hipStreamSynchronize(s));
    const int K2 = a.n_sw;
    std::vector<unsigned long long> h((size_t)rp.G * K2 * 64);
    MFM_HIP_CHECK(hipMemcpy(h.data(), prof_buf.p, h.size() * 8, hipMemcpyDeviceToHost));
    // per phase: mean over (workgroup, factor) and mean over factors of the slowest / fastest workgroup, in us
    const char *names[6] = {"sweep A", "user draw", "sweep B", "barrier 1 (wait)", "item draw", "barrier 2 (wait)"};
    std::fprintf(stderr, "[resident profile] G=%d factors=%d; us per factor: mean | mean of max over workgroups | mean of min\n", rp.G, K2);
    for (int ph = 0; ph < 6; ph++) {
      double sm = 0, smax = 0, smin = 0;
      for (int f = 0; f < K2; f++) {
        double mx = 0, mn = 1e30;
        for (int g = 0; g < rp.G; g++) {
          const unsigned long long *t = &h[((size_t)g * K2 + f) * 64];
          const double d = (double)(t[ph + 1] - t[ph]) * 0.01;
          sm += d;
          mx = std::max(mx, d);
          mn = std::min(mn, d);
        }
        smax += mx;
        smin += mn;
      }
      std::fprintf(stderr, "  %-18s %8.2f | %8.2f | %8.2f\n", names[ph], sm / ((double)rp.G * K2), smax / K2, smin / K2);
    }
    {  // the workgroups that are slowest in sweep B
      std::vector<std::pair<double, int>> tb;
      for (int g = 0; g < rp.G; g++) {
        double d = 0;
        for (int f = 0; f < K2; f++) d += (double)(h[((size_t)g * K2 + f) * 64 + 3] - h[((size_t)g * K2 + f) * 64 + 2]) * 0.01;
        tb.push_back({d / K2, g});
      }
      std::sort(tb.begin(), tb.end());
      std::fprintf(stderr, "  sweep B by workgroup: fastest");
      for (int k = 0; k < 4; k++) std::fprintf(stderr, " wg %d %.1f (%d runs)", tb[k].second, tb[k].first, rp.h_nruns[tb[k].second]);
      std::fprintf(stderr, "; median wg %d %.1f (%d runs); slowest", tb[rp.G / 2].second, tb[rp.G / 2].first, rp.h_nruns[tb[rp.G / 2].second]);
      for (int k = rp.G - 4; k < rp.G; k++) std::fprintf(stderr, " wg %d %.1f (%d runs)", tb[k].second, tb[k].first, rp.h_nruns[tb[k].second]);
      std::fprintf(stderr, "\n");
      if (!rp.h_diag.empty())
        for (int k : {0, 1, rp.G / 2, rp.G - 3, rp.G - 2, rp.G - 1})
        {
          const int g = tb[k].second;
          std::fprintf(stderr, "    wg %d %.1f us: %s; waves:", g, tb[k].first, rp.h_diag[g].c_str());
          for (int w = 0; w < 8; w++) {
            double d = 0;
            for (int f = 0; f < K2; f++) d += (double)(h[((size_t)g * K2 + f) * 64 + 8 + w] - h[((size_t)g * K2 + f) * 64 + 2]) * 0.01;
            std::fprintf(stderr, " %.1f", d / K2);
          }
          std::fprintf(stderr, "\n      batches of waves 4 and 7 (factor 3 alone):");
          for (int w : {0, 20}) {
            std::fprintf(stderr, " |");
            for (int b = 0; b < (rp.RV + rp.RL) / 4; b++)
              std::fprintf(stderr, " %.1f", (double)(h[((size_t)g * K2 + 3) * 64 + 16 + w + b] - h[((size_t)g * K2 + 3) * 64 + 2]) * 0.01);
          }
          std::fprintf(stderr, "\n");
        }
    }
    if (const char *dump = std::getenv("MFM_RES_PROF_DUMP")) {  // one line per workgroup: phase means, then the diagnostics
      if (FILE *fp = std::fopen(dump, "w")) {
        for (int g = 0; g < rp.G; g++) {
          std::fprintf(fp, "%d", g);
          for (int ph = 0; ph < 6; ph++) {
            double d = 0;
            for (int f = 0; f < K2; f++) d += (double)(h[((size_t)g * K2 + f) * 64 + ph + 1] - h[((size_t)g * K2 + f) * 64 + ph]) * 0.01;
            std::fprintf(fp, " %.2f", d / K2);
          }
          std::fprintf(fp, " | %s\n", rp.h_diag.empty() ? "" : rp.h_diag[g].c_str());
        }
        std::fclose(fp);
      }
    }
    {
      const char *nm[4] = {"census barrier", "residual load", "factor loop", "residual store"};
      for (int ph = 0; ph < 4; ph++) {
        double sm = 0, mx = 0;
        for (int g = 0; g < rp.G; g++) {
          const double d = (double)(h[(size_t)g * K2 * 64 + 56 + ph + 1] - h[(size_t)g * K2 * 64 + 56 + ph]) * 0.01;
          sm += d;
          mx = std::max(mx, d);
        }
        std::fprintf(stderr, "  %-18s %8.2f mean | %8.2f max (us, whole launch)\n", nm[ph], sm / rp.G, mx);
      }
      unsigned long long t0 = ~0ull, t1 = 0;
      for (int g = 0; g < rp.G; g++) {
        t0 = std::min(t0, h[(size_t)g * K2 * 64 + 56]);
        t1 = std::max(t1, h[(size_t)g * K2 * 64 + 60]);
      }
      std::fprintf(stderr, "  first workgroup in to last workgroup out: %.1f us\n", (double)(t1 - t0) * 0.01);
    }
    double tot = 0;
    for (int g = 0; g < rp.G; g++) tot += (double)(h[((size_t)g * K2 + K2 - 1) * 64 + 6] - h[(size_t)g * K2 * 64]) * 0.01;
    std::fprintf(stderr, "  factors, start to end: %.1f us (%.2f per factor)\n", tot / rp.G, tot / rp.G / K2);
  }
}

// update_e in slot order (k_res_score). false: this plan / rank is not covered (the caller scores in row order).
static inline bool res_score_supported(const ResPlan &rp, int K) {
  const int KS = (K + 1) & ~1;
  if (!rp.ready || K < 1 || KS > 32) return false;
  const size_t lds = ((size_t)rp.umax * (2 * 16 + 2) + ((rp.umax + 1) & ~1)) * 8 + 8 * 16 + (size_t)rp.umax * 4 + 64;
  return lds <= 160 * 1024 - 512;
}

static inline void run_res_score(hipStream_t s, Timing &tm, ResPlan &rp, int kernel_class, const double *Vt, const double *w,
                                 double w0, int K, const double *y, int64_t nnz, const double *w0p = nullptr) {
  const int KS = (K + 1) & ~1;
  const int64_t n_slots = (int64_t)rp.G * rp.R() * 512;
  if (!rp.y_slots.p) {
    rp.y_slots.alloc((size_t)n_slots);
    hipLaunchKernelGGL(k_res_permute_y, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, s, y, rp.perm.p, n_slots,
                       rp.y_slots.p);
  }
  ResScoreArgs a;
  a.uidw = rp.uidw.p;
  a.headw = rp.headw.p;
  a.first_run = rp.first_run.p;
  a.run_feat = rp.run_feat.p;
  a.wg_user_ptr = rp.wg_user_ptr.p;
  a.wg_fill = rp.wg_fill.p;
  a.user_desc = rp.user_desc.p;
  a.umax = rp.umax;
  a.KS = KS;
  a.Vt = Vt;
  a.w = w;
  a.w0 = w0;
  a.w0p = w0p;
  a.y_slots = rp.y_slots.p;
  a.e_slots = rp.e_slots.p;
  a.sums = rp.sums.p;
  a.dbg = std::getenv("MFM_RES_SCORE_DBG") ? std::atoi(std::getenv("MFM_RES_SCORE_DBG")) : 0;
  static int prof_calls = 0;
  const bool prof = std::getenv("MFM_RES_SCORE_PROF") && ++prof_calls == std::atoi(std::getenv("MFM_RES_SCORE_PROF"));
  DevBuf<unsigned long long> prof_buf;
  a.prof = nullptr;
  if (prof) {
    prof_buf.alloc((size_t)rp.G * 4);
    a.prof = prof_buf.p;
  }
  const size_t lds = ((size_t)rp.umax * (2 * 16 + 2) + ((rp.umax + 1) & ~1)) * 8 + 8 * 16 + (size_t)rp.umax * 4 + 64;
  // algorithmic bytes: y and e in slot order (8 + 8 B / row), the static slot words (1.4 B), one Vt row per run and per user
  TimedLaunch t(tm, s, kernel_class, 17.4 * rp.n_rows + 8.0 * KS * (double)rp.n_runs);
  (void)nnz;
#define MFM_RES_SCORE(NG_)                                                                                              \
  do {                                                                                                                  \
    static DeviceOnce raised;                                                                                           \
    if (raised.need()) {                                                                                                \
      MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_res_score<512, NG_, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        160 * 1024));                                                                   \
      raised.mark();                                                                                                    \
    }                                                                                                                   \
    hipLaunchKernelGGL((k_res_score<512, NG_, 16>), dim3(rp.G), dim3(512), lds, s, a);                                   \
  } while (0)
  const int NG = rp.R() / 16;
  if (NG == 1)
    MFM_RES_SCORE(1);
  else if (NG == 2)
    MFM_RES_SCORE(2);
  else if (NG == 5)
    MFM_RES_SCORE(5);
  else if (NG == 6)
    MFM_RES_SCORE(6);
  else if (NG == 7)
    MFM_RES_SCORE(7);
  else if (NG == 8)
    MFM_RES_SCORE(8);
  else if (NG == 9)
    MFM_RES_SCORE(9);
  else if (NG == 10)
    MFM_RES_SCORE(10);
  else
    throw Error(MFM_ERR_RUNTIME, "internal: no slot-order scorer for this plan");
#undef MFM_RES_SCORE
  MFM_HIP_CHECK(hipGetLastError());
  if (prof) {
    MFM_HIP_CHECK(hipStreamSynchronize(s));
    std::vector<unsigned long long> h((size_t)rp.G * 4);
    MFM_HIP_CHECK(hipMemcpy(h.data(), prof_buf.p, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    double st = 0, lp = 0, en = 0;
    for (int g = 0; g < rp.G; g++) {
      t0 = std::min(t0, h[4 * g]);
      t1 = std::max(t1, h[4 * g + 3]);
      st += (double)(h[4 * g + 1] - h[4 * g]) * 0.01;
      lp += (double)(h[4 * g + 2] - h[4 * g + 1]) * 0.01;
      en += (double)(h[4 * g + 3] - h[4 * g + 2]) * 0.01;
    }
    std::fprintf(stderr, "[k_res_score] first start to last end %.1f us; per workgroup (mean): users staged %.1f, slots %.1f, reduction %.1f us\n",
                 (double)(t1 - t0) * 0.01, st / rp.G, lp / rp.G, en / rp.G);
  }
}

}  // namespace mfm
