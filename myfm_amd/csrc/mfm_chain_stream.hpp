// mfm_chain_stream.hpp -- k_cs_stream: the block-feature sweep of a LARGE relation block (FMTrainer.hpp:276-302 for w, :419-470
// for V; block state beyond one CU's LDS, columns that pairwise share rows) as ONE pipelined launch without batch boundaries.
// Plan and the exactness argument: mfm_chain_plan.hpp. This file: the device side.
//
// Workgroup 0 is the WALKER workgroup (512 threads, roles placed by SIMD -- wavefront w runs on SIMD w % 4):
//   wavefront 0       the walker: the columns in order over their hot entries, on the hot rows' records in LDS (raised priority;
//                     its SIMD mate, wavefront 4, stays idle -- a helper there was starved and became what the walker waited for)
//   wavefronts 1, 5   X: copy the records of rows that leave the LDS and the step's (old, new) pairs out; the two alternate steps
//   wavefronts 2,3,6,7  Y: stage the rows that enter, the cold statistics and the hot entry lists of the coming step; two pairs,
//                     alternating steps, each with the step's static data prefetched one own step ahead
// Workgroups 1..NB own a contiguous range of block rows each: wavefronts 0-3 (S, two pairs alternating steps) take the cold
// statistics of step v and pack the rows entering at v, wavefronts 4-7 (U, two pairs alternating steps: Lw >= 2) apply the cold
// updates of step u and put the leaving rows' records back. Nobody executes a workgroup barrier after the prologue: every
// wavefront runs its own loop over its steps and waits on monotonic per-wavefront counters only --
//   walker(s)  <-  Y(s)  <-  S(s) of every range (global flags) and X(s - RD), X(s - RD - 1) (LDS: slot reuse)
//   S(v)       <-  U(v - Lw) of the same workgroup (LDS)
//   U(u)       <-  X(u) (global counter)             X(u) <- walker(u) (LDS)
// Everything that crosses workgroups (ring slots of entering / leaving records, per-(range, wavefront) partial statistics, the
// (old, new) pairs, the flags; CS_RING steps deep) is written with agent-scope (write-through) stores, 16 bytes an instruction,
// drained with s_waitcnt vmcnt(0) before the flag, and read with 16-byte agent-scope loads; the block-row records themselves are
// only ever touched by their range's workgroup (one CU: its L1 sees its own stores). The sums have a fixed association: a second
// run is bit-identical. What bounds it (profiles/r05_m_cs_stream_timeline.txt): the walker at 0.9-1.0 us a column and the
// S -> Y -> walker -> X -> U -> S round trip of about 11 us over Lw steps are within 5% of each other at the planner's choice.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "mfm_chain_plan.hpp"
#include "mfm_policies.hpp"
#include "mfm_wave.hpp"

namespace mfm {

constexpr int CS_NT = 512;       // threads of every workgroup
constexpr int CS_NWS = 4;        // S wavefronts per range workgroup (the other four are U)
constexpr int CS_MAX_NB = 32;    // row ranges at most
constexpr int CS_SR = 4;         // LDS ring of per-step scalars in the walker (>= RD + 1)
constexpr int CS_U = 8;          // 64-entry rounds of a column's hot entries kept in registers
constexpr int CS_NX = 2, CS_NY = 4;
constexpr int CS_ER = 2;         // steps of hot entry lists in the walker's LDS (>= RD)
constexpr size_t CS_LDS_MAX = 156 * 1024;  // of the CU's 160 KiB

struct CsSync {  // zeroed before every launch
  unsigned long long walk_done;  // steps whose exits and (old, new) pairs are published
  unsigned long long pad0[15];
  unsigned long long s_flag[CS_MAX_NB * CS_NWS];  // per (range, S wavefront): steps whose partials and entering records are published
};

struct CsArgs {
  int n_cols, n_steps, Cg, Lw, NB, RD, max_enter, max_exit, n_slots;
  int ecap;  // hot entries of a column at most, rounded up to whole wavefronts
  int dbg;   // MFM_CS_DBG (experiments): 8 / 16 the walker sees no / at most 64 hot entries per column (wrong results: timing only)
  const int32_t *cols, *col_group;
  const int32_t *cold_ptr, *cold_rc;
  const double *cold_x;
  const int32_t *enter_ptr, *enter_row, *enter_slot;
  const int32_t *exit_ptr, *exit_row, *exit_slot;
  const int32_t *hot_ptr, *hot_slot;
  const double *hot_x;
  double2 *in_ring;   // [CS_RING][max_enter][4]   records entering the LDS
  double2 *out_ring;  // [CS_RING][max_exit][2]    words 0 and 2 of the records leaving it (what a sweep changes)
  double2 *part;      // [CS_RING][NB * CS_NWS][CS_MAX_CG]
  double2 *oldnew;    // [CS_RING][CS_MAX_CG]
  CsSync *sync;
  int *error;
  unsigned long long *trace;  // MFM_CS_TRACE: [3 + 2 NB actors][n_steps][4] raw s_memrealtime stamps of one launch (walker, X, Y, U and S of every range)
  unsigned long long *prof;  // MFM_CB_PROF: [16] s_memrealtime sums (100 MHz): 0 walker waits for Y, 1 walks, 2 steps; 4 Y waits for the
                             // ranges, 5 Y stages; 8 U (range 0, wave 4) waits for X, 9 works; 12 S (range 0, wave 0) waits for U, 13 works
};

__device__ __forceinline__ double2 cs_ld2(const double2 *p) {
  return make_double2(__hip_atomic_load(&p->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                      __hip_atomic_load(&p->y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void cs_st2(double2 *p, double2 v) {
  __hip_atomic_store(&p->x, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&p->y, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 16-byte write-through (agent scope) store as ONE instruction: an 8-byte sc1 store is one fabric write each, 2.7x the time per byte
__device__ __forceinline__ void cs_st16(double2 *p, double2 v) {
  typedef double d2v __attribute__((ext_vector_type(2)));
  d2v w;
  w.x = v.x;
  w.y = v.y;
  // (s_nop 1: a store of more than 8 bytes reads its data registers up to two cycles after issue; the compiler's hazard recogniser
  //  covers its own stores, not this one -- without the wait states the next VALU write to those registers raced the store)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(w) : "memory");
}
// 16-byte agent-scope load as ONE instruction (an 8-byte sc1 access runs at about half the rate per byte). The compiler does not know
// that the result is still in flight: cs_ld16_wait must sit between the loads and the first use of ANY of their results -- the values
// are passed through it, so no use can be scheduled above the wait.
typedef double cs_d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cs_d2v cs_ld16_issue(const double2 *p) {
  cs_d2v v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int N>
__device__ __forceinline__ void cs_ld16_wait(cs_d2v (&v)[N]) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < N; i++) asm volatile("" : "+v"(v[i]));
}
// a wavefront waits until *w >= target: lane 0 polls (global: agent scope; LDS: workgroup scope), bounded -- a lost partner
// raises *error, and every later wait of everybody falls through (the results are then garbage and the host raises)
template <bool LDS>
__device__ __forceinline__ void cs_wait(const void *w, long long target, int *error, bool &dead) {
  if (dead || target <= 0) return;
  unsigned spins = 0;
  // (the bound is wall-clock time on the constant 100 MHz counter, 20 s: on a GPU shared with other work a partner workgroup may be
  //  scheduled late -- that is a slow sweep, not a failed one)
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    long long v;
    if (LDS)
      v = __hip_atomic_load((const int *)w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else
      v = (long long)__hip_atomic_load((const unsigned long long *)w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v >= target) break;
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 1023u) == 0u) {
      if (__builtin_amdgcn_s_memrealtime() - t_start > 2000000000ull || __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        __hip_atomic_store(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dead = true;
        break;
      }
    }
  }
  // (the loads behind the wait must not be hoisted above it)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  asm volatile("" ::: "memory");
}

// The hot rows' records in the walker's LDS, WORD-MAJOR: three 16-byte arrays {q, q_S}, {c, c_S}, {e, e_q} and an 8-byte array of
// cardinalities, indexed by slot -- 56 bytes per row instead of the 80 of a padded 64-byte record (a window one step wider fits),
// and a wavefront's 16-byte gathers of random slots spread over all banks. Same arithmetic as ChainOps<P> (fused multiply-adds).
struct CsSlots {
  d2_t *W0, *W1, *W2;
  double *K;
};
template <class P>
struct CsOps;
template <>
struct CsOps<PBlockV> {
  static __device__ __forceinline__ BlockRec load(const CsSlots &L, int slot) {
    BlockRec s;
    s.qq = L.W0[slot];
    s.cc = L.W1[slot];
    s.ee = L.W2[slot];
    s.kk.x = L.K[slot];
    s.kk.y = 0.0;
    return s;
  }
  static __device__ __forceinline__ void apply(const CsSlots &L, int slot, double x, BlockRec s, double old, double fresh) {
    const double delta = fresh - old, dx = delta * x;
    const double h_B = __builtin_fma(-x, old, s.qq.x);
    s.qq.x = __builtin_fma(delta, x, s.qq.x);
    s.qq.y = __builtin_fma(delta * (fresh + old), x * x, s.qq.y);
    s.ee.x = __builtin_fma(dx, __builtin_fma(h_B, s.kk.x, s.cc.x), s.ee.x);
    s.ee.y = __builtin_fma(dx, __builtin_fma(h_B, s.cc.x, s.cc.y), s.ee.y);
    L.W0[slot] = s.qq;
    L.W2[slot] = s.ee;
  }
};
template <>
struct CsOps<PBlockW> {
  static __device__ __forceinline__ PBlockW::St load(const CsSlots &L, int slot) {
    PBlockW::St s;
    s.e = L.W2[slot].x;
    s.card = L.K[slot];
    return s;
  }
  static __device__ __forceinline__ void apply(const CsSlots &L, int slot, double x, PBlockW::St s, double old, double fresh) {
    ((double *)(L.W2 + slot))[0] = __builtin_fma(x * s.card, fresh - old, s.e);
  }
};

template <class P>
__global__ __launch_bounds__(CS_NT) void k_cs_stream(SweepArgs a, CsArgs g) {
  extern __shared__ double2 cs_lds[];
  constexpr int MC = CS_MAX_CG, R = CS_RING, SR = CS_SR, U = CS_U;
  constexpr int rec2_g = P::REC_DOUBLES / 2;  // 4
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = g.n_cols, ns = g.n_steps, Cg = g.Cg, Lw = g.Lw, NB = g.NB, RD = g.RD;
  const int NP = NB * CS_NWS;
  const int rec2_global = a.rec2;
  bool dead = false;
  if (blockIdx.x == 0) {
    // ================================================== the walker workgroup ==================================================
    const int ecap = g.ecap;                                         // hot entries of a column at most (a multiple of 64)
    const int nsl = max(g.n_slots, 1);
    CsSlots L;                                                       // the hot rows' records, word-major (56 bytes per slot)
    L.W0 = (d2_t *)cs_lds;
    L.W1 = L.W0 + nsl;
    L.W2 = L.W1 + nsl;
    L.K = (double *)(L.W2 + nsl);
    double2 *csum = (double2 *)(L.K + nsl + (nsl & 1));              // [SR][MC]
    double *c_old = (double *)(csum + SR * MC);                      // [SR][MC] each
    double *c_z = c_old + SR * MC, *c_lam = c_z + SR * MC, *c_mu = c_lam + SR * MC, *c_new = c_mu + SR * MC;
    double *e_x = c_new + SR * MC;                                   // [CS_ER][Cg][ecap] the hot entries of CS_ER steps: value ...
    int *e_slot = (int *)(e_x + (size_t)CS_ER * Cg * ecap);              // ... and slot
    int *h_cnt = e_slot + (size_t)CS_ER * Cg * ecap;                     // [SR][MC] hot entries per column
    int *y_steps = h_cnt + SR * MC;                                  // [CS_NY] steps staged, per Y wavefront (16-byte aligned: one read)
    int *w_steps = y_steps + CS_NY;                                  // [1] steps walked
    int *x_steps = w_steps + 1;                                      // [CS_NX] steps whose exits are out (published)
    int *x_read = x_steps + CS_NX;                                   // [CS_NX] steps whose leaving records have been read out of their slots
    if (tid < CS_NY + 1 + 2 * CS_NX) y_steps[tid] = 0;
    __syncthreads();  // (the only barrier: before the roles part)
    // Roles by wavefront. A workgroup's wavefronts go round the CU's four SIMDs (wavefront w on SIMD w % 4): the walker (wavefront 0)
    // shares its SIMD only with the idle wavefront 4 -- a helper next to it was starved of issue slots and became the slowest Y
    // wavefront the walker then waited for. X = wavefronts 1, 5; Y = 2, 3, 6, 7.
    const int role = wv == 0 ? 0 : wv == 4 ? 3 : (wv == 1 || wv == 5) ? 1 : 2;
    const int xw_of = wv == 1 ? 0 : 1, yw_of = wv == 2 ? 0 : wv == 3 ? 1 : wv == 6 ? 2 : 3;
    if (role == 0) {
      __builtin_amdgcn_s_setprio(3);
      // ---- wavefront 0: the columns in order over their hot entries ----
      const bool pf = g.prof != nullptr && lane == 0;
      struct Col {
        int cnt, sl[U];
        double hx[U], S1c, S2c, old, lam, mu, z;
      };
      // a column's staged entries and scalars into registers: every round the lists can hold is requested (no dependence on the
      // column's count: rounds past it read staged zeros or another column's entries and are never used)
      auto load_col = [&](Col &C, int s, int c) {
        const int q = (s % SR) * MC + c, base = ((s % CS_ER) * Cg + c) * ecap;
        C.cnt = h_cnt[q];
#pragma unroll
        for (int u = 0; u < U; u++) {
          C.sl[u] = 0;
          C.hx[u] = 0.0;
          if (u * WAVE < ecap) {  // (uniform, the same for every column)
            C.sl[u] = e_slot[base + u * WAVE + lane];
            C.hx[u] = e_x[base + u * WAVE + lane];
          }
        }
        const double2 cs = csum[q];
        C.S1c = cs.x;
        C.S2c = cs.y;
        C.old = c_old[q];
        C.lam = c_lam[q];
        C.mu = c_mu[q];
        C.z = c_z[q];
      };
      // NR rounds of 64 hot entries, straight-line: only the last round is predicated
      auto body = [&](auto nr_tag, const Col &C, int cnt) -> double {
        constexpr int NR = decltype(nr_tag)::value;
        typename P::St st[NR];
        const bool last_ok = (NR - 1) * WAVE + lane < cnt;
#pragma unroll
        for (int u = 0; u < NR - 1; u++) st[u] = CsOps<P>::load(L, C.sl[u]);
        if (last_ok) st[NR - 1] = CsOps<P>::load(L, C.sl[NR - 1]);
        double h1 = 0.0, h2 = 0.0;
#pragma unroll
        for (int u = 0; u < NR - 1; u++) ChainOps<P>::stats(C.hx[u], st[u], C.old, h1, h2);
        if (last_ok) ChainOps<P>::stats(C.hx[NR - 1], st[NR - 1], C.old, h1, h2);
        wave_allreduce_sum2(h1, h2);
        const double fresh = P::template draw<true>(C.S1c + h1, C.S2c + h2, C.old, a.alpha, C.lam, C.mu, C.z);
#pragma unroll
        for (int u = 0; u < NR - 1; u++) CsOps<P>::apply(L, C.sl[u], C.hx[u], st[u], C.old, fresh);
        if (last_ok) CsOps<P>::apply(L, C.sl[NR - 1], C.hx[NR - 1], st[NR - 1], C.old, fresh);
        return fresh;
      };
      unsigned long long t_wait = 0, t_walk = 0;
      bool have = false;
      int s = 0, c = 0;  // step and column inside the step of the column at hand
      auto column = [&](int k, Col &C, Col &N) {
        const int q = (s % SR) * MC + c;
        if (c == 0) {
          unsigned long long t0 = 0;
          if (pf) t0 = __builtin_amdgcn_s_memrealtime();
          // both Y wavefronts of the step's pair have staged it (a COUNT of arrivals would not do: a fast wavefront runs ahead)
          if (!dead) {
            unsigned spins = 0;
            const unsigned long long t_poll = __builtin_amdgcn_s_memrealtime();  // (20 s on the 100 MHz counter, as cs_wait)
            for (;;) {
              const int y0 = __hip_atomic_load(&y_steps[(s & 1) * 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              const int y1 = __hip_atomic_load(&y_steps[(s & 1) * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              if (min(y0, y1) >= s + 1) break;
              __builtin_amdgcn_s_sleep(1);
              if ((++spins & 1023u) == 0u) {
                if (__builtin_amdgcn_s_memrealtime() - t_poll > 2000000000ull ||
                    __hip_atomic_load(g.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                  __hip_atomic_store(g.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  dead = true;
                  break;
                }
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            asm volatile("" ::: "memory");
          }
          if (pf) {
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
            t_wait += t1 - t0;
            t_walk -= t1;
            if (g.trace) {
              g.trace[((size_t)0 * ns + s) * 4 + 0] = t0;
              g.trace[((size_t)0 * ns + s) * 4 + 1] = t1;
            }
          }
        }
        if (!have) load_col(C, s, c);
        have = c + 1 < Cg && k + 1 < n;  // the next column of the SAME step: its entries and scalars are requested now
        if (have) load_col(N, s, c + 1);
        int cnt = __builtin_amdgcn_readfirstlane(C.cnt);
        if (g.dbg & 8) cnt = 0;              // (timing experiments only, wrong results: no hot entries at all ...
        if (g.dbg & 16) cnt = min(cnt, 64);  //  ... one round at most)
        double fresh;
        switch ((cnt + WAVE - 1) / WAVE) {
          case 0: fresh = P::template draw<true>(C.S1c, C.S2c, C.old, a.alpha, C.lam, C.mu, C.z); break;
          case 1: fresh = body(std::integral_constant<int, 1>(), C, cnt); break;
          case 2: fresh = body(std::integral_constant<int, 2>(), C, cnt); break;
          case 3: fresh = body(std::integral_constant<int, 3>(), C, cnt); break;
          case 4: fresh = body(std::integral_constant<int, 4>(), C, cnt); break;
          case 5: fresh = body(std::integral_constant<int, 5>(), C, cnt); break;
          case 6: fresh = body(std::integral_constant<int, 6>(), C, cnt); break;
          case 7: fresh = body(std::integral_constant<int, 7>(), C, cnt); break;
          case 8: fresh = body(std::integral_constant<int, 8>(), C, cnt); break;
          default: {  // (more hot entries than the registers hold: from the staged lists, twice)
            const int base = ((s % CS_ER) * Cg + c) * ecap;
            double h1 = 0.0, h2 = 0.0;
            for (int i = lane; i < cnt; i += WAVE) ChainOps<P>::stats(e_x[base + i], CsOps<P>::load(L, e_slot[base + i]), C.old, h1, h2);
            wave_allreduce_sum2(h1, h2);
            fresh = P::template draw<true>(C.S1c + h1, C.S2c + h2, C.old, a.alpha, C.lam, C.mu, C.z);
            for (int i = lane; i < cnt; i += WAVE) {
              const int slot = e_slot[base + i];
              CsOps<P>::apply(L, slot, e_x[base + i], CsOps<P>::load(L, slot), C.old, fresh);
            }
          }
        }
        if (lane == 0) c_new[q] = fresh;
        // (a wavefront's LDS operations execute in order: the next column's gathers see this column's updates)
        if (c == Cg - 1 || k == n - 1) {
          if (lane == 0) __hip_atomic_store(w_steps, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (pf) {
            const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
            t_walk += t2;
            if (g.trace) g.trace[((size_t)0 * ns + s) * 4 + 2] = t2;
          }
          s++;
          c = 0;
        } else {
          c++;
        }
      };
      Col A, B;
      for (int k = 0; k < n; k += 2) {
        column(k, A, B);
        if (k + 1 < n) column(k + 1, B, A);
      }
      if (pf) {
        g.prof[0] += t_wait;
        g.prof[1] += t_walk;
        g.prof[2] += (unsigned long long)ns;
      }
      return;
    }
    if (role == 1) {
      // ---- X: what leaves the LDS after step j, and the step's (old, new) pairs ----
      // The two X wavefronts take the steps in turn (wavefront j & 1 publishes step j): reading the leaving records out of their
      // slots, the write-through stores and their drain are a chain of round trips longer than a step of the walker; what is static
      // (list offsets, slots, column ids) is requested one own step ahead. Steps are PUBLISHED in order (walk_done counts them).
      const int xw = xw_of;
      const bool xpf = g.prof != nullptr && lane == 0 && xw == 0;
      constexpr int XP = 8;  // exits per lane kept in registers (beyond: slot by slot)
      int x0_n = 0, x1_n = 0, col_n = -1, slot_n[XP];
      auto prefetch_static = [&](int j) {
        x0_n = x1_n = 0;
        col_n = -1;
#pragma unroll
        for (int t = 0; t < XP; t++) slot_n[t] = -1;
        if (j < ns) {
          x0_n = g.exit_ptr[(size_t)j * NB];
          x1_n = g.exit_ptr[(size_t)(j + 1) * NB];
          if (lane < min(Cg, n - j * Cg)) col_n = g.cols[j * Cg + lane];
#pragma unroll
          for (int t = 0; t < XP; t++) {
            const int x = x0_n + t * WAVE + lane;
            if (x < x1_n) slot_n[t] = g.exit_slot[x];
          }
        }
      };
      prefetch_static(xw);
      for (int j = xw; j < ns; j += 2) {
        const int x0 = x0_n, x1 = x1_n, col = col_n, sl = (j % SR) * MC;
        int slot_p[XP];
#pragma unroll
        for (int t = 0; t < XP; t++) slot_p[t] = slot_n[t];
        prefetch_static(j + 2);
        unsigned long long xt0 = 0, xt1 = 0;
        if (xpf) xt0 = __builtin_amdgcn_s_memrealtime();
        cs_wait<true>(w_steps, j + 1, g.error, dead);
        if (xpf) xt1 = __builtin_amdgcn_s_memrealtime();
        double2 *dst = g.out_ring + (size_t)(j % R) * max(g.max_exit, 1) * 2;
        // the leaving records out of their slots first (the slots are free for the rows entering RD steps later as soon as they
        // have been READ), then the write-through stores
        double2 r0[XP], r2[XP];
#pragma unroll
        for (int t = 0; t < XP; t++) {
          r0[t] = r2[t] = make_double2(0.0, 0.0);
          if (slot_p[t] >= 0) {
            r0[t] = ((const double2 *)L.W0)[slot_p[t]];
            r2[t] = ((const double2 *)L.W2)[slot_p[t]];
          }
        }
        for (int x = x0 + XP * WAVE + lane; x < x1; x += WAVE) {  // (rare: the rest goes slot by slot before the flag)
          const int slot = g.exit_slot[x];
          cs_st16(dst + (size_t)(x - x0) * 2, ((const double2 *)L.W0)[slot]);
          cs_st16(dst + (size_t)(x - x0) * 2 + 1, ((const double2 *)L.W2)[slot]);
        }
        double2 on = make_double2(0.0, 0.0);
        if (col >= 0) on = make_double2(c_old[sl + lane], c_new[sl + lane]);
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the records are in registers
        if (lane == 0) __hip_atomic_store(&x_read[xw], j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (col >= 0) {
          cs_st16(g.oldnew + (size_t)(j % R) * MC + lane, on);
          a.theta[col] = on.y;
        }
#pragma unroll
        for (int t = 0; t < XP; t++) {
          const int x = x0 + t * WAVE + lane;
          if (x < x1) {
            cs_st16(dst + (size_t)(x - x0) * 2, r0[t]);
            cs_st16(dst + (size_t)(x - x0) * 2 + 1, r2[t]);
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // published in step order: behind the other wavefront's step j - 1
        cs_wait<true>(&x_steps[xw ^ 1], j, g.error, dead);
        if (lane == 0) {
          __hip_atomic_store(&g.sync->walk_done, (unsigned long long)(j + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&x_steps[xw], j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (xpf) {
          const unsigned long long xt2 = __builtin_amdgcn_s_memrealtime();
          g.prof[6] += xt1 - xt0;
          g.prof[7] += xt2 - xt1;
          if (g.trace) {
            g.trace[((size_t)1 * ns + j) * 4 + 0] = xt0;
            g.trace[((size_t)1 * ns + j) * 4 + 1] = xt1;
            g.trace[((size_t)1 * ns + j) * 4 + 2] = xt2;
          }
        }
      }
      return;
    }
    if (role == 2) {
      // ---- Y: what enters the LDS before step s, the step's cold statistics, per-column scalars and hot entry lists ----
      // Two pairs of wavefronts take the steps in turn (pair s & 1 stages step s; inside a pair, wavefront i takes the columns
      // i, i + 2, .. and every second wavefront-tile of the entering rows): a step's staging is a chain of three or four memory round
      // trips, longer than the walker needs for the step -- with two steps in flight it is off the critical path. Everything static
      // (column ids, groups, list offsets) is requested one own step ahead, so that no load waits for another inside a step.
      const int yw = yw_of, grp = yw >> 1, yi = yw & 1;
      const bool pf = g.prof != nullptr && lane == 0 && yw == 0;
      unsigned long long t_wait = 0, t_work = 0;
      constexpr int YC = (CS_MAX_CG + 1) / 2;  // columns of a step per Y wavefront at most
      // lane t < YC holds, for column yi + 2 t of the wavefront's NEXT step: first / end of its hot entries, its id and its group
      int hb_n = 0, he_n = 0, j_n = 0, gr_n = 0;
      auto prefetch_static = [&](int s) {
        hb_n = he_n = j_n = gr_n = 0;
        if (s < ns && lane < YC && yi + 2 * lane < min(Cg, n - s * Cg)) {
          const int k = s * Cg + yi + 2 * lane;
          hb_n = g.hot_ptr[k];
          he_n = g.hot_ptr[k + 1];
          j_n = g.cols[k];
          gr_n = g.col_group[k];
        }
      };
      prefetch_static(grp);
      for (int s = grp; s < ns; s += 2) {
        const int e0 = g.enter_ptr[(size_t)s * NB], e1 = g.enter_ptr[(size_t)(s + 1) * NB];
        const int ncs = min(Cg, n - s * Cg), sl = (s % SR) * MC;
        const int hb_v = hb_n, he_v = he_n, j_v = j_n, gr_v = gr_n;
        prefetch_static(s + 2);
        constexpr int EP = 3;  // entering rows per lane and batch
        int slot_p[EP];
#pragma unroll
        for (int t = 0; t < EP; t++) {
          const int e = e0 + (t * 2 + yi) * WAVE + lane;
          slot_p[t] = e < e1 ? g.enter_slot[e] : -1;
        }
        // the scalars and the hot entries of this wavefront's columns (static data: requested before the waits)
        double s_old[YC], s_z[YC], s_lam[YC], s_mu[YC];
        int en_sl[YC][U], en_cnt[YC];
        double en_x[YC][U];
#pragma unroll
        for (int t = 0; t < YC; t++) {
          const int c = yi + 2 * t;
          s_old[t] = s_z[t] = s_lam[t] = s_mu[t] = 0.0;
          en_cnt[t] = 0;
          if (c < ncs) {
            const int hb = __builtin_amdgcn_readlane(hb_v, t), he = __builtin_amdgcn_readlane(he_v, t);
            const int j = __builtin_amdgcn_readlane(j_v, t), gr = __builtin_amdgcn_readlane(gr_v, t);
            en_cnt[t] = he - hb;
            if (lane == 0) {
              s_old[t] = a.theta[j];
              s_z[t] = a.z[j];
              s_lam[t] = a.lambda[gr];
              s_mu[t] = a.mu[gr];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
              const int i = u * WAVE + lane;
              en_sl[t][u] = 0;
              en_x[t][u] = 0.0;
              if (u * WAVE < en_cnt[t] && i < en_cnt[t]) {
                en_sl[t][u] = g.hot_slot[hb + i];
                en_x[t][u] = g.hot_x[hb + i];
              }
            }
          }
        }
        unsigned long long t0 = 0, t1 = 0;
        if (pf) t0 = __builtin_amdgcn_s_memrealtime();
        // every range's S wavefronts have published step s (first: they run ahead of the walker, the poll's round trip hides
        // behind the wait for the slots below)
        if (!dead) {
          unsigned spins = 0;
          const unsigned long long t_poll = __builtin_amdgcn_s_memrealtime();
          for (;;) {
            bool ok = true;
            for (int p = lane; p < NB * 2; p += WAVE)  // (the two S wavefronts of every range that took step s)
              ok = ok && __hip_atomic_load(&g.sync->s_flag[(p >> 1) * CS_NWS + (s & 1) * 2 + (p & 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >=
                             (unsigned long long)(s + 1);
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0u) {
              if (__builtin_amdgcn_s_memrealtime() - t_poll > 2000000000ull ||
                  __hip_atomic_load(g.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                __hip_atomic_store(g.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dead = true;
                break;
              }
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          asm volatile("" ::: "memory");
        }
        // the ranges' partial statistics and the first batch of entering records: one round trip for all of them (16-byte loads)
        constexpr int NPV = (CS_MAX_NB * 2) / WAVE;
        cs_d2v ld[YC * NPV + EP * rec2_g];
        // (every load is issued unconditionally, from a clamped address: a conditional one would leave the compiler a copy of the
        //  not-yet-arrived value to make between the load and the wait)
#pragma unroll
        for (int t = 0; t < YC; t++)
#pragma unroll
          for (int pp = 0; pp < NPV; pp++) {
            const int c = min(yi + 2 * t, MC - 1), p = min(pp * WAVE + lane, NB * 2 - 1);
            ld[t * NPV + pp] = cs_ld16_issue(g.part + ((size_t)(s % R) * (NB * 2) + p) * MC + c);
          }
        const double2 *src = g.in_ring + (size_t)(s % R) * max(g.max_enter, 1) * rec2_g;
#pragma unroll
        for (int t = 0; t < EP; t++) {
          const int e = min(e0 + (t * 2 + yi) * WAVE + lane, max(e1 - 1, e0));
#pragma unroll
          for (int w = 0; w < rec2_g; w++) ld[YC * NPV + t * rec2_g + w] = cs_ld16_issue(src + (size_t)(e - e0) * rec2_g + w);
        }
        cs_ld16_wait(ld);
        double2 pv[YC][NPV], r[EP][rec2_g];
#pragma unroll
        for (int t = 0; t < YC; t++)
#pragma unroll
          for (int pp = 0; pp < NPV; pp++) {
            const bool ok = yi + 2 * t < ncs && pp * WAVE + lane < NB * 2;
            pv[t][pp] = ok ? make_double2(ld[t * NPV + pp].x, ld[t * NPV + pp].y) : make_double2(0.0, 0.0);
          }
#pragma unroll
        for (int t = 0; t < EP; t++)
#pragma unroll
          for (int w = 0; w < rec2_g; w++) r[t][w] = make_double2(ld[YC * NPV + t * rec2_g + w].x, ld[YC * NPV + t * rec2_g + w].y);
        // slot reuse (and the rings of per-step data): the leaving records of step s - RD have been read out of their slots
        // (the X wavefront that took step s - RD, and the other one's step before it: both Y pairs reuse slots of either)
        cs_wait<true>(&x_read[(s - RD) & 1], s - RD + 1, g.error, dead);
        cs_wait<true>(&x_read[(s - RD - 1) & 1], s - RD, g.error, dead);
        if (pf) t1 = __builtin_amdgcn_s_memrealtime();
        // the hot entry lists and the scalars into the LDS
#pragma unroll
        for (int t = 0; t < YC; t++) {
          const int c = yi + 2 * t;
          if (c < ncs) {
            const int base = ((s % CS_ER) * Cg + c) * ecap;
#pragma unroll
            for (int u = 0; u < U; u++)
              if (u * WAVE < en_cnt[t]) {
                e_slot[base + u * WAVE + lane] = en_sl[t][u];
                e_x[base + u * WAVE + lane] = en_x[t][u];
              }
            for (int i = U * WAVE + lane; i < en_cnt[t]; i += WAVE) {  // (beyond the registers: rare)
              const int hb = __builtin_amdgcn_readlane(hb_v, t);
              e_slot[base + i] = g.hot_slot[hb + i];
              e_x[base + i] = g.hot_x[hb + i];
            }
            if (lane == 0) h_cnt[sl + c] = en_cnt[t];
          }
        }
#pragma unroll
        for (int t = 0; t < EP; t++) {
          const int e = e0 + (t * 2 + yi) * WAVE + lane;
          if (e < e1) {
#pragma unroll
            for (int w = 0; w < 1; w++) {
              ((double2 *)L.W0)[slot_p[t]] = r[t][0];
              ((double2 *)L.W1)[slot_p[t]] = r[t][1];
              ((double2 *)L.W2)[slot_p[t]] = r[t][2];
              L.K[slot_p[t]] = r[t][3].x;
            }
          }
        }
        for (int e = e0 + (EP * 2 + yi) * WAVE + lane; e < e1; e += 2 * WAVE) {  // (beyond the first batch)
          const int slot = g.enter_slot[e];
          double2 rr[rec2_g];
#pragma unroll
          for (int w = 0; w < rec2_g; w++) rr[w] = cs_ld2(src + (size_t)(e - e0) * rec2_g + w);
#pragma unroll
          for (int w = 0; w < 1; w++) {
            ((double2 *)L.W0)[slot] = rr[0];
            ((double2 *)L.W1)[slot] = rr[1];
            ((double2 *)L.W2)[slot] = rr[2];
            L.K[slot] = rr[3].x;
          }
        }
#pragma unroll
        for (int t = 0; t < YC; t++) {
          const int c = yi + 2 * t;
          if (c < ncs) {  // (wave-uniform)
            double S1 = 0.0, S2 = 0.0;
#pragma unroll
            for (int pp = 0; pp < (CS_MAX_NB * 2) / WAVE; pp++) {  // (a lane's partials in order; then the fixed tree over the lanes)
              S1 += pv[t][pp].x;
              S2 += pv[t][pp].y;
            }
            wave_allreduce_sum2(S1, S2);
            if (lane == 0) {
              csum[sl + c] = make_double2(S1, S2);
              c_old[sl + c] = s_old[t];
              c_z[sl + c] = s_z[t];
              c_lam[sl + c] = s_lam[t];
              c_mu[sl + c] = s_mu[t];
            }
          }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        if (lane == 0) __hip_atomic_store(&y_steps[yw], s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (pf) {
          const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
          t_wait += t1 - t0;
          t_work += t2 - t1;
          if (g.trace) {
            g.trace[((size_t)2 * ns + s) * 4 + 0] = t0;
            g.trace[((size_t)2 * ns + s) * 4 + 1] = t1;
            g.trace[((size_t)2 * ns + s) * 4 + 2] = t2;
          }
        }
      }
      if (pf) {
        g.prof[4] += t_wait;
        g.prof[5] += t_work;
      }
      return;
    }
    return;
  }
  // ====================================================== a row range ======================================================
  const int b = (int)blockIdx.x - 1;
  double2 *part_w = cs_lds;                          // [CS_NWS][MC]
  double *c_old_w = (double *)(part_w + CS_NWS * MC);  // [CS_NWS][MC]
  int *u_steps = (int *)(c_old_w + CS_NWS * MC);     // [4] steps applied, per U wavefront
  if (tid < 4) u_steps[tid] = 0;
  __syncthreads();  // (the only barrier)
  if (wv < CS_NWS) {
    // ---- S: cold statistics of step v from the records in global memory, the rows entering at v packed for the walker ----
    // Two pairs of wavefronts take the steps in turn (pair v & 1 does step v, each of its two wavefronts half of the range's entries):
    // a step costs a gather of the records, the write-through stores and their drain -- more than the walker needs for the step.
    const int sw = wv, grp = sw >> 1, si = sw & 1;
    const bool pf = g.prof != nullptr && lane == 0 && sw == 0 && (b == 0 || g.trace != nullptr);
    unsigned long long t_wait = 0, t_work = 0;
    double2 *pw = part_w + sw * MC;
    double *co = c_old_w + sw * MC;
    for (int v = grp; v < ns; v += 2) {
      const int lo = g.cold_ptr[(size_t)v * NB + b], hi = g.cold_ptr[(size_t)v * NB + b + 1];
      const int ncs = min(Cg, n - v * Cg);
      const int chunk = (((hi - lo + 1) / 2 + WAVE - 1) / WAVE) * WAVE;
      const int mylo = lo + si * chunk, myhi = min(hi, mylo + chunk);
      constexpr int T = 3;  // tiles whose entries are requested before the wait and whose records are gathered together
      int rc[T];
      double xv[T];
#pragma unroll
      for (int t = 0; t < T; t++) {
        const int p = mylo + t * WAVE + lane;
        rc[t] = -1;
        xv[t] = 0.0;
        if (p < myhi) {
          rc[t] = g.cold_rc[p];
          xv[t] = g.cold_x[p];
        }
      }
      if (lane < MC) {
        pw[lane] = make_double2(0.0, 0.0);
        co[lane] = lane < ncs ? a.theta[g.cols[v * Cg + lane]] : 0.0;
      }
      const int en0 = g.enter_ptr[(size_t)v * NB], en_lo = g.enter_ptr[(size_t)v * NB + b], en_hi = g.enter_ptr[(size_t)v * NB + b + 1];
      int en_row = -1;
      {
        const int e = en_lo + si * WAVE + lane;
        if (e < en_hi) en_row = g.enter_row[e];
      }
      unsigned long long t0 = 0, t1 = 0;
      if (pf) t0 = __builtin_amdgcn_s_memrealtime();
      for (int u = 0; u < 4; u++) {  // (U wavefront u takes the steps of parity u >> 1: its last one that is <= v - Lw)
        const int last = v - Lw - (((v - Lw) - (u >> 1)) & 1);
        cs_wait<true>(&u_steps[u], last + 1, g.error, dead);
      }
      if (pf) t1 = __builtin_amdgcn_s_memrealtime();
      // every record this step needs, requested together: the first tiles' rows and the first entering row of the lane
      typename P::St st[T];
#pragma unroll
      for (int t = 0; t < T; t++)
        if (rc[t] >= 0) st[t] = P::load(a, rc[t] & ((1 << CS_LCOL_SHIFT) - 1));
      double2 er[rec2_g];
      if (en_row >= 0) {
        const double2 *src = (const double2 *)a.state + (int64_t)en_row * rec2_global;
#pragma unroll
        for (int w = 0; w < rec2_g; w++) er[w] = src[w];
      }
      auto tile = [&](int rcv, double x, const typename P::St &stv) {
        const int lc = rcv < 0 ? -1 - lane : (rcv >> CS_LCOL_SHIFT);
        double s1 = 0.0, s2 = 0.0;
        if (rcv >= 0) P::stats(x, stv, co[lc], s1, s2);
        const int lp = dpp_i32<0x138, 0xf>(lc, 0), ln = dpp_i32<0x130, 0xf>(lc, 0);
        const bool head = lane == 0 || lp != lc, tail = lane == 63 || ln != lc;
        int f = head ? 1 : 0;
        wave_segscan2(s1, s2, f);
        if (rcv >= 0 && tail) {  // one lane per column of this tile; a wavefront adds its tiles in program order
          double2 &q = pw[lc];
          q.x += s1;
          q.y += s2;
        }
      };
#pragma unroll
      for (int t = 0; t < T; t++)
        if (mylo + t * WAVE < myhi) tile(rc[t], xv[t], st[t]);
      for (int base = mylo + T * WAVE; base < myhi; base += WAVE) {  // (beyond the first tiles: rare)
        const int p = base + lane;
        const int rcv = p < myhi ? g.cold_rc[p] : -1;
        typename P::St stv;
        if (rcv >= 0) stv = P::load(a, rcv & ((1 << CS_LCOL_SHIFT) - 1));
        tile(rcv, p < myhi ? g.cold_x[p] : 0.0, stv);
      }
      // rows entering the walker's LDS at step v: their records as they are now
      double2 *dst = g.in_ring + (size_t)(v % R) * max(g.max_enter, 1) * rec2_g;
      if (en_row >= 0) {
        const int e = en_lo + si * WAVE + lane;
#pragma unroll
        for (int w = 0; w < rec2_g; w++) cs_st16(dst + (size_t)(e - en0) * rec2_g + w, er[w]);
      }
      for (int e = en_lo + (2 + si) * WAVE + lane; e < en_hi; e += 2 * WAVE) {  // (beyond the first: rare)
        const double2 *src = (const double2 *)a.state + (int64_t)g.enter_row[e] * rec2_global;
        double2 r[rec2_g];
#pragma unroll
        for (int w = 0; w < rec2_g; w++) r[w] = src[w];
#pragma unroll
        for (int w = 0; w < rec2_g; w++) cs_st16(dst + (size_t)(e - en0) * rec2_g + w, r[w]);
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the tails' LDS adds
      if (lane < MC) cs_st16(g.part + ((size_t)(v % R) * (NB * 2) + b * 2 + si) * MC + lane, pw[lane]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0)
        __hip_atomic_store(&g.sync->s_flag[b * CS_NWS + sw], (unsigned long long)(v + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (pf) {
        const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
        t_wait += t1 - t0;
        t_work += t2 - t1;
        if (g.trace) {
          g.trace[((size_t)(3 + NB + b) * ns + v) * 4 + 0] = t0;
          g.trace[((size_t)(3 + NB + b) * ns + v) * 4 + 1] = t1;
          g.trace[((size_t)(3 + NB + b) * ns + v) * 4 + 2] = t2;
        }
      }
    }
    if (pf && b == 0) {
      g.prof[12] += t_wait;
      g.prof[13] += t_work;
    }
    return;
  }
  {
    // ---- U: cold updates of step u with the walker's (old, new), the leaving rows' records back to their rows ----
    // Two pairs of wavefronts take the steps in turn (pair u & 1 does step u, each of its two wavefronts half of the range's entries and
    // leaving rows); list offsets are requested one own step ahead, so that the entries' loads do not wait for them.
    const int uw = wv - CS_NWS, up = uw >> 1, ui = uw & 1;
    const bool pf = g.prof != nullptr && lane == 0 && uw == 0 && (b == 0 || g.trace != nullptr);
    unsigned long long t_wait = 0, t_work = 0;
    int lo_n = 0, hi_n = 0, x0_n = 0, xlo_n = 0, xhi_n = 0;
    auto prefetch_static = [&](int u) {
      lo_n = hi_n = x0_n = xlo_n = xhi_n = 0;
      if (u < ns) {
        lo_n = g.cold_ptr[(size_t)u * NB + b];
        hi_n = g.cold_ptr[(size_t)u * NB + b + 1];
        x0_n = g.exit_ptr[(size_t)u * NB];
        xlo_n = g.exit_ptr[(size_t)u * NB + b];
        xhi_n = g.exit_ptr[(size_t)u * NB + b + 1];
      }
    };
    prefetch_static(up);
    for (int u = up; u < ns; u += 2) {
      const int lo = lo_n, hi = hi_n, x0 = x0_n, x_lo = xlo_n, x_hi = xhi_n;
      prefetch_static(u + 2);
      constexpr int T = 3;
      int rc[T];
      double xv[T];
#pragma unroll
      for (int t = 0; t < T; t++) {
        const int p = lo + (t * 2 + ui) * WAVE + lane;
        rc[t] = -1;
        xv[t] = 0.0;
        if (p < hi) {
          rc[t] = g.cold_rc[p];
          xv[t] = g.cold_x[p];
        }
      }
      int ex_row = -1;
      {
        const int x = x_lo + ui * WAVE + lane;
        if (x < x_hi) ex_row = g.exit_row[x];
      }
      unsigned long long t0 = 0, t1 = 0;
      if (pf) t0 = __builtin_amdgcn_s_memrealtime();
      cs_wait<false>(&g.sync->walk_done, u + 1, g.error, dead);
      if (pf) t1 = __builtin_amdgcn_s_memrealtime();
      const double2 *on = g.oldnew + (size_t)(u % R) * MC;
      auto upd = [&](int rcv, double x) {
        if (rcv < 0) return;
        const int lc = rcv >> CS_LCOL_SHIFT, row = rcv & ((1 << CS_LCOL_SHIFT) - 1);
        const double2 o = cs_ld2(on + lc);
        const typename P::St st = P::load(a, row);
        P::apply(a, row, x, st, o.x, o.y);
      };
      // the leaving record of this lane (requested together with the cold entries' records)
      const double2 *src = g.out_ring + (size_t)(u % R) * max(g.max_exit, 1) * 2;
      double2 xr0 = make_double2(0.0, 0.0), xr2 = xr0;
      if (ex_row >= 0) {
        const int x = x_lo + ui * WAVE + lane;
        xr0 = cs_ld2(src + (size_t)(x - x0) * 2);
        xr2 = cs_ld2(src + (size_t)(x - x0) * 2 + 1);
      }
      double2 o_[T];
      typename P::St st_[T];
#pragma unroll
      for (int t = 0; t < T; t++)
        if (rc[t] >= 0) {
          o_[t] = cs_ld2(on + (rc[t] >> CS_LCOL_SHIFT));
          st_[t] = P::load(a, rc[t] & ((1 << CS_LCOL_SHIFT) - 1));
        }
#pragma unroll
      for (int t = 0; t < T; t++)
        if (rc[t] >= 0) P::apply(a, rc[t] & ((1 << CS_LCOL_SHIFT) - 1), xv[t], st_[t], o_[t].x, o_[t].y);
      for (int p = lo + (T * 2 + ui) * WAVE + lane; p < hi; p += 2 * WAVE) upd(g.cold_rc[p], g.cold_x[p]);
      if (ex_row >= 0) {
        double2 *rec = (double2 *)a.state + (int64_t)ex_row * rec2_global;
        rec[0] = xr0;
        rec[2] = xr2;
      }
      for (int x = x_lo + (2 + ui) * WAVE + lane; x < x_hi; x += 2 * WAVE) {  // (beyond the first: rare)
        const int row = g.exit_row[x];
        const double2 r0 = cs_ld2(src + (size_t)(x - x0) * 2), r2 = cs_ld2(src + (size_t)(x - x0) * 2 + 1);
        double2 *rec = (double2 *)a.state + (int64_t)row * rec2_global;
        rec[0] = r0;
        rec[2] = r2;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(&u_steps[uw], u + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (pf) {
        const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
        t_wait += t1 - t0;
        t_work += t2 - t1;
        if (g.trace) {
          g.trace[((size_t)(3 + b) * ns + u) * 4 + 0] = t0;
          g.trace[((size_t)(3 + b) * ns + u) * 4 + 1] = t1;
          g.trace[((size_t)(3 + b) * ns + u) * 4 + 2] = t2;
        }
      }
    }
    if (pf && b == 0) {
      g.prof[8] += t_wait;
      g.prof[9] += t_work;
    }
  }
}

// LDS of a launch (the walker's need; the ranges use a few hundred bytes of it)
inline int cs_ecap(int max_hot_col) { return std::max(WAVE, ((max_hot_col + WAVE - 1) / WAVE) * WAVE); }
inline size_t cs_lds_bytes(int n_slots, int Cg, int max_hot_col) {
  return (size_t)(std::max(n_slots, 1) + 1) * 56 + (size_t)CS_SR * CS_MAX_CG * (sizeof(double2) + 5 * sizeof(double) + sizeof(int)) +
         (size_t)CS_ER * Cg * cs_ecap(max_hot_col) * (sizeof(double) + sizeof(int)) + (CS_NY + 1 + 2 * CS_NX) * sizeof(int) + 64;
}

}  // namespace mfm
