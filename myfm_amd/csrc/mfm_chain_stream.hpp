// mfm_chain_stream.hpp -- k_cs_stream: the block-feature sweep of a LARGE relation block (FMTrainer.hpp:276-302 for w, :419-470
// for V; block state beyond one CU's LDS, columns that pairwise share rows) as ONE pipelined launch without batch boundaries.
// Plan and the exactness argument: mfm_chain_plan.hpp. This file: the device side.
//
// Workgroup 0 is the WALKER (512 threads): wavefront 0 walks the columns in order over their hot entries, on the hot rows'
// records in LDS; wavefronts 1-2 (X) copy the records of rows that leave the LDS and the step's (old, new) pairs out;
// wavefronts 3-6 (Y) stage the rows that enter and the cold statistics of the coming step. Workgroups 1..NB own a contiguous
// range of block rows each: wavefronts 0-3 (S) take the cold statistics of step v and pack the rows entering at v, wavefronts 4-7
// (U) apply the cold updates of step u and put the leaving rows' records back. Nobody ever executes a workgroup barrier: every
// wavefront runs its own loop over the steps and waits on monotonic counters only --
//   walker(s)  <-  Y(s)  <-  S(s) of every range (global flags) and X(s - RD) (LDS: slot reuse)
//   S(v)       <-  U(v - Lw) of the same workgroup (LDS)
//   U(u)       <-  X(u) (global counter)             X(u) <- walker(u) (LDS)
// Everything that crosses workgroups (ring slots of entering / leaving records, per-(range, wavefront) partial statistics, the
// (old, new) pairs, the flags) is written with agent-scope (write-through) stores, drained with s_waitcnt vmcnt(0) before the
// flag, and read with agent-scope loads; the block-row records themselves are only ever touched by their range's workgroup
// (one CU: its L1 sees its own stores). The sums have a fixed association: a second run is bit-identical.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
constexpr size_t CS_LDS_MAX = 156 * 1024;  // of the CU's 160 KiB

struct CsSync {  // zeroed before every launch
  unsigned long long walk_done;  // steps whose exits and (old, new) pairs are published
  unsigned long long pad0[15];
  unsigned long long s_flag[CS_MAX_NB * CS_NWS];  // per (range, S wavefront): steps whose partials and entering records are published
};

struct CsArgs {
  int n_cols, n_steps, Cg, Lw, NB, RD, max_enter, max_exit, n_slots;
  int ecap;  // hot entries of a column at most, rounded up to whole wavefronts
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
  asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(w) : "memory");
}
// a wavefront waits until *w >= target: lane 0 polls (global: agent scope; LDS: workgroup scope), bounded -- a lost partner
// raises *error, and every later wait of everybody falls through (the results are then garbage and the host raises)
template <bool LDS>
__device__ __forceinline__ void cs_wait(const void *w, long long target, int *error, bool &dead) {
  if (dead || target <= 0) return;
  unsigned spins = 0;
  for (;;) {
    long long v;
    if (LDS)
      v = __hip_atomic_load((const int *)w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else
      v = (long long)__hip_atomic_load((const unsigned long long *)w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v >= target) break;
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 1023u) == 0u) {
      if (spins > (1u << 23) || __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
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

template <class P>
__global__ __launch_bounds__(CS_NT) void k_cs_stream(SweepArgs a, CsArgs g) {
  extern __shared__ double2 cs_lds[];
  constexpr int MC = CS_MAX_CG, R = CS_RING, SR = CS_SR, U = CS_U;
  constexpr int rec2_g = P::REC_DOUBLES / 2;  // 4
  constexpr int rec2_l = rec2_g + 1;          // LDS stride of a record in 16-byte words (bank spread)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = g.n_cols, ns = g.n_steps, Cg = g.Cg, Lw = g.Lw, NB = g.NB, RD = g.RD;
  const int NP = NB * CS_NWS;
  const int rec2_global = a.rec2;
  bool dead = false;
  if (blockIdx.x == 0) {
    // ================================================== the walker workgroup ==================================================
    const int ecap = g.ecap;                                         // hot entries of a column at most (a multiple of 64)
    double2 *recs = cs_lds;                                          // [n_slots][rec2_l]
    double2 *csum = recs + (size_t)max(g.n_slots, 1) * rec2_l;       // [SR][MC]
    double *c_old = (double *)(csum + SR * MC);                      // [SR][MC] each
    double *c_z = c_old + SR * MC, *c_lam = c_z + SR * MC, *c_mu = c_lam + SR * MC, *c_new = c_mu + SR * MC;
    double *e_x = c_new + SR * MC;                                   // [2][Cg][ecap] the hot entries of two steps: value ...
    int *e_slot = (int *)(e_x + (size_t)2 * Cg * ecap);              // ... and slot
    int *h_cnt = e_slot + (size_t)2 * Cg * ecap;                     // [SR][MC] hot entries per column
    int *w_steps = h_cnt + SR * MC;                                  // [1] steps walked
    int *y_count = w_steps + 1;                                      // [1] += 1 per Y wavefront and step staged
    int *x_steps = y_count + 1;                                      // [CS_NX] steps whose exits are out
    if (tid < 2 + CS_NX) w_steps[tid] = 0;
    __syncthreads();  // (the only barrier: before the roles part)
    SweepArgs al = a;
    al.state = recs;
    al.rec2 = rec2_l;
    if (wv == 0) {
      // ---- wavefront 0: the columns in order over their hot entries ----
      const bool pf = g.prof != nullptr && lane == 0;
      struct Col {
        int cnt, sl[U];
        double hx[U], S1c, S2c, old, lam, mu, z;
      };
      auto load_col = [&](Col &C, int k) {
        const int s = k / Cg, c = k - s * Cg, q = (s % SR) * MC + c, base = ((s & 1) * Cg + c) * ecap;
        C.cnt = h_cnt[q];
#pragma unroll
        for (int u = 0; u < U; u++) {
          C.sl[u] = 0;
          C.hx[u] = 0.0;
          if (u * WAVE < C.cnt) {  // (rounds past the column's entries read staged zeros)
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
      unsigned long long t_wait = 0, t_walk = 0;
      bool have = false;
      auto column = [&](int k, Col &C, Col &N) {
        const int s = k / Cg, c = k - s * Cg, q = (s % SR) * MC + c;
        if (c == 0) {
          unsigned long long t0 = 0;
          if (pf) t0 = __builtin_amdgcn_s_memrealtime();
          cs_wait<true>(y_count, (long long)CS_NY * (s + 1), g.error, dead);
          if (pf) {
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
            t_wait += t1 - t0;
            t_walk -= t1;
          }
        }
        if (!have) load_col(C, k);
        have = c + 1 < Cg && k + 1 < n;  // the next column of the SAME step: its entries and scalars are requested now
        if (have) load_col(N, k + 1);
        const int cnt = C.cnt;
        const double old = C.old;
        double fresh;
        if (cnt <= U * WAVE) {
          typename P::St st[U];
          double h1 = 0.0, h2 = 0.0;
#pragma unroll
          for (int u = 0; u < U; u++)
            if (u * WAVE < cnt && u * WAVE + lane < cnt) st[u] = P::load(al, C.sl[u]);
#pragma unroll
          for (int u = 0; u < U; u++)
            if (u * WAVE < cnt) {
              double t1 = 0.0, t2 = 0.0;
              if (u * WAVE + lane < cnt) ChainOps<P>::stats(C.hx[u], st[u], old, t1, t2);
              h1 += t1;
              h2 += t2;
            }
          wave_allreduce_sum2(h1, h2);
          fresh = P::template draw<true>(C.S1c + h1, C.S2c + h2, old, a.alpha, C.lam, C.mu, C.z);
#pragma unroll
          for (int u = 0; u < U; u++)
            if (u * WAVE < cnt && u * WAVE + lane < cnt) ChainOps<P>::apply(al, C.sl[u], C.hx[u], st[u], old, fresh);
        } else {  // (more hot entries than the registers hold: from the staged lists, twice)
          const int base = ((s & 1) * Cg + c) * ecap;
          double h1 = 0.0, h2 = 0.0;
          for (int i = lane; i < cnt; i += WAVE) {
            double t1 = 0.0, t2 = 0.0;
            ChainOps<P>::stats(e_x[base + i], P::load(al, e_slot[base + i]), old, t1, t2);
            h1 += t1;
            h2 += t2;
          }
          wave_allreduce_sum2(h1, h2);
          fresh = P::template draw<true>(C.S1c + h1, C.S2c + h2, old, a.alpha, C.lam, C.mu, C.z);
          for (int i = lane; i < cnt; i += WAVE) {
            const int slot = e_slot[base + i];
            ChainOps<P>::apply(al, slot, e_x[base + i], P::load(al, slot), old, fresh);
          }
        }
        if (lane == 0) c_new[q] = fresh;
        // (a wavefront's LDS operations execute in order: the next column's gathers see this column's updates)
        if (c == Cg - 1 || k == n - 1) {
          if (lane == 0) __hip_atomic_store(w_steps, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (pf) t_walk += __builtin_amdgcn_s_memrealtime();
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
    if (wv <= CS_NX) {
      // ---- X: what leaves the LDS after step j, and the step's (old, new) pairs ----
      const int xw = wv - 1;
      for (int j = 0; j < ns; j++) {
        const int x0 = g.exit_ptr[(size_t)j * NB], x1 = g.exit_ptr[(size_t)(j + 1) * NB];
        const int ncs = min(Cg, n - j * Cg), sl = (j % SR) * MC;
        int col = -1;
        if (xw == 0 && lane < ncs) col = g.cols[j * Cg + lane];
        constexpr int XP = 4;  // exits per lane requested before the wait
        int slot_p[XP];
#pragma unroll
        for (int t = 0; t < XP; t++) {
          const int x = x0 + (t * CS_NX + xw) * WAVE + lane;
          slot_p[t] = x < x1 ? g.exit_slot[x] : -1;
        }
        cs_wait<true>(w_steps, j + 1, g.error, dead);
        double2 *dst = g.out_ring + (size_t)(j % R) * max(g.max_exit, 1) * 2;
#pragma unroll
        for (int t = 0; t < XP; t++) {
          const int x = x0 + (t * CS_NX + xw) * WAVE + lane;
          if (x < x1) {
            cs_st16(dst + (size_t)(x - x0) * 2, recs[(size_t)slot_p[t] * rec2_l]);
            cs_st16(dst + (size_t)(x - x0) * 2 + 1, recs[(size_t)slot_p[t] * rec2_l + 2]);
          }
        }
        for (int x = x0 + (XP * CS_NX + xw) * WAVE + lane; x < x1; x += CS_NX * WAVE) {
          const int slot = g.exit_slot[x];
          cs_st16(dst + (size_t)(x - x0) * 2, recs[(size_t)slot * rec2_l]);
          cs_st16(dst + (size_t)(x - x0) * 2 + 1, recs[(size_t)slot * rec2_l + 2]);
        }
        if (col >= 0) {
          cs_st16(g.oldnew + (size_t)(j % R) * MC + lane, make_double2(c_old[sl + lane], c_new[sl + lane]));
          a.theta[col] = c_new[sl + lane];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(&x_steps[xw], j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (xw == 0) {
          for (int o = 1; o < CS_NX; o++) cs_wait<true>(&x_steps[o], j + 1, g.error, dead);
          if (lane == 0)
            __hip_atomic_store(&g.sync->walk_done, (unsigned long long)(j + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      return;
    }
    if (wv <= CS_NX + CS_NY) {
      // ---- Y: what enters the LDS before step s, the step's cold statistics, per-column scalars and hot entry lists ----
      const int yw = wv - 1 - CS_NX;
      const bool pf = g.prof != nullptr && lane == 0 && yw == 0;
      unsigned long long t_wait = 0, t_work = 0;
      constexpr int YC = (CS_MAX_CG + CS_NY - 1) / CS_NY;  // columns of a step per Y wavefront at most
      // first / end of this wavefront's columns' hot entries, one step ahead (lane t holds column yw + t * CS_NY)
      int hb_n = 0, he_n = 0;
      if (lane < YC && yw + lane * CS_NY < min(Cg, n)) {
        hb_n = g.hot_ptr[yw + lane * CS_NY];
        he_n = g.hot_ptr[yw + lane * CS_NY + 1];
      }
      for (int s = 0; s < ns; s++) {
        const int e0 = g.enter_ptr[(size_t)s * NB], e1 = g.enter_ptr[(size_t)(s + 1) * NB];
        const int ncs = min(Cg, n - s * Cg), sl = (s % SR) * MC;
        const int hb_v = hb_n, he_v = he_n;
        hb_n = he_n = 0;
        if (lane < YC && s + 1 < ns && yw + lane * CS_NY < min(Cg, n - (s + 1) * Cg)) {
          hb_n = g.hot_ptr[(s + 1) * Cg + yw + lane * CS_NY];
          he_n = g.hot_ptr[(s + 1) * Cg + yw + lane * CS_NY + 1];
        }
        constexpr int EP = 3;  // entering rows per lane whose slots are requested before the wait
        int slot_p[EP];
#pragma unroll
        for (int t = 0; t < EP; t++) {
          const int e = e0 + (t * CS_NY + yw) * WAVE + lane;
          slot_p[t] = e < e1 ? g.enter_slot[e] : -1;
        }
        // the scalars and the hot entries of this wavefront's columns (static data: requested before the wait)
        double s_old[YC], s_z[YC], s_lam[YC], s_mu[YC];
        int en_sl[YC][U], en_cnt[YC];
        double en_x[YC][U];
#pragma unroll
        for (int t = 0; t < YC; t++) {
          const int c = yw + t * CS_NY;
          s_old[t] = s_z[t] = s_lam[t] = s_mu[t] = 0.0;
          en_cnt[t] = 0;
          if (c < ncs) {
            const int hb = __builtin_amdgcn_readlane(hb_v, t), he = __builtin_amdgcn_readlane(he_v, t);
            en_cnt[t] = he - hb;
            if (lane == 0) {
              const int k = s * Cg + c, j = g.cols[k], gr = g.col_group[k];
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
        // slot reuse (and the rings of per-step data): the exits of step s - RD are out
        for (int x = 0; x < CS_NX; x++) cs_wait<true>(&x_steps[x], s - RD + 1, g.error, dead);
        // the hot entry lists do not need the ranges: into the LDS now
#pragma unroll
        for (int t = 0; t < YC; t++) {
          const int c = yw + t * CS_NY;
          if (c < ncs) {
            const int base = ((s & 1) * Cg + c) * ecap;
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
        // every range's S wavefronts have published step s
        if (!dead) {
          unsigned spins = 0;
          for (;;) {
            bool ok = true;
            for (int p = lane; p < NP; p += WAVE)
              ok = ok && __hip_atomic_load(&g.sync->s_flag[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned long long)(s + 1);
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0u) {
              if (spins > (1u << 23) || __hip_atomic_load(g.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                __hip_atomic_store(g.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dead = true;
                break;
              }
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          asm volatile("" ::: "memory");
        }
        if (pf) t1 = __builtin_amdgcn_s_memrealtime();
        const double2 *src = g.in_ring + (size_t)(s % R) * max(g.max_enter, 1) * rec2_g;
        {
          double2 r[EP][rec2_g];
#pragma unroll
          for (int t = 0; t < EP; t++) {
            const int e = e0 + (t * CS_NY + yw) * WAVE + lane;
            if (e < e1) {
#pragma unroll
              for (int w = 0; w < rec2_g; w++) r[t][w] = cs_ld2(src + (size_t)(e - e0) * rec2_g + w);
            }
          }
#pragma unroll
          for (int t = 0; t < EP; t++) {
            const int e = e0 + (t * CS_NY + yw) * WAVE + lane;
            if (e < e1) {
#pragma unroll
              for (int w = 0; w < rec2_g; w++) recs[(size_t)slot_p[t] * rec2_l + w] = r[t][w];
            }
          }
        }
        for (int e = e0 + (EP * CS_NY + yw) * WAVE + lane; e < e1; e += CS_NY * WAVE) {
          const int slot = g.enter_slot[e];
          double2 r[rec2_g];
#pragma unroll
          for (int w = 0; w < rec2_g; w++) r[w] = cs_ld2(src + (size_t)(e - e0) * rec2_g + w);
#pragma unroll
          for (int w = 0; w < rec2_g; w++) recs[(size_t)slot * rec2_l + w] = r[w];
        }
#pragma unroll
        for (int t = 0; t < YC; t++) {
          const int c = yw + t * CS_NY;
          if (c < ncs) {  // (wave-uniform)
            double S1 = 0.0, S2 = 0.0;
            for (int p = lane; p < NP; p += WAVE) {  // (partials p, p + 64, ... of a lane in order; then the fixed tree over the lanes)
              const double2 v = cs_ld2(g.part + ((size_t)(s % R) * NP + p) * MC + c);
              S1 += v.x;
              S2 += v.y;
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
        if (lane == 0) __hip_atomic_fetch_add(y_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (pf) {
          t_wait += t1 - t0;
          t_work += __builtin_amdgcn_s_memrealtime() - t1;
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
    const int sw = wv;
    const bool pf = g.prof != nullptr && lane == 0 && sw == 0 && b == 0;
    unsigned long long t_wait = 0, t_work = 0;
    double2 *pw = part_w + sw * MC;
    double *co = c_old_w + sw * MC;
    for (int v = 0; v < ns; v++) {
      const int lo = g.cold_ptr[(size_t)v * NB + b], hi = g.cold_ptr[(size_t)v * NB + b + 1];
      const int ncs = min(Cg, n - v * Cg);
      const int chunk = (((hi - lo + CS_NWS - 1) / CS_NWS + WAVE - 1) / WAVE) * WAVE;
      const int mylo = lo + sw * chunk, myhi = min(hi, mylo + chunk);
      constexpr int T = 2;  // tiles whose entries are requested before the wait
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
        const int e = en_lo + sw * WAVE + lane;
        if (e < en_hi) en_row = g.enter_row[e];
      }
      unsigned long long t0 = 0, t1 = 0;
      if (pf) t0 = __builtin_amdgcn_s_memrealtime();
      for (int u = 0; u < 4; u++) cs_wait<true>(&u_steps[u], v - Lw + 1, g.error, dead);
      if (pf) t1 = __builtin_amdgcn_s_memrealtime();
      auto tile = [&](int rcv, double x) {
        const int lc = rcv < 0 ? -1 - lane : (rcv >> CS_LCOL_SHIFT);
        const int row = rcv < 0 ? -1 : (rcv & ((1 << CS_LCOL_SHIFT) - 1));
        double s1 = 0.0, s2 = 0.0;
        if (row >= 0) {
          const typename P::St st = P::load(a, row);
          P::stats(x, st, co[lc], s1, s2);
        }
        const int lp = dpp_i32<0x138, 0xf>(lc, 0), ln = dpp_i32<0x130, 0xf>(lc, 0);
        const bool head = lane == 0 || lp != lc, tail = lane == 63 || ln != lc;
        int f = head ? 1 : 0;
        wave_segscan2(s1, s2, f);
        if (row >= 0 && tail) {  // one lane per column of this tile; a wavefront adds its tiles in program order
          double2 &q = pw[lc];
          q.x += s1;
          q.y += s2;
        }
      };
#pragma unroll
      for (int t = 0; t < T; t++)
        if (mylo + t * WAVE < myhi) tile(rc[t], xv[t]);
      for (int base = mylo + T * WAVE; base < myhi; base += WAVE) {
        const int p = base + lane;
        tile(p < myhi ? g.cold_rc[p] : -1, p < myhi ? g.cold_x[p] : 0.0);
      }
      // rows entering the walker's LDS at step v: their records as they are now
      double2 *dst = g.in_ring + (size_t)(v % R) * max(g.max_enter, 1) * rec2_g;
      for (int e = en_lo + sw * WAVE + lane; e < en_hi; e += CS_NWS * WAVE) {
        const int row = e == en_lo + sw * WAVE + lane ? en_row : g.enter_row[e];
        const double2 *src = (const double2 *)a.state + (int64_t)row * rec2_global;
        double2 r[rec2_g];
#pragma unroll
        for (int w = 0; w < rec2_g; w++) r[w] = src[w];
#pragma unroll
        for (int w = 0; w < rec2_g; w++) cs_st16(dst + (size_t)(e - en0) * rec2_g + w, r[w]);
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the tails' LDS adds
      if (lane < MC) cs_st16(g.part + ((size_t)(v % R) * NP + b * CS_NWS + sw) * MC + lane, pw[lane]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0)
        __hip_atomic_store(&g.sync->s_flag[b * CS_NWS + sw], (unsigned long long)(v + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (pf) {
        t_wait += t1 - t0;
        t_work += __builtin_amdgcn_s_memrealtime() - t1;
      }
    }
    if (pf) {
      g.prof[12] += t_wait;
      g.prof[13] += t_work;
    }
    return;
  }
  {
    // ---- U: cold updates of step u with the walker's (old, new), the leaving rows' records back to their rows ----
    const int uw = wv - CS_NWS;
    const bool pf = g.prof != nullptr && lane == 0 && uw == 0 && b == 0;
    unsigned long long t_wait = 0, t_work = 0;
    for (int u = 0; u < ns; u++) {
      const int lo = g.cold_ptr[(size_t)u * NB + b], hi = g.cold_ptr[(size_t)u * NB + b + 1];
      constexpr int T = 2;
      int rc[T];
      double xv[T];
#pragma unroll
      for (int t = 0; t < T; t++) {
        const int p = lo + (t * 4 + uw) * WAVE + lane;
        rc[t] = -1;
        xv[t] = 0.0;
        if (p < hi) {
          rc[t] = g.cold_rc[p];
          xv[t] = g.cold_x[p];
        }
      }
      const int x0 = g.exit_ptr[(size_t)u * NB], x_lo = g.exit_ptr[(size_t)u * NB + b], x_hi = g.exit_ptr[(size_t)u * NB + b + 1];
      int ex_row = -1;
      {
        const int x = x_lo + uw * WAVE + lane;
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
#pragma unroll
      for (int t = 0; t < T; t++) upd(rc[t], xv[t]);
      for (int p = lo + (T * 4 + uw) * WAVE + lane; p < hi; p += 4 * WAVE) upd(g.cold_rc[p], g.cold_x[p]);
      const double2 *src = g.out_ring + (size_t)(u % R) * max(g.max_exit, 1) * 2;
      for (int x = x_lo + uw * WAVE + lane; x < x_hi; x += 4 * WAVE) {
        const int row = x == x_lo + uw * WAVE + lane ? ex_row : g.exit_row[x];
        const double2 r0 = cs_ld2(src + (size_t)(x - x0) * 2), r2 = cs_ld2(src + (size_t)(x - x0) * 2 + 1);
        double2 *rec = (double2 *)a.state + (int64_t)row * rec2_global;
        rec[0] = r0;
        rec[2] = r2;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(&u_steps[uw], u + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (pf) {
        t_wait += t1 - t0;
        t_work += __builtin_amdgcn_s_memrealtime() - t1;
      }
    }
    if (pf) {
      g.prof[8] += t_wait;
      g.prof[9] += t_work;
    }
  }
}

// LDS of a launch (the walker's need; the ranges use a few hundred bytes of it)
inline int cs_ecap(int max_hot_col) { return std::max(WAVE, ((max_hot_col + WAVE - 1) / WAVE) * WAVE); }
inline size_t cs_lds_bytes(int n_slots, int Cg, int max_hot_col) {
  return (size_t)std::max(n_slots, 1) * 5 * sizeof(double2) + (size_t)CS_SR * CS_MAX_CG * (sizeof(double2) + 5 * sizeof(double) + sizeof(int)) +
         (size_t)2 * Cg * cs_ecap(max_hot_col) * (sizeof(double) + sizeof(int)) + (2 + CS_NX) * sizeof(int) + 64;
}

}  // namespace mfm
