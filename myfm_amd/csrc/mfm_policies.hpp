// mfm_policies.hpp -- sweep arguments and per-coordinate policies (what one conditional draw needs from the entries of its
// column) shared by the HIP translation units: mfm_hip.hip (level / tile / chain kernels, mfm_kernels.hpp) and mfm_chain.hip
// (the streamed conflict-window chain of large relation blocks, mfm_chain_stream.hpp). Device inline code only: no kernels here.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mfm_wave.hpp"

namespace mfm {

constexpr int WG = 256;         // threads per workgroup
constexpr int CHAIN_WG = 256;   // threads of the sequential-chain kernel (global-state variant)
constexpr int BLOCK_REC = 8;    // doubles per relation-block row record

struct SweepArgs {
  const int64_t *colptr;
  const int32_t *rowidx;
  const double *val;
  void *state;           // double2 eq[N] (main table) or double rec[B][8] (relation block)
  double *theta;         // w or V[:, f], already offset to this matrix's first feature
  const double *z;       // pre-drawn N(0,1) variates, same indexing
  const int32_t *group;  // group_index, same indexing
  const double *lambda;  // [G] for this sweep
  const double *mu;      // [G]
  double alpha;
  int rec2 = 4;          // stride of a relation-block record in 16-byte words (5 when staged in LDS: bank spread)
  // CSR of the same table (PMainVq only: the q-cache entry of a row is rebuilt on the fly)
  const int32_t *r_rowptr = nullptr;
  const int32_t *r_colidx = nullptr;
  const double *r_val = nullptr;
  int r_ell = -1;
  // level whose columns each cover a contiguous row range (a table sorted by this field): first row per
  // column; the row index array is not read and the state loads issue one round trip earlier
  const int32_t *row0 = nullptr;
  // split layout of the latent sweep (PMainVs*): state = e[N], state2 = q[N]
  double *state2 = nullptr;
  // ... its first pass reads e straight from the interleaved {e, q} array (aos, stride 2) and its last pass writes
  // it back there, so that no separate pack / unpack passes are needed
  double2 *aos = nullptr;
  const double *e_src = nullptr;  // PMainVsq: where e is read from (null: state), in units of e_src_stride doubles
  int e_src_stride = 1;
};

struct ChunkDesc {
  int64_t begin;  // first CSC entry
  int32_t len;
  int32_t lcol;  // index into the level's long-column list
};

// ---------------------------------------------------------------------------------------------
// Sweep policies: what one coordinate's conditional needs from the entries of its column.
//   St         per-entry state gathered from memory
//   stats      accumulates the two sufficient statistics
//   draw       the conditional draw from (S1, S2)
//   apply      the scatter update of the entry's state
// ---------------------------------------------------------------------------------------------
// latent factors, main table: FMTrainer.hpp:343-376
struct PMainV {
  static constexpr int R_W16 = 16, R_WG = 16, REC_DOUBLES = 2;
  static constexpr bool QFREE = false;
  static constexpr double BYTES = 44.0, STAT_BYTES = 28.0;  // per nnz: CSC 12 + eq 16 (+ eq 16 write)
  typedef double2 St;
  static __device__ __forceinline__ St load(const SweepArgs &a, int row) { return ((const double2 *)a.state)[row]; }
  static __device__ __forceinline__ void stats(double x, const St &s, double old, double &S1, double &S2) {
    const double h = x * (s.y - x * old);
    S2 += h * h;
    S1 += (-s.x) * h;
  }
  template <bool FAST = false>
  static __device__ __forceinline__ double draw(double S1, double S2, double old, double alpha, double lam, double mu,
                                                double z) {
    double lin = S1 + S2 * old;  // :358
    double sq = S2 * alpha;      // :360
    lin = lin * alpha;           // :361
    sq += lam;                   // :363
    lin += lam * mu;             // :364-365
    return sample_normal_zt<FAST>(sq, lin, z);
  }
  static __device__ __forceinline__ St updated(double x, St s, double old, double fresh) {
    const double delta = fresh - old;
    const double h = x * (s.y - x * old);
    s.y += x * delta;  // :373
    s.x += h * delta;  // :374
    return s;
  }
  static __device__ __forceinline__ void apply(const SweepArgs &a, int row, double x, St s, double old, double fresh) {
    ((double2 *)a.state)[row] = updated(x, s, old, fresh);
  }
  // row-tile path: the {e, q} record staged in LDS
  static __device__ __forceinline__ St from_rec(double2 r) { return r; }
  static __device__ __forceinline__ double2 to_rec(double2, St s) { return s; }
};

// PMainV for the FIRST level of a factor when that level touches every row exactly once (a one-hot field):
// instead of reading q_t from the q-cache, rebuild it from the row's CSR entries,
// q_t = sum_j x_tj v_jf (FMTrainer.hpp:320) -- the separate q-build pass (CSR stream + a partial-line store
// of q for every row) disappears; the level writes (e, q) back anyway.
template <bool UNIT>
struct PMainVq : PMainV {
  static __device__ __forceinline__ St load(const SweepArgs &a, int row) {
    int64_t b, e;
    if (a.r_ell >= 0) {
      b = (int64_t)row * a.r_ell;
      e = b + a.r_ell;
    } else {
      b = a.r_rowptr[row];
      e = a.r_rowptr[row + 1];
    }
    const double ev = ((const double2 *)a.state)[row].x;
    double q = 0.0;
    if (a.r_ell == 2) {  // two one-hot fields: one 8-byte index load, both gathers in flight together
      const int2 ci = *(const int2 *)(a.r_colidx + b);
      const double v0 = a.theta[ci.x], v1 = a.theta[ci.y];
      q = (UNIT ? 1.0 : a.r_val[b]) * v0;
      q += (UNIT ? 1.0 : a.r_val[b + 1]) * v1;
    } else {
      for (int64_t p = b; p < e; p++) q += (UNIT ? 1.0 : a.r_val[p]) * a.theta[a.r_colidx[p]];
    }
    return make_double2(ev, q);
  }
};

// Split ("SoA") layout for the latent sweep of a main table without relation blocks: e[N] and q[N] as two
// arrays for the duration of update_V. With the first level rebuilding q (PMainVsq) and nothing reading q
// after a factor's last level (PMainVsl / k_tile_apply<.., WRITE_Q = false>), the sweep moves 8 bytes per row
// where the interleaved {e, q} layout moves 16: the first level reads only e, the last level writes only e.
struct PMainVs : PMainV {
  static __device__ __forceinline__ St load(const SweepArgs &a, int row) {
    return make_double2(((const double *)a.state)[row], a.state2[row]);
  }
  static __device__ __forceinline__ void apply(const SweepArgs &a, int row, double x, St s, double old, double fresh) {
    const St n = updated(x, s, old, fresh);
    ((double *)a.state)[row] = n.x;
    a.state2[row] = n.y;
  }
};
struct PMainVsl : PMainVs {  // last level of the factor: q is dead
  static __device__ __forceinline__ void apply(const SweepArgs &a, int row, double x, St s, double old, double fresh) {
    ((double *)a.state)[row] = updated(x, s, old, fresh).x;
  }
};
template <bool UNIT>
struct PMainVsq : PMainVs {  // first level: q rebuilt from the row (see PMainVq)
  static __device__ __forceinline__ St load(const SweepArgs &a, int row) {
    int64_t b, e;
    if (a.r_ell >= 0) {
      b = (int64_t)row * a.r_ell;
      e = b + a.r_ell;
    } else {
      b = a.r_rowptr[row];
      e = a.r_rowptr[row + 1];
    }
    const double ev = a.e_src ? a.e_src[(int64_t)row * a.e_src_stride] : ((const double *)a.state)[row];
    double q = 0.0;
    if (a.r_ell == 2) {
      const int2 ci = *(const int2 *)(a.r_colidx + b);
      const double v0 = a.theta[ci.x], v1 = a.theta[ci.y];
      q = (UNIT ? 1.0 : a.r_val[b]) * v0;
      q += (UNIT ? 1.0 : a.r_val[b + 1]) * v1;
    } else {
      for (int64_t p = b; p < e; p++) q += (UNIT ? 1.0 : a.r_val[p]) * a.theta[a.r_colidx[p]];
    }
    return make_double2(ev, q);
  }
};

// PMainVsq for an apply pass that runs AFTER the column's new coefficient was stored (row-sharded mode:
// statistics -> all-reduce -> draw -> apply): the rebuilt q already contains x * v_new.
template <bool UNIT>
struct PMainVsqA : PMainVsq<UNIT> {
  typedef double2 St;
  static __device__ __forceinline__ void apply(const SweepArgs &a, int row, double x, St s, double old, double fresh) {
    const double h = x * (s.y - x * fresh);  // = x (q_old - x v_old)
    ((double *)a.state)[row] = s.x + h * (fresh - old);
    a.state2[row] = s.y;
  }
};

// "q-free" latent sweep for tables with short rows (one-hot designs): the q-cache entry of a row is never
// stored -- it is recomputed from the row's few CSR entries and the current V[:, f] wherever it is needed
// (q_t = sum_j x_tj v_jf, FMTrainer.hpp:320; the increments of :373 are implicit because v is updated in
// place). The sweep then streams a compact residual array e[N] (8 bytes per row instead of the 16-byte
// {e, q} pair), which halves the state traffic of both levels, and the separate q-build pass disappears.
// AFTER = the load happens after the column's new coefficient was written (apply pass of the two-pass
// scattered path): q already contains x * v_new, so h = x (q - x v_new) = x (q_old - x v_old).
template <bool UNIT, bool AFTER>
struct PMainVe : PMainV {
  static constexpr bool QFREE = true;
  static constexpr double BYTES = 28.0, STAT_BYTES = 20.0;  // CSC 4(+8) + row CSR 8 + e 8 (+ e 8 write)
  static __device__ __forceinline__ St load(const SweepArgs &a, int row) {
    int64_t b, e;
    if (a.r_ell >= 0) {
      b = (int64_t)row * a.r_ell;
      e = b + a.r_ell;
    } else {
      b = a.r_rowptr[row];
      e = a.r_rowptr[row + 1];
    }
    double q = 0.0;
    for (int64_t p = b; p < e; p++) q += (UNIT ? 1.0 : a.r_val[p]) * a.theta[a.r_colidx[p]];
    return make_double2(((const double *)a.state)[row], q);
  }
  static __device__ __forceinline__ void apply(const SweepArgs &a, int row, double x, St s, double old, double fresh) {
    const double h = x * (s.y - x * (AFTER ? fresh : old));
    ((double *)a.state)[row] = s.x + h * (fresh - old);  // :374
  }
};
// linear weights, main table: FMTrainer.hpp:237-254
struct PMainW {
  static constexpr int R_W16 = 16, R_WG = 16, REC_DOUBLES = 2;
  static constexpr bool QFREE = false;
  static constexpr double BYTES = 28.0, STAT_BYTES = 20.0;
  typedef double St;
  static __device__ __forceinline__ St load(const SweepArgs &a, int row) { return ((const double2 *)a.state)[row].x; }
  static __device__ __forceinline__ void stats(double x, const St &e, double old, double &S1, double &S2) {
    const double et = e - x * old;  // :242
    S2 += x * x;                    // :246
    S1 += x * et;                   // :248
  }
  template <bool FAST = false>
  static __device__ __forceinline__ double draw(double S1, double S2, double old, double alpha, double lam, double mu,
                                                double z) {
    const double sq = lam + alpha * S2;
    const double lin = -alpha * S1 + lam * mu;
    return sample_normal_zt<FAST>(sq, lin, z);
  }
  static __device__ __forceinline__ St updated(double x, St e, double old, double fresh) {
    e -= x * old;
    e += x * fresh;  // :252
    return e;
  }
  static __device__ __forceinline__ void apply(const SweepArgs &a, int row, double x, St e, double old, double fresh) {
    ((double2 *)a.state)[row].x = updated(x, e, old, fresh);
  }
  static __device__ __forceinline__ St from_rec(double2 r) { return r.x; }
  static __device__ __forceinline__ double2 to_rec(double2 r, St e) { return make_double2(e, r.y); }
};

typedef double d2_t __attribute__((ext_vector_type(2)));  // register-resident 16-byte record: native vector type
                                                          // (arrays of HIP's double2 struct end up in scratch)
struct BlockRec {
  d2_t qq;  // q, q_S
  d2_t cc;  // c, c_S
  d2_t ee;  // e, e_q
  d2_t kk;  // cardinality, unused
};

// latent factors, relation block: FMTrainer.hpp:419-470
struct PBlockV {
  static constexpr int R_W16 = 0, R_WG = 4, REC_DOUBLES = 8;   // 64-byte records: keep the register budget bounded
  static constexpr bool QFREE = false;
  static constexpr double BYTES = 12.0 + 64.0 + 48.0, STAT_BYTES = 12.0 + 64.0;
  typedef BlockRec St;
  static __device__ __forceinline__ St load(const SweepArgs &a, int row) {
    const d2_t *r = (const d2_t *)a.state + (int64_t)row * a.rec2;
    St s;
    s.qq = r[0];
    s.cc = r[1];
    s.ee = r[2];
    s.kk = r[3];
    return s;
  }
  static __device__ __forceinline__ void stats(double x, const St &s, double old, double &S1, double &S2) {
    const double h_B = s.qq.x - x * old;                                      // :432
    double h_squared = h_B * h_B * s.kk.x + 2 * s.cc.x * h_B + s.cc.y;        // :433-436
    h_squared = x * x * h_squared;                                            // :437
    S2 += h_squared;                                                          // :438
    S1 += (-s.ee.x * h_B - s.ee.y) * x;                                       // :439-441
  }
  template <bool FAST = false>
  static __device__ __forceinline__ double draw(double S1, double S2, double old, double alpha, double lam, double mu,
                                                double z) {
    return PMainV::draw<FAST>(S1, S2, old, alpha, lam, mu, z);  // :443-450
  }
  static __device__ __forceinline__ void apply(const SweepArgs &a, int row, double x, St s, double old, double fresh) {
    const double delta = fresh - old;
    const double h_B = s.qq.x - x * old;                      // :456
    s.qq.x += delta * x;                                      // :457
    s.qq.y += delta * (fresh + old) * x * x;                  // :458-459
    s.ee.x += x * delta * (h_B * s.kk.x + s.cc.x);            // :461-464
    s.ee.y += x * delta * (h_B * s.cc.x + s.cc.y);            // :465-468
    d2_t *r = (d2_t *)a.state + (int64_t)row * a.rec2;
    r[0] = s.qq;
    r[2] = s.ee;
  }
};

// linear weights, relation block: FMTrainer.hpp:276-302
struct PBlockW {
  static constexpr int R_W16 = 0, R_WG = 8, REC_DOUBLES = 8;
  static constexpr bool QFREE = false;
  static constexpr double BYTES = 12.0 + 32.0 + 8.0, STAT_BYTES = 12.0 + 32.0;
  struct St {
    double e, card;
  };
  static __device__ __forceinline__ St load(const SweepArgs &a, int row) {
    const double2 *r = (const double2 *)a.state + (int64_t)row * a.rec2;
    St s;
    s.e = r[2].x;
    s.card = r[3].x;
    return s;
  }
  static __device__ __forceinline__ void stats(double x, const St &s, double old, double &S1, double &S2) {
    S2 += (x * x) * s.card;  // :285-287
    S1 += x * s.e;           // :288-289
  }
  template <bool FAST = false>
  static __device__ __forceinline__ double draw(double S1, double S2, double old, double alpha, double lam, double mu,
                                                double z) {
    double lin = -S1;
    lin += S2 * old;                 // :291
    const double sq = lam + alpha * S2;  // :293
    lin = alpha * lin + lam * mu;    // :294
    return sample_normal_zt<FAST>(sq, lin, z);
  }
  static __device__ __forceinline__ void apply(const SweepArgs &a, int row, double x, St s, double old, double fresh) {
    s.e += (x * s.card) * (fresh - old);  // :298-301
    ((double *)a.state)[(int64_t)row * (2 * a.rec2) + 4] = s.e;
  }
};

// Statistics / update of one entry inside a SEQUENTIAL chain (k_chain_lds, the hot walkers of the conflict-batched
// chains): the same formulas with explicit fused multiply-adds -- every instruction of a lone wavefront's column step is
// paid in full, and fusing roughly halves the count. (The level kernels keep the unfused forms of the policies: they are
// bandwidth-bound, and theirs is the arithmetic the CPU oracle is compiled to, -ffp-contract=off.)
template <class P>
struct ChainOps {
  static __device__ __forceinline__ void stats(double x, const typename P::St &s, double old, double &S1, double &S2) {
    P::stats(x, s, old, S1, S2);
  }
  static __device__ __forceinline__ void apply(const SweepArgs &a, int row, double x, typename P::St s, double old, double fresh) {
    P::apply(a, row, x, s, old, fresh);
  }
};
template <>
struct ChainOps<PBlockV> {
  static __device__ __forceinline__ void stats(double x, const BlockRec &s, double old, double &S1, double &S2) {
    const double h_B = __builtin_fma(-x, old, s.qq.x);
    const double t = __builtin_fma(h_B * s.kk.x, h_B, __builtin_fma(s.cc.x + s.cc.x, h_B, s.cc.y));
    S2 = __builtin_fma(x * x, t, S2);
    S1 = __builtin_fma(__builtin_fma(-s.ee.x, h_B, -s.ee.y), x, S1);
  }
  static __device__ __forceinline__ void apply(const SweepArgs &a, int row, double x, BlockRec s, double old, double fresh) {
    const double delta = fresh - old, dx = delta * x;
    const double h_B = __builtin_fma(-x, old, s.qq.x);
    s.qq.x = __builtin_fma(delta, x, s.qq.x);
    s.qq.y = __builtin_fma(delta * (fresh + old), x * x, s.qq.y);
    s.ee.x = __builtin_fma(dx, __builtin_fma(h_B, s.kk.x, s.cc.x), s.ee.x);
    s.ee.y = __builtin_fma(dx, __builtin_fma(h_B, s.cc.x, s.cc.y), s.ee.y);
    d2_t *r = (d2_t *)a.state + (int64_t)row * a.rec2;
    r[0] = s.qq;
    r[2] = s.ee;
  }
};
template <>
struct ChainOps<PBlockW> {
  static __device__ __forceinline__ void stats(double x, const PBlockW::St &s, double old, double &S1, double &S2) {
    S2 = __builtin_fma(x * x, s.card, S2);
    S1 = __builtin_fma(x, s.e, S1);
  }
  static __device__ __forceinline__ void apply(const SweepArgs &a, int row, double x, PBlockW::St s, double old, double fresh) {
    ((double *)a.state)[(int64_t)row * (2 * a.rec2) + 4] = __builtin_fma(x * s.card, fresh - old, s.e);
  }
};

}  // namespace mfm
