// mfm_wave.hpp -- wave64 device helpers shared by the HIP translation units (gfx950): SGPR broadcasts, DPP reductions and
// segmented scans with a fixed association (deterministic), the conditional normal draw.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mfm {

constexpr int WAVE = 64;

// value of lane K (compile-time) broadcast to the wave through SGPRs (v_readlane), no LDS crossbar
__device__ __forceinline__ double readlane_f64(double v, int k) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), k);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int64_t readlane_i64(int64_t v, int k) {
  const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), k);
  const int hi = __builtin_amdgcn_readlane((int)(v >> 32), k);
  return ((int64_t)hi << 32) | (uint32_t)lo;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
// Sum over the 64 lanes, identical in every lane, fixed order. DPP data-parallel primitives (row_shr
// 1/2/4/8 inside the 16-lane rows, row_bcast15 / row_bcast31 across rows; lanes without a source add
// +0.0) instead of six dependent ds_bpermute round trips: the reduction sits on the critical path of
// every conditional draw.
__device__ __forceinline__ double wave_allreduce_sum(double v) {
  v += dpp_f64<0x111, 0xf>(v);  // row_shr:1
  v += dpp_f64<0x112, 0xf>(v);  // row_shr:2
  v += dpp_f64<0x114, 0xf>(v);  // row_shr:4
  v += dpp_f64<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of each row holds the row sum
  v += dpp_f64<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_f64<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
  return readlane_f64(v, 63);
}
// Sums of (a, b) over the 64 lanes at once: v_permlane32_swap folds the two halves of the wavefront so that lanes 0-31
// carry a and lanes 32-63 carry b, then ONE 32-lane DPP reduction serves both (24 instructions instead of 76 for two
// wave_allreduce_sum; fixed order, deterministic). Used where the reduction is on a sequential critical path.
__device__ __forceinline__ void wave_allreduce_sum2(double &a, double &b) {
  typedef unsigned u2_t __attribute__((ext_vector_type(2)));
  const u2_t lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const u2_t hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  // lanes < 32: a[l] + a[l + 32]; lanes >= 32: b[l - 32] + b[l]
  double v = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
  v += dpp_f64<0x111, 0xf>(v);  // row_shr:1
  v += dpp_f64<0x112, 0xf>(v);  // row_shr:2
  v += dpp_f64<0x114, 0xf>(v);  // row_shr:4
  v += dpp_f64<0x118, 0xf>(v);  // row_shr:8
  v += dpp_f64<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3: lane 31 holds sum(a), lane 63 sum(b)
  a = readlane_f64(v, 31);
  b = readlane_f64(v, 63);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int v, int fill) {
  return __builtin_amdgcn_update_dpp(fill, v, CTRL, ROW_MASK, 0xf, false);
}
// Inclusive segmented sum scan over the wave, entirely in the VALU (DPP): (s1, s2) are summed over the
// lanes of a segment up to and including this lane; f != 0 marks "a segment head lies in [row start or
// wave start .. this lane]" and is 1 at heads on entry. In-row steps row_shr 1/2/4/8, then row_bcast15 /
// row_bcast31 carry the running totals across the 16-lane rows. Fixed association: deterministic.
__device__ __forceinline__ void wave_segscan2(double &s1, double &s2, int &f) {
  // lanes without a source (row start / masked rows) read u = 0 and fu = 0: unchanged
  {
    const double u1 = dpp_f64<0x111, 0xf>(s1), u2 = dpp_f64<0x111, 0xf>(s2);
    const int fu = dpp_i32<0x111, 0xf>(f, 0);
    if (!f) { s1 += u1; s2 += u2; }
    f |= fu;
  }
  {
    const double u1 = dpp_f64<0x112, 0xf>(s1), u2 = dpp_f64<0x112, 0xf>(s2);
    const int fu = dpp_i32<0x112, 0xf>(f, 0);
    if (!f) { s1 += u1; s2 += u2; }
    f |= fu;
  }
  {
    const double u1 = dpp_f64<0x114, 0xf>(s1), u2 = dpp_f64<0x114, 0xf>(s2);
    const int fu = dpp_i32<0x114, 0xf>(f, 0);
    if (!f) { s1 += u1; s2 += u2; }
    f |= fu;
  }
  {
    const double u1 = dpp_f64<0x118, 0xf>(s1), u2 = dpp_f64<0x118, 0xf>(s2);
    const int fu = dpp_i32<0x118, 0xf>(f, 0);
    if (!f) { s1 += u1; s2 += u2; }
    f |= fu;
  }
  // rows 1 and 3 take lane 15 of the row before; then rows 2 and 3 take lane 31 (rows 0-1 complete)
  {
    const double u1 = dpp_f64<0x142, 0xa>(s1), u2 = dpp_f64<0x142, 0xa>(s2);
    const int fu = dpp_i32<0x142, 0xa>(f, 0);
    if (!f) { s1 += u1; s2 += u2; }
    f |= fu;
  }
  {
    const double u1 = dpp_f64<0x143, 0xc>(s1), u2 = dpp_f64<0x143, 0xc>(s2);
    const int fu = dpp_i32<0x143, 0xc>(f, 0);
    if (!f) { s1 += u1; s2 += u2; }
    f |= fu;
  }
}

// all threads of the workgroup obtain the same totals; fixed summation tree (deterministic).
template <int NW>
__device__ __forceinline__ void wg_allreduce2(double &a, double &b, double *lds /* [2 * NW] */) {
  wave_allreduce_sum2(a, b);
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    lds[wid * 2] = a;
    lds[wid * 2 + 1] = b;
  }
  __syncthreads();
  a = 0;
  b = 0;
#pragma unroll
  for (int w = 0; w < NW; w++) {
    a += lds[w * 2];
    b += lds[w * 2 + 1];
  }
  __syncthreads();
}

// sample_normal (FMTrainer.hpp:122-125) with the N(0,1) variate supplied
__device__ __forceinline__ double sample_normal_z(double quad, double first, double z) {
  return (first / quad) + z / sqrt(quad);
}
// The same draw for the single-wavefront chains, where every instruction of the draw is on the sequential critical path
// (the IEEE division + square root + division above are ~60 dependent instructions, and a dependent fp64 instruction of
// a lone wavefront costs ~12 cycles -- scripts/ubench/issue_rate.hip): r = quad^(-1/2) by v_rsq_f64 and two coupled
// Newton steps (g -> sqrt(quad), h -> r / 2; quadratic convergence from ~2^-26), then first * r^2 + z * r. Differs from the divisions above by a few ulp. quad = lambda +
// alpha * S2 is a positive normal number.
__device__ __forceinline__ double sample_normal_z_fast(double quad, double first, double z) {
  const double r0 = __builtin_amdgcn_rsq(quad);
  const double f4 = 4.0 * first, z2 = z + z;  // (off the dependent chain of the square root)
  double g = quad * r0, h = 0.5 * r0;
  double e = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, e, g);
  h = __builtin_fma(h, e, h);
  e = __builtin_fma(-h, g, 0.5);
  h = __builtin_fma(h, e, h);  // h = r / 2 to rounding level: two coupled steps square the 2^-26 error of v_rsq_f64 twice
  return __builtin_fma(f4, h, z2) * h;  // first r^2 + z r
}
template <bool FAST>
__device__ __forceinline__ double sample_normal_zt(double quad, double first, double z) {
  return FAST ? sample_normal_z_fast(quad, first, z) : sample_normal_z(quad, first, z);
}

}  // namespace mfm
