// mfm_rng_state.hpp -- the device random stream's state and word-level helpers, shared by mfm_rng.hpp (generator + the
// state-independent draw program) and mfm_latent.hip (the state-dependent latent draws of classification / ordered probit).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mfm {

constexpr int MT_N = 624, MT_M = 397;

struct RngState {
  uint64_t p_gen;   // absolute index of the next output to generate
  uint64_t p_cons;  // absolute index of the next output to consume
  int32_t error;    // 1: the consumer ran past the generated range
  int32_t mt_pos;   // libstdc++ _M_p
  uint32_t mt[MT_N];
};

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

// generate_canonical<double, 53>(mt19937): low word first, sum rounded to nearest, / 2^64
__device__ __forceinline__ double canonical(uint32_t lo, uint32_t hi) {
  const double sum = (double)lo + (double)hi * 4294967296.0;
  double ret = sum * (1.0 / 18446744073709551616.0);
  if (ret >= 1.0) ret = 0.99999999999999988898;  // nextafter(1, 0)
  return ret;
}

}  // namespace mfm
