// mfm_latent_api.hpp -- what mfm_hip.hip sees of the exact latent draws (mfm_latent.hip).
//
// The latent z of probit classification / ordered probit (FMTrainer.hpp:498-521, OProbitSampler.hpp:238-272) are drawn by
// the reference row after row from its one std::mt19937 inside data-dependent rejection loops (util.hpp:15-60). Every
// attempt of every row consumes exactly ONE "quad" of four engine outputs (two generate_canonical<double,53>):
//   left / right truncation, mu < 0 : one Marsaglia polar attempt; accepted when the pair exists and one of its two normals
//                                     passes `z > mu` (the distribution object lives for the call: the second normal of a
//                                     pair IS the next candidate, util.hpp:19-23);
//   left / right truncation, mu >= 0: (u1, u2) -> z = -log(u1) / alpha* + mu, accepted when u2 < exp(-(z - alpha*)^2 / 2);
//   two-sided                       : (u1, u2) -> z = a + (b - a) u1, accepted when u2 < rho(z).
// So the whole draw is ONE monotone lattice path over (row t, quad j): t += A(t, j), j += 1, where A depends on row t and quad
// j only. Paths started at different rows of the same quad never cross and coincide for ever once they meet. mfm_latent.hip
// cuts the quad axis into chunks, walks for every chunk ALL entering rows of a window that contains the true one with
// probability ~1 - 2e-5 (the walkers merge as they meet: "coalescing flows"), composes the chunks' maps, and re-walks the one
// true path of every 1/C-th of a chunk in parallel to write the draws. Same draws, same engine consumption as the sequential
// loop; when a window misses (or scratch overflows) nothing is consumed or written and the caller takes the sequential path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mfm_rng_state.hpp"

namespace mfm {

struct LatentJob {
  hipStream_t stream = nullptr;
  int64_t n = 0;                  // rows of the group, in draw order
  const int32_t *rows = nullptr;  // device: row of position i (nullptr: i itself)
  double2 *eq = nullptr;          // device, row order: .x = score on entry, score - z on exit
  const double *y = nullptr;      // device, row order
  int n_class = 0;                // 0: probit classification (side chosen by y > 0); >= 2: ordered probit
  const double *gamma = nullptr;  // device: n_class - 1 cutpoints
  RngState *state = nullptr;      // device: p_cons is where the draw starts; advanced on success
  const uint32_t *raw = nullptr;  // device: ring of untempered engine outputs
  uint64_t mask = 0;
};

// after the row pass (one host synchronisation): how many quads the draw may need
struct LatentPrep {
  double mean_quads = 0, var_quads = 0;  // sum over the rows of the expected quads / their variance
  int64_t q_cap = 0;                     // quads the run will look at (mean + k sigma + slack): 4 q_cap outputs must exist
  uint64_t p_cons = 0, p_gen = 0;        // the stream's position when the row pass ran
};

struct LatentStats {
  int32_t status = 0;  // 0 ok; 1 window missed the path; 2 snapshot space; 3 walker space; 4 q_cap too small; 5 a row too far out
  int32_t chunks = 0, subs = 0, attempts = 0;  // attempts: 2 = the first one's windows missed the path, the wider ones held it
  int64_t quads_used = 0;   // quads consumed (4 engine outputs each)
  int64_t walkers = 0;      // walkers the flows started with (sum of the windows)
  int64_t lq = 0;
  double ms_rows = 0, ms_quads = 0, ms_flow = 0, ms_final = 0;  // (MFM_LATENT_TIMING=1)
};

class LatentEngine {
 public:
  LatentEngine();
  ~LatentEngine();
  LatentEngine(const LatentEngine &) = delete;
  // row pass: per-row records, expected consumption; synchronises the stream once
  void prepare(const LatentJob &job, LatentPrep *prep);
  // quads, flows, resolution, final pass; synchronises once at the end. Returns stats.status (0: the draws are written and
  // state->p_cons moved; otherwise nothing was written).
  int run(const LatentJob &job, const LatentPrep &prep, LatentStats *stats);

 private:
  struct Impl;
  Impl *im;
};

}  // namespace mfm
