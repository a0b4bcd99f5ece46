// mfm_chain_api.hpp -- what mfm_hip.hip (mfm_plan.hpp) sees of the streamed conflict-window chain, which is compiled in its own
// translation unit (mfm_chain.hip: plan upload, scratch rings, launch of k_cs_stream). Plan and exactness: mfm_chain_plan.hpp;
// kernel: mfm_chain_stream.hpp.
#pragma once
#include <hip/hip_runtime.h>

#include <memory>
#include <vector>

#include "mfm_common.hpp"
#include "mfm_policies.hpp"

namespace mfm {

struct CsStream;  // device-resident plan of one chain run + the rings its launches exchange through

struct CsStreamInfo {
  int n_steps = 0, Cg = 0, Lw = 0, NB = 0, RD = 0, n_slots = 0, max_hot_col = 0, max_enter = 0, max_exit = 0;
  long long n_cold = 0, n_hot = 0;
  double plan_seconds = 0;
};

// csc: the block's CSC as the planners hold it (csc.rows = columns, csc.cols = block rows); run: the chain's columns in sweep
// order. Returns nullptr when no window fits the walker's LDS (the caller keeps the conflict-batched form).
std::shared_ptr<CsStream> cs_stream_build(const HostCsr &csc, const std::vector<int32_t> &run, CsStreamInfo *info);
// One chain run on stream s: latent = update_V's policy (PBlockV), else update_w's (PBlockW). `error`: the ctx's device error
// word (raised by a wait that timed out). Enqueues only.
void cs_stream_launch(hipStream_t s, const SweepArgs &a, const CsStream &st, bool latent, int *error);
const CsStreamInfo &cs_stream_info(const CsStream &st);

}  // namespace mfm
