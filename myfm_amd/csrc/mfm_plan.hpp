// mfm_plan.hpp -- the conflict-free execution plan of a column sweep.
//
// The reference updates the features of a table strictly in index order (FMTrainer.hpp:343, :419).
// Two features whose columns share no row touch disjoint state and commute, so the plan assigns
// level(j) = 1 + max level of the earlier columns sharing a row with j and runs the levels in order:
//   PAR step   one level with enough work: its columns run concurrently, binned by length into
//              wavefront-per-column / workgroup-per-column / chunked long columns;
//   CHAIN step a run of consecutive tiny levels (dense / multi-hot columns, small blocks): one
//              workgroup walks their columns one after the other inside a single launch.
// The draws are identical to the sequential order given the same per-feature variates.
#pragma once
#include <algorithm>
#include <vector>

#include "mfm_common.hpp"
#include "mfm_kernels.hpp"

namespace mfm {

struct ParLevel {
  DevBuf<int32_t> wave_cols, wg_cols, long_cols, long_chunk_ptr;
  DevBuf<ChunkDesc> chunks;
  int n_wave = 0, n_wg = 0, n_long = 0, n_chunks = 0;
  int64_t nnz_wave = 0, nnz_wg = 0, nnz_long = 0;
};

struct ChainRun {
  DevBuf<int32_t> cols;
  int n_cols = 0;
  int64_t nnz = 0;
};

struct Step {
  bool is_chain = false;
  ParLevel par;
  ChainRun chain;
};

struct StepPlan {
  std::vector<Step> steps;
  int n_levels = 0;
  int max_chunks = 0, max_long = 0;
  int64_t launches = 0;

  // a level is "tiny" when running it as its own launches cannot fill the device anyway
  static bool tiny(size_t n_cols, int64_t nnz) { return n_cols <= 8 && nnz <= 16384; }

  void build(const HostCsr &csc, int wave_cap, int wg_cap) {
    std::vector<int32_t> level;
    n_levels = column_levels(csc, level);
    std::vector<std::vector<int32_t>> by_level((size_t)n_levels);
    for (int64_t j = 0; j < csc.rows; j++) by_level[level[j]].push_back((int32_t)j);
    steps.clear();
    launches = 0;
    std::vector<int32_t> run;
    int64_t run_nnz = 0;
    auto flush_run = [&]() {
      if (run.empty()) return;
      steps.emplace_back();
      Step &s = steps.back();
      s.is_chain = true;
      s.chain.n_cols = (int)run.size();
      s.chain.nnz = run_nnz;
      s.chain.cols.upload(run);
      launches += 1;
      run.clear();
      run_nnz = 0;
    };
    for (int32_t l = 0; l < n_levels; l++) {
      int64_t lnnz = 0;
      for (int32_t j : by_level[l]) lnnz += csc.ptr[j + 1] - csc.ptr[j];
      if (tiny(by_level[l].size(), lnnz)) {
        for (int32_t j : by_level[l]) run.push_back(j);
        run_nnz += lnnz;
        continue;
      }
      flush_run();
      steps.emplace_back();
      Step &s = steps.back();
      ParLevel &L = s.par;
      std::vector<int32_t> wv, wg, lg, cptr;
      std::vector<ChunkDesc> ch;
      for (int32_t j : by_level[l]) {
        int64_t len = csc.ptr[j + 1] - csc.ptr[j];
        if (len <= wave_cap) {
          wv.push_back(j);
          L.nnz_wave += len;
        } else if (len <= wg_cap) {
          wg.push_back(j);
          L.nnz_wg += len;
        } else {
          cptr.push_back((int32_t)ch.size());
          for (int64_t b = 0; b < len; b += wg_cap)
            ch.push_back(ChunkDesc{csc.ptr[j] + b, (int32_t)std::min<int64_t>(wg_cap, len - b), (int32_t)lg.size()});
          lg.push_back(j);
          L.nnz_long += len;
        }
      }
      cptr.push_back((int32_t)ch.size());
      L.n_wave = (int)wv.size();
      L.n_wg = (int)wg.size();
      L.n_long = (int)lg.size();
      L.n_chunks = (int)ch.size();
      L.wave_cols.upload(wv);
      L.wg_cols.upload(wg);
      L.long_cols.upload(lg);
      L.long_chunk_ptr.upload(cptr);
      L.chunks.upload(ch.data(), ch.size());
      max_chunks = std::max(max_chunks, L.n_chunks);
      max_long = std::max(max_long, L.n_long);
      launches += (L.n_wave ? 1 : 0) + (L.n_wg ? 1 : 0) + (L.n_long ? 3 : 0);
    }
    flush_run();
  }
};

// scratch shared by every long-column launch of a ctx
struct LongScratch {
  DevBuf<double2> partial, oldnew;
  void reserve(int max_chunks, int max_long) {
    if ((size_t)max_chunks > partial.n) partial.alloc((size_t)max_chunks);
    if ((size_t)max_long > oldnew.n) oldnew.alloc((size_t)max_long);
  }
};

template <class P>
static void run_plan(hipStream_t s, Timing &tm, const StepPlan &plan, const SweepArgs &a, LongScratch &ls, int kc_wave,
                     int kc_wg, int kc_lstats, int kc_ldraw, int kc_lapply, int kc_chain) {
  for (const Step &st : plan.steps) {
    if (st.is_chain) {
      TimedLaunch t(tm, s, kc_chain, P::BYTES * st.chain.nnz);
      hipLaunchKernelGGL((k_chain<P>), dim3(1), dim3(CHAIN_WG), 0, s, a, st.chain.cols.p, st.chain.n_cols);
      continue;
    }
    const ParLevel &L = st.par;
    if (L.n_wave) {
      TimedLaunch t(tm, s, kc_wave, P::BYTES * L.nnz_wave);
      hipLaunchKernelGGL((k_sweep_wave<P>), dim3((L.n_wave + WG / WAVE - 1) / (WG / WAVE)), dim3(WG), 0, s, a,
                         L.wave_cols.p, L.n_wave);
    }
    if (L.n_wg) {
      TimedLaunch t(tm, s, kc_wg, P::BYTES * L.nnz_wg);
      hipLaunchKernelGGL((k_sweep_wg<P>), dim3(L.n_wg), dim3(WG), 0, s, a, L.wg_cols.p);
    }
    if (L.n_long) {
      {
        TimedLaunch t(tm, s, kc_lstats, P::STAT_BYTES * L.nnz_long);
        hipLaunchKernelGGL((k_long_stats<P>), dim3(L.n_chunks), dim3(WG), 0, s, a, L.chunks.p, L.long_cols.p, ls.partial.p);
      }
      {
        TimedLaunch t(tm, s, kc_ldraw, 16.0 * L.n_chunks);
        hipLaunchKernelGGL((k_long_draw<P>), dim3((L.n_long + 63) / 64), dim3(64), 0, s, a, L.long_cols.p,
                           L.long_chunk_ptr.p, L.n_long, ls.partial.p, ls.oldnew.p);
      }
      {
        TimedLaunch t(tm, s, kc_lapply, P::BYTES * L.nnz_long);
        hipLaunchKernelGGL((k_long_apply<P>), dim3(L.n_chunks), dim3(WG), 0, s, a, L.chunks.p, ls.oldnew.p);
      }
    }
  }
  MFM_HIP_CHECK(hipGetLastError());
}

}  // namespace mfm
