// mfm_rng.hpp -- device-side reproduction of the reference's random stream.
//
// The reference draws every variate of the Gibbs iteration from ONE std::mt19937 through libstdc++
// distribution objects (BaseFMTrainer.hpp:193; FMTrainer.hpp:122-125, :142-143, :164-165). For the
// regression / probit-classification sweeps the sequence of engine outputs consumed per iteration
// does not depend on the model state, so the whole stream can be produced ahead of the
// sweeps. Drawing it on the host costs ~140 ms per iteration at the ML-10M shape (2.7 M normals);
// here it is produced on the GPU, on a side stream, bit-compatible with libstdc++:
//   k_mt_generate : the MT19937 recurrence + tempering (one wavefront, state in LDS) -> ring of raw
//                   32-bit outputs in HBM;
//   k_rng_consume : runs the iteration's "draw program" over that ring:
//                   NORMALS(n) = n x `normal_distribution<double>(0,1)(gen)` with a FRESH distribution
//                                each (Marsaglia polar, /usr/include/c++/11/bits/random.tcc:1802-1835:
//                                attempts of 4 outputs, returns y*mult, discards x*mult) -- evaluated
//                                1024 x 4 attempts at a time with a block-wide prefix sum over the
//                                accept flags to keep the sequential semantics;
//                   GAMMA(a)   = the unit-scale variate of `gamma_distribution<double>(a, b)(gen)`
//                                (Marsaglia-Tsang, random.tcc:2337-2392); the caller multiplies by b.
// generate_canonical<double,53> (random.tcc:3348-3380) takes two outputs per uniform, low word first.
// The accept/reject decisions use only +,* and are exact; log() may differ from glibc's in the last
// ulp, so a variate can differ from the host's by 1 ulp (tests/test_gpu_rng.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>

#include "mfm_rng_state.hpp"

namespace mfm {

#ifndef MFM_RNG_CONSUME_THREADS
#define MFM_RNG_CONSUME_THREADS 1024
#endif
constexpr int RNG_CONSUME_THREADS = MFM_RNG_CONSUME_THREADS;
constexpr int RNG_ATT = 4;  // attempts per thread per batch
#ifndef MFM_RNG_SPEC
#define MFM_RNG_SPEC 16  // gamma draws evaluated speculatively at once (k_rng_consume), at most the number of waves
#endif

struct RngOp {
  int32_t kind;   // 0 = NORMALS, 1 = GAMMA
  int32_t dest;   // 0 = hyper variates, 1 = z_w, 2 = z_V
  int64_t count;  // NORMALS: number of draws (GAMMA: 1)
  int64_t offset; // first destination index
  double shape;   // GAMMA: alpha
};

__device__ __forceinline__ uint32_t mt_twist(uint32_t a, uint32_t b, uint32_t c) {
  const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// The MT19937 block update in closed form. With x the old 624-word block, x' the new one and
// G[k] = twist(x[k], x[k+1]) (k <= 622, old words only), the sequential recurrence
//   x'[k] = x[k+397] ^ G[k] (k < 227),  x'[k] = x'[k-227] ^ G[k] (227 <= k < 623)
// unrolls to expressions over OLD words only:
//   k in [0,227)   : x[k+397] ^ G[k]
//   k in [227,454) : x[k+170] ^ G[k-227] ^ G[k]
//   k in [454,623) : x[k-57]  ^ G[k-454] ^ G[k-227] ^ G[k]
//   k = 623        : x'[396] ^ twist(x[623], x'[0]),  x'[396] = x[566]^G[169]^G[396], x'[0] = x[397]^G[0]
// so a whole block is ONE parallel step (624 threads, double-buffered in LDS, one barrier per block)
// instead of three dependent 227-wide fronts.
// The ring holds the UNTEMPERED state words; consumers apply mt_temper() when they read (they run on
// the whole GPU, the generator is the one serial chain and is instruction-issue bound on its CU).
// Word 623 needs two dependent twists: it is computed by lane 0 of an eleventh wave so that no wave
// executes two code paths.
constexpr int MT_GEN_THREADS = 704;
__device__ __forceinline__ uint32_t mt_G(const uint32_t *x, int k) { return mt_twist(x[k], x[k + 1], 0u); }

__global__ __launch_bounds__(MT_GEN_THREADS) void k_mt_generate(RngState *__restrict__ st, uint32_t *__restrict__ raw,
                                                                uint64_t mask, uint64_t need) {
  __shared__ uint32_t buf[2][MT_N + 1];
  const int t = threadIdx.x;
  __builtin_amdgcn_s_setprio(3);
  if (t < MT_N) buf[0][t] = st->mt[t];
  int pos = st->mt_pos;
  uint64_t p_gen = st->p_gen;
  const uint64_t target = st->p_cons + need;
  const uint32_t m32 = (uint32_t)mask;  // the ring has < 2^32 entries
  __syncthreads();
  int cur = 0;
  if (pos < MT_N && p_gen < target) {  // the not yet emitted tail of the current block
    if (t >= pos && t < MT_N) raw[(p_gen + (uint64_t)(t - pos)) & mask] = buf[0][t];
    p_gen += (uint64_t)(MT_N - pos);
    pos = MT_N;
  }
  // element this thread produces: threads 0..622 their own index, thread 640 (wave 10, lane 0) word 623
  const int k = t < 623 ? t : (t == 640 ? 623 : -1);
  uint32_t off = (uint32_t)p_gen & m32;
  while (p_gen < target) {
    const uint32_t *x = buf[cur];
    uint32_t *xn = buf[cur ^ 1];
    if (k >= 0) {
      uint32_t v;
      if (k < 227) {
        v = x[k + 397] ^ mt_G(x, k);
      } else if (k < 454) {
        v = x[k + 170] ^ mt_G(x, k - 227) ^ mt_G(x, k);
      } else if (k < 623) {
        v = x[k - 57] ^ mt_G(x, k - 454) ^ mt_G(x, k - 227) ^ mt_G(x, k);
      } else {
        const uint32_t x0n = x[397] ^ mt_G(x, 0);
        const uint32_t x396n = x[566] ^ mt_G(x, 169) ^ mt_G(x, 396);
        v = mt_twist(x[623], x0n, x396n);
      }
      xn[k] = v;
      raw[(off + (uint32_t)k) & m32] = v;
    }
    p_gen += MT_N;
    off += MT_N;
    cur ^= 1;
    // LDS-only barrier: __syncthreads() would also drain vmcnt (wait for this block's global stores)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  __syncthreads();
  if (t < MT_N) st->mt[t] = buf[cur][t];
  if (t == 0) {
    st->mt_pos = pos;
    st->p_gen = p_gen;
  }
}

// ---- parallel generation with jump-ahead ----------------------------------------------------------------
// One workgroup is a serial chain (~0.33 us per 624-word block: 7.4 ms for the 13.5 M words of an iteration at
// the ML-10M shape -- the wall once the sweeps dropped below that). Here workgroup p produces blocks
// [p * par_blocks, (p + 1) * par_blocks) of the request. To start there it needs the block before its
// first one: it walks 33 blocks from the stored state (20 592 words, in LDS) and applies the jump polynomial
// g_p = x^((p * par_blocks - 1) * 624) mod phi (mfm_mtjump.hpp): word l of the target block is the XOR of
// x[l + i] over the set bits i of g_p -- ~10 k conflict-free LDS reads per lane, no barriers. The new state is
// written to a staging copy (late workgroups must still see the old one) and committed by k_mt_commit.
constexpr int MT_PAR_BLOCKS = 512;  // blocks per workgroup at most (and the request size from which the generator is parallel)
// blocks per workgroup for a request of `blocks`: about 170 workgroups (each takes a CU: 87 KB of LDS for its jump), between
// 64 and 512 blocks, multiples of 16. A workgroup's jump costs ~0.45 ms whatever it generates afterwards (0.4 us per
// block), so the generator is asked for several iterations' worth at a time (MT_GEN_BATCH): one jump per four iterations.
// Round 4: six where an iteration needs at most 2^25 outputs (config 3: 13.5 M; 316 -> 323 it/s, 8: 321, 16: 314,
// profiles/r04_q_res_cus.txt); larger problems keep four (the ring holds the batch: 4 GB of raw words at config 5).
constexpr int MT_GEN_BATCH = 4, MT_GEN_BATCH_SMALL = 6;
static inline int mt_gen_batch(double outputs_per_iteration) {
  if (const char *e = std::getenv("MFM_RNG_GEN_BATCH")) return std::max(1, std::min(16, std::atoi(e)));
  return outputs_per_iteration <= 33554432.0 ? MT_GEN_BATCH_SMALL : MT_GEN_BATCH;
}
static inline int mt_par_blocks_for(int64_t blocks) {
  if (const char *e = std::getenv("MFM_RNG_PAR_BLOCKS")) return std::max(32, std::min(MT_PAR_BLOCKS, std::atoi(e)));
  const int64_t b = ((blocks + 169) / 170 + 15) / 16 * 16;
  return (int)std::max<int64_t>(64, std::min<int64_t>(MT_PAR_BLOCKS, b));
}
constexpr int MT_JUMP_SPAN = 33;  // blocks covering 19937 + 624 words

// PHASE 0: jump + generation in one launch. PHASE 1: the jump only -- workgroup p leaves the block before its first one in
// starts[p] -- and PHASE 2: the generation from those blocks. The jump needs 87 KB of LDS, the generation 5 KB: as two
// launches the ~0.15 ms of generation do not take the LDS of 169 CUs away from the scorer running beside them (update_e
// ran 0.5 instead of 0.3 ms next to the fused launch).
template <int PHASE>
__global__ __launch_bounds__(MT_GEN_THREADS) void k_mt_generate_par(const RngState *__restrict__ st, RngState *__restrict__ st_next,
                                                                    uint32_t *__restrict__ raw, uint64_t mask, uint64_t need,
                                                                    const uint32_t *__restrict__ jump_tab, int par_blocks,
                                                                    uint32_t *__restrict__ starts, uint64_t need_min) {
  extern __shared__ uint32_t lds_seq[];  // [MT_JUMP_SPAN * 624] sequence (PHASE 0, 1), then 2 x 625 generation buffers
  uint32_t *seq = lds_seq;
  uint32_t(*buf)[MT_N + 1] = (uint32_t(*)[MT_N + 1])(lds_seq + (PHASE == 2 ? 0 : MT_JUMP_SPAN * MT_N));
  const int t = threadIdx.x, p = blockIdx.x;
  __builtin_amdgcn_s_setprio(3);
  int pos = st->mt_pos;
  uint64_t p_gen = st->p_gen;
  // `need` outputs ahead of the consumer (several iterations' worth) -- unless the next iteration's `need_min` are there
  // already: then nothing at all (the launch costs a few microseconds; every workgroup takes the same decision)
  const uint64_t target = p_gen >= st->p_cons + need_min ? 0 : st->p_cons + need;
  if (t < MT_N) buf[0][t] = st->mt[t];
  __syncthreads();
  if (pos < MT_N && p_gen < target) {  // the not yet emitted tail of the current block
    if (PHASE != 1 && p == 0 && t >= pos && t < MT_N) raw[(p_gen + (uint64_t)(t - pos)) & mask] = buf[0][t];
    p_gen += (uint64_t)(MT_N - pos);
    pos = MT_N;
  }
  const int64_t nblk = p_gen < target ? (int64_t)((target - p_gen + MT_N - 1) / MT_N) : 0;
  const int64_t b0 = (int64_t)p * par_blocks, b1 = min(nblk, b0 + par_blocks);
  if (nblk == 0) {  // nothing to generate: only the position may have moved
    if (PHASE != 1 && p == 0) {
      if (t < MT_N) st_next->mt[t] = buf[0][t];
      if (t == 0) {
        st_next->mt_pos = pos;
        st_next->p_gen = p_gen;
      }
    }
    return;
  }
  if (b0 >= nblk) return;
  const int k = t < 623 ? t : (t == 640 ? 623 : -1);
  auto step = [&](const uint32_t *x, uint32_t *xn) -> uint32_t {
    uint32_t v = 0;
    if (k >= 0) {
      if (k < 227) {
        v = x[k + 397] ^ mt_G(x, k);
      } else if (k < 454) {
        v = x[k + 170] ^ mt_G(x, k - 227) ^ mt_G(x, k);
      } else if (k < 623) {
        v = x[k - 57] ^ mt_G(x, k - 454) ^ mt_G(x, k - 227) ^ mt_G(x, k);
      } else {
        const uint32_t x0n = x[397] ^ mt_G(x, 0);
        const uint32_t x396n = x[566] ^ mt_G(x, 169) ^ mt_G(x, 396);
        v = mt_twist(x[623], x0n, x396n);
      }
      xn[k] = v;
    }
    return v;
  };
  int cur = 0;
  if (PHASE == 2) {
    if (p > 0) {
      __syncthreads();
      if (t < MT_N) buf[0][t] = starts[(size_t)p * MT_N + t];
      __syncthreads();
    }
  } else if (p > 0) {
    // blocks r = 0 .. 32 after the stored state (all of them generated words: the relation holds for every bit)
    step(buf[0], seq);
    __syncthreads();
    for (int r = 1; r < MT_JUMP_SPAN; r++) {
      step(seq + (r - 1) * MT_N, seq + r * MT_N);
      __syncthreads();
    }
    // block b0 - 1 = block 0 advanced by (p * par_blocks - 1) blocks
    // (the polynomial's 624 words are staged in LDS by one coalesced load -- buf[1] is free until the generation loop --:
    //  a global load per word inside the loop below cost ~350 us per workgroup, twice the generation of its blocks)
    if (t < MT_N) buf[1][t] = jump_tab[(size_t)p * MT_N + t];
    __syncthreads();
    if (t < MT_N) {
      const uint32_t *g = buf[1];
      uint32_t y = 0;
      for (int w0 = 0; w0 < MT_N; w0 += 8) {
        uint32_t gq[8];
#pragma unroll
        for (int i = 0; i < 8; i++) gq[i] = g[w0 + i];  // wave-uniform LDS reads (broadcast), 8 in flight
#pragma unroll
        for (int i = 0; i < 8; i++) {
          uint32_t gw = __builtin_amdgcn_readfirstlane(gq[i]);
          const uint32_t *xs = seq + t + 32 * (w0 + i);
          while (gw) {
            const int b = __builtin_ctz(gw);
            y ^= xs[b];
            gw &= gw - 1;
          }
        }
      }
      buf[0][t] = y;
      if (PHASE == 1) starts[(size_t)p * MT_N + t] = y;
    }
    __syncthreads();
  }
  if (PHASE == 1) return;
  const uint32_t m32 = (uint32_t)mask;
  uint32_t off = (uint32_t)((p_gen + (uint64_t)b0 * MT_N) & mask);
  for (int64_t b = b0; b < b1; b++) {
    const uint32_t *x = buf[cur];
    uint32_t *xn = buf[cur ^ 1];
    const uint32_t v = step(x, xn);
    if (k >= 0) raw[(off + (uint32_t)k) & m32] = v;
    off += MT_N;
    cur ^= 1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  if (b1 == nblk) {  // this workgroup holds the last block: the new generator state
    __syncthreads();
    if (t < MT_N) st_next->mt[t] = buf[cur][t];
    if (t == 0) {
      st_next->mt_pos = MT_N;
      st_next->p_gen = p_gen + (uint64_t)nblk * MT_N;
    }
  }
}
__global__ void k_mt_commit(RngState *__restrict__ st, const RngState *__restrict__ st_next) {
  const int t = threadIdx.x;
  if (t < MT_N) st->mt[t] = st_next->mt[t];
  if (t == 0) {
    st->mt_pos = st_next->mt_pos;
    st->p_gen = st_next->p_gen;
  }
}

struct RawReader {
  const uint32_t *raw;
  uint64_t mask, p;
  // optional window of tempered words in LDS: outputs [wbase, wbase + wlen) (a lone thread walking the ring in HBM pays a
  // dependent global round trip per word: ~4 us per gamma draw)
  const uint32_t *win = nullptr;
  uint64_t wbase = 0;
  uint32_t wlen = 0;
  // outputs from `limit` on do not exist yet (the speculative gamma candidates start at positions the chain may never reach):
  // they are not read -- a fixed word comes back (uniform = 0.75: every rejection loop accepts) and `past` is raised, which ends
  // the draw's loops; the candidate is then marked unusable
  uint64_t limit = ~0ull;
  bool past = false;
  __device__ __forceinline__ uint32_t next() {
    const uint64_t i = p++;
    if (i >= limit) {
      past = true;
      return 0xC0000000u;
    }
    const uint64_t o = i - wbase;
    if (o < (uint64_t)wlen) return win[o];
    return mt_temper(raw[i & mask]);
  }
  __device__ __forceinline__ double uniform() {
    const uint32_t lo = next();
    const uint32_t hi = next();
    return canonical(lo, hi);
  }
};

// std::normal_distribution<double> object state
struct NormalDist {
  bool saved_available = false;
  double saved = 0.0;
  __device__ __forceinline__ double operator()(RawReader &g) {
    double ret;
    if (saved_available) {
      saved_available = false;
      ret = saved;
    } else {
      double x, y, r2;
      do {
        x = 2.0 * g.uniform() - 1.0;
        y = 2.0 * g.uniform() - 1.0;
        r2 = x * x + y * y;
      } while (r2 > 1.0 || r2 == 0.0);
      const double mult = sqrt(-2 * log(r2) / r2);
      saved = x * mult;
      saved_available = true;
      ret = y * mult;
    }
    return ret * 1.0 + 0.0;
  }
};

// unit-scale gamma_distribution<double>(alpha, .)(gen): the caller multiplies by beta. The distribution's own
// normal_distribution member keeps the second value of a polar pair for its next call (random.tcc:2337-2392, :1802-1835).
__device__ __forceinline__ double gamma_unit(RawReader &g, double alpha) {
  const double malpha = alpha < 1.0 ? alpha + 1.0 : alpha;
  const double a1 = malpha - 1.0 / 3.0;
  const double a2 = 1.0 / sqrt(9.0 * a1);
  // (the "second value is available" flag lives in a vector register, pinned by the empty asm: kept as a lane mask across
  //  the divergent loops, the compiler lost it -- a rejected lane drew a new pair instead of taking the saved value, seen
  //  as soon as more than one lane of a wave evaluated draws)
  int have_saved = 0;
  double saved = 0.0;
  double u, v, n;
  for (;;) {
    for (;;) {
      asm volatile("" : "+v"(have_saved));
      if (have_saved != 0) {
        have_saved = 0;
        n = saved;
      } else {
        double x, y, r2;
        do {
          x = 2.0 * g.uniform() - 1.0;
          y = 2.0 * g.uniform() - 1.0;
          r2 = x * x + y * y;
        } while ((r2 > 1.0 || r2 == 0.0) && !g.past);
        const double mult = sqrt(-2 * log(r2) / r2);
        saved = x * mult;
        have_saved = 1;
        n = y * mult;
      }
      asm volatile("" : "+v"(have_saved));
      n = n * 1.0 + 0.0;
      v = 1.0 + a2 * n;
      if (v > 0.0 || g.past) break;
    }
    v = v * v * v;
    u = g.uniform();
    const bool again = u > 1.0 - 0.0331 * n * n * n * n && (log(u) > (0.5 * n * n + a1 * (1.0 - v + log(v))));
    if (!again || g.past) break;
  }
  if (alpha == malpha) return a1 * v;
  do u = g.uniform();
  while (u == 0.0 && !g.past);
  return pow(u, 1.0 / alpha) * a1 * v;
}

__global__ __launch_bounds__(RNG_CONSUME_THREADS) void k_rng_consume(RngState *__restrict__ st,
                                                                     const uint32_t *__restrict__ raw, uint64_t mask,
                                                                     const RngOp *__restrict__ ops, int op_begin,
                                                                     int op_end, double *__restrict__ hv,
                                                                     double *__restrict__ zw, double *__restrict__ zv) {
  constexpr int NW = RNG_CONSUME_THREADS / 64;
  constexpr int WIN = 8192;  // words of the gamma draws' look-ahead window (a draw takes ~10, rarely more than 100)
  __shared__ int s_tot[RNG_ATT][NW];
  __shared__ unsigned long long s_p;
  __shared__ int s_kstar;
  __shared__ uint32_t s_win[WIN];
  __shared__ double s_gval[NW][64];  // speculative gamma draws: value / outputs consumed up to the end, per (op, start)
  __shared__ int s_gend[NW][64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  uint64_t p = st->p_cons;
  const uint64_t p_gen_at_start = st->p_gen;
  uint64_t wbase = 0;
  uint32_t wlen = 0;
  for (int oi = op_begin; oi < op_end; oi++) {
    const RngOp op = ops[oi];
    double *dst = (op.dest == 0 ? hv : (op.dest == 1 ? zw : zv)) + op.offset;
    if (op.kind == 1) {
      // A run of consecutive GAMMA ops. Each draw starts where the one before ended -- a data-dependent number of outputs
      // later -- and a lone lane takes ~4 us per draw (dependent fp64 log / sqrt chains). So up to 16 draws are evaluated
      // at once, SPECULATIVELY: wave j evaluates op j of the run at 64 candidate start positions (a draw consumes an
      // even number of outputs, at least 6), then thread 0 walks the chain start -> end -> next start through the table;
      // where a start falls outside the candidates the walk stops and the next round begins there. Same variates, same
      // consumption as the sequential evaluation.
      int n_run = 1;
      while (oi + n_run < op_end && ops[oi + n_run].kind == 1) n_run++;
      int done_ops = 0;
      while (done_ops < n_run) {  // (every quantity here is workgroup-uniform)
        if (p < wbase || p + 2048 > wbase + wlen) {  // refill: the window starts at the next output
          __syncthreads();
          for (int i = tid; i < WIN; i += RNG_CONSUME_THREADS) s_win[i] = mt_temper(raw[(p + (uint64_t)i) & mask]);
          wbase = p;
          wlen = WIN;
          __syncthreads();
        }
        const int batch = min(MFM_RNG_SPEC < NW ? MFM_RNG_SPEC : NW, n_run - done_ops);
        if (wid < batch && (MFM_RNG_SPEC > 1 || lane == 0)) {
          const RngOp oj = ops[oi + done_ops + wid];
          RawReader g{raw, mask, p + (uint64_t)(6 * wid + 2 * lane), s_win, wbase, wlen, p_gen_at_start};
          s_gval[wid][lane] = gamma_unit(g, oj.shape);
          s_gend[wid][lane] = g.past ? -1 : (int)(g.p - p);  // (-1: the candidate ran into outputs that do not exist yet)
        }
        __syncthreads();
        if (tid == 0) {
          int pos = 0, j = 0;
          for (; j < batch; j++) {
            const int l2 = pos - 6 * j;
            if (l2 < 0 || l2 >= 128) break;  // (l2 is even: every draw consumes an even number of outputs)
            const RngOp oj = ops[oi + done_ops + j];
            double *dj = (oj.dest == 0 ? hv : (oj.dest == 1 ? zw : zv)) + oj.offset;
            dj[0] = s_gval[j][l2 >> 1];
            if (s_gend[j][l2 >> 1] < 0) {  // a draw ON the chain needs outputs that were never generated: the set is invalid
              st->error = 1;
              pos += 6;
            } else {
              pos = s_gend[j][l2 >> 1];
            }
          }
          s_p = p + (uint64_t)pos;
          s_kstar = j;  // draws resolved (>= 1: the first one's start is candidate 0)
        }
        __syncthreads();
        p = s_p;
        done_ops += s_kstar;
        __syncthreads();
      }
      oi += n_run - 1;
      continue;
    }
    int64_t done = 0;
    while (done < op.count) {
      bool acc[RNG_ATT];
      double yv[RNG_ATT], r2v[RNG_ATT];
      unsigned long long bal[RNG_ATT];
#pragma unroll
      for (int r = 0; r < RNG_ATT; r++) {
        const uint64_t base = p + 4ull * (uint64_t)(r * RNG_CONSUME_THREADS + tid);
        const uint32_t u0 = mt_temper(raw[(base + 0) & mask]), u1 = mt_temper(raw[(base + 1) & mask]);
        const uint32_t u2 = mt_temper(raw[(base + 2) & mask]), u3 = mt_temper(raw[(base + 3) & mask]);
        const double x = 2.0 * canonical(u0, u1) - 1.0;
        const double y = 2.0 * canonical(u2, u3) - 1.0;
        const double r2 = x * x + y * y;
        acc[r] = !(r2 > 1.0 || r2 == 0.0);
        yv[r] = y;
        r2v[r] = r2;
        bal[r] = __ballot(acc[r]);
        if (lane == 0) s_tot[r][wid] = __popcll(bal[r]);
      }
      __syncthreads();
      const int64_t needed = op.count - done;
      int64_t run = 0;  // accepted attempts before slice r
      int64_t total = 0;
      int rank[RNG_ATT];
#pragma unroll
      for (int r = 0; r < RNG_ATT; r++) {
        int before = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) {
          const int t = s_tot[r][w];
          before += (w < wid) ? t : 0;
          tot += t;
        }
        rank[r] = (int)run + before + __popcll(bal[r] & ((1ull << lane) - 1ull));
        run += tot;
      }
      total = run;
      const bool last_batch = total >= needed;
#pragma unroll
      for (int r = 0; r < RNG_ATT; r++) {
        if (acc[r] && rank[r] < needed) {
          const double mult = sqrt(-2 * log(r2v[r]) / r2v[r]);
          dst[done + rank[r]] = (yv[r] * mult) * 1.0 + 0.0;
          if (last_batch && rank[r] == needed - 1) s_kstar = r * RNG_CONSUME_THREADS + tid;
        }
      }
      __syncthreads();
      if (last_batch) {
        p += 4ull * (uint64_t)(s_kstar + 1);
        done = op.count;
      } else {
        p += 4ull * (uint64_t)(RNG_ATT * RNG_CONSUME_THREADS);
        done += total;
      }
      __syncthreads();
    }
  }
  if (tid == 0) {
    st->p_cons = p;
    if (p > st->p_gen) st->error = 1;
  }
}


// ---- big NORMALS ops: evaluate the attempts on the whole GPU, then compact (chunked scan) ------------
// Attempt a of the op uses outputs [p + 4a, p + 4a + 4); the op's i-th variate is the i-th accepted
// attempt. k_norm_eval evaluates every attempt of a window that holds >= count accepts (up to a
// 1e-15 tail, checked), k_norm_scan prefix-sums the per-chunk accept counts, k_norm_scatter places the
// accepted candidates by rank and moves the stream position just past the count-th accept.
constexpr int NORM_CHUNK = 1024;  // attempts per workgroup (256 threads x 4)

struct NormScratch {
  unsigned long long p_base;  // stream position of attempt 0
  int32_t insufficient;
  int32_t pad;
};

__global__ __launch_bounds__(256) void k_norm_eval(const RngState *__restrict__ st, const uint32_t *__restrict__ raw,
                                                   uint64_t mask, double *__restrict__ cand,
                                                   unsigned long long *__restrict__ masks, int *__restrict__ counts) {
  __shared__ int s_cnt[4];
  const uint64_t p = st->p_cons;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int cnt = 0;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int64_t a = (int64_t)blockIdx.x * NORM_CHUNK + r * 256 + tid;
    const uint64_t base = p + 4ull * (uint64_t)a;
    const uint32_t u0 = mt_temper(raw[(base + 0) & mask]), u1 = mt_temper(raw[(base + 1) & mask]);
    const uint32_t u2 = mt_temper(raw[(base + 2) & mask]), u3 = mt_temper(raw[(base + 3) & mask]);
    const double x = 2.0 * canonical(u0, u1) - 1.0;
    const double y = 2.0 * canonical(u2, u3) - 1.0;
    const double r2 = x * x + y * y;
    const bool acc = !(r2 > 1.0 || r2 == 0.0);
    if (acc) {
      const double mult = sqrt(-2 * log(r2) / r2);
      cand[a] = (y * mult) * 1.0 + 0.0;
    }
    const unsigned long long b = __ballot(acc);
    if (lane == 0) {
      masks[a >> 6] = b;
      cnt += __popcll(b);
    }
  }
  if (lane == 0) s_cnt[wid] = cnt;
  __syncthreads();
  if (tid == 0) counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// exclusive scan of counts[n_chunks] (in place -> offsets); single workgroup
__global__ __launch_bounds__(1024) void k_norm_scan(const RngState *__restrict__ st, int *__restrict__ counts, int n_chunks,
                                                    int64_t count, NormScratch *__restrict__ ns) {
  __shared__ int s_w[16];
  __shared__ int s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < n_chunks; base += 1024) {
    const int i = base + tid;
    const int v = i < n_chunks ? counts[i] : 0;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    if (lane == 63) s_w[wid] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wid; w++) woff += s_w[w];
    const int carry = s_carry;
    if (i < n_chunks) counts[i] = carry + woff + incl - v;
    __syncthreads();
    if (tid == 1023) s_carry = carry + woff + incl;
    __syncthreads();
  }
  if (tid == 0) {
    ns->p_base = st->p_cons;
    ns->insufficient = (int64_t)s_carry < count ? 1 : 0;
  }
}

__global__ __launch_bounds__(256) void k_norm_scatter(RngState *__restrict__ st, const double *__restrict__ cand,
                                                      const unsigned long long *__restrict__ masks,
                                                      const int *__restrict__ offs, int64_t count,
                                                      const NormScratch *__restrict__ ns, double *__restrict__ dst) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // the 16 ballot words of this chunk, in attempt order: word (r * 4 + wid) covers attempts r*256 + wid*64 ..
  const unsigned long long *m = masks + (int64_t)blockIdx.x * (NORM_CHUNK / 64);
  int64_t before = offs[blockIdx.x];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    int wbefore = 0;
    for (int w = 0; w < 4; w++) {
      const int c = __popcll(m[r * 4 + w]);
      if (w < wid) wbefore += c;
    }
    const unsigned long long b = m[r * 4 + wid];
    const bool acc = (b >> lane) & 1ull;
    const int64_t rank = before + wbefore + __popcll(b & ((1ull << lane) - 1ull));
    const int64_t a = (int64_t)blockIdx.x * NORM_CHUNK + r * 256 + tid;
    if (acc && rank < count) {
      dst[rank] = cand[a];
      if (rank == count - 1) {
        st->p_cons = ns->p_base + 4ull * (unsigned long long)(a + 1);
        if (ns->insufficient) st->error = 1;
      }
    }
    for (int w = 0; w < 4; w++) before += __popcll(m[r * 4 + w]);
  }
  if (blockIdx.x == 0 && tid == 0 && ns->insufficient) st->error = 1;
}

}  // namespace mfm
