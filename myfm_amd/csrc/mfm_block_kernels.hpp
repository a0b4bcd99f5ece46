// mfm_block_kernels.hpp -- relation-block ("block structure", Rendle 2013) device path.
//
// Reference: FMTrainer.hpp:256-313 (linear weights) and :323-340, :378-482 (latent factors);
// caches per block row are RelationWiseCache (definitions.hpp:54-84), kept here as one 64-byte
// record rec[i] = {q, q_S, c, c_S, e, e_q, cardinality, -}.
//
// The O(N) passes through original_to_block are done as follows:
//   statistics + un-sync (:268-275, :401-417): by BLOCK ROW through the inverse map
//       (inv_ptr / inv_rows = for every block row the ascending list of training rows mapped to it),
//       so that every sum has one owner and a fixed order -- no atomics, deterministic;
//   re-sync (:306-311, :473-480): streaming over the training rows, gathering the block record.
#pragma once
#include <chrono>
#include <algorithm>
#include <mutex>
#include <vector>

#include "mfm_common.hpp"
#include "mfm_kernels.hpp"
#include "mfm_plan.hpp"

namespace mfm {

constexpr int INV_WAVE_CAP = 512;    // block rows with at most this many training rows: one wavefront
constexpr int INV_WG_CAP = 16384;    // ... one workgroup; longer ones are chunked

// rec[i].q = sum_l x_il theta_l ; optionally rec[i].q_S = sum_l x_il^2 theta_l^2 ; zero c, c_S, e, e_q.
// FMTrainer.hpp:265-266 / :331-333 / :388-393 (+ the .array() = 0 resets at :262-263, :396-399)
template <bool WITH_QS, bool WAVE_PER_ROW>
__global__ __launch_bounds__(WG) void k_block_rowcache(const int32_t *__restrict__ rowptr,
                                                       const int32_t *__restrict__ colidx,
                                                       const double *__restrict__ val,
                                                       const double *__restrict__ theta, double *__restrict__ rec,
                                                       int64_t B, double *__restrict__ qc = nullptr) {
  int64_t i;
  int lane = 0;
  if (WAVE_PER_ROW) {
    i = (int64_t)blockIdx.x * (WG / WAVE) + (threadIdx.x >> 6);
    lane = threadIdx.x & 63;
  } else {
    i = (int64_t)blockIdx.x * WG + threadIdx.x;
  }
  if (i >= B) return;
  const int32_t b = rowptr[i], e = rowptr[i + 1];
  double q = 0.0, qs = 0.0;
  for (int32_t p = b + lane; p < e; p += (WAVE_PER_ROW ? WAVE : 1)) {
    const double x = val[p], th = theta[colidx[p]];
    q += x * th;
    if (WITH_QS) qs += (x * x) * (th * th);
  }
  if (WAVE_PER_ROW) {
    q = wave_allreduce_sum(q);
    if (WITH_QS) qs = wave_allreduce_sum(qs);
    if (lane != 0) return;
  }
  double2 *r = (double2 *)rec + i * 4;
  r[0] = make_double2(q, qs);
  r[1] = make_double2(0.0, 0.0);
  r[2] = make_double2(0.0, 0.0);
  if (qc) qc[i] = q;  // compact copy for the q-cache build's gather (8-byte stride: a 500 000-row block is 4 MB, not 32)
}

// per-entry body of the statistics + un-sync pass for training row t mapped to a block row whose
// (q_B, q_S) are given. Accumulates into s[4] = {c, c_S, e, e_q} (V) or s[0] = e (w).
template <bool IS_W>
__device__ __forceinline__ void unsync_entry(double2 *__restrict__ eq, int32_t t, double qB, double qS, double *s) {
  double2 v = eq[t];
  if (IS_W) {
    s[0] += v.x;  // FMTrainer.hpp:271
    v.x -= qB;    // :272-273
    eq[t].x = v.x;
  } else {
    const double temp = v.y - qB;  // :402
    s[0] += temp;                  // :403
    s[1] += temp * temp;           // :404
    s[2] += v.x;                   // :405
    s[3] += v.x * temp;            // :406
    v.y = temp;                    // :408
    v.x -= (v.y * qB + 0.5 * qB * qB - 0.5 * qS);  // :412-415
    eq[t] = v;
  }
}

// The same sums taken AFTER the row was un-synced by a streaming pass (k_unsync_update): the stored q is temp, and the
// residual before the un-sync is e + (temp q_B + q_B^2 / 2 - q_S / 2) (:412-415 undone). Read-only.
template <bool IS_W>
__device__ __forceinline__ void unsync_entry_post(const double2 *__restrict__ eq, int32_t t, double qB, double qS, double *s) {
  const double2 v = eq[t];
  if (IS_W) {
    s[0] += v.x + qB;
  } else {
    const double temp = v.y;
    const double e_pre = v.x + (temp * qB + 0.5 * qB * qB - 0.5 * qS);
    s[0] += temp;
    s[1] += temp * temp;
    s[2] += e_pre;
    s[3] += e_pre * temp;
  }
}

template <bool IS_W>
__device__ __forceinline__ void unsync_store(double *__restrict__ rec, int64_t i, const double *s) {
  double2 *r = (double2 *)rec + i * 4;
  if (IS_W) {
    r[2] = make_double2(s[0], 0.0);
  } else {
    r[1] = make_double2(s[0], s[1]);
    r[2] = make_double2(s[2], s[3]);
  }
}

template <bool IS_W, bool POST = false>
__global__ __launch_bounds__(WG) void k_unsync_wave(const int64_t *__restrict__ inv_ptr,
                                                    const int32_t *__restrict__ inv_rows,
                                                    const int32_t *__restrict__ brow, int n, double2 *__restrict__ eq,
                                                    double *__restrict__ rec) {
  const int w = blockIdx.x * (WG / WAVE) + (threadIdx.x >> 6);
  if (w >= n) return;
  const int lane = threadIdx.x & 63;
  const int64_t i = brow[w];
  const double2 qq = ((const double2 *)rec)[i * 4];
  const int64_t b = inv_ptr[i], e = inv_ptr[i + 1];
  double s[4] = {0, 0, 0, 0};
  for (int64_t p = b + lane; p < e; p += WAVE) {
    if (POST) unsync_entry_post<IS_W>(eq, inv_rows[p], qq.x, qq.y, s); else unsync_entry<IS_W>(eq, inv_rows[p], qq.x, qq.y, s);
  }
#pragma unroll
  for (int k = 0; k < (IS_W ? 1 : 4); k++) s[k] = wave_allreduce_sum(s[k]);
  if (lane == 0) unsync_store<IS_W>(rec, i, s);
}

template <bool IS_W, bool POST = false>
__global__ __launch_bounds__(WG) void k_unsync_wg(const int64_t *__restrict__ inv_ptr,
                                                  const int32_t *__restrict__ inv_rows,
                                                  const int32_t *__restrict__ brow, double2 *__restrict__ eq,
                                                  double *__restrict__ rec) {
  __shared__ double lds[2 * WG / WAVE];
  const int64_t i = brow[blockIdx.x];
  const double2 qq = ((const double2 *)rec)[i * 4];
  const int64_t b = inv_ptr[i], e = inv_ptr[i + 1];
  double s[4] = {0, 0, 0, 0};
  for (int64_t p = b + threadIdx.x; p < e; p += WG) {
    if (POST) unsync_entry_post<IS_W>(eq, inv_rows[p], qq.x, qq.y, s); else unsync_entry<IS_W>(eq, inv_rows[p], qq.x, qq.y, s);
  }
  wg_allreduce2<WG / WAVE>(s[0], s[1], lds);
  if (!IS_W) wg_allreduce2<WG / WAVE>(s[2], s[3], lds);
  if (threadIdx.x == 0) unsync_store<IS_W>(rec, i, s);
}

struct InvChunk {
  int64_t begin;
  int32_t len;
  int32_t brow;
};
// long block rows: every chunk un-syncs its rows and leaves partial sums; k_unsync_long_fin adds them
template <bool IS_W, bool POST = false>
__global__ __launch_bounds__(WG) void k_unsync_long(const InvChunk *__restrict__ chunks,
                                                    const int32_t *__restrict__ inv_rows, double2 *__restrict__ eq,
                                                    const double *__restrict__ rec, double *__restrict__ partial) {
  __shared__ double lds[2 * WG / WAVE];
  const InvChunk c = chunks[blockIdx.x];
  const double2 qq = ((const double2 *)rec)[(int64_t)c.brow * 4];
  double s[4] = {0, 0, 0, 0};
  for (int p = threadIdx.x; p < c.len; p += WG) {
    if (POST) unsync_entry_post<IS_W>(eq, inv_rows[c.begin + p], qq.x, qq.y, s); else unsync_entry<IS_W>(eq, inv_rows[c.begin + p], qq.x, qq.y, s);
  }
  wg_allreduce2<WG / WAVE>(s[0], s[1], lds);
  if (!IS_W) wg_allreduce2<WG / WAVE>(s[2], s[3], lds);
  if (threadIdx.x == 0) {
    double *o = partial + (int64_t)blockIdx.x * 4;
    o[0] = s[0];
    o[1] = s[1];
    o[2] = s[2];
    o[3] = s[3];
  }
}
template <bool IS_W>
__global__ void k_unsync_long_fin(const int32_t *__restrict__ lrows, const int32_t *__restrict__ chunk_ptr, int n,
                                  const double *__restrict__ partial, double *__restrict__ rec) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n) return;
  double s[4] = {0, 0, 0, 0};
  for (int c = chunk_ptr[l]; c < chunk_ptr[l + 1]; c++)
    for (int k = 0; k < 4; k++) s[k] += partial[(int64_t)c * 4 + k];
  unsync_store<IS_W>(rec, lrows[l], s);
}

// un-sync of every training row, streaming (no statistics): (PRE: first the re-sync the previous block owes, as in
// k_unsync_stream). The statistics follow read-only through the inverse map (POST forms of the kernels above): a block row's
// scattered training rows are then read, not read-modified-written -- half the random traffic -- and the previous block's
// re-sync pass disappears.
template <bool IS_W, bool PRE>
__global__ __launch_bounds__(WG) void k_unsync_update(const int32_t *__restrict__ map, double2 *__restrict__ eq,
                                                      const double *__restrict__ rec, int64_t N, const int32_t *__restrict__ map_prev,
                                                      const double *__restrict__ rec_prev) {
  const int64_t t = (int64_t)blockIdx.x * WG + threadIdx.x;
  if (t >= N) return;
  double2 v = eq[t];
  const double2 qq = ((const double2 *)rec)[(int64_t)map[t] * 4];
  if (PRE) {
    const double2 qp = ((const double2 *)rec_prev)[(int64_t)map_prev[t] * 4];
    v.x += (v.y * qp.x + 0.5 * qp.x * qp.x - 0.5 * qp.y);
    v.y += qp.x;
  }
  if (IS_W) {
    v.x -= qq.x;
    eq[t].x = v.x;
  } else {
    const double temp = v.y - qq.x;
    v.y = temp;
    v.x -= (v.y * qq.x + 0.5 * qq.x * qq.x - 0.5 * qq.y);
    eq[t] = v;
  }
}

// ---- statistics + un-sync, STREAMING over the training rows (few block rows, each with very many training rows) --------
// The inverse-map kernels above gather eq[t] for the rows of one block row: when a block has ~10^3 rows and the table
// 5 * 10^7, those lists stride through the whole table (a 16-byte access per DRAM page: ~1 TB/s). Here the rows are
// read and written in order (36 B / row, coalesced) and the four sums per block row live in an LDS table of the
// workgroup, B x 32 bytes. Fixed order of every floating-point sum (deterministic, like the rest of the sampler):
//   * a workgroup owns a contiguous range of rows and walks it in steps of UNSYNC_R rows per thread;
//   * after a step's loads and updates the wavefronts add their contributions to the table ONE WAVEFRONT AT A TIME
//     (barriers in between), each with ds_add_f64 in program order (lanes of one instruction that hit the same block
//     row are serialised by the LDS unit in lane order);
//   * the workgroups' tables go to partial[workgroup][B][4] and k_unsync_stream_fin adds them in workgroup order.
// The LDS adds are ~1 row per clock and CU: a fifth of the kernel's memory time.
constexpr int UNSYNC_R = 4;          // rows per thread and step
constexpr int UNSYNC_STREAM_WGS = 1024;
constexpr int64_t UNSYNC_STREAM_MAX_B = 4096;  // 128 KiB table

// PRE: the re-sync of the PREVIOUS block (FMTrainer.hpp:473-480, its map / records) is applied to the row first -- that
// block's own streaming re-sync pass (a read + write of eq) is then not launched
template <bool IS_W, bool PRE = false>
__global__ __launch_bounds__(WG) void k_unsync_stream(const int32_t *__restrict__ map, double2 *__restrict__ eq,
                                                      const double *__restrict__ rec, int64_t N, int B, int64_t rows_per_wg,
                                                      double *__restrict__ partial, const int32_t *__restrict__ map_prev = nullptr,
                                                      const double *__restrict__ rec_prev = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NS = IS_W ? 1 : 4;
  double *tab = (double *)smem;  // [B][NS]
  for (int i = threadIdx.x; i < B * NS; i += WG) tab[i] = 0.0;
  __syncthreads();
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg, r1 = min(N, r0 + rows_per_wg);
  const int wv = threadIdx.x >> 6;
  for (int64_t base = r0; base < r1; base += (int64_t)WG * UNSYNC_R) {
    int key[UNSYNC_R], kprev[UNSYNC_R];
    double s[UNSYNC_R][NS];
    double2 v[UNSYNC_R];
#pragma unroll
    for (int u = 0; u < UNSYNC_R; u++) {
      const int64_t t = base + (int64_t)u * WG + threadIdx.x;
      key[u] = -1;
      kprev[u] = 0;
      if (t < r1) {
        key[u] = map[t];
        if (PRE) kprev[u] = map_prev[t];
        v[u] = eq[t];
      }
    }
    if (PRE) {
#pragma unroll
      for (int u = 0; u < UNSYNC_R; u++)
        if (key[u] >= 0) {
          const double2 qp = ((const double2 *)rec_prev)[(int64_t)kprev[u] * 4];
          v[u].x += (v[u].y * qp.x + 0.5 * qp.x * qp.x - 0.5 * qp.y);  // :476-478 of the previous block
          v[u].y += qp.x;                                             // :479
        }
    }
#pragma unroll
    for (int u = 0; u < UNSYNC_R; u++) {
      const int64_t t = base + (int64_t)u * WG + threadIdx.x;
      if (key[u] >= 0) {
        const double2 qq = ((const double2 *)rec)[(int64_t)key[u] * 4];
        if (IS_W) {
          s[u][0] = v[u].x;   // FMTrainer.hpp:271
          v[u].x -= qq.x;     // :272-273
          eq[t].x = v[u].x;
        } else {
          const double temp = v[u].y - qq.x;  // :402
          s[u][0] = temp;                     // :403
          s[u][1] = temp * temp;              // :404
          s[u][2] = v[u].x;                   // :405
          s[u][3] = v[u].x * temp;            // :406
          v[u].y = temp;                      // :408
          v[u].x -= (v[u].y * qq.x + 0.5 * qq.x * qq.x - 0.5 * qq.y);  // :412-415
          eq[t] = v[u];
        }
      }
    }
    for (int w = 0; w < WG / WAVE; w++) {
      if (wv == w) {
#pragma unroll
        for (int u = 0; u < UNSYNC_R; u++)
          if (key[u] >= 0) {
#pragma unroll
            for (int k = 0; k < NS; k++)
              __hip_atomic_fetch_add(&tab[key[u] * NS + k], s[u][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
      }
      __syncthreads();
    }
  }
  double *o = partial + (int64_t)blockIdx.x * B * NS;
  for (int i = threadIdx.x; i < B * NS; i += WG) o[i] = tab[i];
}
// 32 outputs x 8 slices of the workgroup tables per 256 threads: slice sl adds tables sl, sl + 8, ... in order, the eight
// slice sums are then added in slice order (fixed association)
template <bool IS_W>
__global__ __launch_bounds__(256) void k_unsync_stream_fin(const double *__restrict__ partial, int n_wg, int B,
                                                           double *__restrict__ rec) {
  constexpr int NS = IS_W ? 1 : 4;
  __shared__ double part[8][32];
  const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + o;  // (block row, statistic)
  double acc = 0.0;
  if (i < B * NS)
    for (int g = sl; g < n_wg; g += 8) acc += partial[(int64_t)g * B * NS + i];
  part[sl][o] = acc;
  __syncthreads();
  if (sl == 0 && i < B * NS) {
    double t = part[0][o];
#pragma unroll
    for (int k = 1; k < 8; k++) t += part[k][o];
    const int row = i / NS, k = i - row * NS;
    rec[(int64_t)row * BLOCK_REC + (IS_W ? 4 : 2 + k)] = t;
  }
}

__global__ void k_save_q(const double *__restrict__ rec, int64_t B, double2 *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) out[i] = ((const double2 *)rec)[i * 4];  // (q_B, q_S)
}

// re-sync, streaming over training rows: FMTrainer.hpp:473-480 (V) / :306-311 (w)
template <bool IS_W>
__global__ __launch_bounds__(WG) void k_resync(const int32_t *__restrict__ map, const double *__restrict__ rec,
                                               double2 *__restrict__ eq, int64_t N) {
  const int64_t t = (int64_t)blockIdx.x * WG + threadIdx.x;
  if (t >= N) return;
  const double2 qq = ((const double2 *)rec)[(int64_t)map[t] * 4];
  if (IS_W) {
    eq[t].x += qq.x;
  } else {
    double2 v = eq[t];
    v.x += (v.y * qq.x + 0.5 * qq.x * qq.x - 0.5 * qq.y);
    v.y += qq.x;
    eq[t] = v;
  }
}

// sharded mode: pack / unpack the per-block-row statistics that must be summed over the ranks
// (V: c, c_S, e, e_q = record words 2..5; w: e = word 4; setup: cardinality = word 6)
__global__ void k_rec_pack(const double *__restrict__ rec, double *__restrict__ buf, int64_t B, int first, int n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * n) return;
  buf[i] = rec[(i / n) * BLOCK_REC + first + (i % n)];
}
__global__ void k_rec_unpack(double *__restrict__ rec, const double *__restrict__ buf, int64_t B, int first, int n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * n) return;
  rec[(i / n) * BLOCK_REC + first + (i % n)] = buf[i];
}

// block-level caches of the fused re-score: per block row  bl = sum_l x w_l,
// bq[s] = sum_l x v_ls, bs = sum_s sum_l x^2 v_ls^2   (FM.hpp:81, :104-106, :121-127)
template <int GS, int SPL>
__global__ __launch_bounds__(WG) void k_block_score_cache(const int32_t *__restrict__ rowptr,
                                                          const int32_t *__restrict__ colidx,
                                                          const double *__restrict__ val,
                                                          const double *__restrict__ Vt /* block's first row */,
                                                          const double *__restrict__ w /* block's first */, int K,
                                                          int KS, double *__restrict__ bq, double *__restrict__ bl,
                                                          double *__restrict__ bs, int64_t B) {
  const int64_t i = ((int64_t)blockIdx.x * WG + threadIdx.x) / GS;
  const int lig = threadIdx.x % GS;
  const bool live = i < B;
  double a[SPL], b = 0.0, lin = 0.0;
#pragma unroll
  for (int u = 0; u < SPL; u++) a[u] = 0.0;
  if (live) {
    for (int32_t p = rowptr[i]; p < rowptr[i + 1]; p++) {
      const int32_t j = colidx[p];
      const double x = val[p], x2 = x * x;
      if (lig == 0) lin += x * w[j];
      const double *row = Vt + (int64_t)j * KS;
#pragma unroll
      for (int u = 0; u < SPL; u++) {
        const int s = lig + u * GS;
        if (s < K) {
          const double v = row[s];
          a[u] += x * v;
          b += x2 * (v * v);
        }
      }
    }
  }
#pragma unroll
  for (int m = GS / 2; m >= 1; m >>= 1) b += __shfl_xor(b, m, WAVE);
  if (live) {
#pragma unroll
    for (int u = 0; u < SPL; u++) {
      const int s = lig + u * GS;
      if (s < K) bq[i * KS + s] = a[u];
    }
    if (lig == 0) {
      bl[i] = lin;
      bs[i] = b;
    }
  }
}

// ---------------------------------------------------------------------------------------------
static inline int cdiv_i(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- X_t = X.transpose() (BaseFMTrainer.hpp:61) on the device: a stable radix sort of the stored entries by column.
// Entry order inside a column = ascending entry index = ascending row, exactly the host transpose. The planner (host)
// receives the result by one bulk copy instead of building it with a 20-million-element scatter.
__global__ void k_iota_u32(uint32_t *__restrict__ p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (uint32_t)i;
}
__global__ void k_csc_fill(const uint32_t *__restrict__ perm, const int32_t *__restrict__ rowptr, const double *__restrict__ rval,
                           int64_t nnz, int64_t n_rows, int ell, int32_t *__restrict__ rowidx, double *__restrict__ cval) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nnz) return;
  const uint32_t p = perm[q];
  int64_t row;
  if (ell > 0) {
    row = p / (uint32_t)ell;
  } else {  // last row whose first entry is <= p
    int64_t lo = 0, hi = n_rows;
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if ((uint32_t)rowptr[mid] <= p) lo = mid; else hi = mid;
    }
    row = lo;
  }
  rowidx[q] = (int32_t)row;
  cval[q] = rval[p];
}
__global__ void k_colptr(const int32_t *__restrict__ keys_sorted, int64_t nnz, int64_t D, int64_t *__restrict__ colptr) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j > D) return;
  int64_t lo = 0, hi = nnz;  // first position with key >= j
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys_sorted[mid] < (int32_t)j) lo = mid + 1; else hi = mid;
  }
  colptr[j] = lo;
}

// fills X.colptr / rowidx / cval on the device from X's CSR and returns the host copy the planner reads
static inline HostCsr transpose_device(DevSparse &X, hipStream_t s) {
  HostCsr T;
  T.rows = X.cols;
  T.cols = X.rows;
  const int64_t nnz = X.nnz, D = X.cols;
  X.colptr.alloc((size_t)D + 1);
  X.rowidx.alloc((size_t)std::max<int64_t>(nnz, 1));
  X.cval.alloc((size_t)std::max<int64_t>(nnz, 1));
  T.ptr.assign((size_t)D + 1, 0);
  T.idx.resize((size_t)nnz);
  T.val.resize((size_t)nnz);
  if (nnz == 0) {
    MFM_HIP_CHECK(hipMemsetAsync(X.colptr.p, 0, ((size_t)D + 1) * sizeof(int64_t), s));
    MFM_HIP_CHECK(hipStreamSynchronize(s));
    return T;
  }
  DevBuf<int32_t> keys_out;
  DevBuf<uint32_t> iota, perm;
  keys_out.alloc((size_t)nnz);
  iota.alloc((size_t)nnz);
  perm.alloc((size_t)nnz);
  hipLaunchKernelGGL(k_iota_u32, dim3(cdiv_i(nnz, 256)), dim3(256), 0, s, iota.p, nnz);
  int end_bit = 1;
  while (((int64_t)1 << end_bit) < std::max<int64_t>(D, 2)) end_bit++;
  size_t tmp_bytes = 0;
  MFM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, X.colidx.p, keys_out.p, iota.p, perm.p, (int)nnz, 0, end_bit, s));
  DevBuf<char> tmp;
  tmp.alloc(tmp_bytes);
  MFM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, X.colidx.p, keys_out.p, iota.p, perm.p, (int)nnz, 0, end_bit, s));
  hipLaunchKernelGGL(k_csc_fill, dim3(cdiv_i(nnz, 256)), dim3(256), 0, s, perm.p, X.rowptr.p, X.rval.p, nnz, X.rows,
                     (int)(X.ell_width > 0 ? X.ell_width : 0), X.rowidx.p, X.cval.p);
  hipLaunchKernelGGL(k_colptr, dim3(cdiv_i(D + 1, 256)), dim3(256), 0, s, keys_out.p, nnz, D, X.colptr.p);
  MFM_HIP_CHECK(hipGetLastError());
  MFM_HIP_CHECK(hipMemcpyAsync(T.ptr.data(), X.colptr.p, ((size_t)D + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, s));
  MFM_HIP_CHECK(hipMemcpyAsync(T.idx.data(), X.rowidx.p, (size_t)nnz * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  MFM_HIP_CHECK(hipMemcpyAsync(T.val.data(), X.cval.p, (size_t)nnz * sizeof(double), hipMemcpyDeviceToHost, s));
  MFM_HIP_CHECK(hipStreamSynchronize(s));
  return T;
}

// rows of the training table per block row (cardinality, definitions.hpp:65-68) on the device: a workgroup counts a contiguous
// stretch of the map -- in an LDS table when the block has few rows, else with one global atomic per run of equal indices inside
// a wavefront (a map that follows the table's order hits the same block row for many consecutive rows)
constexpr int64_t CARD_LDS_MAX = 8192;
__global__ void k_block_card(const int32_t *__restrict__ map, int64_t N, int B, bool lds, int32_t *__restrict__ cnt) {
  extern __shared__ int32_t card_lds[];
  const int64_t per = (N + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = min(N, lo + per);
  if (lds) {
    for (int i = threadIdx.x; i < B; i += blockDim.x) card_lds[i] = 0;
    __syncthreads();
    for (int64_t t = lo + threadIdx.x; t < hi; t += blockDim.x) atomicAdd(&card_lds[map[t]], 1);
    __syncthreads();
    for (int i = threadIdx.x; i < B; i += blockDim.x)
      if (card_lds[i]) atomicAdd(&cnt[i], card_lds[i]);
    return;
  }
  const int lane = threadIdx.x & 63;
  for (int64_t t0 = lo + (threadIdx.x & ~63); t0 < hi; t0 += blockDim.x) {
    const int64_t t = t0 + lane;
    const int32_t v = t < hi ? map[t] : -1;
    const int32_t prev = __shfl_up(v, 1);
    const bool head = lane == 0 || v != prev;
    const unsigned long long heads = __ballot(head);
    if (head && v >= 0) {
      const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
      const int len = above ? __ffsll((long long)above) : 64 - lane;
      // (lanes past the end of the stretch carry v = -1: they start a run of their own and are not counted)
      atomicAdd(&cnt[v], len);
    }
  }
}
// descents of the map (0: it is sorted, the block follows the table's row order)
__global__ void k_map_descents(const int32_t *__restrict__ map, int64_t N, int32_t *__restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool d = t > 0 && t < N && map[t] < map[t - 1];
  if (__ballot(d) && (threadIdx.x & 63) == 0) atomicAdd(out, 1);
}
__global__ void k_block_card_rec(const int32_t *__restrict__ cnt, int64_t B, double *__restrict__ rec) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) rec[i * BLOCK_REC + 6] = (double)cnt[i];
}

struct DevBlock {
  int64_t B = 0, Db = 0, nnz = 0;
  int64_t col_off = 0;  // offset of the block's features in the global feature index
  DevSparse X;
  DevBuf<int32_t> map;  // original_to_block (B < 2^31)
  DevBuf<double> rec;   // [B][8]
  DevBuf<double> bq, bl, bs;
  BlockOverflow ov;     // (block MAX_BLOCKS keeps the scorer's pointer arrays of the blocks from there on)
  StepPlan plan_V, plan_W;
  // inverse map + bins of block rows by cardinality
  DevBuf<int64_t> inv_ptr;
  DevBuf<int32_t> inv_rows;
  DevBuf<int32_t> inv_wave, inv_wg, inv_long, inv_long_chunk_ptr;
  DevBuf<InvChunk> inv_chunks;
  DevBuf<double> inv_partial;
  DevBuf<double> comm_buf;  // [B][4] packed statistics for the all-reduce (sharded mode)
  DevBuf<double> qc;        // q_B alone, compact (filled by k_block_rowcache)
  DevBuf<double2> q_saved;  // (q_B, q_S) of the previous factor while its re-sync is folded into the next q-cache build
  // streaming statistics pass (k_unsync_stream): few block rows with very many training rows each
  bool stream_unsync = false;
  bool split_unsync = false;  // streaming un-sync (k_unsync_update) + read-only statistics through the inverse map
  int stream_wgs = 0;
  int64_t stream_rows_per_wg = 0;
  DevBuf<double> stream_partial;  // [stream_wgs][B][4]
  int n_inv_wave = 0, n_inv_wg = 0, n_inv_long = 0, n_inv_chunks = 0;

  // lean: the cell path (mfm_cell.hpp) takes every O(N) pass of this design -- no inverse map, no streaming tables
  void build(const HostCsr &hX, const int32_t *hmap, int64_t N, int KS, hipStream_t s, bool lean = false,
             DevBuf<int32_t> *pre_map = nullptr) {
    B = hX.rows;
    Db = hX.cols;
    nnz = hX.nnz();
    const bool tlog = std::getenv("MFM_SETUP_TIMING") != nullptr;
    double t_prev = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    auto lap = [&](const char *what) {
      if (!tlog) return;
      const double t = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
      std::fprintf(stderr, "[DevBlock::build B=%lld nnz=%lld] %-24s %7.3f s\n", (long long)B, (long long)nnz, what, t - t_prev);
      t_prev = t;
    };
    // X_B^T: on the device (a stable sort of the entries by column, one bulk copy back for the planner) unless the block is tiny
    HostCsr Xt;
    if (nnz >= ((int64_t)1 << 16) && !std::getenv("MFM_HOST_TRANSPOSE")) {
      X.upload(hX, nullptr);
      Xt = transpose_device(X, s);
    } else {
      Xt = transpose_host(hX);
      X.upload(hX, &Xt);
    }
    lap("upload, transpose");
    // (the level schedule of the block's columns is computed on the device from its rows: column_levels_device)
    DevCscView dev_view = DevCscView{X.colptr.p, X.rowidx.p, X.cval.p, s};
    dev_view.rowptr = X.rowptr.p;
    dev_view.colidx = X.colidx.p;
    dev_view.n_rows = B;
    dev_view.n_cols = Db;
    dev_view.ell = (int)X.ell_width;
    plan_V.dev_csc = plan_W.dev_csc = (B > 0 && nnz > 0 && !std::getenv("MFM_HOST_LEVELS")) ? &dev_view : nullptr;
    plan_V.block_plan = plan_W.block_plan = true;
    plan_V.build(Xt, PBlockV::R_W16, PBlockV::R_WG, coop_capacity<PBlockV>());
    lap("plan_V");
    plan_W.build(Xt, PBlockW::R_W16, PBlockW::R_WG, coop_capacity<PBlockW>(), false, false,
                 std::getenv("MFM_NO_PLAN_TWIN") ? nullptr : &plan_V);  // (the level schedule of the same matrix: computed once)
    lap("plan_W");
    plan_V.dev_csc = plan_W.dev_csc = nullptr;  // (the view lives on this frame)
    // rows of a block row far apart in the table (lists longer than a workgroup handles at once) and a table that fits
    // in LDS: the statistics pass streams the training rows (k_unsync_stream) and needs no inverse map. (A map that is
    // sorted -- the block follows the table's row order -- has contiguous lists; those stay with the inverse-map kernels,
    // which then stream as well.)
    // the map goes to the device first: everything that is O(N) about it -- is it sorted, the rows per block row (cardinality,
    // definitions.hpp:65-68), the training rows of every block row (inverse map: a stable sort of the rows by block row) -- is
    // done there; the host keeps what is O(B): list offsets, the length classes of the lists
    if (pre_map && pre_map->n == (size_t)N && N > 0)
      map = std::move(*pre_map);  // (already on the device: the cell planner read it there)
    else
      map.upload(hmap, (size_t)N);
    DevBuf<int32_t> cnt;
    cnt.alloc_zero((size_t)std::max<int64_t>(B, 1) + 1, s);  // (+ 1: the number of descents of the map)
    if (N > 0) {
      const bool lds = B <= CARD_LDS_MAX;
      const int wgs = (int)std::min<int64_t>(1024, (N + 4 * WG - 1) / (4 * WG));
      hipLaunchKernelGGL(k_block_card, dim3(wgs), dim3(WG), lds ? (size_t)B * sizeof(int32_t) : 0, s, map.p, N, (int)B, lds, cnt.p);
      if (!lean) hipLaunchKernelGGL(k_map_descents, dim3(cdiv_i(N, 256)), dim3(256), 0, s, map.p, N, cnt.p + std::max<int64_t>(B, 1));
    }
    rec.alloc_zero((size_t)B * BLOCK_REC, s);
    if (B > 0) hipLaunchKernelGGL(k_block_card_rec, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, cnt.p, B, rec.p);
    std::vector<int64_t> iptr((size_t)B + 1, 0);
    bool sorted = true;
    if (!lean) {
      std::vector<int32_t> hc((size_t)std::max<int64_t>(B, 1) + 1);
      MFM_HIP_CHECK(hipMemcpyAsync(hc.data(), cnt.p, hc.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
      MFM_HIP_CHECK(hipStreamSynchronize(s));
      for (int64_t i = 0; i < B; i++) iptr[i + 1] = iptr[i] + hc[i];
      sorted = hc[(size_t)std::max<int64_t>(B, 1)] == 0;
    } else {
      MFM_HIP_CHECK(hipStreamSynchronize(s));
    }
    stream_unsync = !sorted && B >= 1 && B <= UNSYNC_STREAM_MAX_B && N >= 64 * B && N >= ((int64_t)1 << 20) &&
                    !std::getenv("MFM_NO_UNSYNC_STREAM");  // (short tables: too few workgroups to stream with)
    if (const char *e = std::getenv("MFM_UNSYNC_STREAM_FORCE")) stream_unsync = std::atoi(e) != 0 && B >= 1 && B <= UNSYNC_STREAM_MAX_B;
    // too many block rows for the LDS table, lists scattered over a long table: split form
    split_unsync = !sorted && !stream_unsync && N >= ((int64_t)1 << 20) && !std::getenv("MFM_NO_UNSYNC_SPLIT");
    if (const char *e = std::getenv("MFM_UNSYNC_SPLIT_FORCE")) split_unsync = std::atoi(e) != 0 && !stream_unsync;
    if (lean) stream_unsync = split_unsync = false;
    if (!stream_unsync && !lean && N > 0) {
      // inverse map: the training rows ordered by block row, ascending inside a block row (a stable sort of (block row, t))
      inv_rows.alloc((size_t)N);
      DevBuf<int32_t> keys_out, iota;
      keys_out.alloc((size_t)N);
      iota.alloc((size_t)N);
      hipLaunchKernelGGL(k_iota_u32, dim3(cdiv_i(N, 256)), dim3(256), 0, s, (uint32_t *)iota.p, N);
      int end_bit = 1;
      while (((int64_t)1 << end_bit) < std::max<int64_t>(B, 2)) end_bit++;
      size_t tmp_bytes = 0;
      MFM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, map.p, keys_out.p, iota.p, inv_rows.p, (int)N, 0, end_bit, s));
      DevBuf<char> tmp;
      tmp.alloc(tmp_bytes);
      MFM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, map.p, keys_out.p, iota.p, inv_rows.p, (int)N, 0, end_bit, s));
      MFM_HIP_CHECK(hipStreamSynchronize(s));
      if (std::getenv("MFM_PLAN_CHECK") && hmap) {  // tests: the host's counting sort
        std::vector<int32_t> want((size_t)N), got((size_t)N);
        std::vector<int64_t> cur(iptr.begin(), iptr.end() - 1);
        for (int64_t t = 0; t < N; t++) want[cur[hmap[t]]++] = (int32_t)t;
        MFM_HIP_CHECK(hipMemcpy(got.data(), inv_rows.p, (size_t)N * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (want != got) throw Error(MFM_ERR_RUNTIME, "plan check: device and host inverse block maps differ");
      }
    } else {
      inv_rows.alloc(0);
    }
    std::vector<int32_t> bw, bg, bl_, cptr;
    std::vector<InvChunk> ch;
    for (int64_t i = 0; i < B && !stream_unsync && !lean; i++) {
      int64_t len = iptr[i + 1] - iptr[i];
      if (len <= INV_WAVE_CAP)
        bw.push_back((int32_t)i);
      else if (len <= INV_WG_CAP)
        bg.push_back((int32_t)i);
      else {
        cptr.push_back((int32_t)ch.size());
        for (int64_t b = 0; b < len; b += INV_WG_CAP)
          ch.push_back(InvChunk{iptr[i] + b, (int32_t)std::min<int64_t>(INV_WG_CAP, len - b), (int32_t)i});
        bl_.push_back((int32_t)i);
      }
    }
    cptr.push_back((int32_t)ch.size());
    n_inv_wave = (int)bw.size();
    n_inv_wg = (int)bg.size();
    n_inv_long = (int)bl_.size();
    n_inv_chunks = (int)ch.size();
    lap("map: counts, inverse (device)");
    inv_ptr.upload(iptr);
    inv_wave.upload(bw);
    inv_wg.upload(bg);
    inv_long.upload(bl_);
    inv_long_chunk_ptr.upload(cptr);
    inv_chunks.upload(ch.data(), ch.size());
    inv_partial.alloc((size_t)std::max(n_inv_chunks, 1) * 4);
    bq.alloc_zero((size_t)B * std::max(KS, 1), s);
    bl.alloc_zero((size_t)B, s);
    bs.alloc_zero((size_t)B, s);
    comm_buf.alloc((size_t)std::max<int64_t>(B, 1) * 4);
    if (!std::getenv("MFM_NO_COMPACT_QB")) qc.alloc_zero((size_t)std::max<int64_t>(B, 1), s);
    if (stream_unsync) {
      const int64_t step = (int64_t)WG * UNSYNC_R;
      const int64_t steps = (N + step - 1) / step;
      stream_wgs = (int)std::min<int64_t>(UNSYNC_STREAM_WGS, std::max<int64_t>(steps, 1));
      stream_rows_per_wg = ((steps + stream_wgs - 1) / stream_wgs) * step;
      stream_wgs = (int)((N + stream_rows_per_wg - 1) / stream_rows_per_wg);
      stream_partial.alloc((size_t)std::max(stream_wgs, 1) * (size_t)B * 4);
    }
    lap("uploads, buffers");
  }
  // sum record words [first, first + n) of every block row over the ranks
  void allreduce_fields(hipStream_t s, const Comm &comm, int first, int n) {
    if (!comm.active() || B == 0) return;
    const int64_t cnt = B * n;
    hipLaunchKernelGGL(k_rec_pack, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, rec.p, comm_buf.p, B, first, n);
    comm.allreduce(comm_buf.p, cnt);
    hipLaunchKernelGGL(k_rec_unpack, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, rec.p, comm_buf.p, B, first, n);
    MFM_HIP_CHECK(hipGetLastError());
  }
};


// q_B (and q_S) for one coefficient vector theta (already offset to the block's first feature)
static void block_rowcache(hipStream_t s, Timing &tm, DevBlock &B, const double *theta, bool with_qs) {
  if (B.B == 0) return;
  TimedLaunch t(tm, s, KC_BLOCK_ROWCACHE, 12.0 * B.nnz + 48.0 * B.B);
  const bool wave = B.X.avg_row_nnz > 16.0;
  dim3 grid(wave ? cdiv_i(B.B, WG / WAVE) : cdiv_i(B.B, WG)), block(WG);
#define MFM_RC(QS, WV) \
  hipLaunchKernelGGL((k_block_rowcache<QS, WV>), grid, block, 0, s, B.X.rowptr.p, B.X.colidx.p, B.X.rval.p, theta, B.rec.p, B.B, B.qc.p)
  if (with_qs) {
    if (wave) MFM_RC(true, true); else MFM_RC(true, false);
  } else {
    if (wave) MFM_RC(false, true); else MFM_RC(false, false);
  }
#undef MFM_RC
  MFM_HIP_CHECK(hipGetLastError());
}

template <bool IS_W>
static void block_unsync(hipStream_t s, Timing &tm, DevBlock &B, int64_t N, double2 *eq, DevBlock *pending_resync = nullptr) {
  TimedLaunch t(tm, s, KC_BLOCK_UNSYNC, (IS_W ? 4.0 + 8.0 + 8.0 : 4.0 + 16.0 + 16.0) * N + 32.0 * B.B + (pending_resync ? 4.0 * N : 0.0));
  if (B.stream_unsync && N > 0) {
    constexpr int NS = IS_W ? 1 : 4;
    const size_t lds = (size_t)B.B * NS * sizeof(double);
    static DeviceOnce raised;
    if (raised.need()) {
      MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_unsync_stream<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(UNSYNC_STREAM_MAX_B * 4 * sizeof(double))));
      MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_unsync_stream<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(UNSYNC_STREAM_MAX_B * sizeof(double))));
      raised.mark();
    }
    if (pending_resync && !IS_W) {
      static DeviceOnce raised2;
      if (raised2.need()) {
        MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_unsync_stream<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(UNSYNC_STREAM_MAX_B * 4 * sizeof(double))));
        raised2.mark();
      }
      hipLaunchKernelGGL((k_unsync_stream<false, true>), dim3(B.stream_wgs), dim3(WG), lds, s, B.map.p, eq, B.rec.p, N, (int)B.B,
                         B.stream_rows_per_wg, B.stream_partial.p, pending_resync->map.p, pending_resync->rec.p);
    } else {
      hipLaunchKernelGGL((k_unsync_stream<IS_W>), dim3(B.stream_wgs), dim3(WG), lds, s, B.map.p, eq, B.rec.p, N, (int)B.B,
                         B.stream_rows_per_wg, B.stream_partial.p, nullptr, nullptr);
    }
    hipLaunchKernelGGL((k_unsync_stream_fin<IS_W>), dim3(cdiv_i(B.B * NS, 32)), dim3(256), 0, s, B.stream_partial.p,
                       B.stream_wgs, (int)B.B, B.rec.p);
    MFM_HIP_CHECK(hipGetLastError());
    return;
  }
  if (B.split_unsync && N > 0) {
    if (pending_resync && !IS_W)
      hipLaunchKernelGGL((k_unsync_update<false, true>), dim3(cdiv_i(N, WG)), dim3(WG), 0, s, B.map.p, eq, B.rec.p, N,
                         pending_resync->map.p, pending_resync->rec.p);
    else
      hipLaunchKernelGGL((k_unsync_update<IS_W, false>), dim3(cdiv_i(N, WG)), dim3(WG), 0, s, B.map.p, eq, B.rec.p, N, nullptr, nullptr);
    if (B.n_inv_wave)
      hipLaunchKernelGGL((k_unsync_wave<IS_W, true>), dim3(cdiv_i(B.n_inv_wave, WG / WAVE)), dim3(WG), 0, s, B.inv_ptr.p,
                         B.inv_rows.p, B.inv_wave.p, B.n_inv_wave, eq, B.rec.p);
    if (B.n_inv_wg)
      hipLaunchKernelGGL((k_unsync_wg<IS_W, true>), dim3(B.n_inv_wg), dim3(WG), 0, s, B.inv_ptr.p, B.inv_rows.p, B.inv_wg.p, eq,
                         B.rec.p);
    if (B.n_inv_long) {
      hipLaunchKernelGGL((k_unsync_long<IS_W, true>), dim3(B.n_inv_chunks), dim3(WG), 0, s, B.inv_chunks.p, B.inv_rows.p, eq,
                         B.rec.p, B.inv_partial.p);
      hipLaunchKernelGGL((k_unsync_long_fin<IS_W>), dim3(cdiv_i(B.n_inv_long, 64)), dim3(64), 0, s, B.inv_long.p,
                         B.inv_long_chunk_ptr.p, B.n_inv_long, B.inv_partial.p, B.rec.p);
    }
    MFM_HIP_CHECK(hipGetLastError());
    return;
  }
  if (B.n_inv_wave)
    hipLaunchKernelGGL((k_unsync_wave<IS_W>), dim3(cdiv_i(B.n_inv_wave, WG / WAVE)), dim3(WG), 0, s, B.inv_ptr.p,
                       B.inv_rows.p, B.inv_wave.p, B.n_inv_wave, eq, B.rec.p);
  if (B.n_inv_wg)
    hipLaunchKernelGGL((k_unsync_wg<IS_W>), dim3(B.n_inv_wg), dim3(WG), 0, s, B.inv_ptr.p, B.inv_rows.p, B.inv_wg.p, eq,
                       B.rec.p);
  if (B.n_inv_long) {
    hipLaunchKernelGGL((k_unsync_long<IS_W>), dim3(B.n_inv_chunks), dim3(WG), 0, s, B.inv_chunks.p, B.inv_rows.p, eq,
                       B.rec.p, B.inv_partial.p);
    hipLaunchKernelGGL((k_unsync_long_fin<IS_W>), dim3(cdiv_i(B.n_inv_long, 64)), dim3(64), 0, s, B.inv_long.p,
                       B.inv_long_chunk_ptr.p, B.n_inv_long, B.inv_partial.p, B.rec.p);
  }
  MFM_HIP_CHECK(hipGetLastError());
}

template <bool IS_W>
static void block_resync(hipStream_t s, Timing &tm, DevBlock &B, int64_t N, double2 *eq) {
  if (N == 0) return;
  TimedLaunch t(tm, s, KC_BLOCK_RESYNC, (IS_W ? 4.0 + 8.0 + 8.0 : 4.0 + 16.0 + 16.0) * N);
  hipLaunchKernelGGL((k_resync<IS_W>), dim3(cdiv_i(N, WG)), dim3(WG), 0, s, B.map.p, B.rec.p, eq, N);
  MFM_HIP_CHECK(hipGetLastError());
}

static SweepArgs block_args(DevBlock &B, double *theta_all, const double *z_all, const int32_t *group_all,
                            const double *lam, const double *mu, double alpha) {
  SweepArgs a;
  a.colptr = B.X.colptr.p;
  a.rowidx = B.X.rowidx.p;
  a.val = B.X.cval.p;
  a.state = B.rec.p;
  a.theta = theta_all + B.col_off;
  a.z = z_all + B.col_off;
  a.group = group_all + B.col_off;
  a.lambda = lam;
  a.mu = mu;
  a.alpha = alpha;
  return a;
}

// FMTrainer.hpp:256-313 for one block
static void block_sweep_w(hipStream_t s, Timing &tm, LongScratch &ls, DevBlock &B, int64_t N, double2 *eq, double *w,
                          const double *z, const int32_t *group, const double *lam, const double *mu, double alpha,
                          const Comm &comm) {
  block_rowcache(s, tm, B, w + B.col_off, false);  // :265-266
  block_unsync<true>(s, tm, B, N, eq);             // :268-275
  B.allreduce_fields(s, comm, 4, 1);               // sharded rows: e_B is a sum over all ranks' rows
  SweepArgs a = block_args(B, w, z, group, lam, mu, alpha);
  const SweepClasses kc{KC_BLOCK_SWEEP, KC_BLOCK_SWEEP, KC_BLOCK_SWEEP, KC_BLOCK_SWEEP, KC_BLOCK_SWEEP, KC_BLOCK_SWEEP,
                        KC_BLOCK_SWEEP, KC_BLOCK_SWEEP};
  run_plan<PBlockW>(s, tm, B.plan_W, a, ls, kc, false);  // :276-302
  block_rowcache(s, tm, B, w + B.col_off, false);  // :304-305
  block_resync<true>(s, tm, B, N, eq);             // :306-311
}

// FMTrainer.hpp:378-482 for one block and one factor (q_B / q_S were filled by block_rowcache before
// the q-cache build, :331-333 / :388-393: V_B does not change in between)
// pending: a block whose re-sync is still owed (it was deferred so that THIS block's streaming statistics pass applies it on
// the fly); defer_resync: leave this block's own re-sync to the next block's statistics pass
static void block_sweep_V(hipStream_t s, Timing &tm, LongScratch &ls, DevBlock &B, int64_t N, double2 *eq, double *Vf,
                          const double *zf, const int32_t *group, const double *lamf, const double *muf, double alpha,
                          const Comm &comm, DevBlock *pending = nullptr, bool defer_resync = false) {
  if (pending && !((B.stream_unsync || B.split_unsync) && N > 0)) {  // (cannot be absorbed: apply it now)
    block_resync<false>(s, tm, *pending, N, eq);
    pending = nullptr;
  }
  block_unsync<false>(s, tm, B, N, eq, pending);  // :401-417
  B.allreduce_fields(s, comm, 2, 4);     // sharded rows: c, c_S, e, e_q are sums over all ranks' rows
  SweepArgs a = block_args(B, Vf, zf, group, lamf, muf, alpha);
  const SweepClasses kc{KC_BLOCK_SWEEP, KC_BLOCK_SWEEP, KC_BLOCK_SWEEP, KC_BLOCK_SWEEP, KC_BLOCK_SWEEP, KC_BLOCK_SWEEP,
                        KC_BLOCK_SWEEP, KC_BLOCK_SWEEP};
  run_plan<PBlockV>(s, tm, B.plan_V, a, ls, kc, false);  // :419-470
  if (!defer_resync) block_resync<false>(s, tm, B, N, eq);  // :473-480
}

template <int GS, int SPL>
static void launch_block_score_cache_t(hipStream_t s, DevBlock &B, const double *Vt, const double *w, int K, int KS) {
  hipLaunchKernelGGL((k_block_score_cache<GS, SPL>), dim3(cdiv_i(B.B * GS, WG)), dim3(WG), 0, s, B.X.rowptr.p,
                     B.X.colidx.p, B.X.rval.p, Vt + B.col_off * KS, w + B.col_off, K, KS, B.bq.p, B.bl.p, B.bs.p, B.B);
}
static void launch_block_score_cache(hipStream_t s, DevBlock &B, const double *Vt, const double *w, int K, int KS) {
  if (B.B == 0) return;
  if (K <= 4)
    launch_block_score_cache_t<4, 1>(s, B, Vt, w, K, KS);
  else if (K <= 8)
    launch_block_score_cache_t<8, 1>(s, B, Vt, w, K, KS);
  else if (K <= 16)
    launch_block_score_cache_t<16, 1>(s, B, Vt, w, K, KS);
  else if (K <= 32)
    launch_block_score_cache_t<32, 1>(s, B, Vt, w, K, KS);
  else if (K <= 64)
    launch_block_score_cache_t<64, 1>(s, B, Vt, w, K, KS);
  else if (K <= 128)
    launch_block_score_cache_t<64, 2>(s, B, Vt, w, K, KS);
  else if (K <= 256)
    launch_block_score_cache_t<64, 4>(s, B, Vt, w, K, KS);
  else
    launch_block_score_cache_t<64, 8>(s, B, Vt, w, K, KS);
  MFM_HIP_CHECK(hipGetLastError());
}

}  // namespace mfm
