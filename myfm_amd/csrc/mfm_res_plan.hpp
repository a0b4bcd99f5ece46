// mfm_res_plan.hpp -- the persistent sweep's slot layout (ResPlan, mfm_res.hpp) built ON THE DEVICE from the device-resident CSR.
//
// What the reference does once per fit on the host (`BaseFMTrainer.hpp:58-68`: X and its transpose; `definitions.hpp:58-68`: the
// block maps) is here the layout the kernels of the Gibbs loop want. For a two-field unit-valued table of N rows the layout is a
// handful of O(N) passes -- every row's (user, item) pair, user boundaries, a stable sort of the rows by (workgroup, item), run
// heads and their prefix sums, bit-packed slot words, the item draw's (run, item) lists -- that need neither X_t nor the host: the
// host only takes the decisions that are O(workgroups) or O(items) (where to cut the user ranges, which items a workgroup draws;
// the same code as the host builder's, ResPlan::choose_*), on a few hundred KB copied back. `ResPlan::build` (host threads) stays
// as the checker: with MFM_PLAN_CHECK=1 mfm_finalize builds both and compares every array.
#pragma once
#include <chrono>

#include <hipcub/hipcub.hpp>

#include "mfm_res.hpp"

namespace mfm {

namespace rp {

constexpr int TB = 256;

// rows: user boundary flags, boundaries per user column, rows per item column, the column ranges of both fields
// red: [0] max user column, [1] min item column (starts at INT_MAX)
__global__ void k_rows(const int32_t *__restrict__ colidx, int64_t N, int32_t *__restrict__ bflag, int32_t *__restrict__ ucnt,
                       int32_t *__restrict__ icnt, int32_t *__restrict__ red) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int32_t mu = -1, mi = 0x7fffffff;
  if (r < N) {
    const int2 c = ((const int2 *)colidx)[r];
    const bool b = r == 0 || c.x != colidx[2 * (r - 1)];
    bflag[r] = b ? 1 : 0;
    if (b) atomicAdd(&ucnt[c.x], 1);
    atomicAdd(&icnt[c.y], 1);
    mu = c.x;
    mi = c.y;
  }
  for (int off = 32; off > 0; off >>= 1) {
    mu = max(mu, __shfl_xor(mu, off));
    mi = min(mi, __shfl_xor(mi, off));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(&red[0], mu);
    atomicMin(&red[1], mi);
  }
}

// columns: item / never-occurring flags (for their prefix sums); red[2] counts columns that break the two-field form
__global__ void k_cols(const int32_t *__restrict__ ucnt, const int32_t *__restrict__ icnt, int64_t D0, int32_t *__restrict__ iflag,
                       int32_t *__restrict__ eflag, int32_t *__restrict__ red) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= D0) return;
  const int32_t u = ucnt[j], i = icnt[j];
  if (u > 1 || (u > 0 && i > 0)) atomicAdd(&red[2], 1);
  iflag[j] = i > 0;
  eflag[j] = u == 0 && i == 0;
}

// row-sharded: the levels of the GLOBAL design decide (level 1 = item whether or not this rank has rows of it), and a level-0
// column without rows here is drawn by this rank only when it has rows nowhere (draw_empty)
__global__ void k_cols_given(const int32_t *__restrict__ ucnt, const int32_t *__restrict__ icnt, const int32_t *__restrict__ level,
                             const int32_t *__restrict__ draw_empty, int64_t D0, int32_t *__restrict__ iflag, int32_t *__restrict__ eflag,
                             int32_t *__restrict__ red) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= D0) return;
  const int32_t u = ucnt[j], i = icnt[j], l = level[j];
  if (u > 1 || (u > 0 && l != 0) || (i > 0 && l != 1) || l < 0 || l > 1) atomicAdd(&red[2], 1);
  iflag[j] = l == 1;
  eflag[j] = l == 0 && u == 0 && draw_empty[j] != 0;
}

__global__ void k_col_lists(const int32_t *__restrict__ iflag, const int32_t *__restrict__ eflag, const int32_t *__restrict__ iord,
                            const int32_t *__restrict__ eord, int64_t D0, int32_t *__restrict__ scols, int32_t *__restrict__ empties) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= D0) return;
  if (iflag[j]) scols[iord[j]] = (int32_t)j;
  if (eflag[j]) empties[eord[j]] = (int32_t)j;
}

// users in row order: first row and column of the u-th user (uord1: inclusive prefix sum of the boundary flags)
__global__ void k_users(const int32_t *__restrict__ colidx, const int32_t *__restrict__ bflag, const int32_t *__restrict__ uord1,
                        int64_t N, int32_t *__restrict__ ustart, int32_t *__restrict__ users) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= N) return;
  if (bflag[r]) {
    ustart[uord1[r] - 1] = (int32_t)r;
    users[uord1[r] - 1] = colidx[2 * r];
  }
  if (r == N - 1) ustart[uord1[r]] = (int32_t)N;
}

__device__ __forceinline__ int upper_group(const int32_t *cut, int G, int32_t x) {  // the g with cut[g] <= x < cut[g + 1]
  int lo = 0, hi = G;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (cut[mid] <= x)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

// sort keys: (workgroup, item ordinal) of every row
__global__ void k_keys(const int32_t *__restrict__ colidx, const int32_t *__restrict__ iord, const int32_t *__restrict__ rowcut, int G,
                       int item_bits, int64_t N, uint32_t *__restrict__ key, int32_t *__restrict__ val) {
  __shared__ int32_t cut[1026];
  for (int i = threadIdx.x; i <= G; i += blockDim.x) cut[i] = rowcut[i];
  __syncthreads();
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= N) return;
  const int g = upper_group(cut, G, (int32_t)r);
  key[r] = ((uint32_t)g << item_bits) | (uint32_t)iord[colidx[2 * r + 1]];
  val[r] = (int32_t)r;
}

// sorted positions: slot -> row, slot -> user within the workgroup, run heads
__global__ void k_slots(const uint32_t *__restrict__ key, const int32_t *__restrict__ val, const int32_t *__restrict__ uord1,
                        const int32_t *__restrict__ rowcut, const int32_t *__restrict__ ucut, int item_bits, int R, int NT, int64_t N,
                        int32_t *__restrict__ perm, uint16_t *__restrict__ uid, int32_t *__restrict__ headf) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  const uint32_t k = key[p];
  const int g = (int)(k >> item_bits);
  const int32_t row = val[p];
  const int64_t cap = (int64_t)R * NT;
  const int32_t sidx = (int32_t)(p - rowcut[g]);
  const int t = sidx / R, rr = sidx % R;
  perm[(int64_t)g * cap + (int64_t)rr * NT + t] = row;
  uid[(int64_t)g * cap + sidx] = (uint16_t)(uord1[row] - 1 - ucut[g]);
  headf[p] = (sidx == 0 || k != key[p - 1]) ? 1 : 0;
}

// hrun0[g] = run heads before workgroup g's rows (hrun0[G] = all of them)
__global__ void k_wg_runs(const int32_t *__restrict__ hs, const int32_t *__restrict__ rowcut, int G, int32_t *__restrict__ hrun0) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g > G) return;
  hrun0[g] = rowcut[g] > 0 ? hs[rowcut[g] - 1] : 0;
}

__global__ void k_fill32(int32_t *p, int64_t n, int32_t v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void k_fill_int2(int2 *p, int64_t n, int2 v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// run heads: the run's item (compact list and the workgroup-major padded list), runs per item
__global__ void k_runs(const uint32_t *__restrict__ key, const int32_t *__restrict__ headf, const int32_t *__restrict__ hs,
                       const int32_t *__restrict__ hrun0, const int32_t *__restrict__ run_base, int item_bits, int64_t N,
                       int32_t *__restrict__ run_item_c, int32_t *__restrict__ run_item, int32_t *__restrict__ runs_per_item) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N || !headf[p]) return;
  const uint32_t k = key[p];
  const int g = (int)(k >> item_bits);
  const int32_t item = (int32_t)(k & ((1u << item_bits) - 1u));
  const int32_t kc = hs[p] - 1;
  run_item_c[kc] = item;
  run_item[run_base[g] + kc - hrun0[g]] = item;
  atomicAdd(&runs_per_item[item], 1);
}

// the run holding every thread's first slot (the pad run when the slot is a pad)
__global__ void k_first(const int32_t *__restrict__ hs, const int32_t *__restrict__ rowcut, const int32_t *__restrict__ hrun0,
                        const int32_t *__restrict__ run_base, int G, int R, int NT, int32_t *__restrict__ first_run) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G * NT) return;
  const int g = i / NT, t = i % NT;
  const int32_t fill = rowcut[g + 1] - rowcut[g], s0 = t * R;
  const int32_t nr = hrun0[g + 1] - hrun0[g];
  first_run[i] = run_base[g] + (s0 >= fill ? nr : hs[rowcut[g] + s0] - 1 - hrun0[g]);
}

// the slot words of thread t, slots 16 grp .. 16 grp + 15: users 10 bits each in 5 words, head bits in one
__global__ void k_pack(const uint16_t *__restrict__ uid, const int32_t *__restrict__ headf, const int32_t *__restrict__ rowcut, int G, int R,
                       int NT, int umax, uint32_t *__restrict__ uidw, uint32_t *__restrict__ headw) {
  const int HW = R / 16, UW = 5 * HW;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)G * HW * NT) return;
  const int t = (int)(i % NT), grp = (int)((i / NT) % HW), g = (int)(i / ((int64_t)NT * HW));
  const int64_t cap = (int64_t)R * NT;
  const int32_t fill = rowcut[g + 1] - rowcut[g];
  uint32_t w[5] = {0, 0, 0, 0, 0}, hw = 0;
  for (int q = 0; q < 16; q++) {
    const int32_t sidx = t * R + 16 * grp + q;
    const uint16_t u16 = uid[(int64_t)g * cap + sidx];
    const uint32_t u = u16 == 0xffffu ? (uint32_t)(umax - 1) : u16;
    const int bb = q / 4, k = q % 4;
    if (k < 3) {
      w[bb] |= u << (10 * k);
    } else {
      w[bb] |= (u & 3u) << 30;
      w[4] |= (u >> 2) << (8 * bb);
    }
    // heads: run starts, the first pad slot (it opens the pad run), thread 0's first slot
    const bool head = sidx < fill ? headf[rowcut[g] + sidx] != 0 : sidx == fill;
    if (head || sidx == 0) hw |= 1u << q;
  }
  for (int k = 0; k < 5; k++) uidw[((int64_t)g * UW + 5 * grp + k) * NT + t] = w[k];
  headw[((int64_t)g * HW + grp) * NT + t] = hw;
}

// {feature, group} of the workgroups' users: its own in row order, then the never-occurring columns dealt round-robin
__global__ void k_user_desc(const int32_t *__restrict__ users, const int32_t *__restrict__ empties, const int32_t *__restrict__ ucut,
                            const int32_t *__restrict__ uptr, const int32_t *__restrict__ group, int G, int32_t n_users, int32_t n_emp,
                            int2 *__restrict__ desc) {
  __shared__ int32_t cut[1026];
  for (int i = threadIdx.x; i <= G; i += blockDim.x) cut[i] = ucut[i];
  __syncthreads();
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_users) {
    const int g = upper_group(cut, G, i);
    const int32_t j = users[i];
    desc[uptr[g] + (i - cut[g])] = make_int2(j, group ? group[j] : 0);
  } else if (i < n_users + n_emp) {
    const int32_t k = i - n_users;
    const int g = k % G;
    const int32_t j = empties[k];
    desc[uptr[g] + (cut[g + 1] - cut[g]) + k / G] = make_int2(j, group ? group[j] : 0);
  }
}

__global__ void k_item_desc(const int32_t *__restrict__ scols, const int32_t *__restrict__ group, int32_t n_items, int2 *__restrict__ desc) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_items) desc[i] = make_int2(scols[i], group ? group[scols[i]] : 0);
}

__global__ void k_run_feat(const int32_t *__restrict__ run_item, const int32_t *__restrict__ scols, int32_t n_items, int64_t n_run_item,
                           int64_t n, int32_t *__restrict__ run_feat) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  int32_t f = -1;
  if (r < n_run_item) {
    const int32_t c = run_item[r];
    if (c < n_items) f = scols[c];
  }
  run_feat[r] = f;
}

// item slices of the runs (for the stable sort by slice)
__global__ void k_slice_keys(const int32_t *__restrict__ run_item_c, const int32_t *__restrict__ iptr, int G, int64_t n, uint32_t *__restrict__ key,
                             int32_t *__restrict__ val) {
  __shared__ int32_t cut[1026];
  for (int i = threadIdx.x; i <= G; i += blockDim.x) cut[i] = iptr[i];
  __syncthreads();
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  // (slices may be empty: the LAST slice that starts at or before the item)
  key[k] = (uint32_t)upper_group(cut, G, run_item_c[k]);
  val[k] = (int32_t)k;
}

// the item draw's lists: slice by slice its (run, item within the slice) pairs, source workgroup by source workgroup
__global__ void k_entries(const uint32_t *__restrict__ key, const int32_t *__restrict__ val, const int32_t *__restrict__ run_item_c,
                          const int32_t *__restrict__ hrun0, const int32_t *__restrict__ run_base, const int32_t *__restrict__ iptr,
                          const int32_t *__restrict__ eptr, const int32_t *__restrict__ sstart, int G, int64_t n, int2 *__restrict__ entries) {
  __shared__ int32_t cut[1026];
  for (int i = threadIdx.x; i <= G; i += blockDim.x) cut[i] = hrun0[i];
  __syncthreads();
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int sl = (int)key[q];
  const int32_t kc = val[q];
  const int g = upper_group(cut, G, kc);
  entries[eptr[sl] + (q - sstart[sl])] = make_int2(run_base[g] + (kc - cut[g]), run_item_c[kc] - iptr[sl]);
}

template <class T>
static inline std::vector<T> download(const T *p, size_t n, hipStream_t s) {
  std::vector<T> h(n);
  if (n) MFM_HIP_CHECK(hipMemcpyAsync(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost, s));
  MFM_HIP_CHECK(hipStreamSynchronize(s));
  return h;
}

static inline void inclusive_sum(const int32_t *in, int32_t *out, int64_t n, DevBuf<char> &tmp, hipStream_t s) {
  size_t bytes = 0;
  MFM_HIP_CHECK(hipcub::DeviceScan::InclusiveSum(nullptr, bytes, in, out, (int)n, s));
  if (tmp.n < bytes) tmp.alloc(bytes);
  MFM_HIP_CHECK(hipcub::DeviceScan::InclusiveSum(tmp.p, bytes, in, out, (int)n, s));
}
static inline void exclusive_sum(const int32_t *in, int32_t *out, int64_t n, DevBuf<char> &tmp, hipStream_t s) {
  size_t bytes = 0;
  MFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in, out, (int)n, s));
  if (tmp.n < bytes) tmp.alloc(bytes);
  MFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp.p, bytes, in, out, (int)n, s));
}
static inline void sort_pairs(DevBuf<uint32_t> &k_in, DevBuf<uint32_t> &k_out, DevBuf<int32_t> &v_in, DevBuf<int32_t> &v_out, int64_t n, int bits,
                              DevBuf<char> &tmp, hipStream_t s) {
  size_t bytes = 0;
  MFM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, k_in.p, k_out.p, v_in.p, v_out.p, (int)n, 0, bits, s));
  if (tmp.n < bytes) tmp.alloc(bytes);
  MFM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp.p, bytes, k_in.p, k_out.p, v_in.p, v_out.p, (int)n, 0, bits, s));
}

}  // namespace rp

// X: the main table on the device (CSR; every row two unit entries in ascending column order), group_of: host, may be null
static inline bool res_plan_build_device(ResPlan &rpn, const DevSparse &X, const std::vector<int32_t> *group_of, int n_cu, hipStream_t s,
                                         const std::vector<int32_t> *glevel = nullptr, const std::vector<char> *draw_empty = nullptr) {
  using namespace rp;
  rpn.ready = false;
  const int64_t N = X.rows, D0 = X.cols;
  const int NT = rpn.NT;
  rpn.n_rows = N;
  if (N < 1 || n_cu < 1 || D0 < 2 || !X.unit || X.ell_width != 2) return rpn.fail("shape");
  if (N >= ((int64_t)1 << 31) - 1) return rpn.fail("too many rows");
  auto grid = [](int64_t n) { return dim3((unsigned)((n + TB - 1) / TB)); };
  const bool tlog = std::getenv("MFM_SETUP_TIMING") != nullptr;
  double t_prev = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  auto lap = [&](const char *what) {
    if (!tlog) return;
    MFM_HIP_CHECK(hipStreamSynchronize(s));
    const double t = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    std::fprintf(stderr, "[res_plan_build_device] %-34s %7.2f ms\n", what, (t - t_prev) * 1e3);
    t_prev = t;
  };
  DevBuf<char> tmp;
  DevBuf<int32_t> bflag, ucnt, icnt, red, iflag, eflag, iord, eord, uord1;
  bflag.alloc((size_t)N);
  uord1.alloc((size_t)N);
  ucnt.alloc_zero((size_t)D0, s);
  icnt.alloc_zero((size_t)D0, s);
  iflag.alloc((size_t)D0 + 1);
  eflag.alloc((size_t)D0 + 1);
  iord.alloc((size_t)D0 + 1);
  eord.alloc((size_t)D0 + 1);
  {
    const int32_t r0[4] = {-1, 0x7fffffff, 0, 0};
    red.upload(r0, 4);
  }
  lap("allocations");
  if (tlog) {
    hipLaunchKernelGGL(k_fill32, dim3(1), dim3(TB), 0, s, red.p + 3, (int64_t)1, 0);
    lap("first kernel of the library (code object load)");
  }
  hipLaunchKernelGGL(k_rows, grid(N), dim3(TB), 0, s, X.colidx.p, N, bflag.p, ucnt.p, icnt.p, red.p);
  lap("k_rows");
  const bool given = glevel && draw_empty && (int64_t)glevel->size() == D0 && (int64_t)draw_empty->size() == D0;
  if ((glevel || draw_empty) && !given) return rpn.fail("shape");
  if (given) {
    DevBuf<int32_t> d_level, d_de;
    d_level.upload(*glevel);
    std::vector<int32_t> de32(draw_empty->begin(), draw_empty->end());
    d_de.upload(de32);
    hipLaunchKernelGGL(k_cols_given, grid(D0), dim3(TB), 0, s, ucnt.p, icnt.p, d_level.p, d_de.p, D0, iflag.p, eflag.p, red.p);
    MFM_HIP_CHECK(hipStreamSynchronize(s));
  } else {
    hipLaunchKernelGGL(k_cols, grid(D0), dim3(TB), 0, s, ucnt.p, icnt.p, D0, iflag.p, eflag.p, red.p);
  }
  MFM_HIP_CHECK(hipMemsetAsync(iflag.p + D0, 0, sizeof(int32_t), s));  // (one more element: the totals come out of the exclusive sums)
  MFM_HIP_CHECK(hipMemsetAsync(eflag.p + D0, 0, sizeof(int32_t), s));
  exclusive_sum(iflag.p, iord.p, D0 + 1, tmp, s);
  exclusive_sum(eflag.p, eord.p, D0 + 1, tmp, s);
  inclusive_sum(bflag.p, uord1.p, N, tmp, s);
  lap("rows, columns, scans (enqueued + run)");
  int32_t h_red[4], n_items = 0, n_emp = 0, n_users = 0;
  MFM_HIP_CHECK(hipMemcpyAsync(h_red, red.p, sizeof h_red, hipMemcpyDeviceToHost, s));
  MFM_HIP_CHECK(hipMemcpyAsync(&n_items, iord.p + D0, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  MFM_HIP_CHECK(hipMemcpyAsync(&n_emp, eord.p + D0, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  MFM_HIP_CHECK(hipMemcpyAsync(&n_users, uord1.p + (N - 1), sizeof(int32_t), hipMemcpyDeviceToHost, s));
  MFM_HIP_CHECK(hipStreamSynchronize(s));
  if (!given && h_red[0] >= h_red[1]) return rpn.fail("the two fields' column ranges overlap");
  if (h_red[2] != 0) return rpn.fail("first level not contiguous");
  if (n_users < 1 || n_items < 1) return rpn.fail("empty level");
  rpn.n_items = n_items;
  DevBuf<int32_t> empties, d_ustart, d_users;
  rpn.scols.alloc((size_t)n_items);
  empties.alloc((size_t)std::max(n_emp, 1));
  d_ustart.alloc((size_t)n_users + 1);
  d_users.alloc((size_t)n_users);
  hipLaunchKernelGGL(k_col_lists, grid(D0), dim3(TB), 0, s, iflag.p, eflag.p, iord.p, eord.p, D0, rpn.scols.p, empties.p);
  hipLaunchKernelGGL(k_users, grid(N), dim3(TB), 0, s, X.colidx.p, bflag.p, uord1.p, N, d_ustart.p, d_users.p);
  // the decisions: workgroup cuts at user boundaries (host: O(workgroups log users) on the boundary list)
  std::vector<int64_t> ustart((size_t)n_users + 1), ucut;
  int64_t max_user = 0;
  {
    const std::vector<int32_t> u32 = download(d_ustart.p, (size_t)n_users + 1, s);
    for (size_t u = 0; u < u32.size(); u++) ustart[u] = u32[u];
    for (int32_t u = 0; u < n_users; u++) max_user = std::max(max_user, ustart[u + 1] - ustart[u]);
  }
  if (!rpn.choose_layout(N, ustart, max_user, n_cu, ucut)) return false;
  const int G = rpn.G, R = rpn.R();
  const int64_t cap_slots = (int64_t)NT * R;
  if (G > 1024) return rpn.fail("more than 1024 workgroups");
  std::vector<int32_t> h_rowcut((size_t)G + 1), h_ucut((size_t)G + 1), h_uptr((size_t)G + 1, 0);
  int maxu = 0;
  for (int g = 0; g <= G; g++) {
    h_rowcut[g] = (int32_t)ustart[ucut[g]];
    h_ucut[g] = (int32_t)ucut[g];
  }
  for (int g = 0; g < G; g++) {
    const int nu = (int)(ucut[g + 1] - ucut[g]) + n_emp / G + (g < n_emp % G ? 1 : 0);
    maxu = std::max(maxu, nu);
    h_uptr[g + 1] = h_uptr[g] + nu;
  }
  if (maxu > NT) return rpn.fail("more first-level columns in a workgroup than threads");
  rpn.item_bits = ResPlan::bits_for((int64_t)n_items + 1);
  const int gbits = ResPlan::bits_for(G);
  if (rpn.item_bits + gbits > 32) return rpn.fail("sort key width");
  DevBuf<int32_t> rowcut, d_ucut, hrun0, hs, headf;
  rowcut.upload(h_rowcut);
  d_ucut.upload(h_ucut);
  rpn.wg_user_ptr.upload(h_uptr);
  // rows in (workgroup, item, row) order: a stable sort
  DevBuf<uint32_t> key, key2;
  DevBuf<int32_t> val, val2;
  key.alloc((size_t)N);
  key2.alloc((size_t)N);
  val.alloc((size_t)N);
  val2.alloc((size_t)N);
  hipLaunchKernelGGL(k_keys, grid(N), dim3(TB), 0, s, X.colidx.p, iord.p, rowcut.p, G, rpn.item_bits, N, key.p, val.p);
  sort_pairs(key, key2, val, val2, N, rpn.item_bits + gbits, tmp, s);
  lap("users, cuts, keys, sort");
  DevBuf<uint16_t> uid;
  uid.alloc((size_t)G * cap_slots);
  MFM_HIP_CHECK(hipMemsetAsync(uid.p, 0xff, (size_t)G * cap_slots * sizeof(uint16_t), s));
  rpn.perm.alloc((size_t)G * cap_slots);
  MFM_HIP_CHECK(hipMemsetAsync(rpn.perm.p, 0xff, (size_t)G * cap_slots * sizeof(int32_t), s));
  headf.alloc((size_t)N);
  hs.alloc((size_t)N);
  hipLaunchKernelGGL(k_slots, grid(N), dim3(TB), 0, s, key2.p, val2.p, uord1.p, rowcut.p, d_ucut.p, rpn.item_bits, R, NT, N, rpn.perm.p, uid.p,
                     headf.p);
  inclusive_sum(headf.p, hs.p, N, tmp, s);
  hrun0.alloc((size_t)G + 1);
  hipLaunchKernelGGL(k_wg_runs, dim3((G + 1 + TB - 1) / TB), dim3(TB), 0, s, hs.p, rowcut.p, G, hrun0.p);
  const std::vector<int32_t> h_hrun0 = download(hrun0.p, (size_t)G + 1, s);
  std::vector<int32_t> nruns((size_t)G);
  for (int g = 0; g < G; g++) nruns[g] = h_hrun0[g + 1] - h_hrun0[g];
  const int64_t counter = h_hrun0[G];
  rpn.n_runs = counter;
  if (counter >= ((int64_t)1 << 31) - 2) return rpn.fail("too many runs");
  const std::vector<int32_t> run_base = ResPlan::run_bases(nruns);
  const int32_t zero_run = run_base[G];
  rpn.wg_run_ptr.upload(run_base);
  rpn.wg_nruns.upload(nruns);
  rpn.run_item.alloc((size_t)zero_run + 1);
  hipLaunchKernelGGL(k_fill32, grid(zero_run + 1), dim3(TB), 0, s, rpn.run_item.p, (int64_t)zero_run + 1, n_items);
  DevBuf<int32_t> run_item_c, runs_per_item;
  run_item_c.alloc((size_t)std::max<int64_t>(counter, 1));
  runs_per_item.alloc_zero((size_t)n_items, s);
  hipLaunchKernelGGL(k_runs, grid(N), dim3(TB), 0, s, key2.p, headf.p, hs.p, hrun0.p, rpn.wg_run_ptr.p, rpn.item_bits, N, run_item_c.p,
                     rpn.run_item.p, runs_per_item.p);
  rpn.first_run.alloc((size_t)G * NT);
  hipLaunchKernelGGL(k_first, grid((int64_t)G * NT), dim3(TB), 0, s, hs.p, rowcut.p, hrun0.p, rpn.wg_run_ptr.p, G, R, NT, rpn.first_run.p);
  lap("slots, heads, runs, first");
  // the item draw's slices (host: one pass over the items' run counts)
  std::vector<int32_t> h_slot_ptr((size_t)n_items + 1, 0), h_iptr;
  {
    const std::vector<int32_t> rpi = download(runs_per_item.p, (size_t)n_items, s);
    for (int c = 0; c < n_items; c++) h_slot_ptr[(size_t)c + 1] = h_slot_ptr[c] + rpi[c];
  }
  if (!rpn.choose_item_slices(h_slot_ptr, counter, h_iptr)) return false;
  int imax = 0;
  for (int g = 0; g < G; g++) imax = std::max(imax, h_iptr[g + 1] - h_iptr[g]);
  rpn.umax = std::max(maxu, imax) + 1;
  if (rpn.umax > 1024) return rpn.fail("internal: user field");
  const int UW = 5 * (R / 16), HW = R / 16;
  rpn.uidw.alloc((size_t)G * UW * NT);
  rpn.headw.alloc((size_t)G * HW * NT);
  hipLaunchKernelGGL(k_pack, grid((int64_t)G * HW * NT), dim3(TB), 0, s, uid.p, headf.p, rowcut.p, G, R, NT, rpn.umax, rpn.uidw.p, rpn.headw.p);
  std::vector<int32_t> h_eptr((size_t)G + 1, 0), h_sstart((size_t)G + 1, 0);
  for (int g = 0; g < G; g++) {
    const int64_t cnt = h_slot_ptr[h_iptr[g + 1]] - h_slot_ptr[h_iptr[g]];
    const int64_t e = h_eptr[g] + ((cnt + WAVE - 1) / WAVE) * WAVE;
    if (e >= ((int64_t)1 << 31)) return rpn.fail("too many entries");
    h_eptr[g + 1] = (int32_t)e;
    h_sstart[g + 1] = h_sstart[g] + (int32_t)cnt;
  }
  rpn.wg_item_ptr.upload(h_iptr);
  rpn.ent_ptr.upload(h_eptr);
  rpn.entries.alloc((size_t)h_eptr[G]);
  if (h_eptr[G]) hipLaunchKernelGGL(k_fill_int2, grid(h_eptr[G]), dim3(TB), 0, s, rpn.entries.p, (int64_t)h_eptr[G], make_int2(zero_run, 0));
  if (counter > 0) {
    DevBuf<int32_t> sstart;
    sstart.upload(h_sstart);
    DevBuf<uint32_t> sk, sk2;
    DevBuf<int32_t> sv, sv2;
    sk.alloc((size_t)counter);
    sk2.alloc((size_t)counter);
    sv.alloc((size_t)counter);
    sv2.alloc((size_t)counter);
    hipLaunchKernelGGL(k_slice_keys, grid(counter), dim3(TB), 0, s, run_item_c.p, rpn.wg_item_ptr.p, G, counter, sk.p, sv.p);
    sort_pairs(sk, sk2, sv, sv2, counter, gbits, tmp, s);
    hipLaunchKernelGGL(k_entries, grid(counter), dim3(TB), 0, s, sk2.p, sv2.p, run_item_c.p, hrun0.p, rpn.wg_run_ptr.p, rpn.wg_item_ptr.p,
                       rpn.ent_ptr.p, sstart.p, G, counter, rpn.entries.p);
    MFM_HIP_CHECK(hipStreamSynchronize(s));  // (the sort buffers go out of scope)
  }
  lap("slices, pack, entries");
  // users, items, the scorer's tables
  DevBuf<int32_t> d_group;
  if (group_of && (int64_t)group_of->size() >= D0) d_group.upload(group_of->data(), (size_t)D0);
  rpn.user_desc.alloc((size_t)h_uptr[G]);
  hipLaunchKernelGGL(k_user_desc, grid((int64_t)n_users + n_emp), dim3(TB), 0, s, d_users.p, empties.p, d_ucut.p, rpn.wg_user_ptr.p, d_group.p, G,
                     n_users, n_emp, rpn.user_desc.p);
  rpn.item_desc.alloc((size_t)n_items);
  hipLaunchKernelGGL(k_item_desc, grid(n_items), dim3(TB), 0, s, rpn.scols.p, d_group.p, n_items, rpn.item_desc.p);
  rpn.run_feat.alloc((size_t)zero_run + 1 + 8);
  hipLaunchKernelGGL(k_run_feat, grid((int64_t)zero_run + 9), dim3(TB), 0, s, rpn.run_item.p, rpn.scols.p, n_items, (int64_t)zero_run + 1,
                     (int64_t)zero_run + 9, rpn.run_feat.p);
  {
    std::vector<int32_t> h_fill((size_t)G);
    for (int g = 0; g < G; g++) h_fill[g] = h_rowcut[g + 1] - h_rowcut[g];
    rpn.wg_fill.upload(h_fill);
  }
  rpn.lds_bytes = (size_t)rpn.RL * NT * 8 + (size_t)2 * (NT / WAVE) * rpn.umax * 8 + (size_t)rpn.umax * 16 + (size_t)(NT / WAVE) * 16 +
                  (NT / WAVE) * 4 + 64;
  if (rpn.lds_bytes > 160 * 1024 - 512) return rpn.fail("LDS");
  rpn.e_slots.alloc((size_t)G * cap_slots);
  rpn.sums.alloc((size_t)G);
  rpn.y_slots = DevBuf<double>();
  rpn.h_nruns = nruns;
  rpn.h_diag.clear();
  rpn.partials.alloc((size_t)2 * ((size_t)zero_run + 1));
  MFM_HIP_CHECK(hipMemsetAsync(rpn.partials.p, 0, (size_t)16 * ((size_t)zero_run + 1), s));
  rpn.dv.alloc((size_t)2 * (n_items + 1));
  rpn.bar.alloc(RES_BAR_WORDS);
  MFM_HIP_CHECK(hipStreamSynchronize(s));
  MFM_HIP_CHECK(hipGetLastError());
  lap("descriptors, buffers");
  rpn.ready = true;
  rpn.why.clear();
  return true;
}

// tests (MFM_PLAN_CHECK): every array of the device-built layout against the host-built one
static inline std::string res_plan_compare(const ResPlan &a, const ResPlan &b, hipStream_t s) {
  if (a.G != b.G || a.RV != b.RV || a.RL != b.RL || a.RX != b.RX || a.umax != b.umax || a.item_bits != b.item_bits || a.n_items != b.n_items ||
      a.n_rows != b.n_rows || a.n_runs != b.n_runs || a.lds_bytes != b.lds_bytes)
    return "scalars";
  auto same = [&](const void *p, size_t np, const void *q, size_t nq, size_t elem) {
    if (np != nq) return false;
    std::vector<char> x(np * elem), y(nq * elem);
    if (np) {
      MFM_HIP_CHECK(hipMemcpyAsync(x.data(), p, np * elem, hipMemcpyDeviceToHost, s));
      MFM_HIP_CHECK(hipMemcpyAsync(y.data(), q, nq * elem, hipMemcpyDeviceToHost, s));
    }
    MFM_HIP_CHECK(hipStreamSynchronize(s));
    return x == y;
  };
#define MFM_RP_CMP(f) \
  if (!same(a.f.p, a.f.n, b.f.p, b.f.n, sizeof(*a.f.p))) return #f;
  MFM_RP_CMP(perm)
  MFM_RP_CMP(first_run)
  MFM_RP_CMP(wg_run_ptr)
  MFM_RP_CMP(wg_nruns)
  MFM_RP_CMP(wg_user_ptr)
  MFM_RP_CMP(wg_item_ptr)
  MFM_RP_CMP(ent_ptr)
  MFM_RP_CMP(scols)
  MFM_RP_CMP(entries)
  MFM_RP_CMP(uidw)
  MFM_RP_CMP(headw)
  MFM_RP_CMP(run_item)
  MFM_RP_CMP(user_desc)
  MFM_RP_CMP(item_desc)
  MFM_RP_CMP(run_feat)
  MFM_RP_CMP(wg_fill)
#undef MFM_RP_CMP
  return "";
}

}  // namespace mfm
