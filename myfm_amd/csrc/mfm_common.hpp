// mfm_common.hpp -- host-side plumbing shared by the HIP translation units of libmyfm_hip.so:
// error type, HIP_CHECK, RAII device buffers, pinned staging, event-based kernel timing.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "myfm_hip.h"

namespace mfm {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

// "done once" per DEVICE: kernel function attributes (hipFuncAttributeMaxDynamicSharedMemorySize) belong to the device
// that was current when they were set; a process that opens contexts on several GPUs must raise them on each. Racing
// threads may both find need() true: setting an attribute twice is harmless.
struct DeviceOnce {
  std::atomic<unsigned long long> done[4] = {};
  static int current() {
    int d = 0;
    (void)hipGetDevice(&d);
    return d & 255;
  }
  bool need() const {
    const int d = current();
    return !(done[d >> 6].load(std::memory_order_acquire) & (1ull << (d & 63)));
  }
  void mark() {
    const int d = current();
    done[d >> 6].fetch_or(1ull << (d & 63), std::memory_order_release);
  }
};

#define MFM_HIP_CHECK(expr)                                                                                   \
  do {                                                                                                        \
    hipError_t e__ = (expr);                                                                                  \
    if (e__ != hipSuccess)                                                                                    \
      throw ::mfm::Error(MFM_ERR_DEVICE, std::string(#expr) + " failed: " + hipGetErrorString(e__));          \
  } while (0)

template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  bool owned = true;  // false: a view of another DevBuf's memory (borrow), never freed here
  DevBuf() {}
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n), owned(o.owned) {
    o.p = nullptr;
    o.n = 0;
  }
  DevBuf &operator=(DevBuf &&o) noexcept {
    if (this != &o) {
      release();
      p = o.p;
      n = o.n;
      owned = o.owned;
      o.p = nullptr;
      o.n = 0;
    }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p && owned) (void)hipFree(p);
    p = nullptr;
    n = 0;
    owned = true;
  }
  // non-owning view of `o` (which must outlive this)
  void borrow(const DevBuf &o) {
    release();
    p = o.p;
    n = o.n;
    owned = false;
  }
  void alloc(size_t count) {
    release();
    n = count;
    if (count) MFM_HIP_CHECK(hipMalloc((void **)&p, count * sizeof(T)));
  }
  void alloc_zero(size_t count, hipStream_t s) {
    alloc(count);
    if (count) MFM_HIP_CHECK(hipMemsetAsync(p, 0, count * sizeof(T), s));
  }
  // synchronous upload of pageable host memory (setup path only)
  void upload(const T *src, size_t count) {
    alloc(count);
    if (count) MFM_HIP_CHECK(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
  }
  void upload(const std::vector<T> &v) { upload(v.data(), v.size()); }
};

// A small ring of pinned host buffers for the per-iteration uploads (pre-drawn normals,
// hyper-parameters): memcpy into pinned memory, hipMemcpyAsync on the ctx stream, an event
// per slot so a slot is not overwritten while its copy is still in flight.
struct PinnedRing {
  static constexpr int SLOTS = 3;
  void *buf[SLOTS] = {nullptr, nullptr, nullptr};
  size_t cap[SLOTS] = {0, 0, 0};
  hipEvent_t ev[SLOTS];
  bool ev_valid[SLOTS] = {false, false, false};
  int next = 0;
  ~PinnedRing() {
    for (int i = 0; i < SLOTS; i++) {
      if (buf[i]) (void)hipHostFree(buf[i]);
      if (ev_valid[i]) (void)hipEventDestroy(ev[i]);
    }
  }
  void upload(void *dst_dev, const void *src_host, size_t bytes, hipStream_t s) {
    if (!bytes) return;
    int i = next;
    next = (next + 1) % SLOTS;
    if (ev_valid[i]) MFM_HIP_CHECK(hipEventSynchronize(ev[i]));
    if (cap[i] < bytes) {
      if (buf[i]) MFM_HIP_CHECK(hipHostFree(buf[i]));
      buf[i] = nullptr;
      MFM_HIP_CHECK(hipHostMalloc(&buf[i], bytes, hipHostMallocDefault));
      cap[i] = bytes;
    }
    std::memcpy(buf[i], src_host, bytes);
    MFM_HIP_CHECK(hipMemcpyAsync(dst_dev, buf[i], bytes, hipMemcpyHostToDevice, s));
    if (!ev_valid[i]) {
      MFM_HIP_CHECK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
      ev_valid[i] = true;
    }
    MFM_HIP_CHECK(hipEventRecord(ev[i], s));
  }
};

// ---- kernel classes for the timing / roofline report --------------------------------------
enum KernelClass {
  KC_QBUILD = 0,       // q = X v_f (CSR SpMV)                              FMTrainer.hpp:320-340
  KC_SWEEP_V_LIGHT,    // latent sweep, columns <= 256 entries (wavefront per column)  FMTrainer.hpp:343-376
  KC_SWEEP_V_HEAVY,    // latent sweep, columns <= 4096 entries (wavefront x16 / workgroup per column)
  KC_SWEEP_V_COOP,     // latent sweep, long columns: co-resident chunks, single pass
  KC_SWEEP_V_SCAT,     // latent sweep, scattered level: row-blocked stats / draw / apply
  KC_SWEEP_V_LSTATS,   // latent sweep, huge columns: partial statistics
  KC_SWEEP_V_LDRAW,    //   ... draw
  KC_SWEEP_V_LAPPLY,   //   ... apply
  KC_SWEEP_V_CHAIN,    // latent sweep, sequential chain of tiny levels
  KC_SWEEP_W_LIGHT,    // linear sweep                                      FMTrainer.hpp:237-254
  KC_SWEEP_W_HEAVY,
  KC_SWEEP_W_COOP,
  KC_SWEEP_W_SCAT,
  KC_SWEEP_W_LSTATS,
  KC_SWEEP_W_LDRAW,
  KC_SWEEP_W_LAPPLY,
  KC_SWEEP_W_CHAIN,
  KC_UPDATE_E,         // fused rank-K re-score                             FM.hpp:54-136
  KC_BUILD_VT,         // V -> row-major gather table for the scorer
  KC_REDUCE_E,         // sum e, sum e^2                                    FMTrainer.hpp:138,223
  KC_SHIFT_E,          // e += delta                                        FMTrainer.hpp:227
  KC_GROUP_STATS,      // per-group sums of w / V                           FMTrainer.hpp:150-192
  KC_BLOCK_ROWCACHE,   // q_B, q_S, block-level scorer caches               FMTrainer.hpp:265,331,388
  KC_BLOCK_UNSYNC,     // statistics + un-sync pass through original_to_block  FMTrainer.hpp:268-275,401-417
  KC_BLOCK_RESYNC,     // re-sync pass                                      FMTrainer.hpp:306-311,473-480
  KC_BLOCK_SWEEP,      // block feature sweeps                              FMTrainer.hpp:276-302,419-470
  KC_TN_SAMPLE,        // truncated-normal latent z                         util.hpp:15-78
  KC_OPROBIT_EVAL,     // cutpoint likelihood / gradient / Hessian          OProbitSampler.hpp:389-413
  KC_PREDICT,          // Predictor::predict*                               predictor.hpp:35-147
  KC_SWEEP_V_FUSED,    // latent sweep: last level's apply pass + next factor's first level on the LDS tile
  KC_SWEEP_V_RESIDENT, // latent sweep of a two-field table, all factors in one persistent launch (mfm_res.hpp)
  KC_CELL_PASS,        // latent sweep, cell path: one streaming pass over the rows (apply the pending field + next field's statistics)
  KC_CELL_SMALL,       // ... its table / draw / reduce kernels (block-row or column sized)
  KC_N
};

static const char *const kKernelClassNames[KC_N] = {
    "qbuild_spmv",        "sweep_V_light",     "sweep_V_heavy",      "sweep_V_coop",       "sweep_V_scattered",  "sweep_V_huge_stats",
    "sweep_V_huge_draw",  "sweep_V_huge_apply", "sweep_V_chain",     "sweep_w_light",      "sweep_w_heavy",
    "sweep_w_coop",       "sweep_w_scattered",  "sweep_w_huge_stats", "sweep_w_huge_draw", "sweep_w_huge_apply", "sweep_w_chain",
    "update_e_score",     "build_vt",
    "reduce_e",           "shift_e",           "group_stats",        "block_rowcache",     "block_unsync",
    "block_resync",       "block_sweep",       "tn_sample",          "oprobit_eval",       "predict",
    "sweep_V_fused_next", "sweep_V_resident", "cell_pass",          "cell_small"};

struct Timing {
  bool on = false;
  int only = -1;  // >= 0: only this kernel class is bracketed with events (cheap enough for a timed benchmark region)
  struct Rec {
    hipEvent_t a, b;
    int cls;
  };
  std::vector<Rec> pending;
  std::vector<hipEvent_t> pool;
  double ms[KC_N] = {0};
  int64_t launches[KC_N] = {0};
  double bytes[KC_N] = {0};

  ~Timing() {
    for (auto &r : pending) {
      (void)hipEventDestroy(r.a);
      (void)hipEventDestroy(r.b);
    }
    for (auto e : pool) (void)hipEventDestroy(e);
  }
  hipEvent_t get_event() {
    if (!pool.empty()) {
      hipEvent_t e = pool.back();
      pool.pop_back();
      return e;
    }
    hipEvent_t e;
    MFM_HIP_CHECK(hipEventCreate(&e));
    return e;
  }
  void resolve() {
    for (auto &r : pending) {
      MFM_HIP_CHECK(hipEventSynchronize(r.b));
      float t = 0;
      MFM_HIP_CHECK(hipEventElapsedTime(&t, r.a, r.b));
      ms[r.cls] += t;
      pool.push_back(r.a);
      pool.push_back(r.b);
    }
    pending.clear();
  }
  void reset() {
    resolve();
    for (int i = 0; i < KC_N; i++) {
      ms[i] = 0;
      launches[i] = 0;
      bytes[i] = 0;
    }
  }
};

// debugging aid (MFM_DEBUG_POISON_LDS=1): before every bracketed launch fill the whole LDS of every CU with signalling NaN
// patterns, so that a kernel reading LDS it has not written shows up deterministically instead of only when another
// stream's workgroup happened to leave bits there
static __global__ __launch_bounds__(1024) void k_poison_lds(uint32_t word) {
  extern __shared__ uint32_t poison_words[];
  for (int i = threadIdx.x; i < 160 * 256; i += 1024) poison_words[i] = word;
  __syncthreads();
  if (poison_words[(threadIdx.x * 37) % (160 * 256)] == 1u) __builtin_trap();  // (keeps the stores)
}
inline bool poison_lds(hipStream_t s, int cls, int line, bool clean = false) {  // MFM_DEBUG_POISON_LDS = a kernel class, or -1 for all;
  static const bool on = std::getenv("MFM_DEBUG_POISON_LDS") != nullptr;  // MFM_DEBUG_POISON_LINE = one launch site
  if (!on) return false;
  static const int only = std::atoi(std::getenv("MFM_DEBUG_POISON_LDS"));
  static const int only_line = std::getenv("MFM_DEBUG_POISON_LINE") ? std::atoi(std::getenv("MFM_DEBUG_POISON_LINE")) : -1;
  if (std::getenv("MFM_DEBUG_POISON_LIST")) {
    static std::vector<int> seen;
    if (std::find(seen.begin(), seen.end(), cls * 100000 + line) == seen.end()) {
      seen.push_back(cls * 100000 + line);
      std::fprintf(stderr, "[launch site] class %d line %d\n", cls, line);
    }
  }
  if (only >= 0 && only != cls) return false;
  if (only_line >= 0 && only_line != line) return false;
  static DeviceOnce raised;
  if (raised.need()) {
    MFM_HIP_CHECK(hipFuncSetAttribute((const void *)k_poison_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    raised.mark();
  }
  static const uint32_t word =
      std::getenv("MFM_DEBUG_POISON_WORD") ? (uint32_t)std::strtoul(std::getenv("MFM_DEBUG_POISON_WORD"), nullptr, 16) : 0x7ff4deadu;
  hipLaunchKernelGGL(k_poison_lds, dim3(1024), dim3(1024), 160 * 1024, s, clean ? 0u : word);  // (clean: after the scope)
  return true;
}

// RAII scope: brackets one kernel launch with events when timing is on.
struct TimedLaunch {
  Timing &t;
  hipStream_t s;
  int cls;
  hipEvent_t a;
  bool poisoned = false;
  int site = 0;
  TimedLaunch(Timing &t_, hipStream_t s_, int cls_, double alg_bytes, int line = __builtin_LINE())
      : t(t_), s(s_), cls(cls_), a(nullptr) {
    poisoned = poison_lds(s, cls_, line);
    site = line;
    if (t.on && (t.only < 0 || t.only == cls)) {
      t.launches[cls]++;
      t.bytes[cls] += alg_bytes;
      a = t.get_event();
      MFM_HIP_CHECK(hipEventRecord(a, s));
    }
  }
  ~TimedLaunch() {
    if (poisoned) poison_lds(s, cls, site, true);
    if (t.on && a) {
      hipEvent_t b = t.get_event();
      (void)hipEventRecord(b, s);
      t.pending.push_back({a, b, cls});
    }
  }
};

// ---- host-side sparse helpers ---------------------------------------------------------------
struct HostCsr {
  int64_t rows = 0, cols = 0;
  std::vector<int64_t> ptr;
  std::vector<int32_t> idx;
  std::vector<double> val;
  int64_t nnz() const { return (int64_t)idx.size(); }
};

// [0, n) in contiguous ranges on host threads (the copies of a 10^8-entry design are bound by first-touch page faults of one
// thread otherwise); f(lo, hi) must not throw. Small n: the calling thread alone.
template <class F>
inline void parallel_ranges(int64_t n, F f) {
  const int hw = (int)std::thread::hardware_concurrency();
  const int T = (int)std::max<int64_t>(1, std::min<int64_t>({n >> 20, 16, hw > 0 ? hw : 1}));
  if (T <= 1) {
    f((int64_t)0, n);
    return;
  }
  std::vector<std::thread> pool;
  for (int t = 1; t < T; t++) pool.emplace_back(f, n * t / T, n * (t + 1) / T);
  f((int64_t)0, n / T);
  for (auto &t : pool) t.join();
}

inline HostCsr make_host_csr(int64_t rows, int64_t cols, const int64_t *indptr, const int32_t *indices,
                             const double *data) {
  if (rows < 0 || cols < 0) throw Error(MFM_ERR_INVALID, "negative matrix shape");
  HostCsr X;
  X.rows = rows;
  X.cols = cols;
  if (indptr[0] != 0) throw Error(MFM_ERR_INVALID, "indptr[0] must be 0");
  std::atomic<int> bad(0);
  parallel_ranges(rows, [&](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; i++)
      if (indptr[i + 1] < indptr[i]) bad = 1;
  });
  if (bad) throw Error(MFM_ERR_INVALID, "indptr must be non-decreasing");
  const int64_t nnz = indptr[rows];
  if (nnz >= (int64_t)2147483647) throw Error(MFM_ERR_INVALID, "nnz must be < 2^31 per matrix");
  parallel_ranges(nnz, [&](int64_t lo, int64_t hi) {
    for (int64_t p = lo; p < hi; p++)
      if (indices[p] < 0 || indices[p] >= cols) bad = 1;
  });
  if (bad) throw Error(MFM_ERR_INVALID, "column index out of range");
  // the three copies side by side (each is bound by the first-touch page faults of its thread)
  if (nnz >= ((int64_t)1 << 22)) {
    std::thread t1([&]() { X.ptr.assign(indptr, indptr + rows + 1); });
    std::thread t2([&]() { X.idx.assign(indices, indices + nnz); });
    X.val.assign(data, data + nnz);
    t1.join();
    t2.join();
  } else {
    X.ptr.assign(indptr, indptr + rows + 1);
    X.idx.assign(indices, indices + nnz);
    X.val.assign(data, data + nnz);
  }
  return X;
}

// CSC of X with ascending row order inside every column (= Eigen's X.transpose() stored
// row-major, BaseFMTrainer.hpp:61).
inline HostCsr transpose_host(const HostCsr &X) {
  HostCsr T;
  T.rows = X.cols;
  T.cols = X.rows;
  T.ptr.assign(X.cols + 1, 0);
  for (int64_t p = 0; p < X.nnz(); p++) T.ptr[X.idx[p] + 1]++;
  for (int64_t j = 0; j < X.cols; j++) T.ptr[j + 1] += T.ptr[j];
  T.idx.resize(X.nnz());
  T.val.resize(X.nnz());
  std::vector<int64_t> cur(T.ptr.begin(), T.ptr.end() - 1);
  for (int64_t i = 0; i < X.rows; i++)
    for (int64_t p = X.ptr[i]; p < X.ptr[i + 1]; p++) {
      int64_t q = cur[X.idx[p]]++;
      T.idx[q] = (int32_t)i;
      T.val[q] = X.val[p];
    }
  return T;
}

// SURVEY A.5. csc: one "row" per column, listing the rows it touches.
inline int32_t column_levels(const HostCsr &csc, std::vector<int32_t> &level) {
  std::vector<int32_t> rowlevel((size_t)csc.cols, -1);
  level.assign((size_t)csc.rows, 0);
  int32_t n_levels = 0;
  for (int64_t j = 0; j < csc.rows; j++) {
    int32_t m = -1;
    for (int64_t p = csc.ptr[j]; p < csc.ptr[j + 1]; p++) m = std::max(m, rowlevel[csc.idx[p]]);
    int32_t lv = m + 1;
    level[j] = lv;
    for (int64_t p = csc.ptr[j]; p < csc.ptr[j + 1]; p++) rowlevel[csc.idx[p]] = lv;
    n_levels = std::max(n_levels, lv + 1);
  }
  return n_levels;
}

// the pointers of the blocks beyond MAX_BLOCKS: device arrays, rebuilt (stream-ordered staging copy) whenever they are needed
struct BlockOverflow {
  DevBuf<const int32_t *> map;
  DevBuf<const double *> p0, p1, p2;
  DevBuf<int> stride;
  template <class T>
  static void put(DevBuf<T> &buf, const std::vector<T> &h, PinnedRing &ring, hipStream_t s) {
    if (h.empty()) return;
    if (buf.n < h.size()) buf.alloc(h.size());
    ring.upload(buf.p, h.data(), h.size() * sizeof(T), s);
  }
  template <class T>
  static void put_sync(DevBuf<T> &buf, const std::vector<T> &h) {
    if (h.empty()) return;
    if (buf.n < h.size()) buf.alloc(h.size());
    MFM_HIP_CHECK(hipMemcpy(buf.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  }
};


// Device-resident sparse matrix: CSR (row passes: q-build, re-score) and CSC (column sweeps).
static __global__ void k_iota_mul(int32_t *p, int64_t n, int32_t w) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (int32_t)(i * w);
}
static __global__ void k_fill_f64(double *p, int64_t n, double v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

struct DevSparse {
  int64_t rows = 0, cols = 0, nnz = 0;
  DevBuf<int32_t> rowptr, colidx;
  DevBuf<double> rval;
  DevBuf<int64_t> colptr;
  DevBuf<int32_t> rowidx;
  DevBuf<double> cval;
  double avg_row_nnz = 0;
  bool unit = false;      // every stored value is exactly 1.0 (one-hot designs): kernels skip the val arrays
  int32_t ell_width = -1; // every row has exactly this many entries (>= 0): kernels skip rowptr
  // The CSR straight from the caller's arrays (scipy layout: int64 indptr, int32 indices, double data) -- validated as
  // make_host_csr does, no host copy: the row pointers of a fixed-width table and the values of an all-ones table are written on
  // the device, the rest is copied to it from where it lies.
  void upload_raw(int64_t n_rows, int64_t n_cols, const int64_t *indptr, const int32_t *indices, const double *data) {
    if (n_rows < 0 || n_cols < 0) throw Error(MFM_ERR_INVALID, "negative matrix shape");
    if (indptr[0] != 0) throw Error(MFM_ERR_INVALID, "indptr[0] must be 0");
    rows = n_rows;
    cols = n_cols;
    const int64_t w0 = rows > 0 ? indptr[1] - indptr[0] : -1;
    std::atomic<int> bad(0), not_ell(0), not_unit(0);
    parallel_ranges(rows, [&](int64_t lo, int64_t hi) {
      for (int64_t i = lo; i < hi; i++) {
        const int64_t w = indptr[i + 1] - indptr[i];
        if (w < 0) bad = 1;
        if (w != w0) not_ell = 1;
      }
    });
    if (bad) throw Error(MFM_ERR_INVALID, "indptr must be non-decreasing");
    nnz = indptr[rows];
    if (nnz >= (int64_t)2147483647) throw Error(MFM_ERR_INVALID, "nnz must be < 2^31 per matrix");
    parallel_ranges(nnz, [&](int64_t lo, int64_t hi) {
      bool nu = false;
      for (int64_t p = lo; p < hi; p++) {
        if (indices[p] < 0 || indices[p] >= n_cols) bad = 1;
        nu |= data[p] != 1.0;
      }
      if (nu) not_unit = 1;
    });
    if (bad) throw Error(MFM_ERR_INVALID, "column index out of range");
    unit = nnz > 0 && !not_unit;
    ell_width = rows > 0 && !not_ell ? (int32_t)w0 : -1;
    avg_row_nnz = rows ? (double)nnz / rows : 0;
    rowptr.alloc((size_t)rows + 1);
    if (ell_width >= 0) {
      hipLaunchKernelGGL(k_iota_mul, dim3((unsigned)((rows + 1 + 255) / 256)), dim3(256), 0, 0, rowptr.p, rows + 1, ell_width);
    } else {
      std::vector<int32_t> rp((size_t)rows + 1);
      parallel_ranges(rows + 1, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; i++) rp[i] = (int32_t)indptr[i];
      });
      MFM_HIP_CHECK(hipMemcpy(rowptr.p, rp.data(), rp.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    colidx.upload(indices, (size_t)nnz);
    if (unit) {
      rval.alloc((size_t)nnz);
      hipLaunchKernelGGL(k_fill_f64, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, 0, rval.p, nnz, 1.0);
    } else {
      rval.upload(data, (size_t)nnz);
    }
    MFM_HIP_CHECK(hipDeviceSynchronize());
  }
  // the host copy of what upload_raw put on the device (the host planners, the checkers)
  HostCsr download() const {
    HostCsr X;
    X.rows = rows;
    X.cols = cols;
    std::vector<int32_t> rp((size_t)rows + 1);
    if (rows + 1 > 0) MFM_HIP_CHECK(hipMemcpy(rp.data(), rowptr.p, rp.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    X.ptr.assign(rp.begin(), rp.end());
    X.idx.resize((size_t)nnz);
    X.val.resize((size_t)nnz);
    if (nnz) {
      MFM_HIP_CHECK(hipMemcpy(X.idx.data(), colidx.p, (size_t)nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
      MFM_HIP_CHECK(hipMemcpy(X.val.data(), rval.p, (size_t)nnz * sizeof(double), hipMemcpyDeviceToHost));
    }
    return X;
  }
  void upload_csc(const HostCsr &Xt) {
    colptr.upload(Xt.ptr);
    rowidx.upload(Xt.idx);
    cval.upload(Xt.val);
  }
  void upload(const HostCsr &X, const HostCsr *Xt /* may be null: CSR only */) {
    rows = X.rows;
    cols = X.cols;
    nnz = X.nnz();
    // (three O(nnz) / O(rows) host passes: on host threads for long tables)
    std::atomic<int> not_unit(0), not_ell(0);
    parallel_ranges(nnz, [&](int64_t lo, int64_t hi) {
      for (int64_t p = lo; p < hi && !not_unit.load(std::memory_order_relaxed); p++)
        if (X.val[p] != 1.0) not_unit = 1;
    });
    unit = nnz > 0 && !not_unit;
    ell_width = rows > 0 ? (int32_t)(X.ptr[1] - X.ptr[0]) : -1;
    std::vector<int32_t> rp((size_t)rows + 1);
    const int32_t ew = ell_width;
    parallel_ranges(rows, [&](int64_t lo, int64_t hi) {
      for (int64_t i = lo; i < hi; i++) {
        if (X.ptr[i + 1] - X.ptr[i] != ew) not_ell = 1;
        rp[i] = (int32_t)X.ptr[i];
      }
    });
    if (not_ell) ell_width = -1;
    rp[(size_t)rows] = (int32_t)X.ptr[(size_t)rows];
    rowptr.upload(rp);
    colidx.upload(X.idx);
    if (unit && nnz >= ((int64_t)1 << 20)) {  // (all ones: written on the device instead of copied to it)
      rval.alloc((size_t)nnz);
      hipLaunchKernelGGL(k_fill_f64, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, 0, rval.p, nnz, 1.0);
      MFM_HIP_CHECK(hipDeviceSynchronize());
    } else {
      rval.upload(X.val);
    }
    avg_row_nnz = rows ? (double)nnz / rows : 0;
    if (Xt) {
      colptr.upload(Xt->ptr);
      rowidx.upload(Xt->idx);
      cval.upload(Xt->val);
    }
  }
};

}  // namespace mfm
