#include <thread>
#include <cstdio>
// _myfm.cpp -- the drop-in boundary: a pybind11 module with the names and signatures of the
// reference's `myfm._myfm` (cpp_source/declare_module.hpp:67-404, stubs src/myfm/_myfm.pyi), whose
// trainer drives the MI355X device path through the C ABI of include/myfm_hip.h.
//
// Host side (this file): configuration, validation and error behaviour of the reference, the
// mt19937 draw order of the Gibbs iteration (FMTrainer.hpp / BaseFMTrainer.hpp:135-152), the O(G K)
// hyper-parameter conditionals, the (K-1)-dimensional cutpoint Newton / Metropolis step of ordered
// probit, sample retention and pickling. Device side (libmyfm_hip.so): everything that touches
// the N-row or nnz-sized data. There is no CPU fallback: without a GPU, training / prediction
// raise RuntimeError.
#include <pybind11/functional.h>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <limits>
#include <memory>
#include <random>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "mfm_hostnormals.hpp"
#include "mfm_mtjump.hpp"
#include "myfm_hip.h"

namespace py = pybind11;

// iterations of this process that took mfm_regression_iteration (tests assert the path they mean to test)
static long g_device_hyper_iterations = 0;
using std::vector;
typedef double Real;
typedef py::array_t<double, py::array::c_style | py::array::forcecast> NpF64;

namespace {

// ---- errors: the reference's exception classes (SURVEY 8b "Errors") -------------------------------
[[noreturn]] void throw_code(int code, const char *msg) {
  if (code == MFM_ERR_INVALID) throw std::invalid_argument(msg);
  throw std::runtime_error(msg);
}
void ck(mfm_ctx *c, int code) {
  if (code != MFM_OK) throw_code(code, mfm_last_error(c));
}

// ---- sparse matrices crossing the boundary (pybind11/eigen.h's scipy caster, restated) -----------
struct Csr {
  int64_t rows = 0, cols = 0;
  vector<int64_t> indptr;
  vector<int32_t> indices;
  vector<double> data;
  int64_t nnz() const { return (int64_t)indices.size(); }
};

// The training table as the caller's own arrays (kept alive, converted only when dtype / layout require it): the trainer hands
// the pointers straight to mfm_set_main, which makes the library's one host copy -- no second copy of 3e8 bytes at config 3.
struct CsrView {
  int64_t rows = 0, cols = 0;
  py::array_t<int64_t, py::array::c_style | py::array::forcecast> indptr;
  py::array_t<int32_t, py::array::c_style | py::array::forcecast> indices;
  py::array_t<double, py::array::c_style | py::array::forcecast> data;
  void release() {
    indptr = py::array_t<int64_t, py::array::c_style | py::array::forcecast>();
    indices = py::array_t<int32_t, py::array::c_style | py::array::forcecast>();
    data = py::array_t<double, py::array::c_style | py::array::forcecast>();
  }
};
CsrView csr_view_from_py(const py::handle &obj) {
  py::object sparse = py::module_::import("scipy.sparse");
  py::object m = py::reinterpret_borrow<py::object>(obj);
  if (!py::isinstance(m, sparse.attr("csr_matrix"))) m = sparse.attr("csr_matrix")(m);
  py::tuple shape = m.attr("shape");
  CsrView X;
  X.rows = shape[0].cast<int64_t>();
  X.cols = shape[1].cast<int64_t>();
  X.indptr = py::array_t<int64_t, py::array::c_style | py::array::forcecast>::ensure(m.attr("indptr"));
  X.indices = py::array_t<int32_t, py::array::c_style | py::array::forcecast>::ensure(m.attr("indices"));
  X.data = py::array_t<double, py::array::c_style | py::array::forcecast>::ensure(m.attr("data"));
  if (!X.indptr || !X.indices || !X.data) throw std::invalid_argument("could not convert the sparse matrix to CSR arrays");
  if ((int64_t)X.indptr.size() != X.rows + 1 || X.indices.size() != X.data.size())
    throw std::invalid_argument("inconsistent CSR arrays");
  return X;
}

Csr csr_from_py(const py::handle &obj) {
  py::object sparse = py::module_::import("scipy.sparse");
  py::object m = py::reinterpret_borrow<py::object>(obj);
  if (!py::isinstance(m, sparse.attr("csr_matrix"))) m = sparse.attr("csr_matrix")(m);
  py::tuple shape = m.attr("shape");
  Csr X;
  X.rows = shape[0].cast<int64_t>();
  X.cols = shape[1].cast<int64_t>();
  auto ip = py::array_t<int64_t, py::array::c_style | py::array::forcecast>::ensure(m.attr("indptr"));
  auto ix = py::array_t<int32_t, py::array::c_style | py::array::forcecast>::ensure(m.attr("indices"));
  auto dv = NpF64::ensure(m.attr("data"));
  if (!ip || !ix || !dv) throw std::invalid_argument("could not convert the sparse matrix to CSR arrays");
  X.indptr.assign(ip.data(), ip.data() + ip.size());
  X.indices.assign(ix.data(), ix.data() + ix.size());
  X.data.assign(dv.data(), dv.data() + dv.size());
  if ((int64_t)X.indptr.size() != X.rows + 1) throw std::invalid_argument("malformed CSR matrix");
  return X;
}

py::object csr_to_py(const Csr &X) {
  py::object sparse = py::module_::import("scipy.sparse");
  py::array_t<double> data((py::ssize_t)X.data.size(), X.data.data());
  py::array_t<int32_t> indices((py::ssize_t)X.indices.size(), X.indices.data());
  py::array_t<int64_t> indptr((py::ssize_t)X.indptr.size(), X.indptr.data());
  return sparse.attr("csr_matrix")(py::make_tuple(data, indices, indptr), py::make_tuple(X.rows, X.cols));
}

py::array_t<double> vec_to_np(const vector<double> &v) { return py::array_t<double>((py::ssize_t)v.size(), v.data()); }
vector<double> np_to_vec(const py::handle &h) {
  auto a = NpF64::ensure(h);
  if (!a) throw std::invalid_argument("expected a float64 array");
  return vector<double>(a.data(), a.data() + a.size());
}
// column-major (rows, cols) storage <-> numpy (rows, cols)
py::array_t<double> colmajor_to_np(const vector<double> &v, int64_t rows, int64_t cols) {
  py::array_t<double, py::array::f_style> a({(py::ssize_t)rows, (py::ssize_t)cols});
  std::copy(v.begin(), v.end(), a.mutable_data());
  return std::move(a);
}
vector<double> np_to_colmajor(const py::handle &h, int64_t *rows, int64_t *cols) {
  auto a = py::array_t<double, py::array::f_style | py::array::forcecast>::ensure(h);
  if (!a || a.ndim() != 2) throw std::invalid_argument("expected a 2-d float64 array");
  *rows = a.shape(0);
  *cols = a.shape(1);
  return vector<double>(a.data(), a.data() + a.size());
}

// ---- TaskType / FMLearningConfig / ConfigBuilder (FMLearningConfig.hpp:11-203) -------------------
enum class TaskType { REGRESSION = 0, CLASSIFICATION = 1, ORDERED = 2 };
typedef vector<std::pair<size_t, vector<size_t>>> CutpointGroupType;

struct FMLearningConfig {
  Real alpha_0, beta_0, gamma_0, mu_0, reg_0;
  TaskType task_type;
  Real nu_oprobit;
  bool fit_w0, fit_linear;
  int n_iter, n_kept_samples;
  Real cutpoint_scale;
  vector<size_t> group_index;
  size_t n_groups = 0;
  vector<vector<size_t>> group_vs_feature_index;
  CutpointGroupType cutpoint_groups;
  // Not in the reference (it has ONE generator and nothing to choose): true keeps the trainer's std::mt19937 on the host for the
  // whole fit, so that every variate -- the hyper-parameter / w / V draws AND the latent draws of probit classification and
  // ordered probit, which consume the generator in data-dependent rejection loops row after row (FMTrainer.hpp:498-521,
  // OProbitSampler.hpp:238-272, util.hpp:15-78) -- is the reference's own under the same seed. false (default): the generator is
  // handed to the device; regression chains are still the reference's draw for draw, the latent draws of the other two tasks
  // come from per-row Philox streams (same law, different numbers). ConfigBuilder.set_exact_latent_draws; the environment
  // variable MYFM_AMD_HOST_RNG=1 forces it for every fit of the process.
  bool exact_latent_draws = false;
  // latent_mode (ConfigBuilder.set_latent_mode): 0 "philox" = per-row Philox streams (distributional parity); 1 "host" = the
  // trainer's std::mt19937 stays on the host for every variate of the fit; 2 "exact" = the generator lives on the device as for
  // regression and the latent draws consume ITS stream in the reference's order, evaluated in parallel on the device
  // (csrc/mfm_latent.hpp: coalescing flows) -- the same chain as mode 1 draw for draw. set_exact_latent_draws(true) selects
  // mode 2 (MYFM_AMD_EXACT_ON_HOST=1: mode 1).
  int latent_mode = 2;
  // latent_order (ConfigBuilder.set_latent_row_order; classification, modes "exact" / "host"): the rows in the order in which the
  // reference draws their latent z -- entry i = the row of THIS table that is the caller's row i. Empty: the table's own order.
  // (MyFM*.fit sorts the rows for the device paths and passes the inverse permutation here; ordered probit carries the same
  // information in its cutpoint groups' row lists.)
  vector<int64_t> latent_order;
  bool host_rng() const { return latent_mode == 1 || std::getenv("MYFM_AMD_HOST_RNG") != nullptr; }
  bool exact_dev() const { return latent_mode == 2 && !host_rng() && task_type != TaskType::REGRESSION; }

  // FMLearningConfig.hpp:17-57
  FMLearningConfig(Real alpha_0, Real beta_0, Real gamma_0, Real mu_0, Real reg_0, TaskType task_type, Real nu_oprobit,
                   bool fit_w0, bool fit_linear, const vector<size_t> &group_index, int n_iter, int n_kept_samples,
                   Real cutpoint_scale, const CutpointGroupType &cutpoint_groups)
      : alpha_0(alpha_0), beta_0(beta_0), gamma_0(gamma_0), mu_0(mu_0), reg_0(reg_0), task_type(task_type),
        nu_oprobit(nu_oprobit), fit_w0(fit_w0), fit_linear(fit_linear), n_iter(n_iter), n_kept_samples(n_kept_samples),
        cutpoint_scale(cutpoint_scale), group_index(group_index), cutpoint_groups(cutpoint_groups) {
    std::set<size_t> all_index(group_index.begin(), group_index.end());
    n_groups = all_index.size();
    for (size_t i = 0; i < n_groups; i++) {
      if (all_index.find(i) == all_index.cend()) {
        std::ostringstream ss;
        ss << "No matching index for group index " << i << " found.";
        throw std::invalid_argument(ss.str());
      }
    }
    group_vs_feature_index = vector<vector<size_t>>(n_groups);
    size_t feature_index = 0;
    for (auto g : group_index) group_vs_feature_index[g].push_back(feature_index++);
    if (n_kept_samples < 0) throw std::invalid_argument("n_kept_samples must be non-negative,");
    if (n_iter <= 0) throw std::invalid_argument("n_iter must be positive.");
    if (n_iter < n_kept_samples) throw std::invalid_argument("n_kept_samples must not exceed n_iter.");
  }
};

// FMLearningConfig.hpp:92-201
struct ConfigBuilder {
  Real alpha_0 = 1, beta_0 = 1, gamma_0 = 1, mu_0 = 1, reg_0 = 1;
  int n_iter = 100, n_kept_samples = 10;
  TaskType task_type = TaskType::REGRESSION;
  Real nu_oprobit = 5;
  bool fit_w0 = true, fit_linear = true;
  vector<size_t> group_index;
  Real cutpoint_scale = 10;
  CutpointGroupType cutpoint_groups;
  bool exact_latent_draws = true;   // (see FMLearningConfig::exact_latent_draws)
  int latent_mode = -1;             // -1: follows exact_latent_draws
  vector<int64_t> latent_order;
  ConfigBuilder &set_latent_row_order(const vector<int64_t> &a) { latent_order = a; return *this; }

  ConfigBuilder &set_exact_latent_draws(bool a) { exact_latent_draws = a; return *this; }
  ConfigBuilder &set_latent_mode(const std::string &m) {
    if (m == "philox") latent_mode = 0;
    else if (m == "host") latent_mode = 1;
    else if (m == "exact") latent_mode = 2;
    else throw std::invalid_argument("latent mode must be \"philox\", \"host\" or \"exact\"");
    return *this;
  }
  ConfigBuilder &set_alpha_0(Real a) { alpha_0 = a; return *this; }
  ConfigBuilder &set_beta_0(Real a) { beta_0 = a; return *this; }
  ConfigBuilder &set_gamma_0(Real a) { gamma_0 = a; return *this; }
  ConfigBuilder &set_mu_0(Real a) { mu_0 = a; return *this; }
  ConfigBuilder &set_reg_0(Real a) { reg_0 = a; return *this; }
  ConfigBuilder &set_n_iter(int a) { n_iter = a; return *this; }
  ConfigBuilder &set_n_kept_samples(int a) { n_kept_samples = a; return *this; }
  ConfigBuilder &set_task_type(TaskType a) { task_type = a; return *this; }
  ConfigBuilder &set_group_index(const vector<size_t> a) { group_index = a; return *this; }
  ConfigBuilder &set_identical_groups(size_t n_features) { group_index.assign(n_features, 0); return *this; }
  ConfigBuilder &set_nu_oprobit(size_t nu) { nu_oprobit = (Real)nu; return *this; }
  ConfigBuilder &set_fit_w0(bool a) { fit_w0 = a; return *this; }
  ConfigBuilder &set_fit_linear(bool a) { fit_linear = a; return *this; }
  ConfigBuilder &set_cutpoint_scale(Real a) { cutpoint_scale = a; return *this; }
  ConfigBuilder &set_cutpoint_groups(const CutpointGroupType &a) { cutpoint_groups = a; return *this; }
  FMLearningConfig build() {
    FMLearningConfig c(alpha_0, beta_0, gamma_0, mu_0, reg_0, task_type, nu_oprobit, fit_w0, fit_linear, group_index, n_iter,
                       n_kept_samples, cutpoint_scale, cutpoint_groups);
    c.exact_latent_draws = exact_latent_draws;
    c.latent_mode = latent_mode >= 0 ? latent_mode : (exact_latent_draws ? (std::getenv("MYFM_AMD_EXACT_ON_HOST") ? 1 : 2) : 0);
    c.latent_order = latent_order;
    return c;
  }
};

// ---- RelationBlock (definitions.hpp:30-52) ---------------------------------------------------------
// [0, n) in contiguous ranges on host threads (copies of 10^7..10^8-element index arrays: bound by the first-touch page faults of
// one thread otherwise); f(lo, hi) must not throw
template <class F>
static void host_ranges(int64_t n, F f) {
  const int hw = (int)std::thread::hardware_concurrency();
  const int T = (int)std::max<int64_t>(1, std::min<int64_t>({n >> 20, 16, hw > 0 ? hw : 1}));
  if (T <= 1) {
    f((int64_t)0, n);
    return;
  }
  vector<std::thread> pool;
  for (int t = 1; t < T; t++) pool.emplace_back(f, n * t / T, n * (t + 1) / T);
  f((int64_t)0, n / T);
  for (auto &t : pool) t.join();
}

struct RelationBlock {
  // original_to_block (definitions.hpp:33) as a plain int64 array: what the C ABI takes, filled (and first touched) by several
  // threads -- 5e7 entries per block at config 5; the Python attribute makes a list of it when somebody asks
  std::unique_ptr<int64_t[]> map;
  size_t mapper_size;
  Csr X;
  size_t block_size, feature_size;
  // src: n int64 indices (validated here: definitions.hpp:38-41)
  RelationBlock(const int64_t *src, size_t n, Csr X_)
      : map(new int64_t[std::max<size_t>(n, 1)]), mapper_size(n), X(std::move(X_)), block_size((size_t)X.rows),
        feature_size((size_t)X.cols) {
    std::atomic<int> bad(0);
    int64_t *dst = map.get();
    const int64_t B = (int64_t)block_size;
    host_ranges((int64_t)n, [&](int64_t lo, int64_t hi) {
      bool b = false;
      for (int64_t i = lo; i < hi; i++) {
        b |= src[i] < 0 || src[i] >= B;
        dst[i] = src[i];
      }
      if (b) bad = 1;
    });
    if (bad) throw std::runtime_error("index mapping points to non-existing row.");
  }
  RelationBlock(const vector<size_t> &o2b, Csr X_)
      : RelationBlock(reinterpret_cast<const int64_t *>(o2b.data()), o2b.size(), std::move(X_)) {
    static_assert(sizeof(size_t) == sizeof(int64_t), "original_to_block is reinterpreted as int64");
  }
  const int64_t *map64() const { return map.get(); }
  vector<size_t> original_to_block() const { return vector<size_t>(map.get(), map.get() + mapper_size); }
};
typedef vector<std::shared_ptr<RelationBlock>> Relations;

Relations relations_from_py(const py::handle &h) {
  Relations out;
  for (auto item : py::reinterpret_borrow<py::sequence>(h)) out.push_back(item.cast<std::shared_ptr<RelationBlock>>());
  return out;
}

// util.hpp:147-165
template <class M>
size_t check_row_consistency_return_column(const M &X, const Relations &relations) {
  size_t row = (size_t)X.rows, col = (size_t)X.cols;
  int i = 0;
  for (const auto &rel : relations) {
    if (row != rel->mapper_size) {
      std::ostringstream ss;
      ss << "main table has size " << row << " but the relation[" << i << "] has size " << rel->mapper_size;
      throw std::runtime_error(ss.str());
    }
    col += rel->feature_size;
    i++;
  }
  return col;
}

// The GPU this process works on: MYFM_AMD_DEVICE (one process per GPU sets it to its local rank), default 0.
// Training contexts and prediction designs are created on the same device.
int selected_device() {
  if (const char *e = std::getenv("MYFM_AMD_DEVICE")) return std::atoi(e);
  return 0;
}

// A prediction design resident on the GPU for the duration of one call.
struct DeviceDesign {
  mfm_design *d = nullptr;
  DeviceDesign(const Csr &X, const Relations &rels, int device = -1) {
    if (device < 0) device = selected_device();
    int code = mfm_design_create(device, X.rows, X.cols, X.indptr.data(), X.indices.data(), X.data.data(), &d);
    if (code != MFM_OK) throw_code(code, mfm_global_error());
    for (auto &r : rels) {
      code = mfm_design_add_block(d, r->X.rows, r->X.cols, r->X.indptr.data(), r->X.indices.data(), r->X.data.data(),
                                  r->map64());
      if (code != MFM_OK) {
        std::string msg = mfm_design_last_error(d);
        mfm_design_destroy(d);
        d = nullptr;
        throw_code(code, msg.c_str());
      }
    }
  }
  ~DeviceDesign() {
    if (d) mfm_design_destroy(d);
  }
  void predict(int rank, int S, const double *w0s, const double *ws, const double *Vs, int mode, int n_cut,
               const double *cuts, double *out) {
    int code = mfm_design_predict(d, rank, S, w0s, ws, Vs, mode, n_cut, cuts, out);
    if (code != MFM_OK) throw_code(code, mfm_design_last_error(d));
  }
};

// Kept posterior samples resident on the GPU (include/myfm_hip.h "device-resident posterior samples").
struct DeviceStore {
  mfm_store *st = nullptr;
  DeviceStore(int64_t D, int K) {
    int code = mfm_store_create(selected_device(), D, K, &st);
    if (code != MFM_OK) throw_code(code, mfm_global_error());
  }
  ~DeviceStore() {
    if (st) mfm_store_destroy(st);
  }
  DeviceStore(const DeviceStore &) = delete;
  int size() const { return mfm_store_size(st); }
};

// ---- FM (FM.hpp:10-172) -----------------------------------------------------------------------------
struct FM {
  int n_factors = 0;
  // a kept sample that still lives in the device store (index), untouched by the user since training
  std::shared_ptr<DeviceStore> store;
  int store_idx = -1;
  Real w0 = 0;
  vector<Real> w;               // (D)
  vector<Real> V;               // column-major (D, K)
  vector<vector<Real>> cutpoints;
  bool initialized = false;
  // live sample handed to callbacks: w / V stay on the GPU until somebody looks at them
  std::function<void(FM &)> fetch;
  bool stale = false;
  // live sample only: the training context, and the test designs already uploaded for it (the LibFM-like
  // callbacks score the same X_test object every iteration: libfm.py:85)
  mfm_ctx *live_ctx = nullptr;
  struct CachedDesign {
    py::object X;
    vector<py::object> rels;
    std::shared_ptr<DeviceDesign> design;
  };
  std::shared_ptr<vector<CachedDesign>> design_cache;
  std::shared_ptr<DeviceDesign> cached_design(const py::object &Xo, const py::object &relso) {
    if (!design_cache) design_cache = std::make_shared<vector<CachedDesign>>();
    vector<py::object> rl;
    for (auto item : py::reinterpret_borrow<py::sequence>(relso)) rl.push_back(py::reinterpret_borrow<py::object>(item));
    for (auto &c : *design_cache) {
      if (c.X.ptr() != Xo.ptr() || c.rels.size() != rl.size()) continue;
      bool same = true;
      for (size_t i = 0; i < rl.size(); i++) same = same && c.rels[i].ptr() == rl[i].ptr();
      if (same) return c.design;
    }
    Csr X = csr_from_py(Xo);
    Relations rels = relations_from_py(relso);
    check(X, rels);
    auto dd = std::make_shared<DeviceDesign>(X, rels, live_ctx ? mfm_get_device(live_ctx) : -1);
    if (design_cache->size() >= 4) design_cache->erase(design_cache->begin());
    design_cache->push_back(CachedDesign{Xo, rl, dd});
    return dd;
  }

  FM() {}
  explicit FM(int n_factors) : n_factors(n_factors) {}
  FM(Real w0, vector<Real> w, vector<Real> V, int K, vector<vector<Real>> cutpoints = {})
      : n_factors(K), w0(w0), w(std::move(w)), V(std::move(V)), cutpoints(std::move(cutpoints)), initialized(true) {}
  // a kept sample is a plain host copy
  FM snapshot() {
    FM s;
    s.n_factors = n_factors;
    s.w0 = w0;
    if (stale && fetch) {
      // straight into the copy: the live sample stays device-resident, so callbacks that score it afterwards keep
      // the cached-design path (predict_score) instead of re-uploading the test design every iteration
      fetch(s);
    } else {
      s.w = w;
      s.V = V;
    }
    s.cutpoints = cutpoints;
    s.initialized = initialized;  // (a kept sample is detached from the training context)
    return s;
  }
  void ensure() {
    if (stale && fetch) {
      stale = false;
      fetch(*this);
    }
  }
  int64_t D() const { return (int64_t)w.size(); }

  // FM.hpp:34-45: one persistent normal_distribution; Eigen fills the col-major V in storage order.
  void initialize_weight(int64_t n_features, Real init_std, std::mt19937 &gen) {
    initialized = false;
    const size_t nV = (size_t)n_features * n_factors, nw = (size_t)n_features;
    // The bulk filler restates libstdc++ internals (generate_canonical<double, 53>, the polar method's return order): checked
    // once per process against the plain std::normal_distribution loop it replaces (odd and even counts); a C++ library that
    // differs makes every fit take the plain loop instead of silently changing the initial weights
    static const bool filler_ok = []() {
      for (size_t n : {(size_t)4099, (size_t)1024}) {
        std::mt19937 g1(20240917u), g2(20240917u);
        std::vector<double> a(n), b(n);
        std::normal_distribution<Real> nd;
        for (auto &v : a) v = nd(g1) * 0.1;
        mfm_hostnormals::fill_normals(g2, b.data(), n, 0.1);
        if (std::memcmp(a.data(), b.data(), n * sizeof(double)) != 0) return false;
      }
      return true;
    }();
    if (std::getenv("MYFM_AMD_STD_INIT") || !filler_ok) {  // the plain loop (tests hold the bulk filler against it)
      std::normal_distribution<Real> nd;
      V.resize(nV);
      for (auto &v : V) v = nd(gen) * init_std;
      w.resize(nw);
      for (auto &v : w) v = nd(gen) * init_std;
      w0 = nd(gen) * init_std;
    } else {
      // V, w, w0 are consecutive outputs of one distribution object: drawn as one sequence, then split
      V.resize(nV + nw + 1);
      mfm_hostnormals::fill_normals(gen, V.data(), nV + nw + 1, init_std);
      w.assign(V.begin() + (std::ptrdiff_t)nV, V.begin() + (std::ptrdiff_t)(nV + nw));
      w0 = V[nV + nw];
      V.resize(nV);
    }
    initialized = true;
  }

  // FM.hpp:57-77 checks, then the device scorer
  void check(const Csr &X, const Relations &relations) {
    size_t case_size = (size_t)X.rows, feature_size_all = (size_t)X.cols;
    for (auto const &rel : relations) {
      if (case_size != rel->mapper_size)
        throw std::invalid_argument("Relation blocks have inconsistent mapper size with case_size");
      feature_size_all += rel->feature_size;
    }
    if (feature_size_all != w.size()) {
      std::ostringstream ss;
      ss << "Total feature size mismatch. Should be " << w.size() << ", but got " << feature_size_all << ".";
      throw std::invalid_argument(ss.str());
    }
    if (!initialized) throw std::runtime_error("get_score called before initialization");
  }
  py::array_t<double> predict_score(const py::object &Xo, const py::object &relso) {
    if (live_ctx && stale) {  // score straight from the device-resident state
      auto dd = cached_design(Xo, relso);
      py::array_t<double> out((py::ssize_t)mfm_design_n_rows(dd->d));
      int code = mfm_design_score_ctx(dd->d, live_ctx, out.mutable_data());
      if (code != MFM_OK) throw_code(code, mfm_design_last_error(dd->d));
      return out;
    }
    ensure();
    Csr X = csr_from_py(Xo);
    Relations rels = relations_from_py(relso);
    check(X, rels);
    py::array_t<double> out((py::ssize_t)X.rows);
    DeviceDesign dd(X, rels);
    dd.predict(n_factors, 1, &w0, w.data(), V.data(), 0, 0, nullptr, out.mutable_data());
    return out;
  }
  // FM.hpp:137-162
  py::array_t<double> oprobit_predict_proba(const py::object &Xo, const py::object &relso, size_t cutpoint_index) {
    ensure();
    if (cutpoints.empty()) throw std::runtime_error("No cutpoint available for this FM.");
    Csr X = csr_from_py(Xo);
    Relations rels = relations_from_py(relso);
    check(X, rels);
    const vector<Real> &cp = cutpoints.at(cutpoint_index);
    int n_cpt = (int)cp.size();
    py::array_t<double> out({(py::ssize_t)X.rows, (py::ssize_t)(n_cpt + 1)});
    DeviceDesign dd(X, rels);
    dd.predict(n_factors, 1, &w0, w.data(), V.data(), 2, n_cpt, cp.data(), out.mutable_data());
    return out;
  }
};

// ---- FMHyperParameters (HyperParams.hpp) ------------------------------------------------------------
struct Hyper {
  Real alpha = 1;
  vector<Real> mu_w, lambda_w;  // (G)
  vector<Real> mu_V, lambda_V;  // column-major (G, K)
  size_t G = 0, K = 0;
  Hyper() {}
  Hyper(size_t n_factors, size_t n_groups)
      : mu_w(n_groups), lambda_w(n_groups), mu_V(n_groups * n_factors), lambda_V(n_groups * n_factors), G(n_groups),
        K(n_factors) {}
};

struct LearningHistory {
  vector<Hyper> hypers;
  vector<size_t> n_mh_accept;
  vector<Real> train_log_losses;
};

// ---- Predictor (predictor.hpp:14-167) ---------------------------------------------------------------
struct Predictor {
  size_t rank, feature_size;
  TaskType type;
  vector<FM> samples;
  Predictor(size_t rank, size_t feature_size, TaskType type) : rank(rank), feature_size(feature_size), type(type) {}

  // all samples are the consecutive, unmodified entries [first, first + S) of one device store: predict in place
  std::shared_ptr<DeviceStore> resident(int *first) const {
    if (samples.empty() || !samples[0].store || samples[0].store_idx < 0) return nullptr;
    auto st = samples[0].store;
    *first = samples[0].store_idx;
    for (size_t k = 0; k < samples.size(); k++)
      if (samples[k].store != st || samples[k].store_idx != *first + (int)k) return nullptr;
    return st;
  }
  // mode / cutpoints as mfm_design_predict
  void run_predict(const Csr &X, const Relations &rels, int mode, int n_cut, const vector<double> &cuts, double *out) const {
    DeviceDesign dd(X, rels);
    int first = 0;
    if (auto st = resident(&first)) {
      int code = mfm_design_predict_store(dd.d, st->st, first, (int)samples.size(), mode, n_cut, cuts.data(), out);
      if (code != MFM_OK) throw_code(code, mfm_design_last_error(dd.d));
      return;
    }
    vector<double> w0s, ws, Vs;
    pack(w0s, ws, Vs);
    dd.predict((int)rank, (int)samples.size(), w0s.data(), ws.data(), Vs.data(), mode, n_cut, cuts.data(), out);
  }

  void check_input(const Csr &X, const Relations &relations) const {  // predictor.hpp:24-33
    auto given = check_row_consistency_return_column(X, relations);
    if (feature_size != given) {
      std::ostringstream ss;
      ss << "Told to predict for " << given << " but this->feature_size is " << feature_size;
      throw std::invalid_argument(ss.str());
    }
  }
  void pack(vector<double> &w0s, vector<double> &ws, vector<double> &Vs) const {
    const size_t S = samples.size(), D = feature_size;
    w0s.resize(S);
    ws.resize(S * D);
    Vs.resize(S * D * rank);
    for (size_t s = 0; s < S; s++) {
      const_cast<FM &>(samples[s]).ensure();
      const FM &f = samples[s];
      if (f.w.size() != D || f.V.size() != D * rank) throw std::invalid_argument("feature size mismatch!");
      w0s[s] = f.w0;
      std::copy(f.w.begin(), f.w.end(), ws.begin() + s * D);
      std::copy(f.V.begin(), f.V.end(), Vs.begin() + s * D * rank);
    }
  }
  // predictor.hpp:126-147 (and :35-76: the worker count only changes how the CPU reference splits
  // the samples over threads; here all samples are scored on the GPU)
  py::array_t<double> predict_impl(const py::object &Xo, const py::object &relso, const char *empty_msg) const {
    Csr X = csr_from_py(Xo);
    Relations rels = relations_from_py(relso);
    check_input(X, rels);
    if (samples.empty()) throw std::runtime_error(empty_msg);
    py::array_t<double> out((py::ssize_t)X.rows);
    // regression averages scores, classification Phi(score); ORDERED falls through both branches of
    // predictor.hpp:136-144 and yields zeros
    if (type == TaskType::ORDERED) {
      std::fill(out.mutable_data(), out.mutable_data() + X.rows, 0.0);
      return out;
    }
    run_predict(X, rels, type == TaskType::CLASSIFICATION ? 1 : 0, 0, {}, out.mutable_data());
    return out;
  }
  py::array_t<double> predict(const py::object &X, const py::object &rels) const { return predict_impl(X, rels, "Empty samples!"); }
  py::array_t<double> predict_parallel(const py::object &Xo, const py::object &relso, size_t n_workers) const {
    // predict_parallel applies Phi only for CLASSIFICATION and otherwise averages raw scores (:53-65)
    if (type != TaskType::ORDERED) return predict_impl(Xo, relso, "Told to predict but no sample available.");
    Csr X = csr_from_py(Xo);
    Relations rels = relations_from_py(relso);
    check_input(X, rels);
    if (samples.empty()) throw std::runtime_error("Told to predict but no sample available.");
    py::array_t<double> out((py::ssize_t)X.rows);
    run_predict(X, rels, 0, 0, {}, out.mutable_data());
    return out;
  }
  // predictor.hpp:78-124
  py::array_t<double> predict_parallel_oprobit(const py::object &Xo, const py::object &relso, size_t n_workers,
                                               size_t cutpoint_index) const {
    Csr X = csr_from_py(Xo);
    Relations rels = relations_from_py(relso);
    check_input(X, rels);
    if (samples.empty()) throw std::runtime_error("Told to predict but no sample available.");
    if (type != TaskType::ORDERED) throw std::runtime_error("predict_parallel_oprobit must be called for oprobit model.");
    int n_cpt = (int)samples.at(0).cutpoints.at(cutpoint_index).size();
    vector<double> cuts;
    for (auto &s : samples) {
      const auto &cp = s.cutpoints.at(cutpoint_index);
      if ((int)cp.size() != n_cpt) throw std::runtime_error("inconsistent cutpoint sizes among samples.");
      cuts.insert(cuts.end(), cp.begin(), cp.end());
    }
    py::array_t<double> out({(py::ssize_t)X.rows, (py::ssize_t)(n_cpt + 1)});
    run_predict(X, rels, 2, n_cpt, cuts, out.mutable_data());
    return out;
  }
};

// ---- small dense algebra for the cutpoint sampler (Eigen LLT restated) ------------------------------
void cholesky_lower(const vector<Real> &A, int n, vector<Real> &L) {
  L.assign((size_t)n * n, 0);
  for (int j = 0; j < n; j++) {
    Real d = A[(size_t)j * n + j];
    for (int k = 0; k < j; k++) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
    if (!(d > 0)) d = std::numeric_limits<Real>::quiet_NaN();  // Eigen's LLT carries on with NaNs
    Real ljj = std::sqrt(d);
    L[(size_t)j * n + j] = ljj;
    for (int i = j + 1; i < n; i++) {
      Real s = A[(size_t)i * n + j];
      for (int k = 0; k < j; k++) s -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
      L[(size_t)i * n + j] = s / ljj;
    }
  }
}
void llt_solve(const vector<Real> &L, int n, vector<Real> &b) {
  for (int i = 0; i < n; i++) {
    Real s = b[i];
    for (int k = 0; k < i; k++) s -= L[(size_t)i * n + k] * b[k];
    b[i] = s / L[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    Real s = b[i];
    for (int k = i + 1; k < n; k++) s -= L[(size_t)k * n + i] * b[k];
    b[i] = s / L[(size_t)i * n + i];
  }
}
bool has_nan(const vector<Real> &v) {
  for (Real a : v)
    if (std::isnan(a)) return true;
  return false;
}
Real norm2(const vector<Real> &v) {
  Real s = 0;
  for (Real a : v) s += a * a;
  return std::sqrt(s);
}

// ---- ordered-probit cutpoint sampler, host part (OProbitSampler.hpp:15-481). The O(N) likelihood /
// gradient / Hessian accumulation over the rows (:402-413) and the per-row latent draw (:238-272)
// run on the device; the (K-1)-dimensional reparametrisation, damped Newton search and the
// multivariate-t Metropolis step stay here.
// ---- truncated-normal samplers on the trainer's own std::mt19937 (util.hpp:15-78), for the parity mode
// exact_latent_draws (ConfigBuilder.set_exact_latent_draws / MYFM_AMD_HOST_RNG=1) ONLY: there the latent draws of probit classification / ordered probit consume the generator row
// after row exactly as the reference does (FMTrainer.hpp:498-521, OProbitSampler.hpp:238-272), so that those chains can
// be compared with the CPU sampler draw for draw. The product path draws them on the device (mfm_tasks.hpp).
// The host's window into the DEVICE random stream (latent mode "exact", include/myfm_hip.h mfm_rng_host_read / _advance): the engine
// outputs from the stream's position on, fetched in growing pieces; commit() moves the device position past what was consumed.
struct DeviceStreamWindow {
  mfm_ctx *ctx = nullptr;
  vector<uint32_t> buf;
  size_t pos = 0;
  uint64_t base = 0;   // outputs consumed before buf[0]
  size_t piece = 4096;
  explicit DeviceStreamWindow(mfm_ctx *c) : ctx(c) {}
  uint32_t next() {
    if (pos == buf.size()) {
      base += buf.size();
      buf.resize(piece);
      ck(ctx, mfm_rng_host_read(ctx, base, (int64_t)piece, buf.data()));
      pos = 0;
      piece = std::min<size_t>(piece * 4, (size_t)1 << 24);
    }
    return buf[pos++];
  }
  uint64_t consumed() const { return base + pos; }
  void commit() {
    ck(ctx, mfm_rng_host_advance(ctx, consumed()));
    buf.clear();
    pos = 0;
    base = 0;
    piece = 4096;
  }
};
// UniformRandomBitGenerator over the trainer's host std::mt19937 or over the device stream: libstdc++'s distributions take two
// 32-bit outputs per generate_canonical<double, 53> from either (same range as std::mt19937), so the variates are the same.
struct StreamEngine {
  typedef uint32_t result_type;
  static constexpr result_type min() { return 0u; }
  static constexpr result_type max() { return 0xffffffffu; }
  std::mt19937 *host = nullptr;
  DeviceStreamWindow *dev = nullptr;
  result_type operator()() { return host ? (result_type)(*host)() : dev->next(); }
};

template <class Gen>
static Real host_tn_left(Gen &gen, Real mu_minus) {  // util.hpp:15-38
  if (mu_minus < 0) {
    std::normal_distribution<Real> dist(0, 1);
    for (;;) {
      Real z = dist(gen);
      if (z > mu_minus) return z;
    }
  }
  Real alpha_star = (mu_minus + std::sqrt(mu_minus * mu_minus + 4)) / 2;
  std::uniform_real_distribution<Real> dist(0, 1);
  for (;;) {
    Real z = -std::log(dist(gen)) / alpha_star + mu_minus;
    Real rho = std::exp(-(z - alpha_star) * (z - alpha_star) / 2);
    Real u = dist(gen);
    if (u < rho) return z;
  }
}
template <class Gen>
static Real host_tn_twoside(Gen &gen, Real mu_minus, Real mu_plus) {  // util.hpp:40-62
  std::uniform_real_distribution<Real> proposal(mu_minus, mu_plus);
  std::uniform_real_distribution<Real> acceptance(0, 1);
  for (;;) {
    Real z = proposal(gen);
    Real rho;
    if (mu_minus <= 0 && mu_plus >= 0)
      rho = std::exp(-z * z / 2);
    else if (mu_plus < 0)
      rho = std::exp((mu_plus * mu_plus - z * z) / 2);
    else
      rho = std::exp((mu_minus * mu_minus - z * z) / 2);
    Real u = acceptance(gen);
    if (u < rho) return z;
  }
}
template <class Gen>
static Real host_tn_left(Gen &gen, Real mean, Real sd, Real mu_minus) {  // :63-68
  return mean + sd * host_tn_left(gen, (mu_minus - mean) / sd);
}
template <class Gen>
static Real host_tn_right(Gen &gen, Real mu_plus) { return -host_tn_left(gen, -mu_plus); }  // :70-73
template <class Gen>
static Real host_tn_right(Gen &gen, Real mean, Real sd, Real mu_plus) {                      // :75-79
  return mean + sd * host_tn_right(gen, (mu_plus - mean) / sd);
}

// the parallel evaluation of the exact latent draws refused a draw (status of mfm_*_exact): said once per process -- the draws are
// the same, but a sequential loop over N rows on the host is ~10x slower than the device path, and nobody should have to guess why
static void warn_sequential_latent(int32_t status) {
  static bool said = false;
  if (said) return;
  said = true;
  static const char *why[] = {"", "the draw's path left the prepared windows", "scratch space (snapshots)", "scratch space (walkers)",
                              "more engine outputs needed than prepared",
                              "a score lies more than 1000 standard deviations on the wrong side of its class"};
  std::fprintf(stderr,
               "myfm_amd: exact latent draws: the parallel evaluation refused a draw (status %d: %s); it is made row by row on the "
               "host from the same stream (same draws, slower). GibbsSession.latent_info() counts these.\n",
               (int)status, status >= 1 && status <= 5 ? why[status] : "?");
}

struct OprobitSampler {
  mfm_ctx *ctx;
  int group;  // device-side cutpoint group
  int K;
  Real reg, nu;
  StreamEngine rng;  // the trainer's host generator, or the device stream (latent mode "exact")
  vector<Real> alpha_now, gamma_now, H;
  size_t accept_count = 0;

  OprobitSampler(mfm_ctx *ctx, int group, int K, StreamEngine rng, Real reg, Real nu)
      : ctx(ctx), group(group), K(K), reg(reg), nu(nu), rng(rng) {
    alpha_now.assign(K - 1, 0);
    gamma_now.assign(K - 1, 0);
    alpha_to_gamma(gamma_now, alpha_now);
    H.assign((size_t)(K - 1) * (K - 1), 0);
  }
  int n() const { return K - 1; }
  static void alpha_to_gamma(vector<Real> &target, const vector<Real> &alpha) {  // :95-101
    if (alpha.empty()) return;
    target[0] = alpha[0];
    for (size_t i = 1; i < alpha.size(); i++) target[i] = target[i - 1] + std::exp(alpha[i]);
  }
  static void jacobian_dgamma_dalpha(vector<Real> &J, const vector<Real> &alpha) {  // :74-93
    int m = (int)alpha.size();
    std::fill(J.begin(), J.end(), 0);
    J[0] = 1;
    for (int j = 1; j < m; j++) J[j] = 1;
    for (int i = 1; i < m; i++) {
      Real ed = std::exp(alpha[i]);
      for (int j = i; j < m; j++) J[(size_t)i * m + j] = ed;
    }
  }
  Real log_p_mvt(const vector<Real> &Si, const vector<Real> &mu, Real nu_, const vector<Real> &x) const {  // :49-53
    int m = n();
    Real lp = 0;
    for (int i = 0; i < m; i++) {
      Real s = 0;
      for (int j = 0; j < m; j++) s += Si[(size_t)i * m + j] * (x[j] - mu[j]);
      lp += (x[i] - mu[i]) * s;
    }
    return std::log(1 + lp / nu_) * (-nu_ - m) / 2;
  }
  vector<Real> sample_mvt(const vector<Real> &Si, Real nu_) {  // :55-72
    int m = n();
    vector<Real> result(m);
    std::normal_distribution<Real> base_dist(0, 1);
    std::gamma_distribution<Real> chi_gen(nu_ / 2);
    for (int i = 0; i < m; i++) result[i] = base_dist(rng);
    vector<Real> L;
    cholesky_lower(Si, m, L);
    for (int i = m - 1; i >= 0; i--) {
      Real s = result[i];
      for (int k = i + 1; k < m; k++) s -= L[(size_t)k * m + i] * result[k];
      result[i] = s / L[(size_t)i * m + i];
    }
    Real denom = std::sqrt(chi_gen(rng) * 2 / nu_);
    for (auto &r : result) r /= denom;
    return result;
  }
  // operator(), :389-463 -- rows on the device, the rest here
  Real eval(const vector<Real> &alpha, vector<Real> &dalpha, vector<Real> *Ht) {
    int m = n();
    vector<Real> gamma(m, 0);
    alpha_to_gamma(gamma, alpha);
    vector<Real> J((size_t)m * m);
    jacobian_dgamma_dalpha(J, alpha);
    Real ll = 0;
    dalpha.assign(m, 0);
    if (Ht) Ht->assign((size_t)m * m, 0);
    ck(ctx, mfm_oprobit_eval(ctx, group, gamma.data(), &ll, dalpha.data(), Ht ? Ht->data() : nullptr));
    if (Ht) {
      vector<Real> &Hh = *Ht;
      vector<Real> expAlpha(m);
      for (int k = 0; k < m; k++) expAlpha[k] = std::exp(alpha[k]);
      vector<Real> T((size_t)m * m, 0), R((size_t)m * m, 0);
      for (int a = 0; a < m; a++)
        for (int b = 0; b < m; b++) {
          Real s = 0;
          for (int k = 0; k < m; k++) s += J[(size_t)a * m + k] * Hh[(size_t)k * m + b];
          T[(size_t)a * m + b] = s;
        }
      for (int a = 0; a < m; a++)
        for (int b = 0; b < m; b++) {
          Real s = 0;
          for (int k = 0; k < m; k++) s += T[(size_t)a * m + k] * J[(size_t)b * m + k];
          R[(size_t)a * m + b] = s;
        }
      Hh = R;
      for (int mm = 1; mm < (K - 1); mm++)
        for (int j = 1; j <= mm; j++) Hh[(size_t)j * m + j] += dalpha[mm] * expAlpha[j];
      Hh[0] -= reg;
      for (int mm = 1; mm < (K - 1); mm++) Hh[(size_t)mm * m + mm] -= reg;
      for (auto &h : Hh) h *= -1;
      if (has_nan(Hh)) throw std::runtime_error("H has NaN");
    }
    {
      vector<Real> d2(m, 0);
      for (int a = 0; a < m; a++) {
        Real s = 0;
        for (int k = 0; k < m; k++) s += J[(size_t)a * m + k] * dalpha[k];
        d2[a] = -s;
      }
      dalpha = d2;
    }
    if (has_nan(dalpha)) throw std::runtime_error("dalpha has NaN");
    dalpha[0] += reg * alpha[0];
    ll -= 0.5 * reg * alpha[0] * alpha[0];
    for (int mm = 1; mm < (K - 1); mm++) {
      dalpha[mm] += reg * alpha[mm];
      ll -= 0.5 * reg * alpha[mm] * alpha[mm];
    }
    return -ll;
  }
  void find_minimum(vector<Real> &alpha_hat) {  // :289-357
    int max_iter = 10000;
    Real epsilon = 1e-5, epsilon_rel = 1e-5, delta = 1e-5;
    const int past = 3;
    Real history[past] = {0, 0, 0};
    int m = n();
    vector<Real> alpha_new(alpha_hat), dalpha(alpha_hat), direction(alpha_hat);
    Real ll_current = 0;
    bool first = true;
    int i = 0;
    while (true) {
      if (first) ll_current = eval(alpha_hat, dalpha, &H);
      {
        Real alpha2 = norm2(alpha_hat), dalpha2 = norm2(dalpha);
        if (dalpha2 < epsilon || dalpha2 < epsilon_rel * alpha2) break;
      }
      {
        vector<Real> L;
        cholesky_lower(H, m, L);
        direction = dalpha;
        llt_solve(L, m, direction);
        for (auto &d : direction) d = -d;
      }
      Real step_size = 1;
      int lsc = 0;
      while (true) {
        for (int k = 0; k < m; k++) alpha_new[k] = alpha_hat[k] + step_size * direction[k];
        Real ll_new;
        try {
          ll_new = eval(alpha_new, dalpha, &H);
        } catch (std::runtime_error &) {
          step_size /= 2;
          if (++lsc > 1000) break;  // the reference can spin forever on a persistent NaN; bail out
          continue;
        }
        if (ll_new >= (ll_current * (1 + delta))) {
          step_size /= 2;
        } else {
          alpha_hat = alpha_new;
          ll_current = ll_new;
          break;
        }
        if (++lsc > 1000) break;
      }
      first = false;
      if (i >= past) {
        Real past_loss = history[i % past];
        if (std::abs(past_loss - ll_current) <= delta * std::max(std::max(std::abs(ll_current), std::abs(past_loss)), Real(1)))
          break;
      }
      history[i % past] = ll_current;
      i++;
      if (i >= max_iter) break;
    }
    if (i == max_iter) throw std::runtime_error("Failed to converge. See fail-log.txt");
  }
  void start_sample() {  // :274-279
    vector<Real> alpha_hat(K - 1, 0);
    find_minimum(alpha_hat);
    alpha_now = alpha_hat;
    alpha_to_gamma(gamma_now, alpha_now);
  }
  bool step() {  // :359-387
    vector<Real> alpha_hat = alpha_now;
    vector<Real> gamma(alpha_hat);
    find_minimum(alpha_hat);
    vector<Real> alpha_candidate = sample_mvt(H, nu);
    for (int k = 0; k < n(); k++) alpha_candidate[k] += alpha_hat[k];
    Real ll_candidate, ll_old;
    try {
      ll_candidate = -eval(alpha_candidate, gamma, nullptr);
      ll_old = -eval(alpha_now, gamma, nullptr);
    } catch (std::runtime_error &) {
      return false;
    }
    Real lpc = log_p_mvt(H, alpha_hat, nu, alpha_candidate);
    Real lpo = log_p_mvt(H, alpha_hat, nu, alpha_now);
    Real test_ratio = std::exp(ll_candidate - lpc - ll_old + lpo);
    Real u = std::uniform_real_distribution<Real>{0, 1}(rng);
    if (u < test_ratio) {
      alpha_now = alpha_candidate;
      alpha_to_gamma(gamma_now, alpha_now);
      accept_count++;
      return true;
    }
    return false;
  }
  // exact latent draws: the group's rows in the reference's order and the targets
  const vector<size_t> *host_rows = nullptr;
  const vector<Real> *host_y = nullptr;
  int64_t host_n = 0;
  bool exact_dev = false;         // latent mode "exact": the draws are made on the device from its own stream
  int64_t exact_fallbacks = 0;    // ... draws that took the sequential loop instead (a window of the parallel evaluation missed)
  void sample_z_given_cutpoint(uint64_t seed, uint64_t draw) {  // :238-272
    if (exact_dev) {
      int32_t st = 0;
      ck(ctx, mfm_oprobit_sample_z_exact(ctx, group, gamma_now.data(), &st));
      if (st == 0) return;
      exact_fallbacks++;  // nothing was drawn or consumed: the same draws row after row, below, from the same stream position
      warn_sequential_latent(st);
    } else if (!host_rows) {  // per-row Philox streams on the device
      ck(ctx, mfm_oprobit_sample_z(ctx, group, gamma_now.data(), seed, draw));
      return;
    }
    vector<Real> e((size_t)host_n);
    ck(ctx, mfm_get_e(ctx, e.data()));  // the scores (x_ of the reference)
    const Real deviation = 1;
    for (size_t t : *host_rows) {
      const int c = (int)(*host_y)[t];
      const Real pred = e[t];
      Real z_new;
      if (c == 0)
        z_new = deviation * host_tn_right(rng, (gamma_now[c] - pred) / deviation) + pred;
      else if (c == K - 1)
        z_new = deviation * host_tn_left(rng, (gamma_now[K - 2] - pred) / deviation) + pred;
      else
        z_new = deviation * host_tn_twoside(rng, (gamma_now[c - 1] - pred) / deviation, (gamma_now[c] - pred) / deviation) + pred;
      e[t] -= z_new;
    }
    ck(ctx, mfm_set_e(ctx, e.data()));
    if (exact_dev) rng.dev->commit();
  }
};

// ---- GibbsFMTrainer (BaseFMTrainer.hpp + FMTrainer.hpp) on the device path ---------------------------
// MFM_SETUP_TIMING=1: wall time of the host-side setup stages (stderr)
struct SetupLap {
  const char *who;
  bool on;
  std::chrono::steady_clock::time_point t;
  explicit SetupLap(const char *w) : who(w), on(std::getenv("MFM_SETUP_TIMING") != nullptr), t(std::chrono::steady_clock::now()) {}
  void operator()(const char *what) {
    if (!on) return;
    const auto n = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[%s] %-44s %.3f s\n", who, what, std::chrono::duration<double>(n - t).count());
    t = n;
  }
};

// what myfm_amd.distributed.connect_peers needs of a training context (GibbsSession has the same four methods): handed to the
// `peer_connect` callable of create_train_fm_sharded between mfm_finalize and the first iteration
struct PeerHandle {
  mfm_ctx *ctx = nullptr;
  py::tuple peer_info() {
    int32_t pending = 0;
    void *sum = nullptr, *flag = nullptr;
    int64_t sb = 0, fb = 0;
    ck(ctx, mfm_peer_info(ctx, &pending, &sum, &flag, &sb, &fb));
    return py::make_tuple(pending != 0, (uintptr_t)sum, (uintptr_t)flag, sb, fb);
  }
  py::bytes peer_export() {
    char h[256];
    ck(ctx, mfm_peer_export(ctx, h));
    return py::bytes(h, 256);
  }
  void peer_import(int world, int rank, const std::string &all) {
    if ((int)all.size() != world * 256) throw std::invalid_argument("peer_import: 256 bytes per rank");
    ck(ctx, mfm_peer_import(ctx, world, rank, all.data()));
  }
  void peer_drop() { ck(ctx, mfm_peer_drop(ctx)); }
};

struct FMTrainer {
  mfm_ctx *ctx = nullptr;
  int64_t N = 0, D0 = 0;
  size_t dim_all = 0;
  FMLearningConfig cfg;
  int random_seed;
  std::mt19937 gen_;
  std::mt19937 gen_mh_;  // cutpoint Metropolis draws once gen_ lives on the device
  vector<Real> y;
  vector<Real> n_in_group;
  vector<OprobitSampler> cutpoint_sampler;
  uint64_t latent_draws = 0;  // Philox draw index of the device-side truncated-normal draws
  std::unique_ptr<DeviceStreamWindow> dwin;  // latent mode "exact": the host's draws from the device stream
  int64_t exact_fallbacks = 0;               // ... classification draws that took the sequential loop
  int K = -1;
  vector<Real> zbuf;
  // device-side random stream (include/myfm_hip.h "device-side random stream"): the generator is handed
  // to the GPU after initialize_weight; the host then only scales the pre-drawn unit variates.
  bool device_rng = false;
  vector<Real> hv;
  size_t hv_pos = 0;
  // row-sharded multi-GPU mode: this process holds rows [row_offset, row_offset + N) of
  // N_total; `allreduce(ptr, count)` sums device doubles in place over the ranks
  int64_t N_total = 0, row_offset = 0;
  int shard_rank = 0, shard_world = 1;
  std::string comm_id;  // 128-byte RCCL unique id: the library calls ncclAllReduce itself (mfm_comm_init)
  py::object allreduce;
  py::object peer_connect;  // callable(PeerHandle) or none
  uint64_t stream_ptr = 0;
  vector<int32_t> main_levels;  // level schedule of the GLOBAL main table (sharded mode)
  static int allreduce_trampoline(void *user, void *buf, int64_t count) {
    py::gil_scoped_acquire gil;  // (GibbsSession.step runs without the GIL: sessions of several threads run side by side)
    try {
      (*static_cast<py::object *>(user))((uintptr_t)buf, count);
      return 0;
    } catch (py::error_already_set &e) {
      e.restore();
      PyErr_Print();
      return 1;
    }
  }

  // BaseFMTrainer.hpp:58-105
  FMTrainer(const py::object &Xo, const py::object &relso, const py::object &yo, int random_seed, FMLearningConfig config)
      : cfg(std::move(config)), random_seed(random_seed), gen_(random_seed) {
    SetupLap lap("FMTrainer");
    X_ = csr_view_from_py(Xo);
    lap("view of X");
    rels_ = relations_from_py(relso);
    dim_all = check_row_consistency_return_column(X_, rels_);
    lap("relations");
    y = np_to_vec(yo);
    N = X_.rows;
    N_total = N;
    D0 = X_.cols;
    if (X_.rows != (int64_t)y.size()) {
      std::ostringstream ss;
      ss << "Shape mismatch: X has size " << X_.rows << " and y has size " << y.size();
      throw std::runtime_error(ss.str());
    }
    if (cfg.task_type == TaskType::ORDERED) {
      const size_t rows = (size_t)X_.rows;
      // (the usual case -- ONE group listing every row in order -- is checked by threads without the marker array)
      bool identity = cfg.cutpoint_groups.size() == 1 && cfg.cutpoint_groups[0].second.size() == rows;
      if (identity) {
        const size_t *idx = cfg.cutpoint_groups[0].second.data();
        std::atomic<int> off(0);
        host_ranges((int64_t)rows, [&](int64_t lo, int64_t hi) {
          bool b = false;
          for (int64_t k = lo; k < hi; k++) b |= idx[k] != (size_t)k;
          if (b) off = 1;
        });
        identity = !off;
      }
      vector<bool> existence(identity ? 0 : rows, false);
      if (!identity)
      for (auto &gc : cfg.cutpoint_groups)
        for (size_t k : gc.second) {
          if (k >= rows) throw std::invalid_argument("out of range for cutpoint group config.");
          if (existence[k]) {
            std::stringstream ss;
            ss << "index " << k << " overlapping in cutpoint config.";
            throw std::invalid_argument(ss.str());
          }
          existence[k] = true;
        }
      for (size_t i = 0; i < rows && !identity; i++)
        if (!existence[i]) {
          std::stringstream ss;
          ss << "cutpoint group not specified for " << i << ".";
          throw std::invalid_argument(ss.str());
        }
    }
    lap("y, cutpoint group checks");
    if (cfg.group_index.size() != dim_all) throw std::out_of_range("group_index does not cover all features");  // .at()
    n_in_group.assign(cfg.n_groups, 0);
    for (auto g : cfg.group_index) n_in_group[g] += 1;
  }
  ~FMTrainer() {
    if (ctx) mfm_destroy(ctx);
  }
  FMTrainer(const FMTrainer &) = delete;
  bool comm_active() const { return !comm_id.empty() || (allreduce.ptr() != nullptr && !allreduce.is_none()); }
  // before the device is built: the exact latent draws walk ONE stream over ALL rows in order -- a row-sharded fit keeps the
  // per-row Philox streams (which do not depend on the sharding)
  void resolve_latent_mode() {
    if (cfg.latent_mode == 2 && comm_active()) cfg.latent_mode = 0;
    if (!cfg.latent_order.empty()) {
      if ((int64_t)cfg.latent_order.size() != N) throw std::invalid_argument("latent row order must list every row once");
      vector<bool> seen((size_t)N, false);
      for (auto r : cfg.latent_order) {
        if (r < 0 || r >= N || seen[(size_t)r]) throw std::invalid_argument("latent row order must list every row once");
        seen[(size_t)r] = true;
      }
    }
  }

  // BaseFMTrainer.hpp:107-115
  FM create_FM(int rank, Real init_std) {
    FM fm(rank);
    fm.initialize_weight((int64_t)dim_all, init_std, gen_);
    return fm;
  }
  Hyper create_Hyper(size_t rank) { return Hyper(rank, cfg.n_groups); }

  void build_device(int rank) {
    if (ctx) return;
    K = rank;
    SetupLap lap("build_device");
    if (!cfg.host_rng())  // (the device generator's jump polynomials: beside everything below)
      (void)mfm_rng_prepare((int64_t)cfg.group_index.size(), (int32_t)rank, (int32_t)cfg.n_groups);
    int code = mfm_create(selected_device(), &ctx);
    if (code != MFM_OK) throw_code(code, mfm_global_error());
    if (stream_ptr) ck(ctx, mfm_set_stream(ctx, (void *)stream_ptr));
    if (!comm_id.empty()) {
      if (comm_id.size() != 128) throw std::invalid_argument("comm_id must be the 128 bytes of comm_unique_id()");
      ck(ctx, mfm_comm_init(ctx, comm_id.data(), shard_rank, shard_world));
      ck(ctx, mfm_set_row_offset(ctx, row_offset));
    } else if (allreduce.ptr() != nullptr && !allreduce.is_none()) {
      ck(ctx, mfm_set_allreduce(ctx, &FMTrainer::allreduce_trampoline, &allreduce));
      ck(ctx, mfm_set_shard(ctx, shard_rank, shard_world));
      ck(ctx, mfm_set_row_offset(ctx, row_offset));
    }
    lap("mfm_create (HIP runtime, stream, communicator)");
    if (!main_levels.empty()) ck(ctx, mfm_set_main_levels(ctx, main_levels.data(), (int64_t)main_levels.size()));
    ck(ctx, mfm_set_main(ctx, X_.rows, X_.cols, X_.indptr.data(), X_.indices.data(), X_.data.data(), y.data()));
    X_.release();  // (the library holds its own copy now)
    lap("mfm_set_main");
    for (auto &r : rels_) {
      ck(ctx, mfm_add_block(ctx, r->X.rows, r->X.cols, r->X.indptr.data(), r->X.indices.data(), r->X.data.data(), r->map64()));
    }
    lap("mfm_add_block (all)");
    vector<int32_t> gi(cfg.group_index.begin(), cfg.group_index.end());
    ck(ctx, mfm_set_groups(ctx, gi.data(), (int64_t)gi.size(), (int32_t)cfg.n_groups));
    lap("mfm_set_groups");
    ck(ctx, mfm_finalize(ctx, rank));
    lap("mfm_finalize");
    if (!cfg.latent_order.empty() && cfg.task_type == TaskType::CLASSIFICATION)
      ck(ctx, mfm_set_latent_order(ctx, cfg.latent_order.data(), (int64_t)cfg.latent_order.size()));
    // regression: update_e recomputes the residual after every update_V (:494), nothing reads it in between
    if (cfg.task_type == TaskType::REGRESSION && !std::getenv("MYFM_AMD_KEEP_RESIDUAL")) ck(ctx, mfm_set_residual_policy(ctx, 1));
  }
  void upload(const FM &fm) { ck(ctx, mfm_set_state(ctx, fm.w0, fm.w.data(), fm.V.data())); }
  void download(FM &fm) {
    fm.w.resize(dim_all);
    fm.V.resize(dim_all * (size_t)fm.n_factors);
    ck(ctx, mfm_get_state(ctx, &fm.w0, fm.w.data(), fm.V.data()));
  }

  // FMTrainer.hpp:122-125: a fresh normal_distribution per draw
  Real next_normal() { return device_rng ? hv.at(hv_pos++) : std::normal_distribution<Real>(0, 1)(gen_); }
  Real sample_normal(Real quad, Real first) { return (first / quad) + next_normal() / std::sqrt(quad); }
  // gamma_distribution(shape, scale)(gen_): libstdc++ returns (unit variate) * scale
  Real sample_gamma(Real shape, Real scale) {
    if (device_rng) return hv.at(hv_pos++) * scale;
    return std::gamma_distribution<Real>(shape, scale)(gen_);
  }
  void draw_normals(Real *out, size_t n) {
    for (size_t i = 0; i < n; i++) out[i] = std::normal_distribution<Real>(0, 1)(gen_);
  }

  // Hands gen_ to the device and registers one iteration's draw order (SURVEY 8a "RNG draw order"): the
  // hyper-parameter and w / V draws consume the generator in a state-independent order for every task.
  // The latent z of classification / ordered probit use per-row Philox streams on the device, and the
  // cutpoint sampler's few Metropolis draws (OProbitSampler.hpp:55-72, :378) come from a second host
  // generator (gen_mh_, forked from gen_ before the hand-over): for those tasks parity with the
  // reference is distributional anyway.
  void start_device_rng(int Kf) {
    if (cfg.host_rng() || device_rng) return;
    std::ostringstream os;
    os << gen_;
    std::istringstream is(os.str());
    vector<uint32_t> st(625);
    for (auto &v : st) {
      unsigned long x;
      is >> x;
      v = (uint32_t)x;
    }
    ck(ctx, mfm_rng_seed_mt19937(ctx, st.data(), (int32_t)st[624]));
    const size_t G = cfg.n_groups;
    vector<mfm_rng_op> ops;
    int64_t n = 0;
    auto gamma = [&](Real shape) { ops.push_back(mfm_rng_op{MFM_RNG_GAMMA, 0, 1, n++, shape}); };
    auto normals_hv = [&](int64_t c) {
      ops.push_back(mfm_rng_op{MFM_RNG_NORMALS, 0, c, n, 0.0});
      n += c;
    };
    if (cfg.task_type == TaskType::REGRESSION) gamma((cfg.alpha_0 + N_total) / 2);  // update_alpha
    if (cfg.fit_w0) normals_hv(1);                                              // update_w0
    for (size_t g = 0; g < G; g++) gamma((cfg.alpha_0 + n_in_group[g]) / 2);     // update_lambda_w
    normals_hv((int64_t)G);                                                     // update_mu_w
    if (cfg.fit_linear && dim_all) ops.push_back(mfm_rng_op{MFM_RNG_NORMALS, 1, (int64_t)dim_all, 0, 0.0});  // update_w
    if (Kf > 0) {
      for (int f = 0; f < Kf; f++)
        for (size_t g = 0; g < G; g++) gamma((cfg.alpha_0 + n_in_group[g]) / 2);  // update_lambda_V
      normals_hv((int64_t)G * Kf);                                                // update_mu_V
      if (dim_all) ops.push_back(mfm_rng_op{MFM_RNG_NORMALS, 2, (int64_t)dim_all * Kf, 0, 0.0});  // update_V
    }
    // latent mode "exact": the latent draws consume the same stream after every set (about 1.4 quads of 4 outputs per row)
    if (cfg.exact_dev()) ops.push_back(mfm_rng_op{MFM_RNG_LATENT, 0, N, 0, 0.0});
    ck(ctx, mfm_rng_set_program(ctx, ops.data(), (int32_t)ops.size()));
    hv.assign((size_t)n, 0);
    device_rng = true;
    if (cfg.exact_dev()) return;     // (the first set starts where initialize_e's draws end: first_set())
    ck(ctx, mfm_rng_prefetch(ctx));  // the first iteration's set and the second's (update_all keeps two ahead)
    ck(ctx, mfm_rng_prefetch(ctx));
  }
  // latent mode "exact": ONE set in flight, requested where the stream stands after the state-dependent draws
  void first_set() {
    if (cfg.exact_dev() && device_rng) ck(ctx, mfm_rng_prefetch(ctx));
  }

  void initialize_hyper(Hyper &hyper) {  // FMTrainer.hpp:89-97
    hyper.alpha = 1;
    std::fill(hyper.mu_w.begin(), hyper.mu_w.end(), 0);
    std::fill(hyper.lambda_w.begin(), hyper.lambda_w.end(), 1e-5);
    std::fill(hyper.mu_V.begin(), hyper.mu_V.end(), 0);
    std::fill(hyper.lambda_V.begin(), hyper.lambda_V.end(), 1e-5);
  }

  void initialize_e(FM &fm) {  // FMTrainer.hpp:99-119
    if (cfg.task_type == TaskType::ORDERED) {
      gen_mh_.seed((uint32_t)random_seed ^ 0x9E3779B9u);
      ck(ctx, mfm_score_train(ctx));
      int i = 0;
      cutpoint_sampler.clear();
      cutpoint_sampler.reserve(cfg.cutpoint_groups.size());
      for (auto &c : cfg.cutpoint_groups) {
        // label validation of the OprobitSampler ctor (OProbitSampler.hpp:33-46)
        for (auto r : c.second) {
          int y_label = (int)y[r];
          if (std::abs(y_label - y[r]) > 1e-3) throw std::invalid_argument("y has a floating-point element.");
          if (y_label < 0) throw std::invalid_argument("y has a negative element.");
          if (y_label >= (int)c.first) {
            std::stringstream ss;
            ss << "y[ " << r << "] is greater than " << (c.first - 1) << ".";
            throw std::invalid_argument(ss.str());
          }
        }
        fm.cutpoints.emplace_back(c.first - 1);
        int32_t g = 0;
        bool all_rows = (int64_t)c.second.size() == N;
        if (all_rows)
          for (size_t k = 0; k < c.second.size(); k++)
            if (c.second[k] != k) {
              all_rows = false;
              break;
            }
        vector<int64_t> rows;
        if (!all_rows) rows.assign(c.second.begin(), c.second.end());
        ck(ctx, mfm_oprobit_add_group(ctx, (int32_t)c.first, all_rows ? nullptr : rows.data(), (int64_t)c.second.size(), &g));
        StreamEngine eng;
        if (cfg.exact_dev()) {  // the cutpoint sampler's own draws come from the device stream too (between two sets)
          if (!dwin) dwin.reset(new DeviceStreamWindow(ctx));
          eng.dev = dwin.get();
        } else {
          eng.host = cfg.host_rng() ? &gen_ : &gen_mh_;
        }
        cutpoint_sampler.emplace_back(ctx, g, (int)c.first, eng, cfg.reg_0, cfg.nu_oprobit);
        if (cfg.host_rng() || cfg.exact_dev()) {  // exact latent draws: the rows in the reference's order (host loop / its fall-back)
          cutpoint_sampler[i].host_rows = &c.second;
          cutpoint_sampler[i].host_y = &y;
          cutpoint_sampler[i].host_n = N;
          cutpoint_sampler[i].exact_dev = cfg.exact_dev();
        }
        cutpoint_sampler[i].start_sample();
        OprobitSampler::alpha_to_gamma(fm.cutpoints[i], cutpoint_sampler[i].alpha_now);
        cutpoint_sampler[i].sample_z_given_cutpoint((uint64_t)random_seed, latent_draws++);
        i++;
      }
      return;
    }
    ck(ctx, mfm_update_e_regression(ctx));  // e = score - y
  }

  // ---- one Gibbs iteration, BaseFMTrainer.hpp:135-152 ----
  // MYFM_AMD_HOST_TIMELINE=1: where the host thread spends an iteration (mean microseconds per stage, every 100 iterations)
  struct HostTimeline {
    bool on = std::getenv("MYFM_AMD_HOST_TIMELINE") != nullptr;
    std::chrono::steady_clock::time_point t;
    double acc[8] = {0};
    int n = 0;
    void start() {
      if (on) t = std::chrono::steady_clock::now();
    }
    void mark(int k) {
      if (!on) return;
      const auto now = std::chrono::steady_clock::now();
      acc[k] += std::chrono::duration<double, std::micro>(now - t).count();
      t = now;
    }
    void end() {
      if (!on || ++n % 100) return;
      static const char *names[8] = {"rng acquire", "hyper_stats (sync)", "host draws", "sweep launch", "rng prefetch", "update_e", "", ""};
      std::fprintf(stderr, "[host timeline]");
      for (int k = 0; k < 6; k++) std::fprintf(stderr, " %s %.1f us |", names[k], acc[k] / 100.0);
      std::fprintf(stderr, "\n");
      for (auto &a : acc) a = 0;
    }
  } htl;

  void update_all(FM &fm, Hyper &hyper) {
    const size_t G = cfg.n_groups;
    const int Kf = fm.n_factors;
    htl.start();
    // MYFM_AMD_DEVICE_HYPERS=1, regression on the persistent sweep: the whole iteration is enqueued at once and its
    // hyper-parameters are drawn on the device (mfm_regression_iteration: the same arithmetic on the same variates as below, the
    // chains are bit-identical) -- no read-back / host draws / upload between update_e and the next launch. Opt-in: it does not
    // shorten the iteration (config 3: 335-338 against 338 it/s). What fills the 0.29 ms between two launches is the random
    // stream, not the host: the evaluation of a set's 2.6 M sweep normals runs starved beside the launch on the CUs it leaves free,
    // and the set's single-workgroup draw kernels (0.26 ms in a row) then take the whole gap.
    static const bool host_hypers = [] {  // (default since round 6: the device form; MYFM_AMD_DEVICE_HYPERS=0 keeps the host in the loop)
      const char *e = std::getenv("MYFM_AMD_DEVICE_HYPERS");
      return e != nullptr && std::atoi(e) == 0;
    }();
    if (!host_hypers && device_rng && cfg.task_type == TaskType::REGRESSION && cfg.fit_linear && dim_all && Kf > 0 && !comm_active() &&
        mfm_regression_iteration_ready(ctx) == 1) {
      mfm_hyper_prior pr;
      pr.alpha_0 = cfg.alpha_0;
      pr.beta_0 = cfg.beta_0;
      pr.gamma_0 = cfg.gamma_0;
      pr.mu_0 = cfg.mu_0;
      pr.reg_0 = cfg.reg_0;
      pr.n_total = (double)N_total;
      pr.fit_w0 = cfg.fit_w0 ? 1 : 0;
      pr.reserved = 0;
      Real w0 = cfg.fit_w0 ? fm.w0 : 0;
      ck(ctx, mfm_regression_iteration(ctx, &pr, n_in_group.data(), &hyper.alpha, &w0, hyper.lambda_w.data(), hyper.mu_w.data(),
                                       hyper.lambda_V.data(), hyper.mu_V.data()));
      fm.w0 = w0;
      g_device_hyper_iterations++;
      htl.mark(3);
      htl.end();
      return;
    }
    // every reduction the hyper-parameter updates need, one host synchronisation: sum e / sum e^2 (update_alpha,
    // FMTrainer.hpp:127-145, update_w0 :218-229) and the group sums of w and V (:150-216) -- the latter are taken
    // before update_w / update_V touch w / V, which is where the reference takes them too
    Real sum_e = 0, sum_e2 = 0, e_shift = 0;
    const bool need_alpha = cfg.task_type == TaskType::REGRESSION;
    vector<Real> sum(G), ssd(G), sumV(G * std::max(Kf, 1)), ssdV(G * std::max(Kf, 1));
    ck(ctx, mfm_hyper_stats(ctx, (need_alpha || cfg.fit_w0) ? 1 : 0, hyper.mu_w.data(), hyper.mu_V.data(), &sum_e, &sum_e2,
                            sum.data(), ssd.data(), sumV.data(), ssdV.data()));
    htl.mark(1);
    // this iteration's variates -- asked for AFTER the statistics: those do not need them, and the set is produced behind the
    // previous launch on a side stream (waiting for it first kept the statistics kernels, 0.12 ms with their read-back, off the
    // GPU until the set was there)
    if (device_rng) {
      ck(ctx, mfm_rng_acquire(ctx, hv.data(), (int64_t)hv.size()));
      hv_pos = 0;
    }
    htl.mark(0);
    if (need_alpha) {
      Real exponent = (cfg.alpha_0 + N_total) / 2;
      Real variance = (cfg.beta_0 + sum_e2) / 2;
      hyper.alpha = sample_gamma(exponent, 1 / variance);
    } else {
      hyper.alpha = 1;
    }
    if (!cfg.fit_w0) {
      fm.w0 = 0;
    } else {
      Real w0_lin_term = hyper.alpha * (N_total * fm.w0 - sum_e);  // sum(w0 - e) over all (ranks') rows
      Real w0_quad_term = hyper.alpha * N_total + cfg.reg_0;
      Real w0_new = sample_normal(w0_quad_term, w0_lin_term);
      e_shift = w0_new - fm.w0;  // e += w0' - w0 (:226): applied by the sweep that follows (or right below)
      fm.w0 = w0_new;
    }
    // update_w and update_V as one device call when both run on the device stream: the draws of lambda_V / mu_V that the
    // reference makes between them (BaseFMTrainer.hpp:143-148) read neither w nor e
    const bool fuse_wV = device_rng && cfg.fit_linear && dim_all && Kf > 0;
    if (e_shift != 0 && !fuse_wV) {
      ck(ctx, mfm_shift_e(ctx, e_shift));
      e_shift = 0;
    }
    ck(ctx, mfm_set_w0(ctx, fm.w0));
    // update_lambda_w / update_mu_w (:150-200)
    for (size_t g = 0; g < G; g++) {
      Real alpha = cfg.alpha_0 + n_in_group[g];
      Real beta = cfg.beta_0 + ssd[g];
      hyper.lambda_w[g] = sample_gamma(alpha / 2, 2 / beta);
    }
    for (size_t g = 0; g < G; g++) {
      Real square = hyper.lambda_w[g] * (cfg.gamma_0 + n_in_group[g]);
      Real linear = cfg.gamma_0 * cfg.mu_0 + sum[g];
      linear *= hyper.lambda_w[g];
      hyper.mu_w[g] = sample_normal(square, linear);
    }
    // update_w (:231-314)
    if (!cfg.fit_linear) {
      ck(ctx, mfm_zero_w(ctx));
    } else {
      if (fuse_wV) {
        // (below, with update_V)
      } else if (device_rng && dim_all) {
        ck(ctx, mfm_sweep_w(ctx, hyper.alpha, hyper.lambda_w.data(), hyper.mu_w.data(), nullptr));
      } else {
        zbuf.resize(std::max<size_t>(dim_all, 1));
        draw_normals(zbuf.data(), dim_all);
        ck(ctx, mfm_sweep_w(ctx, hyper.alpha, hyper.lambda_w.data(), hyper.mu_w.data(), zbuf.data()));
      }
    }
    if (Kf > 0) {
      // update_lambda_V / update_mu_V (:202-216): factor outer, group inner
      for (int f = 0; f < Kf; f++)
        for (size_t g = 0; g < G; g++) {
          Real alpha = cfg.alpha_0 + n_in_group[g];
          Real beta = cfg.beta_0 + ssdV[(size_t)f * G + g];
          hyper.lambda_V[(size_t)f * G + g] = sample_gamma(alpha / 2, 2 / beta);
        }
      for (int f = 0; f < Kf; f++)
        for (size_t g = 0; g < G; g++) {
          Real lam = hyper.lambda_V[(size_t)f * G + g];
          Real square = lam * (cfg.gamma_0 + n_in_group[g]);
          Real linear = cfg.gamma_0 * cfg.mu_0 + sumV[(size_t)f * G + g];
          linear *= lam;
          hyper.mu_V[(size_t)f * G + g] = sample_normal(square, linear);
        }
      // update_V (:316-486)
      htl.mark(2);
      if (fuse_wV) {
        ck(ctx, mfm_sweep_wV(ctx, hyper.alpha, e_shift, hyper.lambda_w.data(), hyper.mu_w.data(), nullptr, 0, Kf,
                             hyper.lambda_V.data(), hyper.mu_V.data(), nullptr));
      } else if (device_rng && dim_all) {
        ck(ctx, mfm_sweep_V(ctx, 0, Kf, hyper.alpha, hyper.lambda_V.data(), hyper.mu_V.data(), nullptr));
      } else {
        zbuf.resize(dim_all * (size_t)Kf);
        draw_normals(zbuf.data(), dim_all * (size_t)Kf);
        ck(ctx, mfm_sweep_V(ctx, 0, Kf, hyper.alpha, hyper.lambda_V.data(), hyper.mu_V.data(), zbuf.data()));
      }
    }
    // the variates of the iteration after the next: generated on the side stream from the end of this iteration's latent
    // sweep on (the persistent sweep leaves no CU to anything else), next to update_e and the start of the next iteration
    htl.mark(3);
    const bool exact = cfg.exact_dev() && device_rng;
    if (device_rng && !exact) ck(ctx, mfm_rng_prefetch(ctx));
    htl.mark(4);
    // update_e (:493-522)
    if (cfg.task_type == TaskType::REGRESSION) {
      ck(ctx, mfm_update_e_regression(ctx));
    } else if (cfg.task_type == TaskType::CLASSIFICATION) {
      int32_t st = 1;
      if (exact) {  // FMTrainer.hpp:498-512 on the device stream, evaluated in parallel (csrc/mfm_latent.hip)
        ck(ctx, mfm_update_e_classification_exact(ctx, &st));
        if (st != 0) {  // (nothing drawn or consumed, e holds the scores: row by row below)
          exact_fallbacks++;
          warn_sequential_latent(st);
        }
      }
      if (exact && st == 0) {
      } else if (cfg.host_rng() || exact) {  // FMTrainer.hpp:498-512 row by row, on the trainer's generator / the device stream
        if (!exact) ck(ctx, mfm_score_train(ctx));
        vector<Real> e((size_t)N);
        ck(ctx, mfm_get_e(ctx, e.data()));
        StreamEngine eng;
        if (exact) {
          if (!dwin) dwin.reset(new DeviceStreamWindow(ctx));
          eng.dev = dwin.get();
        } else {
          eng.host = &gen_;
        }
        const bool ordered_rows = !cfg.latent_order.empty();
        for (int64_t i = 0; i < N; i++) {
          const int64_t t = ordered_rows ? cfg.latent_order[(size_t)i] : i;
          const Real pred = e[(size_t)t];
          const Real n = y[(size_t)t] > 0 ? host_tn_left(eng, pred, (Real)1, (Real)0) : host_tn_right(eng, pred, (Real)1, (Real)0);
          e[(size_t)t] -= n;
        }
        ck(ctx, mfm_set_e(ctx, e.data()));
        if (exact) dwin->commit();
      } else {
        ck(ctx, mfm_update_e_classification(ctx, (uint64_t)random_seed, latent_draws++));
      }
    } else {
      ck(ctx, mfm_score_train(ctx));
      int i = 0;
      for (auto &s : cutpoint_sampler) {
        s.step();
        if (exact) dwin->commit();  // (the Metropolis draws came from the device stream: it moves past them)
        OprobitSampler::alpha_to_gamma(fm.cutpoints[i], s.alpha_now);
        s.sample_z_given_cutpoint((uint64_t)random_seed, latent_draws++);
        i++;
      }
    }
    if (exact) ck(ctx, mfm_rng_prefetch(ctx));  // the next iteration's set, from where the latent draws ended
    htl.mark(5);
    htl.end();
  }

  // FMTrainer.hpp:56-87
  std::pair<Predictor, LearningHistory> learn_with_callback(
      FM &fm, Hyper &hyper, const std::function<bool(int, FM *, Hyper *, LearningHistory *)> &cb) {
    std::pair<Predictor, LearningHistory> result{Predictor((size_t)fm.n_factors, dim_all, cfg.task_type), LearningHistory()};
    SetupLap lap("learn_with_callback");
    resolve_latent_mode();
    build_device(fm.n_factors);
    lap("build_device (set_main, blocks, finalize)");
    if (peer_connect.ptr() != nullptr && !peer_connect.is_none()) {  // row-sharded persistent sweep: the ranks' exchange buffers
      PeerHandle h;
      h.ctx = ctx;
      peer_connect(py::cast(h));
    }
    upload(fm);
    initialize_hyper(hyper);
    if (cfg.exact_dev()) start_device_rng(fm.n_factors);  // (initialize_e's latent draws already come from the device stream)
    initialize_e(fm);
    lap("state upload + initialize_e");
    start_device_rng(fm.n_factors);
    first_set();
    lap("device RNG hand-over");
    fm.fetch = [this](FM &f) { this->download(f); };
    fm.live_ctx = ctx;
    result.first.samples.reserve((size_t)cfg.n_kept_samples);
    // kept samples stay on the GPU (device-to-device copy on the training stream, no host transfer inside the loop);
    // MYFM_AMD_HOST_SAMPLES=1 or a store that does not fit: plain host copies as before
    std::shared_ptr<DeviceStore> store;
    // (row-sharded fits too: the model is replicated, every rank keeps its own copy of the samples on its device)
    if (cfg.n_kept_samples > 0 && !std::getenv("MYFM_AMD_HOST_SAMPLES"))
    {
      store = std::make_shared<DeviceStore>((int64_t)dim_all, fm.n_factors);
      if (mfm_store_reserve(store->st, cfg.n_kept_samples) != MFM_OK) store.reset();  // (does not fit: host copies)
    }
    for (int it = 0; it < cfg.n_iter; it++) {
      update_all(fm, hyper);
      fm.stale = true;  // w / V live on the device until somebody reads them
      if (cfg.n_iter <= (it + cfg.n_kept_samples)) {
        if (store && mfm_store_push_ctx(store->st, ctx) == MFM_OK) {
          FM k;
          k.n_factors = fm.n_factors;
          k.w0 = fm.w0;
          k.cutpoints = fm.cutpoints;
          k.initialized = fm.initialized;
          k.store = store;
          k.store_idx = store->size() - 1;
          k.stale = true;
          const size_t D = dim_all;
          k.fetch = [store, D](FM &f) {  // materialise on the host on first access
            f.w.resize(D);
            f.V.resize(D * (size_t)f.n_factors);
            int code = mfm_store_get(store->st, f.store_idx, &f.w0, f.w.data(), f.V.data());
            if (code != MFM_OK) throw_code(code, mfm_store_last_error(store->st));
          };
          result.first.samples.emplace_back(std::move(k));
        } else {
          store.reset();  // (out of device memory: fall back to host copies from here on)
          result.first.samples.emplace_back(fm.snapshot());
        }
      }
      result.second.hypers.emplace_back(hyper);
      bool should_stop = cb(it, &fm, &hyper, &(result.second));
      if (should_stop) break;
    }
    fm.ensure();
    fm.fetch = nullptr;
    fm.live_ctx = nullptr;
    fm.design_cache.reset();
    for (auto &cs : cutpoint_sampler) result.second.n_mh_accept.emplace_back(cs.accept_count);
    return result;
  }

 private:
  CsrView X_;
  Relations rels_;
};

// cpp_source/declare_module.hpp:30-45
std::pair<Predictor, LearningHistory> create_train_fm(size_t n_factor, Real init_std, const py::object &X,
                                                      const py::object &relations, const py::object &y, int random_seed,
                                                      FMLearningConfig &config,
                                                      std::function<bool(int, FM *, Hyper *, LearningHistory *)> cb) {
  SetupLap lap("create_train_fm");
  FMTrainer fm_trainer(X, relations, y, random_seed, config);
  lap("trainer (copy of X, relations, y)");
  auto fm = fm_trainer.create_FM((int)n_factor, init_std);
  lap("initialize_weight");
  auto hyper_param = fm_trainer.create_Hyper((size_t)fm.n_factors);
  return fm_trainer.learn_with_callback(fm, hyper_param, cb);
}

// create_train_fm over row shards (not part of the reference's surface): every rank passes its contiguous slice
// of the rows (X, y, every original_to_block), the level schedule of the GLOBAL main table, and either the RCCL id
// (comm_id, native all-reduce) or an all-reduce callable. Every rank returns the same Predictor / history.
std::pair<Predictor, LearningHistory> create_train_fm_sharded(size_t n_factor, Real init_std, const py::object &X,
                                                              const py::object &relations, const py::object &y, int random_seed,
                                                              FMLearningConfig &config,
                                                              std::function<bool(int, FM *, Hyper *, LearningHistory *)> cb,
                                                              int rank, int world, int64_t n_total_rows, int64_t row_offset,
                                                              const py::object &main_levels, const std::string &comm_id,
                                                              py::object allreduce, uint64_t stream, py::object peer_connect) {
  FMTrainer t(X, relations, y, random_seed, config);
  t.allreduce = allreduce;
  t.peer_connect = peer_connect;
  t.comm_id = comm_id;
  t.shard_rank = rank;
  t.shard_world = world;
  t.N_total = n_total_rows;
  t.row_offset = row_offset;
  t.stream_ptr = stream;
  auto lv = py::array_t<int32_t, py::array::c_style | py::array::forcecast>::ensure(main_levels);
  if (!lv) throw std::invalid_argument("main_levels must be an int32 array");
  t.main_levels.assign(lv.data(), lv.data() + lv.size());
  auto fm = t.create_FM((int)n_factor, init_std);
  auto hyper_param = t.create_Hyper((size_t)fm.n_factors);
  return t.learn_with_callback(fm, hyper_param, cb);
}

// A steppable training session: not part of the reference's surface; bench.py and the parity tests
// use it to time / inspect single Gibbs iterations of exactly the loop create_train_fm runs.
struct GibbsSession {
  std::unique_ptr<FMTrainer> trainer;
  FM fm;
  Hyper hyper;
  int it = 0;
  GibbsSession(size_t n_factor, Real init_std, const py::object &X, const py::object &relations, const py::object &y,
               int random_seed, FMLearningConfig &config, py::object allreduce, int64_t n_total_rows, int64_t row_offset,
               uint64_t stream, py::object main_levels, const std::string &comm_id, int rank, int world)
      : trainer(new FMTrainer(X, relations, y, random_seed, config)) {
    trainer->allreduce = allreduce;
    trainer->comm_id = comm_id;
    trainer->shard_rank = rank;
    trainer->shard_world = world;
    if (!main_levels.is_none()) {
      auto lv = py::array_t<int32_t, py::array::c_style | py::array::forcecast>::ensure(main_levels);
      if (!lv) throw std::invalid_argument("main_levels must be an int32 array");
      trainer->main_levels.assign(lv.data(), lv.data() + lv.size());
    }
    if (n_total_rows > 0) trainer->N_total = n_total_rows;
    trainer->row_offset = row_offset;
    trainer->stream_ptr = stream;
    SetupLap lap("GibbsSession");
    fm = trainer->create_FM((int)n_factor, init_std);
    hyper = trainer->create_Hyper((size_t)fm.n_factors);
    lap("create_FM / create_Hyper");
    trainer->resolve_latent_mode();
    trainer->build_device(fm.n_factors);
    lap("build_device (set_main, blocks, finalize)");
    trainer->upload(fm);
    trainer->initialize_hyper(hyper);
    if (trainer->cfg.exact_dev()) trainer->start_device_rng(fm.n_factors);
    trainer->initialize_e(fm);
    lap("state upload + initialize_e");
    trainer->start_device_rng(fm.n_factors);
    trainer->first_set();
    lap("device RNG hand-over");
    fm.fetch = [this](FM &f) { this->trainer->download(f); };
    fm.live_ctx = trainer->ctx;
  }
  void step() {
    trainer->update_all(fm, hyper);
    fm.stale = true;
    it++;
  }
  void synchronize() { ck(trainer->ctx, mfm_synchronize(trainer->ctx)); }
  py::array_t<double> residual() {
    py::array_t<double> e((py::ssize_t)trainer->N);
    ck(trainer->ctx, mfm_get_e(trainer->ctx, e.mutable_data()));
    return e;
  }
  void timing_enable(bool on) { ck(trainer->ctx, mfm_timing_enable(trainer->ctx, on ? 1 : 0)); }
  // only this kernel class is bracketed with events ("" : all)
  void timing_select(const std::string &name) {
    int idx = -1;
    if (!name.empty()) {
      for (int i = 0; i < mfm_timing_n_classes(); i++)
        if (name == mfm_timing_class_name(i)) idx = i;
      if (idx < 0) throw std::invalid_argument("unknown kernel class " + name);
    }
    ck(trainer->ctx, mfm_timing_select(trainer->ctx, idx));
  }
  void timing_reset() { ck(trainer->ctx, mfm_timing_reset(trainer->ctx)); }
  py::dict timing() {
    py::dict out;
    for (int c = 0; c < mfm_timing_n_classes(); c++) {
      double ms = 0, by = 0;
      int64_t n = 0;
      ck(trainer->ctx, mfm_timing_get(trainer->ctx, c, &ms, &n, &by));
      if (n) out[py::str(mfm_timing_class_name(c))] = py::make_tuple(ms, n, by);
    }
    return out;
  }
  py::tuple plan_info() {
    int64_t a = 0, b = 0;
    mfm_plan_info(trainer->ctx, &a, &b);
    return py::make_tuple(a, b);
  }
  // latent mode "exact": geometry of the last parallel draw and how many draws took the sequential loop instead
  py::dict latent_info() {
    int64_t v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    ck(trainer->ctx, mfm_latent_stats(trainer->ctx, v));
    int64_t fb = trainer->exact_fallbacks;
    for (auto &cs : trainer->cutpoint_sampler) fb += cs.exact_fallbacks;
    py::dict d;
    d["mode"] = trainer->cfg.exact_dev() ? "exact" : (trainer->cfg.host_rng() ? "host" : "philox");
    d["status"] = v[0];
    d["chunks"] = v[1];
    d["sub_chunks"] = v[2];
    d["quads_per_chunk"] = v[3];
    d["quads_consumed"] = v[4];
    d["walkers_started"] = v[5];
    d["attempts"] = v[6];
    d["sequential_fallbacks"] = fb;
    return d;
  }
};

}  // namespace

PYBIND11_MODULE(_myfm, m) {
  m.doc() = "MI355X backend for myfm (same surface as the reference's myfm._myfm).";

  py::enum_<TaskType>(m, "TaskType", py::arithmetic())
      .value("REGRESSION", TaskType::REGRESSION)
      .value("CLASSIFICATION", TaskType::CLASSIFICATION)
      .value("ORDERED", TaskType::ORDERED);

  py::class_<FMLearningConfig>(m, "FMLearningConfig");

  py::class_<RelationBlock, std::shared_ptr<RelationBlock>>(m, "RelationBlock", "The RelationBlock Class.")
      .def(py::init([](const py::object &o2b, const py::object &data) {
             // (an integer numpy array is taken without a per-element Python conversion: maps of 10^7..10^8 rows)
             if (py::isinstance<py::array>(o2b)) {
               auto arr = py::array_t<int64_t, py::array::c_style | py::array::forcecast>::ensure(o2b);
               if (!arr || arr.ndim() != 1) throw std::invalid_argument("original_to_block must be a 1-d integer array");
               return std::make_shared<RelationBlock>(arr.data(), (size_t)arr.size(), csr_from_py(data));
             }
             return std::make_shared<RelationBlock>(o2b.cast<vector<size_t>>(), csr_from_py(data));
           }),
           py::arg("original_to_block"), py::arg("data"))
      .def_property_readonly("original_to_block", &RelationBlock::original_to_block)
      .def_property_readonly("original_to_block_array",  // (extension: the map as an int64 array, no Python list)
                             [](const RelationBlock &b) {
                               py::array_t<int64_t> a((py::ssize_t)b.mapper_size);
                               if (b.mapper_size) std::memcpy(a.mutable_data(), b.map.get(), b.mapper_size * sizeof(int64_t));
                               return a;
                             })
      .def_property_readonly("data", [](const RelationBlock &b) { return csr_to_py(b.X); })
      .def_readonly("mapper_size", &RelationBlock::mapper_size)
      .def_readonly("block_size", &RelationBlock::block_size)
      .def_readonly("feature_size", &RelationBlock::feature_size)
      .def("__repr__",
           [](const RelationBlock &b) {
             std::ostringstream ss;
             ss << "<RelationBlock with mapper size = " << b.mapper_size << ", block data size = " << b.block_size
                << ", feature size = " << b.feature_size << ">";
             return ss.str();
           })
      .def(py::pickle([](const RelationBlock &b) { return py::make_tuple(b.original_to_block(), csr_to_py(b.X)); },
                      [](py::tuple t) {
                        if (t.size() != 2) throw std::runtime_error("invalid state for Relationblock.");
                        return std::make_shared<RelationBlock>(t[0].cast<vector<size_t>>(), csr_from_py(t[1]));
                      }));

  py::class_<ConfigBuilder>(m, "ConfigBuilder")
      .def(py::init<>())
      .def("set_alpha_0", &ConfigBuilder::set_alpha_0)
      .def("set_beta_0", &ConfigBuilder::set_beta_0)
      .def("set_gamma_0", &ConfigBuilder::set_gamma_0)
      .def("set_mu_0", &ConfigBuilder::set_mu_0)
      .def("set_reg_0", &ConfigBuilder::set_reg_0)
      .def("set_n_iter", &ConfigBuilder::set_n_iter)
      .def("set_n_kept_samples", &ConfigBuilder::set_n_kept_samples)
      .def("set_task_type", &ConfigBuilder::set_task_type)
      .def("set_nu_oprobit", &ConfigBuilder::set_nu_oprobit)
      .def("set_fit_w0", &ConfigBuilder::set_fit_w0)
      .def("set_fit_linear", &ConfigBuilder::set_fit_linear)
      .def("set_group_index", &ConfigBuilder::set_group_index)
      .def("set_identical_groups", &ConfigBuilder::set_identical_groups)
      .def("set_cutpoint_scale", &ConfigBuilder::set_cutpoint_scale)
      .def("set_exact_latent_draws", &ConfigBuilder::set_exact_latent_draws, py::return_value_policy::reference_internal)
      .def("set_latent_mode", &ConfigBuilder::set_latent_mode, py::return_value_policy::reference_internal)
      .def("set_latent_row_order",
           [](ConfigBuilder &b, const py::object &o) -> ConfigBuilder & {
             auto arr = py::array_t<int64_t, py::array::c_style | py::array::forcecast>::ensure(o);
             if (!arr || arr.ndim() != 1) throw std::invalid_argument("latent row order must be a 1-d integer array");
             b.latent_order.assign(arr.data(), arr.data() + arr.size());
             return b;
           },
           py::return_value_policy::reference_internal)
      .def("set_cutpoint_groups",
           // [(n_class, row indices)]: the reference's list-of-lists (declare_module.hpp:139-156), and numpy index arrays
           // without a per-element Python conversion (5e7 rows at config 5)
           [](ConfigBuilder &b, const py::object &groups) -> ConfigBuilder & {
             CutpointGroupType g;
             for (auto item : groups) {
               py::sequence t = py::reinterpret_borrow<py::sequence>(item);
               if (py::len(t) != 2) throw std::invalid_argument("cutpoint group: (n_class, row indices) expected");
               const size_t n_class = t[0].cast<size_t>();
               py::object rows = t[1];
               vector<size_t> v;
               if (py::isinstance<py::array>(rows)) {
                 auto a = py::array_t<int64_t, py::array::c_style | py::array::forcecast>::ensure(rows);
                 if (!a) throw std::invalid_argument("cutpoint group: integer row indices expected");
                 const int64_t *p = a.data();
                 std::atomic<int> neg(0);
                 host_ranges((int64_t)a.size(), [&](int64_t lo, int64_t hi) {
                   bool b = false;
                   for (int64_t k = lo; k < hi; k++) b |= p[k] < 0;
                   if (b) neg = 1;
                 });
                 if (neg) throw std::invalid_argument("cutpoint group: negative row index");
                 v.assign(p, p + a.size());  // (one pass: no zero fill before the copy)
               } else {
                 v = rows.cast<vector<size_t>>();
               }
               g.emplace_back(n_class, std::move(v));
             }
             b.cutpoint_groups = std::move(g);
             return b;
           },
           py::return_value_policy::reference_internal)
      .def("build", &ConfigBuilder::build);

  py::class_<FM>(m, "FM")
      .def_property(
          "w0", [](FM &f) { return f.w0; },
          [](FM &f, Real v) {
            f.ensure();
            f.store_idx = -1;
            f.w0 = v;
          })
      .def_property(
          "w",
          [](FM &f) {
            f.ensure();
            return vec_to_np(f.w);
          },
          [](FM &f, const py::object &v) {
            f.ensure();
            f.store_idx = -1;
            f.w = np_to_vec(v);
          })
      .def_property(
          "V",
          [](FM &f) {
            f.ensure();
            return colmajor_to_np(f.V, f.D(), f.n_factors);
          },
          [](FM &f, const py::object &v) {
            f.ensure();
            f.store_idx = -1;
            int64_t r, c;
            f.V = np_to_colmajor(v, &r, &c);
          })
      .def_property(
          "cutpoints",
          [](FM &f) {
            py::list out;
            for (auto &c : f.cutpoints) out.append(vec_to_np(c));
            return out;
          },
          [](FM &f, const py::object &v) {
            f.cutpoints.clear();
            for (auto item : py::reinterpret_borrow<py::sequence>(v)) f.cutpoints.push_back(np_to_vec(item));
          })
      .def("predict_score", &FM::predict_score)
      .def("oprobit_predict_proba", &FM::oprobit_predict_proba)
      .def("__repr__",
           [](FM &f) {
             f.ensure();
             std::ostringstream ss;
             ss << "<Factorization Machine sample with feature size = " << f.w.size() << ", rank = " << f.n_factors << ">";
             return ss.str();
           })
      .def(py::pickle(
          [](FM &f) {
            f.ensure();
            py::list cps;
            for (auto &c : f.cutpoints) cps.append(vec_to_np(c));
            return py::make_tuple(f.w0, vec_to_np(f.w), colmajor_to_np(f.V, f.D(), f.n_factors), cps);
          },
          [](py::tuple t) {
            if (t.size() != 3 && t.size() != 4) throw std::runtime_error("invalid state for FM.");
            int64_t r, c;
            vector<Real> V = np_to_colmajor(t[2], &r, &c);
            vector<vector<Real>> cps;
            if (t.size() == 4)
              for (auto item : py::reinterpret_borrow<py::sequence>(t[3])) cps.push_back(np_to_vec(item));
            return FM(t[0].cast<Real>(), np_to_vec(t[1]), std::move(V), (int)c, std::move(cps));
          }));

  py::class_<Hyper>(m, "FMHyperParameters")
      .def_readonly("alpha", &Hyper::alpha)
      .def_property_readonly("mu_w", [](const Hyper &h) { return vec_to_np(h.mu_w); })
      .def_property_readonly("lambda_w", [](const Hyper &h) { return vec_to_np(h.lambda_w); })
      .def_property_readonly("mu_V", [](const Hyper &h) { return colmajor_to_np(h.mu_V, h.G, h.K); })
      .def_property_readonly("lambda_V", [](const Hyper &h) { return colmajor_to_np(h.lambda_V, h.G, h.K); })
      .def(py::pickle(
          [](const Hyper &h) {
            return py::make_tuple(h.alpha, vec_to_np(h.mu_w), vec_to_np(h.lambda_w), colmajor_to_np(h.mu_V, h.G, h.K),
                                  colmajor_to_np(h.lambda_V, h.G, h.K));
          },
          [](py::tuple t) {
            if (t.size() != 5) throw std::runtime_error("invalid state for FMHyperParameters.");
            Hyper h;
            h.alpha = t[0].cast<Real>();
            h.mu_w = np_to_vec(t[1]);
            h.lambda_w = np_to_vec(t[2]);
            int64_t r, c;
            h.mu_V = np_to_colmajor(t[3], &r, &c);
            h.lambda_V = np_to_colmajor(t[4], &r, &c);
            h.G = (size_t)r;
            h.K = (size_t)c;
            return h;
          }));

  py::class_<Predictor>(m, "Predictor")
      .def_readonly("samples", &Predictor::samples)
      .def("predict", &Predictor::predict)
      .def("predict_parallel", &Predictor::predict_parallel)
      .def("predict_parallel_oprobit", &Predictor::predict_parallel_oprobit)
      .def(py::pickle(
          [](const Predictor &p) { return py::make_tuple(p.rank, p.feature_size, static_cast<int>(p.type), p.samples); },
          [](py::tuple t) {
            if (t.size() != 4) throw std::runtime_error("invalid state for FMHyperParameters.");
            Predictor p(t[0].cast<size_t>(), t[1].cast<size_t>(), static_cast<TaskType>(t[2].cast<int>()));
            p.samples = t[3].cast<vector<FM>>();
            return p;
          }));

  py::class_<FMTrainer>(m, "FMTrainer")
      .def(py::init<const py::object &, const py::object &, const py::object &, int, FMLearningConfig>())
      .def("create_FM", &FMTrainer::create_FM)
      .def("create_Hyper", &FMTrainer::create_Hyper);

  py::class_<LearningHistory>(m, "LearningHistory")
      .def_readonly("hypers", &LearningHistory::hypers)
      .def_readonly("train_log_losses", &LearningHistory::train_log_losses)
      .def_readonly("n_mh_accept", &LearningHistory::n_mh_accept)
      .def(py::pickle([](const LearningHistory &h) { return py::make_tuple(h.hypers, h.train_log_losses, h.n_mh_accept); },
                      [](py::tuple t) {
                        if (t.size() != 3) throw std::runtime_error("invalid state for LearningHistory.");
                        LearningHistory r;
                        r.hypers = t[0].cast<vector<Hyper>>();
                        r.train_log_losses = t[1].cast<vector<Real>>();
                        r.n_mh_accept = t[2].cast<vector<size_t>>();
                        return r;
                      }));

  // The callback is called after every iteration (FMTrainer.hpp:78-83) -- unless the callable carries an integer attribute
  // `myfm_every` (the estimators set it on the reference's DEFAULT callback, which only acts every `callback_default_freq`
  // iterations, base.py:179-205): then only on iterations 0, every, 2 every, ... and the last one, and the other iterations never
  // leave the C++ loop.
  m.def(
      "create_train_fm",
      [](size_t n_factor, Real init_std, const py::object &X, const py::object &relations, const py::object &y, int random_seed,
         FMLearningConfig &config, py::object cb) {
        int every = 1;
        if (py::hasattr(cb, "myfm_every")) every = std::max(1, cb.attr("myfm_every").cast<int>());
        auto f = cb.cast<std::function<bool(int, FM *, Hyper *, LearningHistory *)>>();
        if (every == 1) return create_train_fm(n_factor, init_std, X, relations, y, random_seed, config, f);
        const int n_iter = config.n_iter;
        return create_train_fm(n_factor, init_std, X, relations, y, random_seed, config,
                               [f, every, n_iter](int it, FM *fm, Hyper *hy, LearningHistory *h) {
                                 if (it % every != 0 && it != n_iter - 1) return false;
                                 return f(it, fm, hy, h);
                               });
      },
      "create and train fm.", py::return_value_policy::move);

  // extensions beyond the reference's surface (bench / tests)
  py::class_<GibbsSession>(m, "GibbsSession")
      .def(py::init<size_t, Real, const py::object &, const py::object &, const py::object &, int, FMLearningConfig &,
                    py::object, int64_t, int64_t, uint64_t, py::object, const std::string &, int, int>(),
           py::arg("rank"), py::arg("init_std"), py::arg("X"), py::arg("relations"), py::arg("y"), py::arg("random_seed"),
           py::arg("config"), py::arg("allreduce") = py::none(), py::arg("n_total_rows") = 0, py::arg("row_offset") = 0,
           py::arg("stream") = 0, py::arg("main_levels") = py::none(), py::arg("comm_id") = py::bytes(""),
           py::arg("shard_rank") = 0, py::arg("shard_world") = 1)
      .def("comm_stats",
           [](GibbsSession &s) {
             int64_t c = 0, d = 0;
             mfm_comm_stats(s.trainer->ctx, &c, &d);
             return py::make_tuple(c, d);
           })
      .def("comm_info",
           [](GibbsSession &s) {
             int32_t n = 0;
             char path[512];
             if (mfm_comm_info(s.trainer->ctx, &n, path, sizeof(path)) != MFM_OK) throw std::runtime_error(mfm_global_error());
             return py::make_tuple((int)n, std::string(path));
           },
           "(ranks of the library's own RCCL communicator (ncclCommCount), path of the librccl.so it bound); (0, '') without one")
      .def("step", &GibbsSession::step, py::call_guard<py::gil_scoped_release>())
      .def("synchronize", &GibbsSession::synchronize)
      .def("residual", &GibbsSession::residual)
      .def("timing_enable", &GibbsSession::timing_enable)
      .def("timing_select", &GibbsSession::timing_select)
      .def("timing_reset", &GibbsSession::timing_reset)
      .def("timing", &GibbsSession::timing)
      .def("plan_info", &GibbsSession::plan_info)
      .def("latent_info", &GibbsSession::latent_info)
      .def("plan_flags", [](GibbsSession &s) { return mfm_plan_flags(s.trainer->ctx); })
      // row-sharded persistent sweep (myfm_hip.h: mfm_peer_*): this rank's exchange buffers, every rank's buffers
      .def("peer_info",
           [](GibbsSession &s) {
             int32_t pending = 0;
             void *sum = nullptr, *flag = nullptr;
             int64_t sb = 0, fb = 0;
             ck(s.trainer->ctx, mfm_peer_info(s.trainer->ctx, &pending, &sum, &flag, &sb, &fb));
             return py::make_tuple(pending != 0, (uintptr_t)sum, (uintptr_t)flag, sb, fb);
           })
      .def("peer_set",
           [](GibbsSession &s, int world, int rank, const std::vector<uintptr_t> &sums, const std::vector<uintptr_t> &flags) {
             if ((int)sums.size() != world || (int)flags.size() != world) throw std::invalid_argument("peer_set: one buffer pair per rank");
             std::vector<void *> a, b;
             for (auto v : sums) a.push_back((void *)v);
             for (auto v : flags) b.push_back((void *)v);
             ck(s.trainer->ctx, mfm_peer_set(s.trainer->ctx, world, rank, a.data(), b.data()));
           })
      .def("peer_model_info",
           [](GibbsSession &s) {
             void *w = nullptr, *V = nullptr;
             ck(s.trainer->ctx, mfm_peer_model_info(s.trainer->ctx, &w, &V));
             return py::make_tuple((uintptr_t)w, (uintptr_t)V);
           })
      .def("peer_set_model",
           [](GibbsSession &s, int world, int rank, const std::vector<uintptr_t> &ws, const std::vector<uintptr_t> &Vs) {
             if ((int)ws.size() != world || (int)Vs.size() != world) throw std::invalid_argument("peer_set_model: one buffer pair per rank");
             std::vector<void *> a, b;
             for (auto v : ws) a.push_back((void *)v);
             for (auto v : Vs) b.push_back((void *)v);
             ck(s.trainer->ctx, mfm_peer_set_model(s.trainer->ctx, world, rank, a.data(), b.data()));
           })
      .def("peer_export",
           [](GibbsSession &s) {
             char h[256];
             ck(s.trainer->ctx, mfm_peer_export(s.trainer->ctx, h));
             return py::bytes(h, 256);
           })
      .def("peer_drop", [](GibbsSession &s) { ck(s.trainer->ctx, mfm_peer_drop(s.trainer->ctx)); })
      .def("peer_import",
           [](GibbsSession &s, int world, int rank, const std::string &all) {
             if ((int)all.size() != world * 256) throw std::invalid_argument("peer_import: 256 bytes per rank");
             ck(s.trainer->ctx, mfm_peer_import(s.trainer->ctx, world, rank, all.data()));
           })
      .def_property_readonly("fm", [](GibbsSession &s) -> FM & { return s.fm; }, py::return_value_policy::reference_internal)
      .def_property_readonly("hyper", [](GibbsSession &s) -> Hyper & { return s.hyper; },
                             py::return_value_policy::reference_internal)
      .def_readonly("iteration", &GibbsSession::it);
  m.def("create_train_fm_sharded", &create_train_fm_sharded, "create_train_fm over row shards (one process per GPU).",
        py::arg("rank"), py::arg("init_std"), py::arg("X"), py::arg("relations"), py::arg("y"), py::arg("random_seed"),
        py::arg("config"), py::arg("callback"), py::arg("shard_rank"), py::arg("shard_world"), py::arg("n_total_rows"),
        py::arg("row_offset"), py::arg("main_levels"), py::arg("comm_id") = py::bytes(""), py::arg("allreduce") = py::none(),
        py::arg("stream") = 0, py::arg("peer_connect") = py::none(), py::return_value_policy::move);
  py::class_<PeerHandle>(m, "PeerHandle", "The exchange buffers of a row-sharded training context (myfm_amd.distributed.connect_peers).")
      .def("peer_info", &PeerHandle::peer_info)
      .def("peer_export", &PeerHandle::peer_export)
      .def("peer_import", &PeerHandle::peer_import)
      .def("peer_drop", &PeerHandle::peer_drop);
  // fit()'s row sort without numpy's argsort + fancy indexing (6 s for 1e7 shuffled rows): a stable counting
  // sort of the rows by their first stored column, and a threaded gather of the CSR rows in that order
  m.def("row_order_by_first_column",
        [](py::array_t<int64_t, py::array::c_style | py::array::forcecast> indptr,
           py::array_t<int32_t, py::array::c_style | py::array::forcecast> indices, int64_t n_cols) {
          const int64_t n = (int64_t)indptr.size() - 1;
          const int64_t *ip = indptr.data();
          const int32_t *ix = indices.data();
          py::array_t<int64_t> order((size_t)std::max<int64_t>(n, 0));
          int64_t *o = order.mutable_data();
          vector<int64_t> cnt((size_t)n_cols + 1, 0);
          for (int64_t t = 0; t < n; t++) {
            if (ip[t + 1] <= ip[t]) throw std::invalid_argument("row_order_by_first_column: a row has no stored entry");
            const int32_t j = ix[ip[t]];
            if (j < 0 || j >= n_cols) throw std::invalid_argument("row_order_by_first_column: column index out of range");
            cnt[(size_t)j + 1]++;
          }
          for (int64_t j = 0; j < n_cols; j++) cnt[j + 1] += cnt[j];
          for (int64_t t = 0; t < n; t++) o[cnt[ix[ip[t]]]++] = t;
          return order;
        },
        py::arg("indptr"), py::arg("indices"), py::arg("n_cols"));
  m.def("permute_csr_rows",
        [](py::array_t<int64_t, py::array::c_style | py::array::forcecast> indptr,
           py::array_t<int32_t, py::array::c_style | py::array::forcecast> indices,
           py::array_t<double, py::array::c_style | py::array::forcecast> data,
           py::array_t<int64_t, py::array::c_style | py::array::forcecast> order) {
          const int64_t n = (int64_t)order.size();
          const int64_t *ip = indptr.data(), *od = order.data();
          const int32_t *ix = indices.data();
          const double *dv = data.data();
          const int64_t n_src = (int64_t)indptr.size() - 1;
          py::array_t<int64_t> nptr((size_t)n + 1);
          int64_t *np_ = nptr.mutable_data();
          np_[0] = 0;
          for (int64_t t = 0; t < n; t++) {
            if (od[t] < 0 || od[t] >= n_src) throw std::invalid_argument("permute_csr_rows: row index out of range");
            np_[t + 1] = np_[t] + (ip[od[t] + 1] - ip[od[t]]);
          }
          py::array_t<int32_t> nidx((size_t)np_[n]);
          py::array_t<double> nval((size_t)np_[n]);
          int32_t *ni = nidx.mutable_data();
          double *nv = nval.mutable_data();
          const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
          const int64_t chunk = (n + hw - 1) / hw;
          {
            py::gil_scoped_release rel;
            vector<std::thread> th;
            for (unsigned k = 0; k < hw; k++)
              th.emplace_back([=]() {
                const int64_t b = (int64_t)k * chunk, e = std::min<int64_t>(n, b + chunk);
                for (int64_t t = b; t < e; t++) {
                  const int64_t s0 = ip[od[t]], len = ip[od[t] + 1] - s0, d0 = np_[t];
                  for (int64_t q = 0; q < len; q++) {
                    ni[d0 + q] = ix[s0 + q];
                    nv[d0 + q] = dv[s0 + q];
                  }
                }
              });
            for (auto &t : th) t.join();
          }
          return py::make_tuple(nptr, nidx, nval);
        },
        py::arg("indptr"), py::arg("indices"), py::arg("data"), py::arg("order"));
  m.def("comm_unique_id", []() {
    char id[128];
    if (mfm_comm_unique_id(id) != MFM_OK) throw std::runtime_error(mfm_global_error());
    return py::bytes(id, 128);
  });
  m.def("device_count", []() { return mfm_device_count(); });
  m.def("backend_version", []() { return std::string(mfm_version()); });
  // host-only self-test of the jump-ahead polynomials the parallel generator uses (csrc/mfm_mtjump.hpp) against
  // std::mt19937 itself: tempering is linear, so the relation x_{m+J} = XOR_{g_i = 1} x_{m+i} holds for the
  // engine's outputs too. Returns the number of mismatching outputs among 624 (0 = correct).
  m.def("device_hyper_iterations", []() { return g_device_hyper_iterations; },
        "Gibbs iterations of this process whose hyper-parameters were drawn on the device (MYFM_AMD_DEVICE_HYPERS=1)");
  m.def("mt_jump_selftest", [](int blocks_per_wg, int p, unsigned seed) {
    std::vector<uint32_t> tab;
    if (!mfm::mtjump::build_jump_table(blocks_per_wg, p, tab)) return -1;
    const uint64_t J = (uint64_t)((int64_t)p * blocks_per_wg - 1) * 624u;
    std::mt19937 gen(seed);
    std::vector<uint32_t> x((size_t)J + 19937 + 2 * 624);
    for (auto &v : x) v = (uint32_t)gen();
    const uint32_t *g = tab.data() + (size_t)p * mfm::mtjump::JUMP_WORDS32;
    const size_t m0 = 624;  // past the seeded block
    int bad = 0;
    for (int l = 0; l < 624; l++) {
      uint32_t yv = 0;
      for (int i = 0; i < 19937; i++)
        if ((g[i >> 5] >> (i & 31)) & 1u) yv ^= x[m0 + l + i];
      bad += yv != x[m0 + l + J];
    }
    return bad;
  });
  // host-only self-test of the bulk normal filler (csrc/mfm_hostnormals.hpp) against one persistent
  // std::normal_distribution<double> on the same std::mt19937: (values, values of the plain loop, the next raw output of
  // either engine afterwards). `discard` outputs are taken first so that the sequence starts inside a state block.
  m.def("host_normals_selftest", [](unsigned seed, int64_t discard, int64_t count, double scale, int threads) {
    std::mt19937 g1(seed), g2(seed);
    g1.discard((unsigned long long)discard);
    g2.discard((unsigned long long)discard);
    py::array_t<double> fast((py::ssize_t)count), plain((py::ssize_t)count);
    mfm_hostnormals::fill_normals(g1, fast.mutable_data(), (size_t)count, scale, threads);
    std::normal_distribution<double> nd;
    double *pp = plain.mutable_data();
    for (int64_t i = 0; i < count; i++) pp[i] = nd(g2) * scale;
    return py::make_tuple(fast, plain, (uint32_t)g1(), (uint32_t)g2());
  });
}
