// myfm_oracle.cpp -- TEST INFRASTRUCTURE ONLY (parity oracle + timed CPU baseline).
//
// Eigen-free, single-threaded CPU restatement of the Gibbs sampler of tohtsky/myFM
// (reference at /root/reference, read-only). Nothing under myfm_amd/ may include, link,
// import or call this file: only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg use it, and only as the checker / the CPU baseline.
//
// PARITY STATUS: "parity unpinned" against a real build of the reference. The reference
// core needs Eigen 3.4.0, which setup.py:20-49 downloads at build time; Eigen is not in
// /root/reference and not in this image, so include/myfm/*.hpp cannot be compiled here and
// the reference ships no numeric golden vectors. This restatement follows the reference
// loop-for-loop (same update order, same std::mt19937 + libstdc++ normal/gamma/uniform
// distribution objects constructed at the same places, same ascending-index summation
// order where the reference's order is defined by its own loops) and is pinned by the
// identities the reference's own tests assert (tests/test_oracle_*.py):
//   flat == blocked (tests/regression/test_block.py:136-149), predict == running mean of
//   predict_score (tests/regression/test_fit.py:39), closed-form FM score
//   (tests/test_utils.py:16-25), statistical recovery bounds (test_fit.py:43-72,
//   test_classification.py:56-70, test_oprobit_1dim.py:34-38).
// Eigen's vectorised dense reductions (e.g. e.array().square().sum()) use an unspecified
// association order, so even a real build could only be matched to fp64 round-off there.
//
// Every function cites the reference file:line it restates. File paths are relative to
// /root/reference/.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace orc {

typedef double Real;
using std::vector;

// ---------------------------------------------------------------------------------------
// Sparse containers. include/myfm/definitions.hpp:16-24: row-major sparse, int32 inner
// indices, col-major dense.
// ---------------------------------------------------------------------------------------
struct Csr {
  int64_t rows = 0, cols = 0;
  vector<int64_t> ptr;  // rows + 1
  vector<int32_t> idx;
  vector<Real> val;
  int64_t nnz() const { return (int64_t)idx.size(); }
};

// X.transpose() assigned to a row-major matrix (BaseFMTrainer.hpp:61, definitions.hpp:59):
// row j of the result lists the entries of column j of X in ascending row order.
static Csr transpose(const Csr &X) {
  Csr T;
  T.rows = X.cols;
  T.cols = X.rows;
  T.ptr.assign(X.cols + 1, 0);
  for (int64_t p = 0; p < X.nnz(); p++) T.ptr[X.idx[p] + 1]++;
  for (int64_t j = 0; j < X.cols; j++) T.ptr[j + 1] += T.ptr[j];
  T.idx.resize(X.nnz());
  T.val.resize(X.nnz());
  vector<int64_t> cur(T.ptr.begin(), T.ptr.end() - 1);
  for (int64_t i = 0; i < X.rows; i++) {
    for (int64_t p = X.ptr[i]; p < X.ptr[i + 1]; p++) {
      int64_t q = cur[X.idx[p]]++;
      T.idx[q] = (int32_t)i;
      T.val[q] = X.val[p];
    }
  }
  return T;
}

// Eigen row-major sparse * dense vector: one sequential accumulation per row.
static void spmv(const Csr &X, const Real *v, Real *out) {
  for (int64_t i = 0; i < X.rows; i++) {
    Real s = 0;
    for (int64_t p = X.ptr[i]; p < X.ptr[i + 1]; p++) s += X.val[p] * v[X.idx[p]];
    out[i] = s;
  }
}
// X.cwiseAbs2() * v.array().square().matrix()
static void spmv_sq(const Csr &X, const Real *v, Real *out) {
  for (int64_t i = 0; i < X.rows; i++) {
    Real s = 0;
    for (int64_t p = X.ptr[i]; p < X.ptr[i + 1]; p++) {
      Real x = X.val[p], w = v[X.idx[p]];
      s += (x * x) * (w * w);
    }
    out[i] = s;
  }
}

// include/myfm/definitions.hpp:30-52
struct RelationBlock {
  vector<int64_t> original_to_block;
  Csr X;
  int64_t block_size = 0, feature_size = 0;
};

// include/myfm/definitions.hpp:54-84
struct RelationWiseCache {
  Csr X_t;
  vector<Real> cardinality, q, q_S, c, c_S, e, e_q;
  explicit RelationWiseCache(const RelationBlock &src) : X_t(transpose(src.X)) {
    size_t B = (size_t)src.X.rows;
    cardinality.assign(B, 0);
    q.assign(B, 0);
    q_S.assign(B, 0);
    c.assign(B, 0);
    c_S.assign(B, 0);
    e.assign(B, 0);
    e_q.assign(B, 0);
    for (auto v : src.original_to_block) cardinality[v] += 1;
  }
};

enum Task { REGRESSION = 0, CLASSIFICATION = 1, ORDERED = 2 };

// include/myfm/FMLearningConfig.hpp:17-78
struct Config {
  Real alpha_0 = 1, beta_0 = 1, gamma_0 = 1, mu_0 = 1, reg_0 = 1;
  int task = REGRESSION;
  Real nu_oprobit = 5;
  bool fit_w0 = true, fit_linear = true;
  vector<int32_t> group_index;
  int n_groups = 0;
  vector<vector<int64_t>> group_vs_feature_index;
  vector<std::pair<int, vector<int64_t>>> cutpoint_groups;

  // FMLearningConfig.hpp:29-45
  void finalize() {
    n_groups = 0;
    for (auto g : group_index) n_groups = std::max(n_groups, (int)g + 1);
    vector<char> seen(n_groups, 0);
    for (auto g : group_index) seen[g] = 1;
    for (int g = 0; g < n_groups; g++)
      if (!seen[g]) throw std::invalid_argument("No matching index for group index " + std::to_string(g) + " found.");
    group_vs_feature_index.assign(n_groups, {});
    for (size_t j = 0; j < group_index.size(); j++) group_vs_feature_index[group_index[j]].push_back((int64_t)j);
  }
};

// include/myfm/FM.hpp:10-172. V is column-major D x K (definitions.hpp:17).
struct FM {
  int n_factors = 0;
  int64_t D = 0;
  Real w0 = 0;
  vector<Real> w;
  vector<Real> V;  // V[f * D + j]
  vector<vector<Real>> cutpoints;

  // FM.hpp:34-45. One persistent normal_distribution (polar pairs are both consumed);
  // Eigen evaluates unaryExpr over a col-major matrix in storage order: V[:,0], V[:,1], ...
  void initialize_weight(int64_t n_features, Real init_std, std::mt19937 &gen) {
    D = n_features;
    std::normal_distribution<Real> nd;
    V.resize((size_t)D * n_factors);
    for (size_t k = 0; k < V.size(); k++) V[k] = nd(gen) * init_std;
    w.resize((size_t)D);
    for (int64_t j = 0; j < D; j++) w[j] = nd(gen) * init_std;
    w0 = nd(gen) * init_std;
  }

  // FM.hpp:54-136
  void predict_score_write_target(Real *target, const Csr &X, const vector<RelationBlock> &relations) const {
    int64_t case_size = X.rows;
    int64_t feature_size_all = X.cols;
    for (auto const &rel : relations) {
      if (case_size != (int64_t)rel.original_to_block.size())
        throw std::invalid_argument("Relation blocks have inconsistent mapper size with case_size");
      feature_size_all += rel.feature_size;
    }
    if (feature_size_all != D)
      throw std::invalid_argument("Total feature size mismatch. Should be " + std::to_string(D) + ", but got " +
                                  std::to_string(feature_size_all) + ".");
    // FM.hpp:78-87
    spmv(X, w.data(), target);
    for (int64_t i = 0; i < case_size; i++) target[i] = w0 + target[i];
    int64_t offset = X.cols;
    vector<Real> block_cache;
    for (auto const &rel : relations) {
      block_cache.assign((size_t)rel.block_size, 0);
      spmv(rel.X, w.data() + offset, block_cache.data());
      for (int64_t t = 0; t < case_size; t++) target[t] += block_cache[rel.original_to_block[t]];
      offset += rel.feature_size;
    }
    // FM.hpp:89-135
    vector<Real> q_cache((size_t)case_size);
    for (int f = 0; f < n_factors; f++) {
      const Real *vf = V.data() + (size_t)f * D;
      spmv(X, vf, q_cache.data());
      offset = X.cols;
      for (auto const &rel : relations) {
        block_cache.assign((size_t)rel.block_size, 0);
        spmv(rel.X, vf + offset, block_cache.data());
        offset += rel.feature_size;
        for (int64_t t = 0; t < case_size; t++) q_cache[t] += block_cache[rel.original_to_block[t]];
      }
      for (int64_t t = 0; t < case_size; t++) target[t] += q_cache[t] * q_cache[t] * 0.5;
      offset = X.cols;
      spmv_sq(X, vf, q_cache.data());
      for (auto const &rel : relations) {
        block_cache.assign((size_t)rel.block_size, 0);
        spmv_sq(rel.X, vf + offset, block_cache.data());
        offset += rel.feature_size;
        for (int64_t t = 0; t < case_size; t++) q_cache[t] += block_cache[rel.original_to_block[t]];
      }
      for (int64_t t = 0; t < case_size; t++) target[t] -= q_cache[t] * 0.5;
    }
  }
};

// include/myfm/HyperParams.hpp:13-19. mu_V / lambda_V are G x K column-major: [f * G + g].
struct Hyper {
  Real alpha = 1;
  vector<Real> mu_w, lambda_w, mu_V, lambda_V;
  int G = 0, K = 0;
  Hyper() {}
  Hyper(int K_, int G_) : mu_w(G_), lambda_w(G_), mu_V((size_t)G_ * K_), lambda_V((size_t)G_ * K_), G(G_), K(K_) {}
};

// ---------------------------------------------------------------------------------------
// include/myfm/util.hpp:15-78 -- truncated normal samplers.
// ---------------------------------------------------------------------------------------
static Real sample_truncated_normal_left(std::mt19937 &gen, Real mu_minus) {  // util.hpp:15-37
  if (mu_minus < 0) {
    std::normal_distribution<Real> dist(0, 1);
    while (true) {
      Real z = dist(gen);
      if (z > mu_minus) return z;
    }
  } else {
    Real alpha_star = (mu_minus + std::sqrt(mu_minus * mu_minus + 4)) / 2;
    std::uniform_real_distribution<Real> dist(0, 1);
    while (true) {
      Real z = -std::log(dist(gen)) / alpha_star + mu_minus;
      Real rho = std::exp(-(z - alpha_star) * (z - alpha_star) / 2);
      Real u = dist(gen);
      if (u < rho) return z;
    }
  }
}
static Real sample_truncated_normal_twoside(std::mt19937 &gen, Real mu_minus, Real mu_plus) {  // util.hpp:39-60
  std::uniform_real_distribution<Real> proposal(mu_minus, mu_plus);
  std::uniform_real_distribution<Real> acceptance(0, 1);
  Real rho;
  while (true) {
    Real z = proposal(gen);
    if ((mu_minus <= 0) && (mu_plus >= 0)) {
      rho = std::exp(-z * z / 2);
    } else if (mu_plus < 0) {
      rho = std::exp((mu_plus * mu_plus - z * z) / 2);
    } else {
      rho = std::exp((mu_minus * mu_minus - z * z) / 2);
    }
    Real u = acceptance(gen);
    if (u < rho) return z;
  }
}
static Real sample_truncated_normal_left(std::mt19937 &gen, Real mean, Real std_, Real mu_minus) {  // util.hpp:61-66
  return mean + std_ * sample_truncated_normal_left(gen, (mu_minus - mean) / std_);
}
static Real sample_truncated_normal_right(std::mt19937 &gen, Real mu_plus) {  // util.hpp:68-71
  return -sample_truncated_normal_left(gen, -mu_plus);
}
static Real sample_truncated_normal_right(std::mt19937 &gen, Real mean, Real std_, Real mu_plus) {  // util.hpp:73-78
  return mean + std_ * sample_truncated_normal_right(gen, (mu_plus - mean) / std_);
}

// ---------------------------------------------------------------------------------------
// erf / erfcx. The reference uses the vendored Faddeeva package (cpp_source/Faddeeva.cc,
// MIT, S. G. Johnson) -- a Chebyshev-table erfcx. That table is not restated here; this
// oracle evaluates the same functions from libm erfc for small |x| and the Laplace
// continued fraction erfcx(x) = (1/sqrt(pi)) / (x + (1/2)/(x + 1/(x + (3/2)/(x + ...))))
// for large x. tests/test_oracle_special.py checks it against the real Faddeeva.cc built
// into oracle/_ref/ (<= 5e-15 relative).
// ---------------------------------------------------------------------------------------
static Real erfcx_pos(Real x) {
  if (x < 3.0) return std::exp(x * x) * std::erfc(x);
  if (x > 5e7) return 0.5641895835477562869 / x;  // 1/sqrt(pi)/x, next term < 1e-16 relative
  // backward recurrence of the continued fraction; 60 terms are ample for x >= 3
  int n = (x < 5) ? 90 : (x < 10 ? 50 : 25);
  Real t = x;
  for (int k = n; k >= 1; k--) t = x + (0.5 * k) / t;
  return 0.5641895835477562869 / t;
}
static Real erfcx(Real x) {
  if (x >= 0) return erfcx_pos(x);
  if (x < -26.7) return std::numeric_limits<Real>::infinity();
  return 2 * std::exp(x * x) - erfcx_pos(-x);
}

// ---------------------------------------------------------------------------------------
// Small dense helpers for the ordered-probit sampler (Eigen LLT restated).
// Matrices are n x n row-major vectors here; all of them are symmetric where it matters.
// ---------------------------------------------------------------------------------------
static bool cholesky_lower(const vector<Real> &A, int n, vector<Real> &L) {
  L.assign((size_t)n * n, 0);
  for (int j = 0; j < n; j++) {
    Real d = A[(size_t)j * n + j];
    for (int k = 0; k < j; k++) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
    if (!(d > 0)) {
      // Eigen's LLT does not throw on a non-PD matrix; it carries NaNs on. Mirror that.
      d = std::numeric_limits<Real>::quiet_NaN();
    }
    Real ljj = std::sqrt(d);
    L[(size_t)j * n + j] = ljj;
    for (int i = j + 1; i < n; i++) {
      Real s = A[(size_t)i * n + j];
      for (int k = 0; k < j; k++) s -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
      L[(size_t)i * n + j] = s / ljj;
    }
  }
  return true;
}
// solve A x = b with A = L L^T
static void llt_solve(const vector<Real> &L, int n, vector<Real> &b) {
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

// ---------------------------------------------------------------------------------------
// include/myfm/OProbitSampler.hpp:15-481
// ---------------------------------------------------------------------------------------
struct OprobitSampler {
  static constexpr Real SQRT2 = 1.4142135623730951;
  static constexpr Real SQRTPI = 1.7724538509055159;
  static constexpr Real SQRT2PI = SQRT2 * SQRTPI;
  static constexpr Real PI = 3.141592653589793;

  vector<Real> *x_;        // trainer's e_train (OProbitSampler.hpp:465)
  const vector<Real> *y_;  // trainer's y
  int K;
  vector<int64_t> indices_;
  Real reg, nu;
  std::mt19937 *rng;
  vector<Real> alpha_now, gamma_now;
  vector<Real> H;  // (K-1)x(K-1)
  vector<Real> zmins, zmaxs;
  vector<size_t> histogram;
  size_t accept_count = 0;

  // OProbitSampler.hpp:25-47
  OprobitSampler(vector<Real> &x, const vector<Real> &y, int K_, const vector<int64_t> &indices, std::mt19937 &rng_,
                 Real reg_, Real nu_)
      : x_(&x), y_(&y), K(K_), indices_(indices), reg(reg_), nu(nu_), rng(&rng_), zmins(K_), zmaxs(K_), histogram(K_) {
    alpha_now.assign(K - 1, 0);
    gamma_now.assign(K - 1, 0);
    alpha_to_gamma(gamma_now, alpha_now);
    H.assign((size_t)(K - 1) * (K - 1), 0);
    for (auto i : indices_) {
      int y_label = (int)y[i];
      if (std::abs(y_label - y[i]) > 1e-3) throw std::invalid_argument("y has a floating-point element.");
      if (y_label < 0) throw std::invalid_argument("y has a negative element.");
      if (y_label >= K)
        throw std::invalid_argument("y[ " + std::to_string(i) + "] is greater than " + std::to_string(K - 1) + ".");
      histogram[y_label]++;
    }
  }

  int n() const { return K - 1; }
  Real &Hm(vector<Real> &M, int i, int j) const { return M[(size_t)i * n() + j]; }

  // OProbitSampler.hpp:49-53
  Real log_p_mvt(const vector<Real> &SigmaInverse, const vector<Real> &mu, Real nu_, const vector<Real> &x) const {
    int m = n();
    Real log_p = 0;
    for (int i = 0; i < m; i++) {
      Real s = 0;
      for (int j = 0; j < m; j++) s += SigmaInverse[(size_t)i * m + j] * (x[j] - mu[j]);
      log_p += (x[i] - mu[i]) * s;
    }
    return std::log(1 + log_p / nu_) * (-nu_ - m) / 2;
  }

  // OProbitSampler.hpp:55-72
  vector<Real> sample_mvt(const vector<Real> &SigmaInverse, Real nu_) {
    int m = n();
    vector<Real> result(m);
    std::normal_distribution<Real> base_dist(0, 1);
    std::gamma_distribution<Real> chi_gen(nu_ / 2);
    for (int i = 0; i < m; i++) result[i] = base_dist(*rng);
    vector<Real> L;
    cholesky_lower(SigmaInverse, m, L);
    // L.matrixU().solve(result): U = L^T, back substitution
    for (int i = m - 1; i >= 0; i--) {
      Real s = result[i];
      for (int k = i + 1; k < m; k++) s -= L[(size_t)k * m + i] * result[k];
      result[i] = s / L[(size_t)i * m + i];
    }
    Real denom = std::sqrt(chi_gen(*rng) * 2 / nu_);
    for (int i = 0; i < m; i++) result[i] /= denom;
    return result;
  }

  // OProbitSampler.hpp:74-93 (fix_gamma0 == false)
  static void jacobian_dgamma_dalpha(vector<Real> &J, const vector<Real> &alpha) {
    int m = (int)alpha.size();
    std::fill(J.begin(), J.end(), 0);
    J[0] = 1;
    for (int j = 1; j < m; j++) J[j] = 1;
    for (int i = 1; i < m; i++) {
      Real ed = std::exp(alpha[i]);
      for (int j = i; j < m; j++) J[(size_t)i * m + j] = ed;
    }
  }
  // OProbitSampler.hpp:95-101
  static void alpha_to_gamma(vector<Real> &target, const vector<Real> &alpha) {
    if (alpha.empty()) return;
    target[0] = alpha[0];
    for (size_t i = 1; i < alpha.size(); i++) target[i] = target[i - 1] + std::exp(alpha[i]);
  }

  // OProbitSampler.hpp:111-181
  void safe_ldiff(Real x, Real y, Real &loss, Real &dx, Real &dy, vector<Real> *Ht, int label) const {
    Real denominator, exp_factor;
    if (y > 0) {
      exp_factor = std::exp((y * y - x * x) / 2);
      denominator = erfcx(y / SQRT2) - exp_factor * erfcx(x / SQRT2);
      loss -= y * y / 2;
      loss += std::log(denominator / 2);
      dx += (2 / SQRT2PI) * exp_factor / denominator;
      dy -= (2 / SQRT2PI) / denominator;
      if (Ht) {
        Hm(*Ht, label, label) += -(SQRT2PI * x * denominator * std::exp((y * y - x * x) / 2) + 2 * std::exp(y * y - x * x)) /
                                 denominator / denominator / PI;
        Hm(*Ht, label - 1, label - 1) += (SQRT2PI * y * denominator - 2) / denominator / denominator / PI;
        Real off_diag = 2 * std::exp((y * y - x * x) / 2) / PI / denominator / denominator;
        Hm(*Ht, label, label - 1) += off_diag;
        Hm(*Ht, label - 1, label) += off_diag;
      }
    } else if (x < 0) {
      loss -= x * x / 2;
      exp_factor = std::exp((x * x - y * y) / 2);
      denominator = erfcx(-x / SQRT2) - exp_factor * erfcx(-y / SQRT2);
      loss += std::log(denominator / 2);
      dx += (2 / SQRT2PI) / denominator;
      dy -= (2 / SQRT2PI) * exp_factor / denominator;
      if (Ht) {
        Hm(*Ht, label, label) += -(SQRT2PI * x * denominator + 2) / PI / denominator / denominator;
        Hm(*Ht, label - 1, label - 1) +=
            (SQRT2PI * y * exp_factor * denominator - 2 * (exp_factor * exp_factor)) / PI / denominator / denominator;
        Real off_diag = 2 * exp_factor / PI / denominator / denominator;
        Hm(*Ht, label, label - 1) += off_diag;
        Hm(*Ht, label - 1, label) += off_diag;
      }
    } else {
      denominator = std::erf(x / SQRT2) - std::erf(y / SQRT2);
      Real expxx = std::exp(-x * x / 2);
      Real expyy = std::exp(-y * y / 2);
      dx += 2 * expxx / denominator / SQRT2PI;
      dy -= 2 * expyy / denominator / SQRT2PI;
      loss += std::log(denominator / 2);
      if (Ht) {
        Hm(*Ht, label, label) += -(SQRT2PI * x * denominator * expxx + 2 * expxx * expxx) / PI / denominator / denominator;
        Hm(*Ht, label - 1, label - 1) +=
            -(-SQRT2PI * y * denominator * expyy + 2 * expyy * expyy) / PI / denominator / denominator;
        Real off_diag = 2 * expxx * expyy / PI / denominator / denominator;
        Hm(*Ht, label, label - 1) += off_diag;
        Hm(*Ht, label - 1, label) += off_diag;
      }
    }
  }
  // OProbitSampler.hpp:183-208
  void safe_lcdf(Real x, Real &loss, Real &dx, vector<Real> *Ht, int label) const {
    Real denominator, exp_factor;
    if (x > 1) {
      exp_factor = std::exp(-x * x / 2);
      denominator = 1 + std::erf(x / SQRT2);
      dx += (2 / SQRT2PI) * exp_factor / denominator;
      loss += std::log(denominator / 2);
      if (Ht)
        Hm(*Ht, label, label) +=
            -(SQRT2PI * x * denominator * exp_factor + 2 * exp_factor * exp_factor) / PI / denominator / denominator;
    } else {
      denominator = erfcx(-x / SQRT2);
      dx += (2 / SQRT2PI) / denominator;
      loss -= x * x / 2;
      loss += std::log(denominator / 2);
      if (Ht) Hm(*Ht, label, label) += -(SQRT2PI * x * denominator + 2) / PI / denominator / denominator;
    }
  }
  // OProbitSampler.hpp:210-236
  void safe_lccdf(Real x, Real &loss, Real &dx, vector<Real> *Ht, int label) const {
    Real denominator;
    if (x > -1) {
      denominator = erfcx(x / SQRT2);
      dx -= (2 / SQRT2PI) / denominator;
      loss += std::log(denominator / 2);
      loss -= x * x / 2;
      if (Ht) Hm(*Ht, label - 1, label - 1) += (SQRT2PI * x * denominator - 2) / denominator / denominator / PI;
    } else {
      denominator = 1 - std::erf(x / SQRT2);
      dx -= (2 / SQRT2PI) * std::exp(-x * x / 2) / denominator;
      loss += std::log(denominator / 2);
      if (Ht) {
        Real exp_factor = std::exp(-(x * x) / 2);
        Hm(*Ht, label - 1, label - 1) +=
            -(-SQRT2PI * x * denominator * exp_factor + 2 * exp_factor * exp_factor) / PI / denominator / denominator;
      }
    }
  }

  // OProbitSampler.hpp:238-272
  void sample_z_given_cutpoint() {
    std::fill(zmins.begin(), zmins.end(), std::numeric_limits<Real>::max());
    std::fill(zmaxs.begin(), zmaxs.end(), std::numeric_limits<Real>::lowest());
    Real deviation = 1;
    vector<Real> &x = *x_;
    const vector<Real> &y = *y_;
    for (int64_t t : indices_) {
      int class_index = (int)y[t];
      Real pred_score = x[t];
      Real z_new;
      if (class_index == 0) {
        z_new = deviation * sample_truncated_normal_right(*rng, (gamma_now[class_index] - pred_score) / deviation) + pred_score;
        zmaxs[0] = std::max(zmaxs[0], z_new);
      } else if (class_index == (K - 1)) {
        z_new = deviation * sample_truncated_normal_left(*rng, (gamma_now[K - 2] - pred_score) / deviation) + pred_score;
        zmins[K - 1] = std::min(zmins[K - 1], z_new);
      } else {
        z_new = deviation * sample_truncated_normal_twoside(*rng, (gamma_now[class_index - 1] - pred_score) / deviation,
                                                            (gamma_now[class_index] - pred_score) / deviation) +
                pred_score;
        zmins[class_index] = std::min(zmins[class_index], z_new);
        zmaxs[class_index] = std::max(zmaxs[class_index], z_new);
      }
      x[t] -= z_new;
    }
  }

  // OProbitSampler.hpp:274-279
  void start_sample() {
    vector<Real> alpha_hat(K - 1, 0);
    find_minimum(alpha_hat);
    alpha_now = alpha_hat;
    alpha_to_gamma(gamma_now, alpha_now);
  }

  static Real norm2(const vector<Real> &v) {
    Real s = 0;
    for (Real a : v) s += a * a;
    return std::sqrt(s);
  }

  // OProbitSampler.hpp:289-357
  void find_minimum(vector<Real> &alpha_hat) {
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
        if (std::abs(past_loss - ll_current) <=
            delta * std::max(std::max(std::abs(ll_current), std::abs(past_loss)), Real(1)))
          break;
      }
      history[i % past] = ll_current;
      i++;
      if (i >= max_iter) break;
    }
    if (i == max_iter) throw std::runtime_error("Failed to converge. See fail-log.txt");
  }

  // OProbitSampler.hpp:359-387
  bool step() {
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
    Real log_p_transition_candidate = log_p_mvt(H, alpha_hat, nu, alpha_candidate);
    Real log_p_transition_old = log_p_mvt(H, alpha_hat, nu, alpha_now);
    Real test_ratio = std::exp(ll_candidate - log_p_transition_candidate - ll_old + log_p_transition_old);
    Real u = std::uniform_real_distribution<Real>{0, 1}(*rng);
    if (u < test_ratio) {
      alpha_now = alpha_candidate;
      alpha_to_gamma(gamma_now, alpha_now);
      accept_count++;
      return true;
    }
    return false;
  }

  static bool has_nan(const vector<Real> &v) {
    for (Real a : v)
      if (std::isnan(a)) return true;
    return false;
  }

  // the row loop of operator() (OProbitSampler.hpp:402-413): log-likelihood, d/dgamma and the gamma-space Hessian
  // accumulated over the group's rows in index order (what the device's mfm_oprobit_eval returns)
  void eval_rows(const vector<Real> &gamma, Real &ll, vector<Real> &dgamma, vector<Real> *Ht) const {
    const vector<Real> &x = *x_;
    const vector<Real> &y = *y_;
    for (auto i : indices_) {
      int label = (int)y[i];
      if (label == 0) {
        safe_lcdf(gamma[0] - x[i], ll, dgamma[0], Ht, label);
      } else if (label == (K - 1)) {
        safe_lccdf(gamma[K - 2] - x[i], ll, dgamma[K - 2], Ht, label);
      } else {
        safe_ldiff(gamma[label] - x[i], gamma[label - 1] - x[i], ll, dgamma[label], dgamma[label - 1], Ht, label);
      }
    }
  }

  // OProbitSampler.hpp:389-463 (operator())
  Real eval(const vector<Real> &alpha, vector<Real> &dalpha, vector<Real> *Ht) {
    int m = n();
    vector<Real> gamma(m, 0);
    std::fill(dalpha.begin(), dalpha.end(), 0);
    alpha_to_gamma(gamma, alpha);
    vector<Real> J((size_t)m * m);
    jacobian_dgamma_dalpha(J, alpha);
    Real ll = 0;
    if (Ht) std::fill(Ht->begin(), Ht->end(), 0);
    eval_rows(gamma, ll, dalpha, Ht);
    if (Ht) {
      vector<Real> &Hh = *Ht;
      vector<Real> expAlpha(m);
      for (int k = 0; k < m; k++) expAlpha[k] = std::exp(alpha[k]);
      // H = J * H * J^T
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
    // dalpha = -J * dalpha
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
};

// ---------------------------------------------------------------------------------------
// include/myfm/BaseFMTrainer.hpp:18-196 + include/myfm/FMTrainer.hpp:22-524
// ---------------------------------------------------------------------------------------
struct Trainer {
  Csr X, X_t;
  vector<RelationBlock> relations;
  vector<RelationWiseCache> relation_caches;
  int64_t dim_all = 0;
  vector<Real> y;
  int64_t n_train = 0;
  vector<Real> e_train, q_train;
  Config cfg;
  std::mt19937 gen_;
  FM fm;
  Hyper hyper;
  vector<OprobitSampler> cutpoint_sampler;
  // optional trace of every latent draw of update_V (kernel-level known-answer tests)
  bool trace_on = false;
  vector<Real> trace;  // (factor, feature, square_coeff, linear_coeff, v_new) x n

  Trainer() : gen_(0) {}
  Trainer(const Trainer &o)
      : X(o.X), X_t(o.X_t), relations(o.relations), relation_caches(o.relation_caches), dim_all(o.dim_all), y(o.y),
        n_train(o.n_train), e_train(o.e_train), q_train(o.q_train), cfg(o.cfg), gen_(o.gen_), fm(o.fm), hyper(o.hyper),
        cutpoint_sampler(o.cutpoint_sampler), trace_on(o.trace_on), trace(o.trace) {
    for (auto &cs : cutpoint_sampler) {
      cs.x_ = &e_train;
      cs.y_ = &y;
      cs.rng = &gen_;
    }
  }

  // BaseFMTrainer.hpp:58-105 (+ util.hpp:147-165)
  void construct(int seed) {
    X_t = transpose(X);
    dim_all = X.cols;
    int i = 0;
    for (auto &rel : relations) {
      if (X.rows != (int64_t)rel.original_to_block.size())
        throw std::runtime_error("main table has size " + std::to_string(X.rows) + " but the relation[" + std::to_string(i) +
                                 "] has size " + std::to_string(rel.original_to_block.size()));
      dim_all += rel.feature_size;
      i++;
    }
    n_train = X.rows;
    e_train.assign((size_t)n_train, 0);
    q_train.assign((size_t)n_train, 0);
    gen_.seed((uint32_t)seed);
    relation_caches.clear();
    for (auto &rel : relations) relation_caches.emplace_back(rel);
    if (X.rows != (int64_t)y.size())
      throw std::runtime_error("Shape mismatch: X has size " + std::to_string(X.rows) + " and y has size " +
                               std::to_string(y.size()));
    if ((int64_t)cfg.group_index.size() != dim_all) throw std::invalid_argument("group_index size mismatch");
    if (cfg.task == ORDERED) {
      vector<char> existence((size_t)X.rows, 0);
      for (auto &gc : cfg.cutpoint_groups)
        for (auto k : gc.second) {
          if (k >= X.rows) throw std::invalid_argument("out of range for cutpoint group config.");
          if (existence[k]) throw std::invalid_argument("index " + std::to_string(k) + " overlapping in cutpoint config.");
          existence[k] = 1;
        }
      for (int64_t r = 0; r < X.rows; r++)
        if (!existence[r]) throw std::invalid_argument("cutpoint group not specified for " + std::to_string(r) + ".");
    }
  }

  // declare_module.hpp:41-44 + FMTrainer.hpp:64-65
  void start(int rank, Real init_std) {
    fm = FM();
    fm.n_factors = rank;
    fm.initialize_weight(dim_all, init_std, gen_);  // BaseFMTrainer.hpp:107-111
    hyper = Hyper(rank, cfg.n_groups);              // BaseFMTrainer.hpp:113-115
    initialize_hyper();
    initialize_e();
  }

  // FMTrainer.hpp:89-97
  void initialize_hyper() {
    hyper.alpha = 1;
    std::fill(hyper.mu_w.begin(), hyper.mu_w.end(), 0);
    std::fill(hyper.lambda_w.begin(), hyper.lambda_w.end(), 1e-5);
    std::fill(hyper.mu_V.begin(), hyper.mu_V.end(), 0);
    std::fill(hyper.lambda_V.begin(), hyper.lambda_V.end(), 1e-5);
  }

  // FMTrainer.hpp:99-119
  void initialize_e() {
    fm.predict_score_write_target(e_train.data(), X, relations);
    if (cfg.task == ORDERED) {
      int i = 0;
      cutpoint_sampler.clear();
      cutpoint_sampler.reserve(cfg.cutpoint_groups.size());
      for (auto &c : cfg.cutpoint_groups) {
        fm.cutpoints.emplace_back((size_t)(c.first - 1));
        cutpoint_sampler.emplace_back(e_train, y, c.first, c.second, gen_, cfg.reg_0, cfg.nu_oprobit);
        cutpoint_sampler[i].start_sample();
        OprobitSampler::alpha_to_gamma(fm.cutpoints[i], cutpoint_sampler[i].alpha_now);
        cutpoint_sampler[i].sample_z_given_cutpoint();
        i++;
      }
      return;
    }
    for (int64_t t = 0; t < n_train; t++) e_train[t] -= y[t];
  }

  // FMTrainer.hpp:122-125 -- a fresh normal_distribution per draw.
  Real sample_normal(Real quad, Real first) { return (first / quad) + std::normal_distribution<Real>(0, 1)(gen_) / std::sqrt(quad); }

  // FMTrainer.hpp:127-145
  void update_alpha() {
    if (cfg.task == CLASSIFICATION || cfg.task == ORDERED) {
      hyper.alpha = 1;
      return;
    }
    Real e_all = 0;
    for (int64_t t = 0; t < n_train; t++) e_all += e_train[t] * e_train[t];
    Real exponent = (cfg.alpha_0 + X.rows) / 2;
    Real variance = (cfg.beta_0 + e_all) / 2;
    hyper.alpha = std::gamma_distribution<Real>(exponent, 1 / variance)(gen_);
  }

  // FMTrainer.hpp:150-169. mu/lambda/weight point at one factor's column (or w).
  void update_lambda_generic(const Real *mu, Real *lambda, const Real *weight) {
    int g = 0;
    for (const auto &feats : cfg.group_vs_feature_index) {
      Real mean = mu[g];
      Real alpha = cfg.alpha_0 + feats.size();
      Real beta = cfg.beta_0;
      for (auto j : feats) {
        Real dev = weight[j] - mean;
        beta += dev * dev;
      }
      lambda[g] = std::gamma_distribution<Real>(alpha / 2, 2 / beta)(gen_);
      g++;
    }
  }
  // FMTrainer.hpp:174-192
  void update_mu_generic(Real *mu, const Real *lambda, const Real *weight) {
    int g = 0;
    for (const auto &feats : cfg.group_vs_feature_index) {
      size_t n_feature_in_groups = feats.size();
      Real square = lambda[g] * (cfg.gamma_0 + n_feature_in_groups);
      Real linear = cfg.gamma_0 * cfg.mu_0;
      for (auto j : feats) linear += weight[j];
      linear *= lambda[g];
      mu[g] = sample_normal(square, linear);
      g++;
    }
  }
  void update_lambda_w() { update_lambda_generic(hyper.mu_w.data(), hyper.lambda_w.data(), fm.w.data()); }  // :194-196
  void update_mu_w() { update_mu_generic(hyper.mu_w.data(), hyper.lambda_w.data(), fm.w.data()); }          // :198-200
  void update_lambda_V() {                                                                                  // :202-208
    int G = cfg.n_groups;
    for (int f = 0; f < fm.n_factors; f++)
      update_lambda_generic(hyper.mu_V.data() + (size_t)f * G, hyper.lambda_V.data() + (size_t)f * G,
                            fm.V.data() + (size_t)f * fm.D);
  }
  void update_mu_V() {  // :210-216
    int G = cfg.n_groups;
    for (int f = 0; f < fm.n_factors; f++)
      update_mu_generic(hyper.mu_V.data() + (size_t)f * G, hyper.lambda_V.data() + (size_t)f * G,
                        fm.V.data() + (size_t)f * fm.D);
  }

  // FMTrainer.hpp:218-229
  void update_w0() {
    if (!cfg.fit_w0) {
      fm.w0 = 0;
      return;
    }
    Real s = 0;
    for (int64_t t = 0; t < n_train; t++) s += (fm.w0 - e_train[t]);
    Real w0_lin_term = hyper.alpha * s;
    Real w0_quad_term = hyper.alpha * n_train + cfg.reg_0;
    Real w0_new = sample_normal(w0_quad_term, w0_lin_term);
    for (int64_t t = 0; t < n_train; t++) e_train[t] += (w0_new - fm.w0);
    fm.w0 = w0_new;
  }

  // FMTrainer.hpp:231-314
  void update_w() {
    if (!cfg.fit_linear) {
      std::fill(fm.w.begin(), fm.w.end(), 0);
      return;
    }
    // main table, :237-254
    for (int64_t j = 0; j < X.cols; j++) {
      int group = cfg.group_index[j];
      const Real w_old = fm.w[j];
      for (int64_t p = X_t.ptr[j]; p < X_t.ptr[j + 1]; p++) e_train[X_t.idx[p]] -= X_t.val[p] * w_old;
      Real lambda = hyper.lambda_w[group], mu = hyper.mu_w[group];
      Real sq = 0;
      for (int64_t p = X_t.ptr[j]; p < X_t.ptr[j + 1]; p++) sq += X_t.val[p] * X_t.val[p];
      Real square_term = lambda + hyper.alpha * sq;
      Real dot = 0;
      for (int64_t p = X_t.ptr[j]; p < X_t.ptr[j + 1]; p++) dot += X_t.val[p] * e_train[X_t.idx[p]];
      Real linear_term = -hyper.alpha * dot + lambda * mu;
      Real w_new = sample_normal(square_term, linear_term);
      for (int64_t p = X_t.ptr[j]; p < X_t.ptr[j + 1]; p++) e_train[X_t.idx[p]] += X_t.val[p] * w_new;
      fm.w[j] = w_new;
    }
    // relation blocks, :256-313
    int64_t offset = X.cols;
    for (size_t r = 0; r < relations.size(); r++) {
      RelationBlock &rd = relations[r];
      RelationWiseCache &rc = relation_caches[r];
      std::fill(rc.e.begin(), rc.e.end(), 0);
      spmv(rd.X, fm.w.data() + offset, rc.q.data());
      for (int64_t t = 0; t < n_train; t++) {
        int64_t i = rd.original_to_block[t];
        rc.e[i] += e_train[t];
        e_train[t] -= rc.q[i];
      }
      for (int64_t l = 0; l < rd.feature_size; l++) {
        int group = cfg.group_index[offset + l];
        const Real w_old = fm.w[offset + l];
        Real lambda = hyper.lambda_w[group], mu = hyper.mu_w[group];
        Real square_term = 0;
        for (int64_t p = rc.X_t.ptr[l]; p < rc.X_t.ptr[l + 1]; p++)
          square_term += (rc.X_t.val[p] * rc.X_t.val[p]) * rc.cardinality[rc.X_t.idx[p]];
        Real dot = 0;
        for (int64_t p = rc.X_t.ptr[l]; p < rc.X_t.ptr[l + 1]; p++) dot += rc.X_t.val[p] * rc.e[rc.X_t.idx[p]];
        Real linear_term = -dot;
        linear_term += square_term * w_old;
        square_term = lambda + hyper.alpha * square_term;
        linear_term = hyper.alpha * linear_term + lambda * mu;
        Real w_new = sample_normal(square_term, linear_term);
        fm.w[offset + l] = w_new;
        for (int64_t p = rc.X_t.ptr[l]; p < rc.X_t.ptr[l + 1]; p++)
          rc.e[rc.X_t.idx[p]] += (rc.X_t.val[p] * rc.cardinality[rc.X_t.idx[p]]) * (w_new - w_old);
      }
      spmv(rd.X, fm.w.data() + offset, rc.q.data());
      for (int64_t t = 0; t < n_train; t++) e_train[t] += rc.q[rd.original_to_block[t]];
      offset += rd.feature_size;
    }
  }

  // FMTrainer.hpp:319-483, one factor
  void update_V_factor(int f) {
    const int G = cfg.n_groups;
    Real *Vf = fm.V.data() + (size_t)f * fm.D;
    const Real *lamf = hyper.lambda_V.data() + (size_t)f * G;
    const Real *muf = hyper.mu_V.data() + (size_t)f * G;
    // :320
    spmv(X, Vf, q_train.data());
    // :323-340
    {
      int64_t offset = X.cols;
      for (size_t r = 0; r < relations.size(); r++) {
        const RelationBlock &rd = relations[r];
        RelationWiseCache &rc = relation_caches[r];
        spmv(rd.X, Vf + offset, rc.q.data());
        for (int64_t t = 0; t < n_train; t++) q_train[t] += rc.q[rd.original_to_block[t]];
        offset += rd.feature_size;
      }
    }
    // main table, :343-376
    for (int64_t j = 0; j < X_t.rows; j++) {
      int g = cfg.group_index[j];
      Real v_old = Vf[j];
      Real square_coeff = 0, linear_coeff = 0;
      for (int64_t p = X_t.ptr[j]; p < X_t.ptr[j + 1]; p++) {
        int64_t t = X_t.idx[p];
        Real x = X_t.val[p];
        Real h = x * (q_train[t] - x * v_old);
        square_coeff += h * h;
        linear_coeff += (-e_train[t]) * h;
      }
      linear_coeff += square_coeff * v_old;
      square_coeff *= hyper.alpha;
      linear_coeff *= hyper.alpha;
      square_coeff += lamf[g];
      linear_coeff += lamf[g] * muf[g];
      Real v_new = sample_normal(square_coeff, linear_coeff);
      if (trace_on) {
        trace.push_back(f);
        trace.push_back((Real)j);
        trace.push_back(square_coeff);
        trace.push_back(linear_coeff);
        trace.push_back(v_new);
      }
      Vf[j] = v_new;
      for (int64_t p = X_t.ptr[j]; p < X_t.ptr[j + 1]; p++) {
        int64_t t = X_t.idx[p];
        Real x = X_t.val[p];
        Real h = x * (q_train[t] - x * v_old);
        q_train[t] += x * (v_new - v_old);
        e_train[t] += h * (v_new - v_old);
      }
    }
    // relations, :378-482
    int64_t offset = X.cols;
    for (size_t r = 0; r < relations.size(); r++) {
      const RelationBlock &rd = relations[r];
      RelationWiseCache &rc = relation_caches[r];
      spmv_sq(rd.X, Vf + offset, rc.q_S.data());  // :388-393
      std::fill(rc.c.begin(), rc.c.end(), 0);
      std::fill(rc.c_S.begin(), rc.c_S.end(), 0);
      std::fill(rc.e.begin(), rc.e.end(), 0);
      std::fill(rc.e_q.begin(), rc.e_q.end(), 0);
      for (int64_t t = 0; t < n_train; t++) {  // :401-417
        int64_t i = rd.original_to_block[t];
        Real temp = (q_train[t] - rc.q[i]);
        rc.c[i] += temp;
        rc.c_S[i] += temp * temp;
        rc.e[i] += e_train[t];
        rc.e_q[i] += e_train[t] * temp;
        q_train[t] -= rc.q[i];
        e_train[t] -= (q_train[t] * rc.q[i] + 0.5 * rc.q[i] * rc.q[i] - 0.5 * rc.q_S[i]);
      }
      for (int64_t l = 0; l < rd.feature_size; l++) {  // :419-470
        int g = cfg.group_index[offset + l];
        Real v_old = Vf[offset + l];
        Real square_coeff = 0, linear_coeff = 0;
        for (int64_t p = rc.X_t.ptr[l]; p < rc.X_t.ptr[l + 1]; p++) {
          int64_t i = rc.X_t.idx[p];
          Real x_il = rc.X_t.val[p];
          Real h_B = (rc.q[i] - x_il * v_old);
          Real h_squared = h_B * h_B * rc.cardinality[i] + 2 * rc.c[i] * h_B + rc.c_S[i];
          h_squared = x_il * x_il * h_squared;
          square_coeff += h_squared;
          linear_coeff += (-rc.e[i] * h_B - rc.e_q[i]) * x_il;
        }
        linear_coeff += square_coeff * v_old;
        square_coeff *= hyper.alpha;
        linear_coeff *= hyper.alpha;
        square_coeff += lamf[g];
        linear_coeff += lamf[g] * muf[g];
        Real v_new = sample_normal(square_coeff, linear_coeff);
        if (trace_on) {
          trace.push_back(f);
          trace.push_back((Real)(offset + l));
          trace.push_back(square_coeff);
          trace.push_back(linear_coeff);
          trace.push_back(v_new);
        }
        Real delta = v_new - v_old;
        Vf[offset + l] = v_new;
        for (int64_t p = rc.X_t.ptr[l]; p < rc.X_t.ptr[l + 1]; p++) {
          int64_t i = rc.X_t.idx[p];
          const Real x_il = rc.X_t.val[p];
          Real h_B = rc.q[i] - x_il * v_old;
          rc.q[i] += delta * x_il;
          rc.q_S[i] += delta * (v_new + v_old) * x_il * x_il;
          rc.e[i] += x_il * delta * (h_B * rc.cardinality[i] + rc.c[i]);
          rc.e_q[i] += x_il * delta * (h_B * rc.c[i] + rc.c_S[i]);
        }
      }
      for (int64_t t = 0; t < n_train; t++) {  // :473-480
        int64_t i = rd.original_to_block[t];
        e_train[t] += (q_train[t] * rc.q[i] + 0.5 * rc.q[i] * rc.q[i] - 0.5 * rc.q_S[i]);
        q_train[t] += rc.q[i];
      }
      offset += rd.feature_size;
    }
  }
  void update_V() {
    for (int f = 0; f < fm.n_factors; f++) update_V_factor(f);
  }

  // FMTrainer.hpp:493-522
  void update_e() {
    fm.predict_score_write_target(e_train.data(), X, relations);
    if (cfg.task == REGRESSION) {
      for (int64_t t = 0; t < n_train; t++) e_train[t] -= y[t];
    } else if (cfg.task == CLASSIFICATION) {
      Real zero = 0, std_ = 1;
      for (int64_t t = 0; t < n_train; t++) {
        Real gt = y[t], pred = e_train[t], nn;
        if (gt > 0)
          nn = sample_truncated_normal_left(gen_, pred, std_, zero);
        else
          nn = sample_truncated_normal_right(gen_, pred, std_, zero);
        e_train[t] -= nn;
      }
    } else {
      int i = 0;
      for (auto &s : cutpoint_sampler) {
        s.step();
        OprobitSampler::alpha_to_gamma(fm.cutpoints[i], s.alpha_now);
        s.sample_z_given_cutpoint();
        i++;
      }
    }
  }

  // BaseFMTrainer.hpp:135-152
  void update_all() {
    update_alpha();
    update_w0();
    update_lambda_w();
    update_mu_w();
    update_w();
    update_lambda_V();
    update_mu_V();
    update_V();
    update_e();
  }
};

}  // namespace orc

// =======================================================================================
// C interface for ctypes (tests/, bench.py cpu_baseline, __graft_entry__.smoke only).
// =======================================================================================
static thread_local std::string g_err;
#define ORC_TRY try {
#define ORC_CATCH(ret)                 \
  }                                    \
  catch (std::invalid_argument & ex) { \
    g_err = ex.what();                 \
    return ret == 0 ? -2 : ret;        \
  }                                    \
  catch (std::exception & ex) {        \
    g_err = ex.what();                 \
    return ret == 0 ? -1 : ret;        \
  }

static orc::Csr make_csr(int64_t rows, int64_t cols, const int64_t *indptr, const int32_t *indices, const double *data) {
  orc::Csr X;
  X.rows = rows;
  X.cols = cols;
  X.ptr.assign(indptr, indptr + rows + 1);
  int64_t nnz = indptr[rows];
  X.idx.assign(indices, indices + nnz);
  X.val.assign(data, data + nnz);
  return X;
}

extern "C" {

const char *orc_last_error() { return g_err.c_str(); }

orc::Trainer *orc_new() { return new orc::Trainer(); }
void orc_free(orc::Trainer *t) { delete t; }
orc::Trainer *orc_clone(const orc::Trainer *t) { return new orc::Trainer(*t); }

int orc_set_main(orc::Trainer *t, int64_t N, int64_t D0, const int64_t *indptr, const int32_t *indices, const double *data,
                 const double *y, int64_t ny) {
  ORC_TRY
  t->X = make_csr(N, D0, indptr, indices, data);
  t->y.assign(y, y + ny);
  return 0;
  ORC_CATCH(0)
}

// definitions.hpp:34-42
int orc_add_block(orc::Trainer *t, int64_t B, int64_t Db, const int64_t *indptr, const int32_t *indices, const double *data,
                  const int64_t *map, int64_t n_map) {
  ORC_TRY
  orc::RelationBlock rb;
  rb.X = make_csr(B, Db, indptr, indices, data);
  rb.block_size = B;
  rb.feature_size = Db;
  rb.original_to_block.assign(map, map + n_map);
  for (auto c : rb.original_to_block)
    if (c < 0 || c >= B) throw std::runtime_error("index mapping points to non-existing row.");
  t->relations.push_back(std::move(rb));
  return 0;
  ORC_CATCH(0)
}

int orc_set_config(orc::Trainer *t, int task, double alpha_0, double beta_0, double gamma_0, double mu_0, double reg_0,
                   int fit_w0, int fit_linear, double nu_oprobit, const int32_t *group_index, int64_t n_group_index) {
  ORC_TRY
  orc::Config &c = t->cfg;
  c.task = task;
  c.alpha_0 = alpha_0;
  c.beta_0 = beta_0;
  c.gamma_0 = gamma_0;
  c.mu_0 = mu_0;
  c.reg_0 = reg_0;
  c.fit_w0 = fit_w0 != 0;
  c.fit_linear = fit_linear != 0;
  c.nu_oprobit = nu_oprobit;
  c.group_index.assign(group_index, group_index + n_group_index);
  c.finalize();
  return 0;
  ORC_CATCH(0)
}

int orc_add_cutpoint_group(orc::Trainer *t, int n_class, const int64_t *rows, int64_t n_rows) {
  ORC_TRY
  t->cfg.cutpoint_groups.emplace_back(n_class, std::vector<int64_t>(rows, rows + n_rows));
  return 0;
  ORC_CATCH(0)
}

// trainer ctor + create_FM + create_Hyper + initialize_hyper + initialize_e
int orc_start(orc::Trainer *t, int rank, double init_std, int seed) {
  ORC_TRY
  t->construct(seed);
  t->start(rank, init_std);
  return 0;
  ORC_CATCH(0)
}

int orc_step(orc::Trainer *t) {
  ORC_TRY
  t->update_all();
  return 0;
  ORC_CATCH(0)
}
// sub-steps, BaseFMTrainer.hpp:135-152 order: 0 alpha, 1 w0, 2 lambda_w, 3 mu_w, 4 w, 5 lambda_V, 6 mu_V, 7 V, 8 e
int orc_substep(orc::Trainer *t, int which) {
  ORC_TRY
  switch (which) {
    case 0: t->update_alpha(); break;
    case 1: t->update_w0(); break;
    case 2: t->update_lambda_w(); break;
    case 3: t->update_mu_w(); break;
    case 4: t->update_w(); break;
    case 5: t->update_lambda_V(); break;
    case 6: t->update_mu_V(); break;
    case 7: t->update_V(); break;
    case 8: t->update_e(); break;
    default: throw std::invalid_argument("bad substep");
  }
  return 0;
  ORC_CATCH(0)
}
int orc_update_V_factor(orc::Trainer *t, int f) {
  ORC_TRY
  t->update_V_factor(f);
  return 0;
  ORC_CATCH(0)
}

int64_t orc_dim_all(const orc::Trainer *t) { return t->dim_all; }
int orc_n_groups(const orc::Trainer *t) { return t->cfg.n_groups; }

void orc_get_fm(const orc::Trainer *t, double *w0, double *w, double *V) {
  *w0 = t->fm.w0;
  std::memcpy(w, t->fm.w.data(), t->fm.w.size() * sizeof(double));
  if (!t->fm.V.empty()) std::memcpy(V, t->fm.V.data(), t->fm.V.size() * sizeof(double));
}
void orc_set_fm(orc::Trainer *t, double w0, const double *w, const double *V) {
  t->fm.w0 = w0;
  std::memcpy(t->fm.w.data(), w, t->fm.w.size() * sizeof(double));
  if (!t->fm.V.empty()) std::memcpy(t->fm.V.data(), V, t->fm.V.size() * sizeof(double));
}
// alpha, mu_w[G], lambda_w[G], mu_V[G*K] (f-major: [f*G+g]), lambda_V[G*K]
void orc_get_hyper(const orc::Trainer *t, double *alpha, double *mu_w, double *lambda_w, double *mu_V, double *lambda_V) {
  const orc::Hyper &h = t->hyper;
  *alpha = h.alpha;
  std::memcpy(mu_w, h.mu_w.data(), h.mu_w.size() * sizeof(double));
  std::memcpy(lambda_w, h.lambda_w.data(), h.lambda_w.size() * sizeof(double));
  if (!h.mu_V.empty()) {
    std::memcpy(mu_V, h.mu_V.data(), h.mu_V.size() * sizeof(double));
    std::memcpy(lambda_V, h.lambda_V.data(), h.lambda_V.size() * sizeof(double));
  }
}
void orc_set_hyper(orc::Trainer *t, double alpha, const double *mu_w, const double *lambda_w, const double *mu_V,
                   const double *lambda_V) {
  orc::Hyper &h = t->hyper;
  h.alpha = alpha;
  std::memcpy(h.mu_w.data(), mu_w, h.mu_w.size() * sizeof(double));
  std::memcpy(h.lambda_w.data(), lambda_w, h.lambda_w.size() * sizeof(double));
  if (!h.mu_V.empty()) {
    std::memcpy(h.mu_V.data(), mu_V, h.mu_V.size() * sizeof(double));
    std::memcpy(h.lambda_V.data(), lambda_V, h.lambda_V.size() * sizeof(double));
  }
}
void orc_get_e(const orc::Trainer *t, double *e) { std::memcpy(e, t->e_train.data(), t->e_train.size() * sizeof(double)); }
void orc_get_q(const orc::Trainer *t, double *q) { std::memcpy(q, t->q_train.data(), t->q_train.size() * sizeof(double)); }
void orc_set_e(orc::Trainer *t, const double *e) { std::memcpy(t->e_train.data(), e, t->e_train.size() * sizeof(double)); }

int orc_n_cutpoint_groups(const orc::Trainer *t) { return (int)t->fm.cutpoints.size(); }
int orc_cutpoint_size(const orc::Trainer *t, int g) { return (int)t->fm.cutpoints[g].size(); }
void orc_get_cutpoints(const orc::Trainer *t, int g, double *out) {
  std::memcpy(out, t->fm.cutpoints[g].data(), t->fm.cutpoints[g].size() * sizeof(double));
}
int64_t orc_mh_accept(const orc::Trainer *t, int g) { return (int64_t)t->cutpoint_sampler[g].accept_count; }

// draws the trainer's generator exactly as FMTrainer.hpp:122-125 does (fresh distribution per
// draw) -- used on a clone to learn the z-stream the original will consume next.
void orc_rng_sample_normals(orc::Trainer *t, int64_t n, double *out) {
  for (int64_t i = 0; i < n; i++) out[i] = std::normal_distribution<double>(0, 1)(t->gen_);
}
double orc_rng_gamma(orc::Trainer *t, double shape, double scale) { return std::gamma_distribution<double>(shape, scale)(t->gen_); }
uint32_t orc_rng_raw(orc::Trainer *t) { return t->gen_(); }
// the generator's 624 state words followed by its position index (libstdc++ operator<< layout)
void orc_rng_state(const orc::Trainer *t, uint32_t *out625) {
  std::ostringstream os;
  os << t->gen_;
  std::istringstream is(os.str());
  for (int i = 0; i < 625; i++) {
    unsigned long v;
    is >> v;
    out625[i] = (uint32_t)v;
  }
}

// ... and back (tests: a small trainer replays the variate stream a large one is about to consume, without cloning it)
void orc_rng_set_state(orc::Trainer *t, const uint32_t *in625) {
  std::ostringstream os;
  for (int i = 0; i < 625; i++) os << (unsigned long)in625[i] << (i < 624 ? " " : "");
  std::istringstream is(os.str());
  is >> t->gen_;
}

void orc_trace_enable(orc::Trainer *t, int on) {
  t->trace_on = on != 0;
  t->trace.clear();
}
int64_t orc_trace_size(const orc::Trainer *t) { return (int64_t)t->trace.size(); }
void orc_trace_get(const orc::Trainer *t, double *out) { std::memcpy(out, t->trace.data(), t->trace.size() * sizeof(double)); }

// FM.hpp:47-52 on an arbitrary design (main CSR + blocks registered on a scratch trainer `d`)
int orc_predict_score(const orc::Trainer *d, double w0, const double *w, const double *V, int rank, double *out) {
  ORC_TRY
  orc::FM fm;
  fm.n_factors = rank;
  int64_t D = d->X.cols;
  for (auto &r : d->relations) D += r.feature_size;
  fm.D = D;
  fm.w0 = w0;
  fm.w.assign(w, w + D);
  fm.V.assign(V, V + (size_t)D * rank);
  fm.predict_score_write_target(out, d->X, d->relations);
  return 0;
  ORC_CATCH(0)
}

// util.hpp samplers + special functions, for unit tests
double orc_tn_left(uint32_t seed, double mu_minus) {
  std::mt19937 g(seed);
  return orc::sample_truncated_normal_left(g, mu_minus);
}
double orc_tn_twoside(uint32_t seed, double lo, double hi) {
  std::mt19937 g(seed);
  return orc::sample_truncated_normal_twoside(g, lo, hi);
}
void orc_tn_left_many(uint32_t seed, double mu_minus, int64_t n, double *out) {
  std::mt19937 g(seed);
  for (int64_t i = 0; i < n; i++) out[i] = orc::sample_truncated_normal_left(g, mu_minus);
}
void orc_tn_twoside_many(uint32_t seed, double lo, double hi, int64_t n, double *out) {
  std::mt19937 g(seed);
  for (int64_t i = 0; i < n; i++) out[i] = orc::sample_truncated_normal_twoside(g, lo, hi);
}
double orc_erfcx(double x) { return orc::erfcx(x); }

// OprobitSampler::operator() (OProbitSampler.hpp:389-463) on given scores x[n] and labels y[n], rows[n_rows] of one
// cutpoint group. gamma-space part (the row loop, :402-413): ll, dgamma[K-1], Hg[(K-1)^2]; and the full
// alpha-space result of operator(): returns -ll (with the prior), dalpha[K-1], Ha[(K-1)^2]. Any output may be NULL.
int orc_oprobit_eval(int n_class, const double *alpha, double reg, const double *x, const double *y, int64_t n,
                     const int64_t *rows, int64_t n_rows, double *ll_rows, double *dgamma, double *Hg, double *neg_ll,
                     double *dalpha, double *Ha) {
  ORC_TRY
  std::vector<double> xv(x, x + n), yv(y, y + n);
  std::vector<int64_t> idx(rows, rows + n_rows);
  std::mt19937 gen(0);
  orc::OprobitSampler s(xv, yv, n_class, idx, gen, reg, 5.0);
  const int m = n_class - 1;
  std::vector<double> a(alpha, alpha + m), gamma(m, 0.0), dg(m, 0.0), H((size_t)m * m, 0.0);
  orc::OprobitSampler::alpha_to_gamma(gamma, a);
  double ll = 0;
  s.eval_rows(gamma, ll, dg, &H);
  if (ll_rows) *ll_rows = ll;
  if (dgamma) std::copy(dg.begin(), dg.end(), dgamma);
  if (Hg) std::copy(H.begin(), H.end(), Hg);
  std::vector<double> da(m, 0.0), Hh((size_t)m * m, 0.0);
  double v = s.eval(a, da, &Hh);
  if (neg_ll) *neg_ll = v;
  if (dalpha) std::copy(da.begin(), da.end(), dalpha);
  if (Ha) std::copy(Hh.begin(), Hh.end(), Ha);
  return 0;
  ORC_CATCH(0)
}

}  // extern "C"
