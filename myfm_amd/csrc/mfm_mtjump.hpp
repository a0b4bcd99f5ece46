// mfm_mtjump.hpp -- jump-ahead polynomials for MT19937 (host side, computed once per context).
//
// The word sequence x_m of std::mt19937 is linear over GF(2): every bit sequence satisfies the recurrence
// whose characteristic polynomial phi(x) (degree 19937) is that of the state transition. Hence for any J
//     x_{m+J} = XOR_{i : g_i = 1} x_{m+i},   g(x) = x^J mod phi(x),  deg g < 19937,
// which lets workgroup p of the parallel generator (k_mt_generate_par, mfm_rng.hpp) start p * MT_PAR_BLOCKS
// blocks ahead of the stream position from 33 consecutive blocks of the sequence instead of walking there
// (Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer: "Efficient jump ahead for F2-linear random number
// generators", 2008 -- the plain polynomial form, no sliding window). phi is obtained with Berlekamp-Massey
// from 2 * 19937 bits of one output bit of the generator itself, so nothing is tabulated.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <future>
#include <thread>
#include <mutex>
#include <vector>

namespace mfm {
namespace mtjump {

constexpr int L = 19937;            // degree of phi
constexpr int W = (2 * L + 63) / 64 + 1;  // words for a product
constexpr int WL = (L + 64) / 64;   // words for a polynomial of degree <= L

typedef std::vector<uint64_t> Poly;  // bit i = coefficient of x^i

static inline bool bit(const Poly &p, int i) { return (p[(size_t)i >> 6] >> (i & 63)) & 1u; }
static inline void flip(Poly &p, int i) { p[(size_t)i >> 6] ^= (uint64_t)1 << (i & 63); }

// dst ^= src << shift   (src has n words; dst must hold n + shift/64 + 1 words)
static inline void xor_shifted(uint64_t *dst, const uint64_t *src, int n, int shift) {
  const int ws = shift >> 6, bs = shift & 63;
  if (bs == 0) {
    for (int i = 0; i < n; i++) dst[i + ws] ^= src[i];
  } else {
    uint64_t carry = 0;
    for (int i = 0; i < n; i++) {
      dst[i + ws] ^= (src[i] << bs) | carry;
      carry = src[i] >> (64 - bs);
    }
    dst[n + ws] ^= carry;
  }
}

// plain MT19937 word sequence (untempered state words), libstdc++ / reference recurrence
struct HostMt {
  uint32_t s[624];
  int p = 624;
  explicit HostMt(uint32_t seed) {
    s[0] = seed;
    for (int i = 1; i < 624; i++) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + (uint32_t)i;
  }
  uint32_t next_word() {
    if (p == 624) {
      for (int k = 0; k < 624; k++) {
        const uint32_t y = (s[k] & 0x80000000u) | (s[(k + 1) % 624] & 0x7fffffffu);
        s[k] = s[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      p = 0;
    }
    return s[p++];
  }
};

// characteristic polynomial (bits 0..L, bit L set) by Berlekamp-Massey on one bit of the word sequence
static inline Poly characteristic_polynomial() {
  const int N = 2 * L + 64;
  HostMt g(5489u);
  std::vector<uint8_t> s((size_t)N);
  for (int i = 0; i < 2000; i++) (void)g.next_word();  // well past the seeding block
  for (int i = 0; i < N; i++) s[i] = (uint8_t)(g.next_word() & 1u);
  const int nw = (N + 63) / 64 + 1;
  Poly C((size_t)nw, 0), B((size_t)nw, 0), T, R((size_t)nw, 0);  // R bit k = s[n - k]
  C[0] = B[0] = 1;
  int Lc = 0, m = 1;
  for (int n = 0; n < N; n++) {
    // R <<= 1, insert s[n]
    uint64_t carry = s[n];
    const int rw = std::min(nw, (n >> 6) + 2);
    for (int i = 0; i < rw; i++) {
      const uint64_t nc = R[i] >> 63;
      R[i] = (R[i] << 1) | carry;
      carry = nc;
    }
    uint64_t acc = 0;
    const int cw = (Lc >> 6) + 1;
    for (int i = 0; i < cw; i++) acc ^= C[i] & R[i];
    if (__builtin_parityll(acc) == 0) {
      m++;
    } else if (2 * Lc <= n) {
      T = C;
      xor_shifted(C.data(), B.data(), nw - (m >> 6) - 1, m);
      Lc = n + 1 - Lc;
      B.swap(T);
      m = 1;
    } else {
      xor_shifted(C.data(), B.data(), nw - (m >> 6) - 1, m);
      m++;
    }
  }
  Poly phi((size_t)WL + 1, 0);
  if (Lc != L) return Poly();  // (cannot happen for MT19937)
  for (int k = 0; k <= L; k++)
    if (bit(C, L - k)) flip(phi, k);
  return phi;
}

struct Field {
  Poly phi;  // bits 0..L
  bool ok() const { return !phi.empty(); }
  void reduce(Poly &prod) const {  // prod: < 2L bits -> < L bits
    for (int d = 2 * L - 2; d >= L; d--)
      if (bit(prod, d)) xor_shifted(prod.data(), phi.data(), WL, d - L);
  }
  Poly mulmod(const Poly &a, const Poly &b) const {
    Poly prod((size_t)W + 2, 0);
    for (int w = 0; w < WL; w++) {
      uint64_t aw = a[w];
      while (aw) {
        const int i = __builtin_ctzll(aw);
        xor_shifted(prod.data(), b.data(), WL, w * 64 + i);
        aw &= aw - 1;
      }
    }
    reduce(prod);
    prod.resize((size_t)WL + 1);
    return prod;
  }
  Poly xpow(uint64_t J) const {  // x^J mod phi
    Poly r((size_t)WL + 1, 0);
    r[0] = 1;
    for (int b = 63; b >= 0; b--) {
      if ((J >> b) == 0) continue;
      r = mulmod(r, r);
      if ((J >> b) & 1u) {
        Poly t((size_t)WL + 2, 0);
        xor_shifted(t.data(), r.data(), WL, 1);
        if (bit(t, L)) xor_shifted(t.data(), phi.data(), WL, 0);
        t.resize((size_t)WL + 1);
        r.swap(t);
      }
    }
    return r;
  }
};

constexpr int JUMP_WORDS32 = 624;  // 19968 bits per table entry (device layout: uint32 words, bit i of entry)

// entries p = 1 .. n_entries: g_p = x^((p * blocks_per_wg - 1) * 624) mod phi, as 624 uint32 words each (entry 0 unused)
static inline bool build_jump_table(int blocks_per_wg, int n_entries, std::vector<uint32_t> &out) {
  Field F;
  F.phi = characteristic_polynomial();
  if (!F.ok()) return false;
  out.assign((size_t)(n_entries + 1) * JUMP_WORDS32, 0u);
  if (n_entries < 1) return true;
  // g_(p+1) = g_p h with h = x^(blocks_per_wg * 624): T independent chains g_(p+T) = g_p h^T, one per thread (a product of
  // two 19937-bit polynomials mod phi costs ~3 ms)
  Poly h, g1;
  {
    std::thread th([&]() { h = F.xpow((uint64_t)blocks_per_wg * 624u); });
    g1 = F.xpow((uint64_t)(blocks_per_wg - 1) * 624u);
    th.join();
  }
  const int hw = (int)std::thread::hardware_concurrency();
  const int T = std::max(1, std::min({n_entries, 8, hw > 0 ? hw : 1}));
  std::vector<Poly> seed((size_t)T);
  seed[0] = g1;
  Poly hT = h;
  for (int i = 1; i < T; i++) {
    seed[i] = F.mulmod(seed[i - 1], h);
    hT = F.mulmod(hT, h);
  }
  auto emit = [&](int p, const Poly &g) {
    uint32_t *dst = out.data() + (size_t)p * JUMP_WORDS32;
    for (int w = 0; w < JUMP_WORDS32 / 2; w++) {
      const uint64_t v = w < (int)g.size() ? g[w] : 0;
      dst[2 * w] = (uint32_t)v;
      dst[2 * w + 1] = (uint32_t)(v >> 32);
    }
  };
  auto chain = [&](int i) {
    Poly g = seed[i];
    for (int p = 1 + i; p <= n_entries; p += T) {
      emit(p, g);
      if (p + T <= n_entries) g = F.mulmod(g, hT);
    }
  };
  std::vector<std::thread> pool;
  for (int i = 1; i < T; i++) pool.emplace_back(chain, i);
  chain(0);
  for (auto &t : pool) t.join();
  return true;
}

// Process-wide cache: the table for n entries is a prefix of the table for more (entry p depends on p only), and building
// it costs ~0.3 s of host time (Berlekamp-Massey + n polynomial products). prefetch() starts the computation on a helper
// thread (mfm_finalize calls it as soon as the problem size is known); get() waits for it / extends it.
struct JumpCache {
  std::mutex mu, pmu;  // mu: the table; pmu: the `pending` future
  int blocks = 0, n = -1;
  std::vector<uint32_t> tab;
  std::future<void> pending;
  int pend_blocks = 0, pend_n = -1;  // what `pending` computes (under pmu)
  static JumpCache &inst() {
    static JumpCache c;
    return c;
  }
  void compute(int blocks_per_wg, int n_entries) {
    std::vector<uint32_t> t;
    const bool ok = build_jump_table(blocks_per_wg, n_entries, t);
    std::lock_guard<std::mutex> g(mu);
    if (ok && (blocks != blocks_per_wg || n_entries > n)) {
      blocks = blocks_per_wg;
      n = n_entries;
      tab.swap(t);
    }
  }
  void prefetch(int blocks_per_wg, int n_entries) {
    {
      std::lock_guard<std::mutex> g(mu);
      if (blocks == blocks_per_wg && n >= n_entries) return;
    }
    // `pending` is only touched under pmu (two contexts finalizing at once, or a prefetch racing get()); the wait itself
    // runs on a moved-out future, outside the lock
    std::unique_lock<std::mutex> pl(pmu);
    if (pending.valid()) {
      if (pend_blocks == blocks_per_wg && pend_n >= n_entries) return;  // (already being computed: mfm_rng_prepare, then mfm_finalize)
      std::future<void> prev = std::move(pending);
      pl.unlock();
      prev.wait();
      pl.lock();
      {
        std::lock_guard<std::mutex> g(mu);
        if (blocks == blocks_per_wg && n >= n_entries) return;
      }
    }
    if (!pending.valid()) {
      pend_blocks = blocks_per_wg;
      pend_n = n_entries;
      pending = std::async(std::launch::async, [this, blocks_per_wg, n_entries]() { compute(blocks_per_wg, n_entries); });
    }
  }
  void wait_pending() {
    std::future<void> prev;
    {
      std::lock_guard<std::mutex> pl(pmu);
      if (pending.valid()) prev = std::move(pending);
    }
    if (prev.valid()) prev.wait();
  }
  bool get(int blocks_per_wg, int n_entries, std::vector<uint32_t> &out) {
    wait_pending();
    {
      std::lock_guard<std::mutex> g(mu);
      if (blocks == blocks_per_wg && n >= n_entries) {
        out.assign(tab.begin(), tab.begin() + (size_t)(n_entries + 1) * JUMP_WORDS32);
        return true;
      }
    }
    compute(blocks_per_wg, n_entries);
    std::lock_guard<std::mutex> g(mu);
    if (blocks != blocks_per_wg || n < n_entries) return false;
    out.assign(tab.begin(), tab.begin() + (size_t)(n_entries + 1) * JUMP_WORDS32);
    return true;
  }
};

}  // namespace mtjump
}  // namespace mfm
