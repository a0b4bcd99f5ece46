// Bulk standard normals from a std::mt19937, bit for bit what ONE persistent libstdc++ std::normal_distribution<double>
// returns call after call (FM::initialize_weight, FM.hpp:34-45: V, then w, then w0 from the same distribution object), but
// produced the way the device engine of mfm_rng.hpp produces its variates: the raw 32-bit stream in bulk, the polar attempts
// evaluated side by side on host threads, the accepted ones compacted in order.
//
// libstdc++ (bits/random.tcc): an attempt draws x = 2 c() - 1, y = 2 c() - 1 with c = generate_canonical<double, 53> (two
// engine outputs: (u0 + u1 2^32) / 2^64, clamped below 1) until 0 < r2 = x^2 + y^2 <= 1; mult = sqrt(-2 log(r2) / r2); the
// call returns y mult and keeps x mult for the next call. Attempt a therefore uses outputs [4a, 4a + 4) of the stream,
// whatever the attempts before it decided -- which is what makes them independent.
//
// Weight initialisation of config 5 (D K = 35 M normals) took 0.85 s through std::normal_distribution (x87 long double in
// generate_canonical, one engine call per word); this takes ~0.15 s on 8 threads.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <sstream>
#include <thread>
#include <vector>

namespace mfm_hostnormals {

// the mt19937 recurrence on a plain array (same constants as std::mt19937; the compiler vectorises the three loops)
struct Mt {
  static constexpr int N = 624, M = 397;
  uint32_t s[N];
  int pos = N;  // outputs taken from the current block (N: regenerate before the next output)
  void from(const std::mt19937 &g) {
    std::ostringstream os;
    os << g;
    std::istringstream is(os.str());
    for (int i = 0; i < N; i++) {
      unsigned long x;
      is >> x;
      s[i] = (uint32_t)x;
    }
    unsigned long p;
    is >> p;
    pos = (int)p;
  }
  void to(std::mt19937 &g) const {
    std::ostringstream os;
    for (int i = 0; i < N; i++) os << s[i] << ' ';
    os << pos;
    std::istringstream is(os.str());
    is >> g;
  }
  void twist() {
    auto mix = [](uint32_t a, uint32_t b) {
      const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
      return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    };
    for (int k = 0; k < N - M; k++) s[k] = s[k + M] ^ mix(s[k], s[k + 1]);
    for (int k = N - M; k < N - 1; k++) s[k] = s[k + M - N] ^ mix(s[k], s[k + 1]);
    s[N - 1] = s[M - 1] ^ mix(s[N - 1], s[0]);
    pos = 0;
  }
  static inline uint32_t temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
  // the next n outputs
  void fill(uint32_t *out, size_t n) {
    size_t i = 0;
    while (i < n) {
      if (pos >= N) twist();
      const size_t take = std::min(n - i, (size_t)(N - pos));
      for (size_t k = 0; k < take; k++) out[i + k] = temper(s[pos + k]);
      pos += (int)take;
      i += take;
    }
  }
  // move back by n outputs inside what fill() just produced is not possible in general: callers regenerate from a saved copy
};

inline double canonical(uint32_t u0, uint32_t u1) {
  const double sum = (double)u0 + (double)u1 * 4294967296.0;
  double r = sum / 18446744073709551616.0;
  if (r >= 1.0) r = std::nextafter(1.0, 0.0);
  return r;
}

// out[0 .. count): what `std::normal_distribution<double> nd; for (i) out[i] = nd(gen) * scale;` gives; gen is left where
// that loop leaves it (a value the distribution would still hold in reserve when count is odd is dropped, as it is when the
// reference's distribution object goes out of scope).
inline void fill_normals(std::mt19937 &gen, double *out, size_t count, double scale, int n_threads = 0) {
  if (count == 0) return;
  if (n_threads <= 0) {
    n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 16) n_threads = 16;
  }
  Mt mt;
  mt.from(gen);
  const size_t pairs = (count + 1) / 2;
  constexpr size_t BLOCK = (size_t)1 << 18;  // attempts per round (4 MB of raw words: stays in L2 / L3)
  std::vector<uint32_t> raw(4 * BLOCK);
  std::vector<double> c0(BLOCK), c1(BLOCK);
  std::vector<uint8_t> ok(BLOCK);
  size_t done = 0;  // accepted attempts so far
  while (done < pairs) {
    // not more attempts than could be needed if every one were accepted, at least a few thousand: the stream position must
    // end exactly behind the last attempt USED, so a round never evaluates past what the remaining pairs can consume ...
    // unless some are rejected -- the surplus is handled below by regenerating from the saved state
    const size_t want = pairs - done;
    size_t na = std::min(BLOCK, want + want / 3 + 64);
    const Mt saved = mt;
    mt.fill(raw.data(), 4 * na);
    auto eval = [&](size_t a0, size_t a1) {
      for (size_t a = a0; a < a1; a++) {
        const uint32_t *u = raw.data() + 4 * a;
        const double x = 2.0 * canonical(u[0], u[1]) - 1.0;
        const double y = 2.0 * canonical(u[2], u[3]) - 1.0;
        const double r2 = x * x + y * y;
        const bool acc = !(r2 > 1.0 || r2 == 0.0);
        ok[a] = acc;
        if (acc) {
          const double mult = std::sqrt(-2 * std::log(r2) / r2);
          c0[a] = (y * mult) * 1.0 + 0.0;
          c1[a] = (x * mult) * 1.0 + 0.0;
        }
      }
    };
    const int nt = na < 4096 ? 1 : n_threads;
    if (nt == 1) {
      eval(0, na);
    } else {
      std::vector<std::thread> th;
      const size_t per = (na + nt - 1) / nt;
      for (int t = 0; t < nt; t++) {
        const size_t a0 = std::min(na, (size_t)t * per), a1 = std::min(na, a0 + per);
        if (a0 < a1) th.emplace_back(eval, a0, a1);
      }
      for (auto &t : th) t.join();
    }
    size_t used = na;  // attempts of this round that belong to the sequence
    for (size_t a = 0; a < na; a++) {
      if (!ok[a]) continue;
      const size_t i = 2 * done;
      out[i] = c0[a] * scale;
      if (i + 1 < count) out[i + 1] = c1[a] * scale;
      done++;
      if (done == pairs) {
        used = a + 1;
        break;
      }
    }
    if (used < na) {  // the stream stops behind the last attempt used: replay exactly that many outputs from the saved state
      mt = saved;
      size_t left = 4 * used;
      while (left) {
        const size_t take = std::min(left, raw.size());
        mt.fill(raw.data(), take);
        left -= take;
      }
    }
  }
  mt.to(gen);
}

}  // namespace mfm_hostnormals
