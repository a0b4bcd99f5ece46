// mfm_chain_plan.hpp -- host planner of the STREAMED conflict-window chain (k_cs_stream, mfm_chain_stream.hpp): the block-feature
// sweep of a relation block whose state does not fit one CU's LDS (FMTrainer.hpp:276-302 for w, :419-470 for V) as ONE pipelined
// launch without batch boundaries. Pure host code (no HIP): the plan can be built and its data flow emulated on a CPU
// (cs_emulate, tests/test_chain_plan_cpu.py).
//
// The sweep visits the columns of the run one after the other; column k reads and rewrites the records of the block rows it has
// entries in. Time is counted in STEPS of Cg consecutive columns. An entry (column k, row r) is COLD when the row's touch before
// and its touch after are at least Lw steps away, else HOT:
//   * a hot row lives in the walker workgroup's LDS from the step of the first touch of its cluster (touches chained by step
//     distances < Lw) to the step of the last one: one wavefront walks the columns in order over their hot entries only;
//   * cold statistics of step v (S(v)) are taken from the records in global memory by the row-range workgroups, as soon as the
//     cold updates and the write-backs of step v - Lw (U(v - Lw)) are in; cold updates of step u need the walker's draws of step u.
// So the only cycle is  walker(s) <- S(s) <- U(s - Lw) <- walker(s - Lw):  a pipeline Lw steps deep in which nobody waits as long as
// a hop (walker -> ranges -> walker) takes less than Lw steps. Exact: every record sees the same updates in the same order as
// in the sequential sweep; only the association of the floating-point sums of a column's statistics differs (fixed, reproducible).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <string>
#include <vector>

namespace mfm {

constexpr int CS_MAX_CG = 4;       // columns per step
constexpr int CS_RING = 8;         // ring slots of everything the two sides exchange (>= Lw + 1)
constexpr int CS_MAX_LW = CS_RING - 1;
constexpr int CS_LCOL_SHIFT = 27;  // cold entry word: row | (column inside the step) << 27

struct CsParams {
  int Cg = 4;         // columns per step
  int Lw = 3;         // window in steps
  int NB = 16;        // row ranges (one workgroup each)
  int RD = 2;         // an LDS slot freed by an exit of step u may be taken by an enter of step >= u + RD
  int cap = 1900;     // LDS slots of the walker
};

struct CsPlanHost {
  CsParams prm;
  int n_cols = 0, n_steps = 0;
  int64_t n_rows = 0;
  int n_slots = 0;                    // slots in use at most
  int max_enter = 0, max_exit = 0;    // per step (ring slot sizes)
  int max_hot_col = 0;                // hot entries of a column at most
  int64_t n_cold = 0, n_hot = 0;
  // cold entries by (step, range), by column inside
  std::vector<int32_t> cold_ptr;      // [n_steps * NB + 1]
  std::vector<int32_t> cold_rc;       // row | lcol << CS_LCOL_SHIFT
  std::vector<double> cold_x;
  // rows entering / leaving the LDS by (step, range); position inside the step = ring position
  std::vector<int32_t> enter_ptr, enter_row, enter_slot;  // [n_steps * NB + 1], [..], [..]
  std::vector<int32_t> exit_ptr, exit_row, exit_slot;
  // hot entries by column position
  std::vector<int32_t> hot_ptr, hot_slot;  // [n_cols + 1], [..]
  std::vector<double> hot_x;
  std::string why;  // not built: the reason
  bool ok = false;
};

// csc_ptr / csc_idx / csc_val: the block's CSC (column j -> ascending rows); run: the chain's columns in sweep order.
// Returns a plan with ok = false when the hot rows do not fit `cap` slots (the caller tries a smaller window or another form).
inline CsPlanHost cs_build_plan(const int64_t *csc_ptr, const int32_t *csc_idx, const double *csc_val, int64_t n_rows,
                                const std::vector<int32_t> &run, const CsParams &prm) {
  CsPlanHost P;
  P.prm = prm;
  P.n_rows = n_rows;
  const int n = (int)run.size(), Cg = prm.Cg, Lw = prm.Lw, NB = prm.NB;
  P.n_cols = n;
  if (n < 1 || n_rows < 1 || Cg < 1 || Cg > CS_MAX_CG || Lw < 1 || Lw > CS_MAX_LW || NB < 1 || prm.RD < 1 ||
      n_rows >= ((int64_t)1 << CS_LCOL_SHIFT)) {
    P.why = "parameters";
    return P;
  }
  const int ns = (n + Cg - 1) / Cg;
  P.n_steps = ns;
  int64_t nnz = 0;
  for (int k = 0; k < n; k++) nnz += csc_ptr[run[k] + 1] - csc_ptr[run[k]];
  if (nnz >= (int64_t)2147483647) {
    P.why = "too many entries";
    return P;
  }
  // entries in sweep order: e = (k, p); prev / next touch of the entry's row as STEPS
  std::vector<int32_t> eptr((size_t)n + 1, 0);
  for (int k = 0; k < n; k++) eptr[k + 1] = eptr[k] + (int32_t)(csc_ptr[run[k] + 1] - csc_ptr[run[k]]);
  const int32_t NONE = std::numeric_limits<int32_t>::min() / 2;
  std::vector<int32_t> prev_step((size_t)nnz), next_step((size_t)nnz), last((size_t)n_rows, NONE);
  for (int k = 0; k < n; k++) {
    const int64_t b = csc_ptr[run[k]];
    for (int32_t q = eptr[k]; q < eptr[k + 1]; q++) {
      const int32_t r = csc_idx[b + (q - eptr[k])];
      prev_step[q] = last[r];
      last[r] = k / Cg;
    }
  }
  std::fill(last.begin(), last.end(), -NONE);
  for (int k = n - 1; k >= 0; k--) {
    const int64_t b = csc_ptr[run[k]];
    for (int32_t q = eptr[k + 1] - 1; q >= eptr[k]; q--) {
      const int32_t r = csc_idx[b + (q - eptr[k])];
      next_step[q] = last[r];
      last[r] = k / Cg;
    }
  }
  auto bucket = [&](int32_t r) { return (int)(((int64_t)r * NB) / n_rows); };
  // pass 1: counts per (step, range), enters / exits per (step, range), hot per column
  std::vector<int32_t> ccnt((size_t)ns * NB + 1, 0), encnt((size_t)ns * NB + 1, 0), excnt((size_t)ns * NB + 1, 0);
  P.hot_ptr.assign((size_t)n + 1, 0);
  for (int k = 0; k < n; k++) {
    const int s = k / Cg;
    const int64_t b = csc_ptr[run[k]];
    int hot = 0;
    for (int32_t q = eptr[k]; q < eptr[k + 1]; q++) {
      const int32_t r = csc_idx[b + (q - eptr[k])];
      const bool near_prev = s - prev_step[q] < Lw, near_next = next_step[q] - s < Lw;  // (a second touch inside the step: distance 0)
      if (!near_prev && !near_next) {
        ccnt[(size_t)s * NB + bucket(r) + 1]++;
      } else {
        hot++;
        if (!near_prev) encnt[(size_t)s * NB + bucket(r) + 1]++;
        if (!near_next) excnt[(size_t)s * NB + bucket(r) + 1]++;
      }
    }
    P.hot_ptr[k + 1] = P.hot_ptr[k] + hot;
    P.max_hot_col = std::max(P.max_hot_col, hot);
  }
  for (size_t i = 1; i < ccnt.size(); i++) {
    ccnt[i] += ccnt[i - 1];
    encnt[i] += encnt[i - 1];
    excnt[i] += excnt[i - 1];
  }
  P.cold_ptr = ccnt;
  P.enter_ptr = encnt;
  P.exit_ptr = excnt;
  P.n_cold = ccnt.back();
  P.n_hot = P.hot_ptr[n];
  for (int s = 0; s < ns; s++) {
    P.max_enter = std::max(P.max_enter, encnt[(size_t)(s + 1) * NB] - encnt[(size_t)s * NB]);
    P.max_exit = std::max(P.max_exit, excnt[(size_t)(s + 1) * NB] - excnt[(size_t)s * NB]);
  }
  // pass 2: slots (a stack of free slots; a slot freed at step u comes back at step u + RD) and the lists
  P.cold_rc.resize((size_t)P.n_cold);
  P.cold_x.resize((size_t)P.n_cold);
  P.enter_row.resize((size_t)encnt.back());
  P.enter_slot.resize((size_t)encnt.back());
  P.exit_row.resize((size_t)excnt.back());
  P.exit_slot.resize((size_t)excnt.back());
  P.hot_slot.resize((size_t)P.n_hot);
  P.hot_x.resize((size_t)P.n_hot);
  std::vector<int32_t> ccur(ccnt.begin(), ccnt.end() - 1), encur(encnt.begin(), encnt.end() - 1), excur(excnt.begin(), excnt.end() - 1);
  std::vector<int32_t> slot_of((size_t)n_rows, -1);
  std::vector<int32_t> free_slots;                          // usable now
  std::vector<std::vector<int32_t>> pending((size_t)prm.RD + 1);  // freed at step u: usable from u + RD on (ring by step)
  int next_new = 0;
  for (int s = 0; s < ns; s++) {
    {  // slots freed at step s - RD come back
      std::vector<int32_t> &back = pending[(size_t)s % (prm.RD + 1)];
      // (descending push: the lowest slot is popped first -- a deterministic, compact assignment)
      std::sort(back.begin(), back.end(), std::greater<int32_t>());
      free_slots.insert(free_slots.end(), back.begin(), back.end());
      back.clear();
    }
    std::vector<int32_t> &freed = pending[(size_t)(s + prm.RD) % (prm.RD + 1)];
    // enters of the step first (every column of the step): a row enters at the START of its first step
    for (int k = s * Cg; k < std::min(n, (s + 1) * Cg); k++) {
      const int64_t b = csc_ptr[run[k]];
      for (int32_t q = eptr[k]; q < eptr[k + 1]; q++) {
        const int32_t r = csc_idx[b + (q - eptr[k])];
        const bool near_prev = s - prev_step[q] < Lw, near_next = next_step[q] - s < Lw;
        if ((near_prev || near_next) && !near_prev) {
          int32_t sl;
          if (!free_slots.empty()) {
            sl = free_slots.back();
            free_slots.pop_back();
          } else {
            sl = next_new++;
          }
          slot_of[r] = sl;
          const int32_t at = encur[(size_t)s * NB + bucket(r)]++;
          P.enter_row[at] = r;
          P.enter_slot[at] = sl;
        }
      }
    }
    if (next_new > prm.cap) {
      P.why = "hot rows exceed the LDS slots";
      P.n_slots = next_new;
      return P;
    }
    for (int k = s * Cg; k < std::min(n, (s + 1) * Cg); k++) {
      const int64_t b = csc_ptr[run[k]];
      int32_t hq = P.hot_ptr[k];
      for (int32_t q = eptr[k]; q < eptr[k + 1]; q++) {
        const int32_t r = csc_idx[b + (q - eptr[k])];
        const double x = csc_val[b + (q - eptr[k])];
        const bool near_prev = s - prev_step[q] < Lw, near_next = next_step[q] - s < Lw;
        if (!near_prev && !near_next) {
          const int32_t at = ccur[(size_t)s * NB + bucket(r)]++;  // (columns of a step arrive in order: by column inside (step, range))
          P.cold_rc[at] = r | ((int32_t)(k - s * Cg) << CS_LCOL_SHIFT);
          P.cold_x[at] = x;
        } else {
          P.hot_slot[hq] = slot_of[r];
          P.hot_x[hq] = x;
          hq++;
          if (!near_next) {  // last touch of the cluster: the row leaves at the END of this step
            const int32_t at = excur[(size_t)s * NB + bucket(r)]++;
            P.exit_row[at] = r;
            P.exit_slot[at] = slot_of[r];
            freed.push_back(slot_of[r]);
            slot_of[r] = -1;
          }
        }
      }
    }
  }
  P.n_slots = next_new;
  P.ok = true;
  return P;
}

// What a plan with steps of Cg columns, a window of Lw steps and slot reuse delay RD would need, without building it: LDS slots, most
// hot entries of a column, hot entries in all (one pass over the entries; the touches are taken once per run by cs_touches).
struct CsTouches {
  std::vector<int32_t> eptr;        // [n + 1] entries per column position
  std::vector<int32_t> prev, next;  // per entry: column POSITION of the row's touch before / after (-2^30 / 2^30: none)
};
inline CsTouches cs_touches(const int64_t *csc_ptr, const int32_t *csc_idx, int64_t n_rows, const std::vector<int32_t> &run) {
  CsTouches T;
  const int n = (int)run.size();
  T.eptr.assign((size_t)n + 1, 0);
  for (int k = 0; k < n; k++) T.eptr[k + 1] = T.eptr[k] + (int32_t)(csc_ptr[run[k] + 1] - csc_ptr[run[k]]);
  const int32_t NONE = 1 << 30;
  T.prev.resize((size_t)T.eptr[n]);
  T.next.resize((size_t)T.eptr[n]);
  std::vector<int32_t> last((size_t)n_rows, -NONE);
  for (int k = 0; k < n; k++) {
    const int64_t b = csc_ptr[run[k]];
    for (int32_t q = T.eptr[k]; q < T.eptr[k + 1]; q++) {
      const int32_t r = csc_idx[b + (q - T.eptr[k])];
      T.prev[q] = last[r];
      last[r] = k;
    }
  }
  std::fill(last.begin(), last.end(), NONE);
  for (int k = n - 1; k >= 0; k--) {
    const int64_t b = csc_ptr[run[k]];
    for (int32_t q = T.eptr[k + 1] - 1; q >= T.eptr[k]; q--) {
      const int32_t r = csc_idx[b + (q - T.eptr[k])];
      T.next[q] = last[r];
      last[r] = k;
    }
  }
  return T;
}
struct CsNeed {
  int n_slots = 0, max_hot_col = 0;
  int64_t n_hot = 0;
};
inline CsNeed cs_estimate(const CsTouches &T, int Cg, int Lw, int RD) {
  CsNeed N;
  const int n = (int)T.eptr.size() - 1, ns = (n + Cg - 1) / Cg;
  const int32_t NONE = 1 << 30;
  std::vector<int32_t> en((size_t)ns + 1, 0), ex((size_t)ns + RD + 1, 0);
  for (int k = 0; k < n; k++) {
    const int s = k / Cg;
    int hot = 0;
    for (int32_t q = T.eptr[k]; q < T.eptr[k + 1]; q++) {
      const bool near_prev = T.prev[q] > -NONE && s - T.prev[q] / Cg < Lw, near_next = T.next[q] < NONE && T.next[q] / Cg - s < Lw;
      if (near_prev || near_next) {
        hot++;
        if (!near_prev) en[s]++;
        if (!near_next) ex[s + RD]++;  // (the slot comes back RD steps later)
      }
    }
    N.n_hot += hot;
    N.max_hot_col = std::max(N.max_hot_col, hot);
  }
  int live = 0;
  for (int s = 0; s < ns; s++) {
    live += en[s] - ex[s];
    N.n_slots = std::max(N.n_slots, live);
  }
  return N;
}

// The largest window (in steps) whose hot rows fit, for the given step width: Lw = lw_max, lw_max - 1, ... 1.
inline CsPlanHost cs_build_plan_fit(const int64_t *csc_ptr, const int32_t *csc_idx, const double *csc_val, int64_t n_rows,
                                    const std::vector<int32_t> &run, CsParams prm, int lw_max, int lw_min = 1) {
  CsPlanHost P;
  for (int lw = std::min(lw_max, CS_MAX_LW); lw >= lw_min; lw--) {
    prm.Lw = lw;
    P = cs_build_plan(csc_ptr, csc_idx, csc_val, n_rows, run, prm);
    if (P.ok) return P;
  }
  return P;
}

// ---- emulation of the launch's data flow on the host (tests) --------------------------------------------------------------------
// A toy policy on one double per row, q: statistics S = sum x q, draw new = f(S, k) (a fixed pseudo-random map), update
// q += x (new - old). The emulation runs the four actors in the EARLIEST order their flags allow -- after the walker's step j:
// X(j) (exits out), U(j) (cold updates + write-backs), S(j + Lw) (cold statistics + enter packs of step j + Lw), Y (enter staging
// as soon as slot reuse allows) -- on separate copies of what each actor can see (records in global memory, LDS slots, ring
// slots), so that a plan that lets an actor read something before it is final, or reuse a slot or a ring position too early,
// gives other numbers than the plain sequential sweep. Returns the largest relative difference of the coefficients.
inline double cs_emulate(const CsPlanHost &P, const int64_t *csc_ptr, const int32_t *csc_idx, const double *csc_val,
                         const std::vector<int32_t> &run, std::vector<double> q0, std::string *err = nullptr) {
  const int n = P.n_cols, ns = P.n_steps, Cg = P.prm.Cg, Lw = P.prm.Lw, NB = P.prm.NB, RD = P.prm.RD, R = CS_RING;
  auto draw = [](double S, int k) { return std::sin(S * 1.7 + 0.3 * k) * 0.9 + 0.05 * std::cos(0.11 * k); };
  std::vector<double> theta0((size_t)n);
  for (int k = 0; k < n; k++) theta0[k] = 0.2 * std::cos(0.37 * k);
  // reference: the sequential sweep
  std::vector<double> qr = q0, tr = theta0;
  for (int k = 0; k < n; k++) {
    const int64_t b = csc_ptr[run[k]], e = csc_ptr[run[k] + 1];
    double S = 0;
    for (int64_t p = b; p < e; p++) S += csc_val[p] * qr[csc_idx[p]];
    const double fresh = draw(S, k);
    for (int64_t p = b; p < e; p++) qr[csc_idx[p]] += csc_val[p] * (fresh - tr[k]);
    tr[k] = fresh;
  }
  // the pipeline
  const double NaN = std::numeric_limits<double>::quiet_NaN();
  std::vector<double> G = q0;                                  // records in global memory
  std::vector<double> lds((size_t)std::max(P.n_slots, 1), NaN);  // walker LDS
  std::vector<double> in_ring((size_t)R * std::max(P.max_enter, 1), NaN), out_ring((size_t)R * std::max(P.max_exit, 1), NaN);
  std::vector<double> part((size_t)R * NB * Cg, NaN), oldnew((size_t)R * Cg * 2, NaN);
  std::vector<double> csum((size_t)R * Cg, NaN);               // staged by Y
  std::vector<double> th = theta0;
  int s_done = 0, u_done = 0, y_done = 0, x_done = 0;
  auto fail = [&](const char *m) {
    if (err) *err = m;
    return 1e300;
  };
  auto run_S = [&](int v) {  // needs U(v - Lw)
    if (v - Lw >= 0 && u_done < v - Lw + 1) return false;
    for (int b = 0; b < NB; b++) {
      double acc[CS_MAX_CG] = {0};
      for (int32_t p = P.cold_ptr[(size_t)v * NB + b]; p < P.cold_ptr[(size_t)v * NB + b + 1]; p++) {
        const int32_t r = P.cold_rc[p] & ((1 << CS_LCOL_SHIFT) - 1), lc = P.cold_rc[p] >> CS_LCOL_SHIFT;
        acc[lc] += P.cold_x[p] * G[r];
      }
      for (int c = 0; c < Cg; c++) part[((size_t)(v % R) * NB + b) * Cg + c] = acc[c];
      const int32_t e0 = P.enter_ptr[(size_t)v * NB];
      for (int32_t e = P.enter_ptr[(size_t)v * NB + b]; e < P.enter_ptr[(size_t)v * NB + b + 1]; e++)
        in_ring[(size_t)(v % R) * std::max(P.max_enter, 1) + (e - e0)] = G[P.enter_row[e]];
    }
    s_done = v + 1;
    return true;
  };
  auto run_Y = [&](int s) {  // needs S(s) of every range and the exits of steps <= s - RD copied out
    if (s_done < s + 1 || (s - RD >= 0 && x_done < s - RD + 1)) return false;
    const int32_t e0 = P.enter_ptr[(size_t)s * NB];
    for (int32_t e = e0; e < P.enter_ptr[(size_t)(s + 1) * NB]; e++) lds[P.enter_slot[e]] = in_ring[(size_t)(s % R) * std::max(P.max_enter, 1) + (e - e0)];
    for (int c = 0; c < Cg; c++) {
      double t = 0;
      for (int b = 0; b < NB; b++) t += part[((size_t)(s % R) * NB + b) * Cg + c];
      csum[(size_t)(s % R) * Cg + c] = t;
    }
    y_done = s + 1;
    return true;
  };
  for (int v = 0; v < std::min(Lw, ns); v++)
    if (!run_S(v)) return fail("S(v < Lw) blocked");
  for (int s = 0; s < std::min(RD, ns); s++)
    if (!run_Y(s)) {
      // (RD > Lw: Y of these steps waits for S; run it on demand below)
      break;
    }
  for (int j = 0; j < ns; j++) {
    while (y_done < j + 1)
      if (!run_Y(y_done)) return fail("walker blocked: enters / statistics of its step cannot be staged");
    // walker(j)
    for (int k = j * Cg; k < std::min(n, (j + 1) * Cg); k++) {
      double S = csum[(size_t)(j % R) * Cg + (k - j * Cg)];
      for (int32_t h = P.hot_ptr[k]; h < P.hot_ptr[k + 1]; h++) S += P.hot_x[h] * lds[P.hot_slot[h]];
      const double fresh = draw(S, k);
      for (int32_t h = P.hot_ptr[k]; h < P.hot_ptr[k + 1]; h++) lds[P.hot_slot[h]] += P.hot_x[h] * (fresh - th[k]);
      oldnew[((size_t)(j % R) * Cg + (k - j * Cg)) * 2] = th[k];
      oldnew[((size_t)(j % R) * Cg + (k - j * Cg)) * 2 + 1] = fresh;
      th[k] = fresh;
    }
    {  // X(j): exits out (the slots become garbage for everybody else)
      const int32_t x0 = P.exit_ptr[(size_t)j * NB];
      for (int32_t x = x0; x < P.exit_ptr[(size_t)(j + 1) * NB]; x++) {
        out_ring[(size_t)(j % R) * std::max(P.max_exit, 1) + (x - x0)] = lds[P.exit_slot[x]];
        lds[P.exit_slot[x]] = NaN;
      }
      x_done = j + 1;
    }
    {  // U(j)
      const int32_t x0 = P.exit_ptr[(size_t)j * NB];
      for (int b = 0; b < NB; b++) {
        for (int32_t p = P.cold_ptr[(size_t)j * NB + b]; p < P.cold_ptr[(size_t)j * NB + b + 1]; p++) {
          const int32_t r = P.cold_rc[p] & ((1 << CS_LCOL_SHIFT) - 1), lc = P.cold_rc[p] >> CS_LCOL_SHIFT;
          const double *on = &oldnew[((size_t)(j % R) * Cg + lc) * 2];
          G[r] += P.cold_x[p] * (on[1] - on[0]);
        }
        for (int32_t x = P.exit_ptr[(size_t)j * NB + b]; x < P.exit_ptr[(size_t)j * NB + b + 1]; x++)
          G[P.exit_row[x]] = out_ring[(size_t)(j % R) * std::max(P.max_exit, 1) + (x - x0)];
      }
      u_done = j + 1;
    }
    if (j + Lw < ns && !run_S(j + Lw)) return fail("S blocked behind U");
    while (y_done < ns && run_Y(y_done)) {
    }
  }
  double worst = 0;
  for (int k = 0; k < n; k++) worst = std::max(worst, std::fabs(th[k] - tr[k]) / std::max(1.0, std::fabs(tr[k])));
  for (int64_t r = 0; r < P.n_rows; r++) {
    const double d = std::fabs(G[r] - qr[r]) / std::max(1.0, std::fabs(qr[r]));
    if (!(d <= worst)) worst = std::isnan(d) ? 1e300 : d;
  }
  return worst;
}

}  // namespace mfm
