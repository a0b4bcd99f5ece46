// mfm_cell.hpp -- "cell path" of update_V (FMTrainer.hpp:315-482) for designs whose training rows are tuples of indices.
//
// When the main table is a row of unit-valued one-hot fields (every row has exactly one stored 1.0 per field) every Gibbs
// "field" of the design -- a one-hot main field (a level of :343-376) or a relation block (:378-482) -- sees a training row t
// only through ONE index: the field's column in the row, or original_to_block[t]. Fields that share the index (the user-side
// block is mapped by the user column, the item-side block by the item column) share an index STREAM. Then for factor f
//
//     q_t = sum over fields F of T_F[idx_{s(F)}(t)],     T_F = V[:, f] of the field's columns, or the block's q_B   (:320-340)
//
// is a sum of a handful of table look-ups, and the q-cache (8 bytes per row read + written by every pass of the reference) never
// has to exist in HBM: all that lives per row is the residual e_t and a packed record of the row's indices. For a field F with
// q_other = q_t - T_F[idx(t)] both forms of the conditional need the same four sums per index value i (:351-356 with x = 1:
// S2 = c_S, S1 = -e_q; :401-407 for a block):
//
//     c_i = sum q_other,  c_S,i = sum q_other^2,  e_i = sum e_t,  e_q,i = sum e_t q_other       over the rows with idx(t) = i
//
// and its update changes the rows by   e_t += q_other * d1[i] + d2[i]   with (d1, d2) = (v' - v, 0) for a main field (:371-375)
// and (q_B' - q_B, (q_B'^2 - q_B^2)/2 - (q_S' - q_S)/2) for a block (the un-sync :408-415 and re-sync :473-480 taken together).
// So update_V is, per factor, ONE streaming pass over (e, index record) per field: the pass applies the update of the field
// before it and takes the statistics of its own field; between two passes a column- or block-row-sized kernel draws (main
// field) or the block's feature sweep runs on its 64-byte records exactly as in the generic path (mfm_block_kernels.hpp).
//
// Row layout ("cell order"): the rows are cut into G groups of consecutive values of the SORTED stream U (the table is sorted
// by its first field), one workgroup per group; inside a group the rows are ordered by the index of the one large scattered
// stream I (items). A group's U indices are few enough for LDS tables (values, deltas, statistics accumulators -- statistics by
// ds_add_f64 in a fixed order: the waves of a workgroup take turns, a wave adds in program order). Along I the rows of one index
// value are consecutive: their sums are a wave-level segmented scan (DPP) and leave as ONE partial per (group, item) cell,
// reduced over the groups in group order by the draw / reduce kernel. Small streams C (context blocks, <= a few thousand
// values) have their whole tables in LDS. Every floating-point sum has a fixed order: the chain is bit-reproducible.
#pragma once
#include <functional>
#include <string>
#include <vector>

#include "mfm_common.hpp"

namespace mfm {

constexpr int CELL_MAX_STREAMS = 4;    // index streams of a row (u16 slots of its record; the I stream may be a separate int32)
constexpr int CELL_MAX_FIELDS = 16;
constexpr int CELL_NT = 1024;          // threads of a pass workgroup
constexpr int CELL_NW = CELL_NT / 64;  // its waves = row chunks of a group
#ifndef MFM_CELL_R
#define MFM_CELL_R 6
#endif
constexpr int CELL_R = MFM_CELL_R;     // 64-row windows per wave and step (MI355X, config 5: 4 -> 6 takes a pass from 280 to 235 us, 8 spills)
constexpr int64_t CELL_SMALL_MAX = 4096;          // largest index cardinality kept as a whole LDS table
constexpr size_t CELL_LDS_BYTES = 156 * 1024;     // LDS a pass workgroup may use

enum CellStreamType { CELL_U = 0, CELL_I = 1, CELL_C = 2 };

struct CellBlockIn {  // a relation block as the planner sees it (host)
  const int32_t *map;
  int64_t B;
};

struct CellField {
  int stream = 0;
  int kind = 0;      // 0: one-hot main field, 1: relation block
  int64_t n = 0;     // coefficients of the main field / rows of the block
  int64_t base = 0;  // main field: its first column in the feature index;  block: its position in the block list
};

struct CellStream {
  int type = CELL_C;
  int slot = -1;  // u16 slot of the row record (-1: the int32 item array)
  int64_t card = 0;
  std::vector<int> fields;
};

struct CellSrc {  // where a field's current table T_F lives: element i at p[i * stride]
  const double *p = nullptr;
  int stride = 1;
};

struct CellScoreSrc {  // a field's inputs of the scorer: block: q = X_B V_B [B][KS], lin = X_B w_B, ss = sum_f sum_l x^2 v^2;
  const double *q = nullptr, *lin = nullptr, *ss = nullptr;  // main field: lin = w + base (q and ss come from Vt)
};

struct CellPlan {
  bool ready = false;
  std::string why;
  int64_t N = 0, Npad = 0;  // rows / positions of the padded step layout
  int G = 0;
  int sU = -1, sI = -1;
  bool item32 = false;
  int64_t umax = 0;  // most U values in one group
  std::vector<CellStream> streams;
  std::vector<CellField> fields;  // in Gibbs order: main fields, then blocks
  // rows in cell order
  DevBuf<uint2> ix;        // 4 x u16 per row: the row's index in every LDS stream (U: group-local), and in I when it fits
  DevBuf<int32_t> item;    // I index per row when it needs more than 16 bits
  DevBuf<int32_t> perm;    // cell position -> training row
  DevBuf<int32_t> chunk_len;  // [G * CELL_NW] rows of every wave chunk
  DevBuf<int32_t> grp_base;   // [G + 1] first position of every group in the padded step layout
  DevBuf<int32_t> grp_u0;  // [G + 1] first U value of every group
  DevBuf<int32_t> grp_steps;  // [G] steps of the group's longest chunk
  DevBuf<double> e;        // the residual in cell order (valid inside mfm_sweep_V)
  // per-pass tables
  DevBuf<double> QA[CELL_MAX_STREAMS], QS[CELL_MAX_STREAMS];  // per stream: q without the pending / without the statistics field
  DevBuf<double> packI;    // [card_I][4] = (QA, QS, d1, d2) gathered per row
  DevBuf<double2> DP;      // (d1, d2) of the pending field, by index value
  DevBuf<double> stat;     // direct statistics of a main field on U [card][2]
  DevBuf<double> dense;    // row-sharded: a field's sums [index values][sums], all-reduced over the ranks before they are used
  DevBuf<double> stat1;    // update_w: sum e per index value of the field whose statistics were taken
  DevBuf<double> cnt[CELL_MAX_FIELDS];  // update_w: rows per column of every main field (cell_counts)
  bool cnt_ready = false;
  DevBuf<double> cells1, cells2, cells4;  // [G][card_I][1 | 2 | 4] partials of an I field (never-written cells stay zero)
  DevBuf<double> cpart;    // [G][card_C][4] partial tables of a C field
  // scorer tables (cell_score)
  DevBuf<double> scoreQ[CELL_MAX_STREAMS], scoreLS[CELL_MAX_STREAMS], vss;
  // which tables cell_prep has already built: a stream's content version (bumped by cell_touch when one of its fields' tables
  // changes) and the field left out, per side (0: QA, 1: QS)
  int64_t ver[CELL_MAX_STREAMS] = {1, 1, 1, 1}, ver_next = 2;
  int64_t have_ver[2][CELL_MAX_STREAMS] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  int have_ex[2][CELL_MAX_STREAMS] = {{-2, -2, -2, -2}, {-2, -2, -2, -2}};
  void touch(int field) { ver[fields[field].stream] = ver_next++; }  // the field's table changed
  void touch_all() {
    for (auto &v : ver) v = ver_next++;
  }
  bool fail(const char *w) {
    why = w;
    ready = false;
    return false;
  }
  bool fail(const std::string &w) { return fail(w.c_str()); }
  size_t lds_bytes(int P, int F, bool sw, int *off = nullptr, bool linear = false) const;  // LDS of the pass (P, F: field numbers or -1)
};

// planner (host): X = the main table in CSR (rows sorted by the first field), blocks in Gibbs order
// row-sharded: this rank's place among `world` ranks, sum_ranks = sum a host vector over the ranks
bool cell_plan_build(CellPlan &cp, const HostCsr &X, const std::vector<CellBlockIn> &blocks, int n_cu, hipStream_t s, int rank = 0,
                     int world = 1, const std::function<void(std::vector<double> &)> &sum_ranks = nullptr);
// the same plan built on the device (one GPU): X = the main table in device CSR, the blocks' maps as device int32 arrays
struct CellBlockDev {
  const int32_t *map;
  int64_t B;
};
bool cell_plan_build_device(CellPlan &cp, const DevSparse &X, const std::vector<CellBlockDev> &blocks, int n_cu, hipStream_t s, int rank = 0,
                            int world = 1, const std::function<void(std::vector<double> &)> &sum_ranks = nullptr);
std::string cell_plan_compare(const CellPlan &a, const CellPlan &b, hipStream_t s);  // "" or the first array that differs
void cell_pack_e(hipStream_t s, CellPlan &cp, const double2 *eq);
void cell_unpack_e(hipStream_t s, CellPlan &cp, double2 *eq);
// per-stream tables of a pass from the fields' current tables: QA = q without field exA (skipped when !doA), QS = q without
// field exS (skipped when !doS); dp_to_I: the pending field lives on I, copy its (d1, d2) next to the gathered values
void cell_prep(hipStream_t s, Timing &tm, CellPlan &cp, const std::vector<CellSrc> &cur, bool doA, int exA, bool doS, int exS, bool dp_to_I);
// one pass over the rows: apply the pending field P (-1: none), statistics of field F (-1: none); sw: QA / QS belong to
// different factors. out / out_stride / ns_out: where a U field's sums go (index i at out[i * out_stride + 0..ns))
void cell_pass(hipStream_t s, Timing &tm, CellPlan &cp, int P, int F, bool sw, double *out_u, int out_stride, bool linear = false);
// main field F: draw its columns from the statistics the pass left (FMTrainer.hpp:357-369), write V and DP
void cell_draw_main(hipStream_t s, Timing &tm, CellPlan &cp, int F, double *Vf, const double *zf, const int32_t *group, const double *lam,
                    const double *mu, double alpha, const double *dense = nullptr);
// row-sharded: a field's sums over this rank's groups -> dense [n][ns] (a U field: written by the pass); dense -> record words
void cell_stats_dense(hipStream_t s, Timing &tm, CellPlan &cp, int F, int ns, double *dense);
void cell_dense_to_rec(hipStream_t s, Timing &tm, CellPlan &cp, int F, int ns, const double *dense, double *rec, int w0);
// block F on an I or C stream: sum the partials over the groups into rec[i].{c, c_S, e, e_q} (a U block: written by the pass)
void cell_block_stats(hipStream_t s, Timing &tm, CellPlan &cp, int F, double *rec);
// block F after its feature sweep: DP[i] = (q' - q, (q'^2 - q^2)/2 - (q_S' - q_S)/2) from the saved and the new (q, q_S)
void cell_block_delta(hipStream_t s, Timing &tm, CellPlan &cp, int F, const double *rec, const double2 *saved);

// update_w on the cell layout: rows per main column (once), sums of an I / C field over the groups, the draw of a main field's
// columns (FMTrainer.hpp:237-254) from (sum e, n), a block's row update q_B' - q_B
void cell_counts(hipStream_t s, Timing &tm, CellPlan &cp);
void cell_sum1(hipStream_t s, Timing &tm, CellPlan &cp, int F, double *dst, int dst_stride);
void cell_draw_main_w(hipStream_t s, Timing &tm, CellPlan &cp, int F, double *w, const double *z, const int32_t *group, const double *lam,
                      const double *mu, double alpha, const double *se = nullptr);
void cell_block_delta_w(hipStream_t s, Timing &tm, CellPlan &cp, int F, const double *rec, const double2 *saved);
// update_e on the cell layout: eq[t].x = score_t (- y_t when y is given). V: factor-major [K][D], Vt: its row-major copy [D][KS]
void cell_score(hipStream_t s, Timing &tm, CellPlan &cp, const std::vector<CellScoreSrc> &src, const double *V, const double *Vt, int64_t D,
                int K, int KS, double w0, const double *y, double2 *eq);

}  // namespace mfm
