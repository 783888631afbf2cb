// bn254_msm.h -- the G1 multi-scalar multiplications of the PlonK verifier as ROWS (round 4): what one lane of k_g1_msm_rows does, and the plan that
// deals the terms of a launch to rows.  Replaces bn::AffineG1::msm / `AffineG1 * Fr` at plonk/verify.rs:284 and plonk/kzg.rs:82,161,169,175.
//
// A launch evaluates up to two sums  S_s = sum_t k_t P_t  per item (proof).  Its terms are of three kinds:
//   variable   P_t comes with the proof (commitments, openings): the scalar arrives GLV-decomposed, k = +-k1 +- k2 lambda with 128-bit halves, and is
//              walked by the joint two-bit-window form of bn254_rlc.h over the 128 joint bit positions -- by ONE row, or by two rows that take the low and
//              the high positions (the high row doubles its result a more times), which shortens the chain where a launch is latency-bound; or -- large
//              launches, where the total work counts -- by a JOINT row that walks up to MSM_MAX_JOINT terms of a sum together (Straus): one pair of doublings
//              per step for all of them instead of one pair per term;
//   fixed      P_t belongs to the verifying key: a window table of the point (MSM_FW_WINDOWS = 20 windows of MSM_FW_BITS = 13 bits, 8191 multiples each, built on the
//              device: bn254_k_comb.hip form 2; until round 5 byte windows, 32 x 255) turns the term into at most 20 complete mixed additions, and those additions are
//              dealt out window by window -- to the LOW rows of the variable terms, which have time to spare while the high rows double, and to rows of their own;
//   unit       k_t = +-1 (the -H of the KZG check): one mixed addition.
// Every row produces one projective point; k_g1_sum_affine adds the rows of a sum.  All formulas are the complete ones of bn254_curve.h, so no scalar or
// point a prover chooses reaches an exceptional case.  The plan is a pure function of (term kinds, items, lane budget): bn254_dbg_msm_plan exports it and
// tests/test_capi_cpu.py checks that every launch fits its buffers; tests/hostsim evaluates planned rows on the CPU against the oracle.
#pragma once
#include <cstring>
#include <utility>
#include "bn254_rlc.h"
#include "bn254_fw.h"

namespace bn254 {

#define MSM_MAX_ROWS 32
#define MSM_MAX_FIXED 16
#define MSM_MAX_JOINT 8        // variable terms of one joint row
#define MSM_TERM_DWORDS 26       // MsmTerm (bn254_plonk.hpp): 18 digits of the affine point, 8 scalar words
// (window tables of the key's points: MSM_FW_BITS / MSM_FW_WINDOWS / MSM_FW_ENTRIES in bn254_fw.h)
struct MsmRow {
  int8_t var_term;      // term whose GLV halves this row walks, or -1
  uint8_t pos_lo, pos_hi;   // joint bit positions [pos_lo, pos_hi) of the 128, both even; the row's result carries the factor 2^pos_lo
  int8_t unit_term;     // term added as +-P, or -1
  uint8_t sum;          // the sum the row belongs to
  uint8_t glv_slot;     // which of the launch's window-table scratch rows it uses (variable rows only)
  uint16_t fw_lo, fw_hi;    // windows [fw_lo, fw_hi) of its sum's fixed terms, flattened: window q = digit q % MSM_FW_WINDOWS of fixed term q / MSM_FW_WINDOWS
  uint8_t n_joint;      // > 0: a JOINT row -- var_term is then the INDEX of its first term in var_list[sum], it walks n_joint consecutive ones over all 128
                        // positions with shared doublings; their window tables are the slots glv_slot .. glv_slot + n_joint - 1
  uint8_t reserved;
};
struct MsmPlan {
  int32_t n_rows, n_var_rows;
  int32_t first[2], count[2];                       // rows of sum s: [first[s], first[s] + count[s])
  int32_t n_fixed[2];
  int8_t fixed_term[2][MSM_MAX_FIXED];              // term index of the f-th fixed term of sum s (its scalar: 8 canonical words in the term's k[])
  int8_t fixed_tab[2][MSM_MAX_FIXED];               // and which of the key's window tables it reads
  int8_t var_list[2][MSM_MAX_FIXED];                // the variable terms of sum s in the order given (joint rows index it)
  MsmRow row[MSM_MAX_ROWS];
};
// what the caller says about a launch: per sum the term indices of each kind
struct MsmShape {
  int n_sums;
  int n_var[2], n_unit[2], n_fixed[2];
  int8_t var_term[2][MSM_MAX_FIXED], unit_term[2][4], fixed_term[2][MSM_MAX_FIXED], fixed_tab[2][MSM_MAX_FIXED];
};
// cost model of the planner: multiply-adds of one lane per loop body of k_g1_msm_rows as counted in the gfx950 code object (tools/count_mads.py `components`): a two-bit
// step (2 doublings + 1 addition from the lane's table), a doubling, a complete mixed addition (fixed-base byte window / unit term), the lane's 15-entry table
// (round 5, the sum-of-products G1 formulas of bn254_curve.h; until round 4: 4461 / 1243 / 1815 / 24149)
#define MSM_COST_STEP 4079
#define MSM_COST_DBL 1169
#define MSM_COST_MIXED 1580
#define MSM_COST_TABLE 21404
#define MSM_COST_JADD (MSM_COST_STEP - 2 * MSM_COST_DBL)     // the addition of a step without its two doublings: a joint row's cost per term and step
// Rows for `n_pad` items (a multiple of 64: a wavefront never straddles two rows) within `lane_budget` lanes (one wavefront per SIMD: 65536).
//   * While twice the variable terms fit the budget the launch is latency-bound -- it lasts as long as its LONGEST row -- and every variable term is SPLIT over a
//     low row (joint bit positions [0, a)) and a high row ([a, 128), which doubles its result a more times).  Per sum the planner tries every even a and keeps
//     the one with the shortest longest row: the fixed windows of the sum go to its low rows as far as those stay below the high rows, then to rows of their own
//     as far as the budget has rows left, and whatever is still left is spread over the low rows.  (a = 64 with the fixed windows riding on the low rows was
//     round 4's first form; a sum without fixed windows is level at a = 88, one with spare rows for its windows too: 0.82 -> 0.74 ms at 4096 proofs.)
//   * Otherwise (a large launch: throughput) every variable term is one row and the fixed windows form rows of about a variable row's cost.
// Unit terms ride on the first rows of their sum.  Returns false if the shape needs more than MSM_MAX_ROWS rows.
struct MsmSplit { int a, low_each, own_rows, own_each; long worst; };
// the best split of one sum: L variable terms, wf fixed windows, `spare` rows of the budget not yet taken
inline MsmSplit msm_best_split(int L, int wf, long spare, int force_a = 0) {
  MsmSplit best = {64, 0, 0, 0, -1};
  for (int a = force_a ? force_a : 2; a <= (force_a ? force_a : 126); a += 2) {
    const long hi = MSM_COST_TABLE + (long)(128 - a) / 2 * MSM_COST_STEP + (long)a * MSM_COST_DBL, lo = MSM_COST_TABLE + (long)a / 2 * MSM_COST_STEP;
    MsmSplit c = {a, 0, 0, 0, hi > lo ? hi : lo};
    int rest = wf;
    if (rest > 0) {
      const long room = hi > lo ? (hi - lo) / MSM_COST_MIXED : 0;              // windows a low row takes without becoming the longest row
      c.low_each = (int)((long)(rest + L - 1) / L < room ? (rest + L - 1) / L : room);
      rest -= c.low_each * L;
      if (rest > 0) {
        const long own_cap = c.worst / MSM_COST_MIXED;
        long want = (rest + own_cap - 1) / own_cap;
        c.own_rows = (int)(want < spare ? want : (spare > 0 ? spare : 0));
        if (c.own_rows > 0) { c.own_each = (rest + c.own_rows - 1) / c.own_rows; if (c.own_each > own_cap) c.own_each = (int)own_cap; rest -= c.own_rows * c.own_each; }
        if (rest > 0) c.low_each += (rest + L - 1) / L;                        // no rows left: the low rows get longer
      }
      const long lo_full = lo + (long)c.low_each * MSM_COST_MIXED, own = (long)c.own_each * MSM_COST_MIXED;
      if (lo_full > c.worst) c.worst = lo_full;
      if (own > c.worst) c.worst = own;
    }
    // fewer rows of their own on a tie (a row of the sum costs k_g1_sum_affine one more addition)
    if (best.worst < 0 || c.worst < best.worst || (c.worst == best.worst && c.own_rows < best.own_rows)) best = c;
  }
  return best;
}
inline bool msm_plan_build(MsmPlan& p, const MsmShape& sh, size_t n_pad, size_t lane_budget, int force_a = 0 /* experiments: the split position, even, 2..126 */,
                           int joint_g = 0 /* > 1: unsplit launches walk up to that many variable terms of a sum in one JOINT row (large launches) */) {
  std::memset(&p, 0, sizeof p);
  if (sh.n_sums < 1 || sh.n_sums > 2 || n_pad == 0) return false;
  int total_var = 0;
  for (int s = 0; s < sh.n_sums; s++) {
    if (sh.n_fixed[s] > MSM_MAX_FIXED || sh.n_var[s] > MSM_MAX_FIXED || sh.n_unit[s] > 4 || sh.n_fixed[s] < 0 || sh.n_var[s] < 0 || sh.n_unit[s] < 0) return false;
    total_var += sh.n_var[s];
  }
  const bool split = total_var > 0 && (size_t)(2 * total_var) * n_pad <= lane_budget;
  if (joint_g > MSM_MAX_JOINT) joint_g = MSM_MAX_JOINT;
  const bool joint = !split && joint_g > 1;
  const long chain_full = MSM_COST_TABLE + 64L * MSM_COST_STEP;
  long spare = split ? (long)(lane_budget / n_pad) - 2 * total_var : MSM_MAX_ROWS;             // rows the budget still has
  int r = 0, slot = 0;
  for (int s = 0; s < sh.n_sums; s++) {
    p.first[s] = r;
    p.n_fixed[s] = sh.n_fixed[s];
    for (int f = 0; f < sh.n_fixed[s]; f++) { p.fixed_term[s][f] = sh.fixed_term[s][f]; p.fixed_tab[s][f] = sh.fixed_tab[s][f]; }
    const int wf = MSM_FW_WINDOWS * sh.n_fixed[s], L = sh.n_var[s];
    for (int t = 0; t < L; t++) p.var_list[s][t] = sh.var_term[s][t];
    int low_each = 0, own_rows = 0, own_each = 0, a = 64;
    // joint rows: the L terms in ceil(L / joint_g) rows of sizes that differ by at most one
    const int j_rows = joint && L > 0 ? (L + joint_g - 1) / joint_g : 0;
    if (split && L > 0) {
      // rows left for THIS sum's windows: what the budget has, minus nothing -- later sums take what remains (PlonK: only the first sum of a launch has fixed terms)
      const MsmSplit b = msm_best_split(L, wf, spare > MSM_MAX_ROWS - r - 2 * L ? MSM_MAX_ROWS - r - 2 * L : spare, force_a);
      a = b.a; low_each = b.low_each; own_rows = b.own_rows; own_each = b.own_each;
    } else if (wf > 0 && j_rows) {
      // joint rows carry the sum's fixed windows themselves, levelled (below): no rows of their own -- a large launch is over when its LAST row is, and a short
      // row of windows beside long joint rows would leave its lanes idle
    } else if (wf > 0) {
      const int own_cap = (int)((split ? MSM_COST_TABLE + 32L * MSM_COST_STEP + 64L * MSM_COST_DBL : chain_full) / MSM_COST_MIXED);   // windows of a row that has nothing else to do
      own_rows = (wf + own_cap - 1) / own_cap;
      if (split && own_rows > spare) own_rows = spare > 0 ? (int)spare : 1;                    // (a sum without variable terms in a split launch)
      own_each = (wf + own_rows - 1) / own_rows;
    }
    spare -= own_rows;
    int q = 0, unit_i = 0;                                                                      // next fixed window / unit term to hand out
    auto fixed_slice = [&](MsmRow& w, int want) { w.fw_lo = (uint16_t)q; q = q + want < wf ? q + want : wf; w.fw_hi = (uint16_t)q; };
    auto blank = [&](MsmRow& w) { w.var_term = -1; w.unit_term = -1; w.pos_lo = w.pos_hi = 0; w.sum = (uint8_t)s; w.glv_slot = 0; w.fw_lo = w.fw_hi = 0; w.n_joint = 0; w.reserved = 0; };
    if (j_rows) {
      // the terms over the rows, sizes differing by at most one; then the windows poured onto the rows with fewer terms until all are level, the rest spread evenly
      int gsz[MSM_MAX_FIXED], win[MSM_MAX_FIXED];
      for (int k = 0, t0 = 0; k < j_rows; k++) { gsz[k] = (L - t0 + (j_rows - k) - 1) / (j_rows - k); t0 += gsz[k]; win[k] = 0; }
      const long per_term = MSM_COST_TABLE + 64L * MSM_COST_JADD;
      int left = wf;
      for (int k = 0; k < j_rows && left > 0; k++) {
        const int room = (int)((long)(gsz[0] - gsz[k]) * per_term / MSM_COST_MIXED);             // gsz[0] is the largest group
        win[k] = room < left ? room : left; left -= win[k];
      }
      for (int k = 0; left > 0; k = (k + 1) % j_rows) { const int share = (left + (j_rows - k) - 1) / (j_rows - k); win[k] += share; left -= share; }
      for (int k = 0, t0 = 0; k < j_rows; k++) {
        if (r + 1 > MSM_MAX_ROWS) return false;
        MsmRow& w = p.row[r++];
        blank(w);
        w.var_term = (int8_t)t0; w.n_joint = (uint8_t)gsz[k]; w.pos_lo = 0; w.pos_hi = 128; w.glv_slot = (uint8_t)slot; slot += gsz[k]; t0 += gsz[k];
        if (unit_i < sh.n_unit[s]) w.unit_term = sh.unit_term[s][unit_i++];
        fixed_slice(w, win[k]);
      }
    }
    for (int t = 0; t < (j_rows ? 0 : L); t++) {
      if (r + 2 > MSM_MAX_ROWS) return false;
      MsmRow& lo = p.row[r++];
      blank(lo);
      lo.var_term = sh.var_term[s][t]; lo.pos_lo = 0; lo.pos_hi = (uint8_t)(split ? a : 128); lo.glv_slot = (uint8_t)slot++;
      if (unit_i < sh.n_unit[s]) lo.unit_term = sh.unit_term[s][unit_i++];
      fixed_slice(lo, low_each);
      if (split) {
        MsmRow& hi = p.row[r++];
        blank(hi);
        hi.var_term = sh.var_term[s][t]; hi.pos_lo = (uint8_t)a; hi.pos_hi = 128; hi.glv_slot = (uint8_t)slot++;
      }
    }
    for (int k = 0; k < own_rows || q < wf || unit_i < sh.n_unit[s]; k++) {
      if (r + 1 > MSM_MAX_ROWS) return false;
      MsmRow& o = p.row[r++];
      blank(o);
      if (unit_i < sh.n_unit[s]) o.unit_term = sh.unit_term[s][unit_i++];
      fixed_slice(o, k + 1 >= own_rows ? wf : own_each);
    }
    if (r == p.first[s]) {                                                                      // an empty sum is one row that yields the identity
      if (r + 1 > MSM_MAX_ROWS) return false;
      blank(p.row[r++]);
    }
    p.count[s] = r - p.first[s];
  }
  p.n_rows = r; p.n_var_rows = slot;
  return true;
}
// the longest row of a plan, in the planner's cost units = multiply-adds of a lane (tests; the roofline of the bench line)
inline int msm_plan_chain(const MsmPlan& p) {
  int worst = 0;
  for (int r = 0; r < p.n_rows; r++) {
    const MsmRow& w = p.row[r];
    int c = (w.fw_hi - w.fw_lo) * MSM_COST_MIXED + (w.unit_term >= 0 ? MSM_COST_MIXED : 0);
    if (w.n_joint) c += w.n_joint * (MSM_COST_TABLE + 64 * MSM_COST_JADD) + 128 * MSM_COST_DBL;
    else if (w.var_term >= 0) c += MSM_COST_TABLE + (w.pos_hi - w.pos_lo) / 2 * MSM_COST_STEP + w.pos_lo * MSM_COST_DBL;
    if (c > worst) worst = c;
  }
  return worst;
}

// ---- a variable term over the joint bit positions [pos_lo, pos_hi) of its GLV halves: sum over those positions, times 2^pos_lo ------------------------
// (+-k1 +- k2 lambda) P restricted to the positions, TWO bits of each half per step: acc <- 4 acc + (d1 P1 + d2 P2), d1, d2 in 0..3, from a table of the 15
// non-zero combinations i P1 + j P2 at index 4 i + j (P1 = +-P, P2 = +-phi(P), bn254_rlc.h).  (pos_hi - pos_lo) / 2 steps of (2 doublings + 1 addition) + 13
// point operations for the table, against one (doubling + addition) per position for the one-bit form: 0.75 of the chain.  The table lives where TAB puts
// it (k_g1_msm_rows: 15 x 28 dwords of global memory per lane -- one contiguous 108-byte read per step, issued before the doublings; the host test keeps it
// in an array).  The halves are first shifted so that position pos_hi - 1 is the top bit.  pos_lo, pos_hi even and uniform over the wavefront.
// the 15 non-zero combinations i P1 + j P2 at index 4 i + j, P1 = +-P, P2 = +-phi(P)
template <class TAB>
BN_HD void glv_w2_table(const G1Aff& P, bool neg1, bool neg2, TAB& tab) {
  G1Aff P1 = P, P2;
  P1.y = fp_select(neg1, fp_neg(P.y), P.y);
  P2.x = fp_mul(P.x, fp_from_limbs(BN_GLV_BETA)); P2.y = fp_select(neg2, fp_neg(P.y), P.y);
  {
    // the table is built THROUGH the table: the multiples of P1 and P2 go out as they are made, the nine mixed entries are sums of entries read back -- no array
    // of points stays live (as local arrays the eight points would be 216 registers or, indexed by the loop counters, 864 bytes of scratch memory)
    G1Proj t = g1_from_affine(P1); tab.put(4, t);
    G1Proj d = g1_dbl(t); tab.put(8, d); tab.put(12, g1_add_mixed(d, P1));
    t = g1_from_affine(P2); tab.put(1, t);
    d = g1_dbl(t); tab.put(2, d); tab.put(3, g1_add_mixed(d, P2));
    tab.fence();
    for (int i = 1; i < 4; i++)
      for (int j = 1; j < 4; j++) tab.put(4 * i + j, g1_add(tab.get(4 * i), tab.get(j)));
  }
  tab.fence();
}
template <class TAB>
BN_HD G1Proj g1_mul_glv_w2_range(const G1Aff& P, const uint32_t k1[4], bool neg1, const uint32_t k2[4], bool neg2, int pos_lo, int pos_hi, TAB& tab) {
  glv_w2_table(P, neg1, neg2, tab);
  uint32_t a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; i++) { a[i] = k1[i]; b[i] = k2[i]; }
  for (int s = 128 - pos_hi; s >= 32; s -= 32) {
    a[3] = a[2]; a[2] = a[1]; a[1] = a[0]; a[0] = 0;
    b[3] = b[2]; b[2] = b[1]; b[1] = b[0]; b[0] = 0;
  }
  const int bs = (128 - pos_hi) & 31;
  if (bs) {
#pragma unroll
    for (int i = 3; i > 0; i--) { a[i] = (a[i] << bs) | (a[i - 1] >> (32 - bs)); b[i] = (b[i] << bs) | (b[i - 1] >> (32 - bs)); }
    a[0] <<= bs; b[0] <<= bs;
  }
  G1Proj acc = g1_identity();
  const int steps = (pos_hi - pos_lo) / 2;
  for (int step = 0; step < steps; step++) {
    const uint32_t idx = ((a[3] >> 30) << 2) | (b[3] >> 30);
#pragma unroll
    for (int i = 3; i > 0; i--) { a[i] = (a[i] << 2) | (a[i - 1] >> 30); b[i] = (b[i] << 2) | (b[i - 1] >> 30); }
    a[0] <<= 2; b[0] <<= 2;
    const G1Proj q = tab.get(idx != 0 ? idx : 1u);
    acc = g1_dbl(g1_dbl(acc));
    const G1Proj c = g1_add(acc, q);
    const bool take = idx != 0;
    acc.x = fp_select(take, c.x, acc.x); acc.y = fp_select(take, c.y, acc.y); acc.z = fp_select(take, c.z, acc.z);
  }
  for (int d = 0; d < pos_lo; d++) acc = g1_dbl(acc);
  return acc;
}

// the whole term (W words per half) in one go: what a single row of an unsplit launch computes (tests/hostsim)
template <int W, class TAB>
BN_HD G1Proj g1_mul_glv_w2(const G1Aff& P, const uint32_t* k1, bool neg1, const uint32_t* k2, bool neg2, TAB& tab) {
  uint32_t a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < W; i++) { a[i] = k1[i]; b[i] = k2[i]; }
  return g1_mul_glv_w2_range(P, a, neg1, b, neg2, 0, 32 * W, tab);
}

// ---- one row ------------------------------------------------------------------------------------------------------------------------------------------------
// IO supplies the item's data: term(t, P, k1, k2, flags) -- the affine point, the GLV halves and the flag byte (bit 0: the point is the identity, bits 1 / 2:
// signs; for a unit term bit 1 is the sign) of term t; scalar_digit(t, w) -- digit w (MSM_FW_BITS bits, weight 2^(MSM_FW_BITS w)) of the canonical scalar of fixed term t;
// entry(tab, w, d) -- multiple d + 1 of window w of key table `tab`.
// ---- a joint row: sum_j (+-k1_j +- k2_j lambda) P_j for n_joint terms of a sum, Straus' way -- every term gets its 15-entry table (slot j of the row's scratch), then
// 64 steps of  acc <- 4 acc + sum_j T_j[digit_j]: ONE pair of doublings per step whatever the number of terms.  The scalar words are re-read per step (io.kword:
// eight words per term would be 56 registers for seven terms) and, like the table entry of the NEXT addition, fetched while the current addition runs.
template <class IO, class TAB>
BN_HD G1Proj msm_joint_eval(const MsmPlan& plan, const MsmRow& row, IO& io, TAB& glv) {
  const int s = row.sum, g = row.n_joint;
  uint32_t dead = 0;                                   // bit j: term j's point is the identity, its digits count as zero
  for (int j = 0; j < g; j++) {
    G1Aff P; uint32_t k1[4], k2[4]; uint32_t fl;
    io.term(plan.var_list[s][row.var_term + j], P, k1, k2, fl);
    if (fl & 1) dead |= 1u << j;
    TAB tj = glv.slot(j);
    glv_w2_table(P, (fl & 2) != 0, (fl & 4) != 0, tj);
  }
  // digit of term j at step st: bits 127 - 2 st and 126 - 2 st of each half (the position is odd: both bits sit in one word)
  auto digit = [&](int st, int j) -> uint32_t {
    const int ph = 127 - 2 * st, w = ph >> 5, sh = (ph & 31) - 1, t = plan.var_list[s][row.var_term + j];
    const uint32_t idx = (((io.kword(t, 0, w) >> sh) & 3u) << 2) | ((io.kword(t, 1, w) >> sh) & 3u);
    return ((dead >> j) & 1u) ? 0u : idx;
  };
  G1Proj acc = g1_identity();
  uint32_t idx = digit(0, 0);
  G1Proj q = glv.slot(0).get(idx != 0 ? idx : 1u);
  int st = 0, j = 0;
  for (int k = 0; k < 64 * g; k++) {
    int jn = j + 1, stn = st;
    if (jn == g) { jn = 0; stn = st + 1; }
    uint32_t idx_n = 0; G1Proj qn = q;
    if (stn < 64) { idx_n = digit(stn, jn); qn = glv.slot(jn).get(idx_n != 0 ? idx_n : 1u); }
    if (j == 0) acc = g1_dbl(g1_dbl(acc));
    const G1Proj c = g1_add(acc, q);
    const bool take = idx != 0;
    acc.x = fp_select(take, c.x, acc.x); acc.y = fp_select(take, c.y, acc.y); acc.z = fp_select(take, c.z, acc.z);
    idx = idx_n; q = qn; j = jn; st = stn;
  }
  return acc;
}

template <class IO, class TAB>
BN_HD G1Proj msm_row_eval(const MsmPlan& plan, int r, IO& io, TAB& glv) {
  const MsmRow& row = plan.row[r];
  G1Proj acc = g1_identity();
  if (row.n_joint) acc = msm_joint_eval(plan, row, io, glv);
  else if (row.var_term >= 0) {
    G1Aff P; uint32_t k1[4], k2[4]; uint32_t fl;
    io.term(row.var_term, P, k1, k2, fl);
    if (fl & 1) {
#pragma unroll
      for (int k = 0; k < 4; k++) { k1[k] = 0; k2[k] = 0; }
    }
    acc = g1_mul_glv_w2_range(P, k1, (fl & 2) != 0, k2, (fl & 4) != 0, row.pos_lo, row.pos_hi, glv);
  }
  if (row.unit_term >= 0) {
    G1Aff P; uint32_t k1[4], k2[4]; uint32_t fl;
    io.term(row.unit_term, P, k1, k2, fl);
    P.y = fp_select((fl & 2) != 0, fp_neg(P.y), P.y);
    const G1Proj c = g1_add_mixed(acc, P);
    const bool take = (fl & 1) == 0;
    acc.x = fp_select(take, c.x, acc.x); acc.y = fp_select(take, c.y, acc.y); acc.z = fp_select(take, c.z, acc.z);
  }
  const int s = row.sum;
  if (row.fw_lo < row.fw_hi) {
    // the next window's digit and table entry are fetched while the current addition runs
    int q = row.fw_lo;
    uint32_t dig = io.scalar_digit(plan.fixed_term[s][q / MSM_FW_WINDOWS], q % MSM_FW_WINDOWS);
    G1Aff e = io.entry(plan.fixed_tab[s][q / MSM_FW_WINDOWS], q % MSM_FW_WINDOWS, dig ? dig - 1 : 0);
    for (; q < row.fw_hi; q++) {
      uint32_t dn = 0; G1Aff en = e;
      if (q + 1 < row.fw_hi) {
        const int qn = q + 1;
        dn = io.scalar_digit(plan.fixed_term[s][qn / MSM_FW_WINDOWS], qn % MSM_FW_WINDOWS);
        en = io.entry(plan.fixed_tab[s][qn / MSM_FW_WINDOWS], qn % MSM_FW_WINDOWS, dn ? dn - 1 : 0);
      }
      const G1Proj c = g1_add_mixed(acc, e);
      const bool take = dig != 0;
      acc.x = fp_select(take, c.x, acc.x); acc.y = fp_select(take, c.y, acc.y); acc.z = fp_select(take, c.z, acc.z);
      dig = dn; e = en;
    }
  }
  return acc;
}

}  // namespace bn254
