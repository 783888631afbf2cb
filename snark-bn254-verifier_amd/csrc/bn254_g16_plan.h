// bn254_g16_plan.h -- how a Groth16 batch is cut into workspace chunks, sub-batches (parts) and launch forms, and what a (key, device) context allocates
// for a reservation: PURE functions of the sizes, shared by the code that allocates (bn254_capi.hip::ensure_dev), the code that enqueues
// (g16_enqueue_exact, bn254_launch_g16) and the probe tests/test_capi_cpu.py reads through bn254_dbg_g16_plan.  The launch form of a batch follows the
// BATCH's size, the buffers the RESERVATION's: the property test walks batch sizes against reservations and asserts that every launch fits (the class of
// the PlonK scratch overflow of round 3, found there by a sweep instead of a test).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "bn254_kernels.h"

namespace bn254 {

// ---- what ensure_dev allocates for a reservation of n proofs against a key with `key_inputs` public inputs ------------------------------------------------
struct G16Alloc {
  size_t ws_proofs;          // proofs the workspace holds (G16_WS_BYTES_PER_PROOF each)
  size_t msm_part_proofs;    // keys with many inputs: proofs the partial-sum / digit buffer holds per launch (0: the key has none)
  size_t msm_chunks;         // partial sums per proof
  size_t msm_part_bytes;     // chunks * 27 dwords * proofs
  size_t msm_digit_bytes;    // comb tables: G16_COMB_COLS * key_inputs * proofs u16
};
inline size_t g16_round256(size_t n) { return (n + 255) / 256 * 256; }
inline G16Alloc g16_alloc_for(size_t n, size_t key_inputs, bool comb) {
  G16Alloc a = {0, 0, 0, 0, 0};
  a.ws_proofs = g16_round256(n) < (size_t)G16_MAX_BATCH ? g16_round256(n) : (size_t)G16_MAX_BATCH;   // larger batches run in chunks of G16_MAX_BATCH
  if (key_inputs > (size_t)G16_WIDE_MSM_MIN_INPUTS) {
    a.msm_part_proofs = n < (size_t)G16_WIDE_MSM_MAX_PROOFS ? g16_round256(n) : (size_t)G16_WIDE_MSM_MAX_PROOFS;
    a.msm_chunks = (key_inputs + G16_WIDE_MSM_INPUTS_PER_LANE - 1) / G16_WIDE_MSM_INPUTS_PER_LANE;
    a.msm_part_bytes = a.msm_chunks * 27 * a.msm_part_proofs * sizeof(int32_t);
    a.msm_digit_bytes = comb ? (size_t)G16_COMB_COLS * key_inputs * a.msm_part_proofs * sizeof(uint16_t) : 0;
  }
  return a;
}

// ---- the form of ONE launch (bn254_launch_g16) --------------------------------------------------------------------------------------------------------------
enum { G16_FORM_LANES = 0, G16_FORM_COOP = 1, G16_FORM_LATENCY = 2 };
struct G16Form {
  int form;          // cooperative kernels (twelve lanes per proof), lane kernels, or the lane kernels' latency mode (three Miller chains on three streams)
  int run_steps;     // lane kernels: Miller steps per k_miller_run launch (0: one launch per step)
  bool wide;         // the public-input MSM runs as (proof, chunk) lanes through the partial-sum buffer
};
// n: proofs of the launch; n_public: inputs per proof AS PASSED; inputs_match_key: n_public + 1 == len(vk.K); has_msm_part: the caller handed a partial-sum buffer;
// part_of_larger: one of several sub-batches; have_split_streams: the caller provided the two extra streams of the latency mode
inline G16Form g16_launch_form(size_t n, size_t n_public, bool inputs_match_key, bool has_msm_part, bool part_of_larger, bool have_split_streams, bool coop_on, int run_steps_env) {
  G16Form f;
  f.wide = has_msm_part && inputs_match_key && n_public > (size_t)G16_WIDE_MSM_MIN_INPUTS;
  const bool coop = coop_on && !part_of_larger && n <= (size_t)COOP12_MAX_PROOFS && (f.wide || n_public <= (size_t)G16_WIDE_MSM_MIN_INPUTS);
  f.form = coop ? G16_FORM_COOP : (have_split_streams && n <= (size_t)G16_SPLIT_MAX_PROOFS) ? G16_FORM_LATENCY : G16_FORM_LANES;
  // steps of the Miller loop per launch: a batch that is ONE sub-batch takes the whole loop; sub-batches of a larger batch that are a single generation of
  // workgroups run better in a few shorter launches (profiles/r03_run_steps_sweep.txt)
  f.run_steps = run_steps_env >= 0 ? run_steps_env : !part_of_larger ? 88 : (n <= 65536 ? 11 : n <= 131072 ? 22 : n <= 262144 ? 44 : 88);
  return f;
}

// ---- one workspace chunk (at most G16_MAX_BATCH proofs) of a batch: its sub-batches -----------------------------------------------------------------------
#define G16_MAX_PARTS 32
struct G16Part {
  size_t first, count;       // proofs [first, first + count) of the chunk: also the part's position in the workspace (proof units)
  int stream_slot;           // 0: the caller's stream, 1..3: auxiliary streams (concurrent parts); -1: not concurrent (the caller's stream)
};
struct G16ChunkPlan {
  int parts; size_t per;     // sub-batches and their nominal size (a multiple of 256)
  bool wide, concurrent, split_small;
  size_t max_launch;
  G16Part part[G16_MAX_PARTS];
};
// m: proofs of the chunk (<= G16_MAX_BATCH); key_inputs: len(vk.K) - 1; n_public: as passed by the caller; n_streams: BN254_STREAMS; single_stream: the device's
// sub-batch streams were measured NOT to overlap (they share a hardware queue): one sub-batch where one launch can hold the chunk
inline bool g16_plan_chunk(G16ChunkPlan& p, size_t m, size_t key_inputs, size_t n_public, int n_streams, bool single_stream) {
  p.wide = n_public == key_inputs && n_public > (size_t)G16_WIDE_MSM_MIN_INPUTS;
  p.max_launch = p.wide ? (size_t)G16_WIDE_MSM_MAX_PROOFS : (size_t)G16_MAX_LAUNCH;
  // Up to 65 536 proofs are one wavefront per SIMD at most: one sub-batch, and up to COOP12_MAX_PROOFS the cooperative kernels take the batch whole.  Above that,
  // n_streams sub-batches side by side (the tail of one sub-batch's kernel overlaps the head of the other's).  Keys with many inputs share ONE partial-sum
  // buffer between the launches of a batch, so their launches stay on the caller's stream and cover at most G16_WIDE_MSM_MAX_PROOFS proofs each.
  int parts = (!p.wide && n_streams > 1 && !single_stream && m > 65536) ? n_streams : 1;
  while ((m + parts - 1) / parts > p.max_launch) parts++;      // 32-bit workspace offsets per launch
  if (parts > G16_MAX_PARTS) return false;
  p.parts = parts;
  p.concurrent = !p.wide && n_streams > 1 && !single_stream && parts > 1;
  p.split_small = m <= (size_t)G16_SPLIT_MAX_PROOFS;
  p.per = g16_round256((m + parts - 1) / parts);
  int k = 0;
  for (int pi = 0; pi < parts; pi++) {
    const size_t lo = (size_t)pi * p.per, hi = lo + p.per < m ? lo + p.per : m;
    if (lo >= hi) break;
    p.part[k].first = lo; p.part[k].count = hi - lo; p.part[k].stream_slot = p.concurrent ? pi % 4 : -1;
    k++;
  }
  p.parts = k;
  return true;
}

// ---- BN254_FLAG_RLC: the group status bytes of one workspace chunk --------------------------------------------------------------------------------------
// A chunk of m proofs runs as `parts` launch parts; part pi forms rlc_plan(its proofs).groups groups whose status bytes sit in a region rounded up to 256
// (k_rlc_* grids are multiples of 256 lanes).  g16_rlc_alloc(m) is what rlc_ensure allocates for a chunk of m proofs, g16_rlc_need what the parts address.
inline size_t g16_rlc_alloc(size_t m) { return g16_round256(m) + 1024; }
inline int g16_rlc_parts(size_t m, int n_streams) {
  int parts = (n_streams > 1 && m >= (size_t)n_streams * 16384) ? n_streams : 1;
  while ((m + parts - 1) / parts > (size_t)G16_MAX_LAUNCH) parts++;
  return parts;
}
inline int g16_rlc_share(size_t part_n, int log2_group, int log2_share_env, size_t min_lanes) {
  int log2_share = log2_share_env < log2_group ? log2_share_env : log2_group;
  while (log2_share > 0 && (part_n >> log2_share) < (min_lanes < 1 ? 1 : min_lanes)) log2_share--;   // sharing needs enough lanes to fill the GPU
  return log2_share;
}
inline size_t g16_rlc_need(size_t m, int n_streams, int log2_group, int log2_share_env, size_t min_lanes) {
  const int parts = g16_rlc_parts(m, n_streams);
  const size_t per = g16_round256((m + parts - 1) / parts);
  size_t off = 0;
  for (int pi = 0; pi < parts; pi++) {
    const size_t lo = (size_t)pi * per, hi = lo + per < m ? lo + per : m;
    if (lo >= hi) break;
    const RlcPlan pl = rlc_plan((uint32_t)(hi - lo), log2_group, g16_rlc_share(hi - lo, log2_group, log2_share_env, min_lanes));
    off += g16_round256(pl.groups);
  }
  return off;
}

}  // namespace bn254
