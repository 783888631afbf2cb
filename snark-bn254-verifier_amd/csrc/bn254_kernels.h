// bn254_kernels.h -- host-visible launch interface of bn254_kernels.hip (internal to the library).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// status byte values: identical to include/bn254_verify.h (static_asserted in bn254_capi.hip)
#define BN254_ST_REJECT 0
#define BN254_ST_ACCEPT 1
#define BN254_ST_NOT_MEMBER 2
#define BN254_ST_NOT_ON_CURVE 3
#define BN254_ST_NOT_IN_SUBGROUP 4
#define BN254_ST_INPUT_LEN 5
#define BN254_ST_MALFORMED 6
#define BN254_ST_PENDING 0x80  // internal: no error so far (low 6 bits: deferred error of point C)
#define BN254_ST_LINF 0x40     // internal: the public-input point L is the identity
#define BN254_ST_LINF2 0x20    // internal (PlonK pairing check, RLC group stage): the G1 point of the second fixed pair is the identity
#define BN254_ST_LINF3 0x10    // internal (RLC group stage): the G1 point of the third fixed pair is the identity

#define MSM_ENTRY_DWORDS 20    // affine G1 point, 2 x 9 limbs + 2 pad: 80-byte entries, 16-byte aligned
#include "bn254_fw.h"   // MSM_FW_BITS / MSM_FW_WINDOWS / MSM_FW_ENTRIES: the window tables of the key points of the PlonK MSMs
#define FIXED_LINE_DWORDS 54   // one precomputed Miller step of a fixed G2 argument: m, c, xi*c (3 Fp2)
#define G16_WS_ELEMS 130       // Fp elements per proof in the workspace (bn254_vm.h: VE_COUNT)
#define G16_WS_BYTES_PER_PROOF (G16_WS_ELEMS * 36)
#define G16_MAX_BATCH 1048576  // proofs per chunk of a larger batch (one workspace)
#define G16_MAX_LAUNCH 786432  // proofs per kernel launch: the workspace is addressed with 32-bit buffer offsets, 130 * 36 * n < 2^32

// every workspace byte offset (SGPR row offset + VGPR lane offset) must fit the 32-bit offset arithmetic of the buffer instructions, and the
// out-of-range lane offset 0xfffffffc must stay above every valid one so that the descriptor's bounds check (num_records = the launch's
// workspace bytes) drops the dead lanes' accesses
static_assert((unsigned long long)G16_MAX_LAUNCH * G16_WS_BYTES_PER_PROOF < 0xfffffffcull, "workspace of one launch exceeds 32-bit buffer offsets");
// workspace elements the PlonK path writes its two G1 points to (bn254_vm.h: VE_LX, VE_CX)
#define VE_LX_ELEM 8
#define VE_CX_ELEM 6
struct G16LaunchArgs {
  const uint8_t* proofs; size_t stride;
  const uint8_t* inputs; int n_public;
  size_t n;                 // <= G16_MAX_LAUNCH
  int32_t* ws;              // G16_WS_BYTES_PER_PROOF * n bytes
  uint8_t* status;          // n bytes
  const int32_t* msm_tab;   // n_public * 32 * 255 entries of MSM_ENTRY_DWORDS
  const int32_t* k0;        // 18 dwords: affine K[0]
  const int32_t* gtab;      // BN_ATE_STEPS * FIXED_LINE_DWORDS: lines of the G2 argument paired with L
  const int32_t* dtab;      // same for the one paired with C
  const int32_t* target;    // 108 dwords: the GT element the product must equal, w-power (k) order
  int inputs_match_key;     // n_public + 1 == len(vk.K)
  int strict_scalars = 0;   // BN254_FLAG_STRICT_SCALARS: inputs >= r -> NOT_MEMBER
  int msm_comb = 0;         // msm_tab of a key with many inputs is in comb form
  uint16_t* msm_digits = nullptr;   // comb form: G16_COMB_COLS * n_public * n column digits (scratch)
  int part_of_larger = 0;   // this launch is one of several sub-batches of a larger batch: never the cooperative (small-batch) kernels
  int32_t* msm_part;        // wide keys: ceil(n_public / G16_WIDE_MSM_INPUTS_PER_LANE) * 27 * n dwords of partial sums, else nullptr
  // small batches (n <= G16_SPLIT_MAX_PROOFS): two extra streams and three events (fork, join, join) let the three pairs run their
  // Miller loops as three concurrent chains (the GPU is mostly idle at such sizes: latency, not throughput, is what counts)
  hipStream_t split_streams[2] = {nullptr, nullptr};
  hipEvent_t split_ev[3] = {nullptr, nullptr, nullptr};
};
#define G16_SPLIT_MAX_PROOFS 16384
#define G16_WIDE_MSM_MIN_INPUTS 16      // above this many public inputs the MSM runs as (proof, chunk) lanes + a reduction
#define G16_WIDE_MSM_INPUTS_PER_LANE 16
#define G16_COMB_TEETH 13               // comb tables of keys with many inputs (bn254_host.hpp::build_comb_table): 13 teeth x 20 columns >= 256 bits
#define G16_COMB_COLS 20
// column digit of a scalar given as eight 32-bit words (w[k]: bits 32 k .. 32 k + 31): bit t of the digit is bit col + G16_COMB_COLS * t of the scalar
// (bits from 256 up are zero).  Shared by k_g16_comb_digits and the host-side check (bn254_dbg_comb_mul).
static inline __host__ __device__ uint32_t g16_comb_digit(const uint32_t w[8], int col) {
  uint32_t idx = 0;
#pragma unroll
  for (int t = 0; t < G16_COMB_TEETH; t++) {
    const int base = G16_COMB_COLS * t;                   // static; the bit is base + col: at most one word further
    const int wi = base >> 5, sh = base & 31;
    const uint64_t two = ((uint64_t)(wi + 1 < 8 ? w[wi + 1 < 8 ? wi + 1 : 7] : 0u) << 32) | w[wi < 8 ? wi : 7];
    const uint32_t bit = wi < 8 ? (uint32_t)((two >> (sh + col)) & 1u) : 0u;
    idx |= bit << t;
  }
  return idx;
}
#define G16_WIDE_MSM_MAX_PROOFS 65536   // proofs per launch on the wide path (bounds the partial-sum buffer: 442 MB at 1024 inputs)
// kernel kinds of the Groth16 path (one launch per Fp12-level operation of the verification program)
enum {
  KID_PREPARE, KID_SUBGROUP, KID_VM_INIT, KID_F12_SQR, KID_MUL_LINE_FIXED, KID_F12_MUL,
  KID_CYCLO_SQR, KID_F12_CONJ, KID_F12_FROB, KID_F12_INV, KID_COMPARE, KID_F12_COPY, KID_CYCLO_SQR_N, KID_MILLER_DBL_VAR, KID_MILLER_ADD_VAR, KID_MSM_PARTIAL, KID_MSM_REDUCE, KID_MUL_LINE_FIXED2, KID_MILLER_SQR_DBL_VAR, KID_MILLER_STEP_DBL, KID_MILLER_STEP_ADD, KID_COOP_G16, KID_MILLER_RUN, KID_COUNT
};
extern const char* const bn254_kernel_kind_names[KID_COUNT];
// optional per-launch timing: every launch whose kind is in `mask` is bracketed by two events from the pool
struct G16Prof {
  uint32_t mask;        // bit k: time launches of kind k
  hipEvent_t* ev;       // 2 * cap events
  uint8_t* kid;         // kind of pair i
  int cap, used;
};
hipError_t bn254_launch_g16(const G16LaunchArgs& a, hipStream_t s, hipEvent_t* ev, G16Prof* prof);
// bn254_k_miller.hip: one whole step of the shared Miller loop (kind 0: doubling, with the squaring of f when do_sqr; 1..4: additions)
void bn254_launch_miller_step(bool do_sqr, int kind, int32_t* ws, uint32_t n, const uint8_t* status, unsigned grid, hipStream_t s, int et, int eb, int e, int epa,
                              const int32_t* t0, int ep0, int inf0, const int32_t* t1, int ep1, int inf1);
// a run of steps [s_begin, s_end) in one launch (bn254_vm.h::vm_miller_run; tab0 / tab1: the WHOLE line tables); kinds: the step table, a nibble per step
struct MillerKinds { uint8_t nib[44]; };
void bn254_launch_miller_run(const MillerKinds& kinds, int s_begin, int s_end, int32_t* ws, uint32_t n, const uint8_t* status, unsigned grid, hipStream_t s, int et, int eb,
                             int e, int epa, const int32_t* tab0, int ep0, int inf0, const int32_t* tab1, int ep1, int inf1);
// the same for two table-driven pairs and no variable pair (bn254_vm.h::vm_miller_run_fixed2)
void bn254_launch_miller_run_fixed2(const MillerKinds& kinds, int s_begin, int s_end, int32_t* ws, uint32_t n, const uint8_t* status, unsigned grid, hipStream_t s, int e,
                                    const int32_t* tab0, int ep0, int inf0, const int32_t* tab1, int ep1, int inf1);
// fixed-base tables of a key built on the device (bn254_k_comb.hip).  form 0: comb tables (points * 8192 entries), 1: byte-window tables (points * 32 * 255 entries), both of
// MSM_ENTRY_DWORDS dwords; pts = `points` affine points (18 dwords each, device memory); scratch: teeth_plane = 27 * points * teeth dwords, teeth_aff = 18 * points * teeth
// dwords, plane = 27 * points * entries dwords (bn254_tab_build_teeth / _entries: 13 / 8192 and 256 / 8192)
size_t bn254_tab_build_teeth(int form);
size_t bn254_tab_build_entries(int form);
size_t bn254_tab_build_out_entries(int form);     // entries of the finished table per point: 8192 | 32 * 255 | 16 * 65535 (form 2: 16-bit windows, bn254_msm.h)
hipError_t bn254_launch_tab_build(int form, const int32_t* pts, uint32_t points, int32_t* table, int32_t* teeth_plane, int32_t* teeth_aff, int32_t* plane, hipStream_t s);
// RLC batch mode (bn254_rlc.h): one launch part = n <= G16_MAX_LAUNCH proofs forming plan.groups groups
#include "bn254_rlc_plan.h"
struct RlcLaunchArgs {
  uint32_t key[11];         // ChaCha20 key (8 words) + nonce (3 words), fresh per call
  uint32_t counter_base;    // global index of this part's first proof (weights are a function of the global index)
  bn254::RlcPlan plan;
  uint8_t* grp_status;      // >= round_up(plan.groups, 256) bytes
  const int32_t* btab;      // line table of the G2 argument paired with alpha
  const int32_t* rlc_tab;   // window tables of -alpha and K[0]: 2 * 32 * 255 entries of MSM_ENTRY_DWORDS
  const int32_t* one;       // 108 dwords: 1 in GT
};
hipError_t bn254_launch_g16_rlc(const G16LaunchArgs& a, const RlcLaunchArgs& r, hipStream_t s);
hipError_t bn254_launch_gather_rows(uint8_t* dst, const uint8_t* src, size_t src_stride, uint32_t row_bytes, const uint32_t* idx, uint32_t m, hipStream_t s);
hipError_t bn254_launch_scatter_status(uint8_t* status, const uint8_t* fb_status, const uint32_t* idx, uint32_t m, hipStream_t s);
#define G1_GLV_TAB_BYTES_PER_LANE (16 * 28 * 4)
// PlonK's G1 multi-scalar multiplications as rows of a plan (bn254_msm.h, bn254_k_msm.hip)
namespace bn254 { struct MsmPlan; }
size_t bn254_g1_msm_scratch_lanes(const bn254::MsmPlan& plan, size_t n);   // lanes of window-table scratch (G1_GLV_TAB_BYTES_PER_LANE each) a launch over n items needs
hipError_t bn254_launch_g1_msm_rows(const bn254::MsmPlan& plan, const int32_t* terms, const uint8_t* flags, size_t n, int n_terms, int32_t* part, int32_t* glv_tab,
                                    const int32_t* tabs, hipStream_t s);
hipError_t bn254_launch_g1_sum_rows(const bn254::MsmPlan& plan, const int32_t* part, size_t n, uint32_t* out_words, uint8_t* out_inf, int32_t* ws, uint8_t* status, int e_x, int inf_bit,
                                    int e_x_b, int inf_bit_b, hipStream_t s);
hipError_t bn254_launch_pairing2_fixed(int32_t* ws, uint8_t* status, size_t n, const int32_t* tab0, const int32_t* tab1, const int32_t* target_one,
                                       int reject_code, hipStream_t s, hipStream_t aux, hipEvent_t ev_fork, hipEvent_t ev_join);
// cooperative layout for small batches (bn254_coop12.hip): twelve lanes per proof (one Fp number of every Fp12 value per lane), 39 KB of LDS per
// wavefront, the whole Miller loop / final exponentiation in one launch.  (The first generation, six lanes per proof, was retired in round 3:
// 2.95 ms against 2.11 ms at 4096 proofs; DESIGN.md section 5.4 keeps its measurements.)
#define COOP_T_ELEM 46          // = VE_S2: where the cooperative Miller loop leaves the running G2 point for k_g16_subgroup
// Groth16: passes of 1024 wavefronts x 5 proofs, 1.98 ms each.  6 passes = 11.8 ms against 12.9 ms of the lane kernels (one sub-batch, k_miller_run) at 30 720 proofs;
// 7 passes = 13.7 ms against 13.0 ms at 32 768 (profiles/r03_mid_batch_sweep.txt; until k_miller_run the hand-over was at 40 960)
#define COOP12_MAX_PROOFS 30720
// two-pair check of the PlonK path: its lane form is still one launch per operation, the cooperative kernel keeps the range it had
#define COOP12_MAX_PROOFS_FIXED 40960
hipError_t bn254_coop12_miller_g16(int32_t* ws, uint8_t* status, size_t n, const int32_t* tab0, const int32_t* tab1, const uint8_t* inputs, int n_public,
                                   int inputs_match_key, const int32_t* msm_tab, const int32_t* k0, int l_from_ws, int fuse_final_exp, const int32_t* target, hipStream_t s);
hipError_t bn254_coop12_final_exp(int32_t* ws, uint8_t* status, size_t n, hipStream_t s);
hipError_t bn254_coop12_miller_fixed(int32_t* ws, uint8_t* status, size_t n, int n_pairs, const int32_t* tab0, const int32_t* tab1, const int32_t* tab2,
                                     int e_p0, int e_p1, int e_p2, int inf0, int inf1, int inf2, int fuse_final_exp, const int32_t* target, int reject_code, hipStream_t s);
static inline size_t bn254_coop_max_proofs() { return COOP12_MAX_PROOFS; }
static inline size_t bn254_coop_max_proofs_fixed() { return COOP12_MAX_PROOFS_FIXED; }
double bn254_measure_valu_sustained(double ms_target);   // the same kernel back to back for ms_target milliseconds, one interval
double bn254_measure_valu_peak(int reps);   // lane-level v_mad_u64_u32 per second of the current device at four wavefronts per SIMD
hipError_t bn254_launch_dbg_fp_mul(const uint8_t* a, const uint8_t* b, uint8_t* o, size_t n, hipStream_t s);
hipError_t bn254_launch_dbg_fp12_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* o, size_t n, int32_t* ws, uint8_t* status, hipStream_t s);
// e(P_i, Q_i): needs a workspace of G16_WS_BYTES_PER_PROOF * n bytes and the step program
hipError_t bn254_launch_dbg_pairing(const uint8_t* g1, const uint8_t* g2, uint8_t* o, size_t n, int32_t* ws, uint8_t* status, hipStream_t s);
hipError_t bn254_launch_dbg_g2_ate(const uint8_t* g1, const uint8_t* g2, uint8_t* o, size_t n, int32_t* ws, uint8_t* status, hipStream_t s);
hipError_t bn254_launch_dbg_g2_subgroup(const uint8_t* g2, uint8_t* o, size_t n, hipStream_t s);
