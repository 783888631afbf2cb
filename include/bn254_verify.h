/*
 * bn254_verify.h -- C ABI of the MI355X-native batch BN254 verifier (libbn254_verify_amd.so).
 *
 * This is the drop-in boundary for the hot path of succinctlabs/snark-bn254-verifier.  The reference has no FFI of its
 * own; its only surface is the Rust API (paths relative to /root/reference):
 *     Groth16Verifier::verify(proof:&[u8], vk:&[u8], public_inputs:&[Fr]) -> Result<bool, Groth16Error>   verifier/src/lib.rs:44-49
 *     PlonkVerifier::verify(...)                                                                         verifier/src/lib.rs:69-74
 * and, below it, the `bn` crate calls that do all the work (groth16/verify.rs:70-77).  The entry points here are what a
 * thin Rust `extern "C"` wrapper binds to keep that surface and add `verify_batch(&[proof], &vk, &[[Fr]])`
 * (INTEGRATION.md shows the binding).  Plain pointers and sizes only; no C++ or torch types.
 *
 * Conventions
 *   - All field elements cross the boundary as 32-byte big-endian integers, exactly as in gnark files.
 *   - Return value = infrastructure status (BN254_OK or a negative BN254_E_* code).  Per-proof outcomes are reported
 *     only through status bytes (BN254_REJECT ... below); nothing panics, unlike the reference's unwrap()s.
 *   - The caller owns every buffer it passes; the library owns the opaque prepared-vk handle.
 *   - A prepared vk is immutable and may be shared between threads.  The library keeps one workspace per (key, device): batches
 *     against the same key and device are serialised by the library itself (enqueue under a per-device lock, each batch waits
 *     on the previous batch's completion event before it touches the workspace), whatever streams the callers use.
 *     (PlonK keys hand out contexts instead and run calls side by side: see there.)  A handle must not be freed while a call that uses it is in flight
 *     on another thread; the free functions wait for the GPU work the handle has enqueued and release its device memory.
 *   - There is NO CPU fallback: every verify entry point runs the HIP kernels and fails with BN254_E_NO_DEVICE if
 *     no gfx950 device is usable.
 */
#ifndef BN254_VERIFY_H
#define BN254_VERIFY_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- per-proof status bytes (one per proof, written to status[]) ------------------------------------------------
 * Mapping to the reference's observable behaviour (verifier/src/...):                                            */
enum {
  BN254_REJECT = 0,              /* Ok(false): pairing equation does not hold           groth16/verify.rs:77          */
  BN254_ACCEPT = 1,              /* Ok(true)                                                                           */
  BN254_ERR_NOT_MEMBER = 2,      /* a proof coordinate >= p: Field(NotMember), a panic via unwrap   converter.rs:85-86,144-147, lib.rs:45 */
  BN254_ERR_NOT_ON_CURVE = 3,    /* Group(NotOnCurve)                                     converter.rs:87,152          */
  BN254_ERR_NOT_IN_SUBGROUP = 4, /* Group(NotInSubgroup), G2 point B only                 converter.rs:152             */
  BN254_ERR_INPUT_LEN = 5,       /* Err(PrepareInputsFailed): len(inputs)+1 != len(vk.K)  groth16/verify.rs:54-56      */
  BN254_ERR_MALFORMED = 6,       /* everything else the reference turns into a panic: short buffer, flag 0b00, no square root */
  BN254_ERR_OPENING_MISMATCH = 7,/* PlonK Error::OpeningPolyMismatch                      plonk/verify.rs:212-214      */
  BN254_ERR_PAIRING_FAILED = 8,  /* PlonK Error::PairingCheckFailed                       plonk/kzg.rs:185-187         */
  BN254_ERR_BSB22_MISMATCH = 9,  /* PlonK Error::Bsb22CommitmentMismatch                  plonk/verify.rs:52-54        */
  BN254_ERR_INVERSE = 10         /* PlonK Error::InverseNotFound                          plonk/verify.rs:106          */
};

/* ---- infrastructure return codes ----------------------------------------------------------------------------- */
enum {
  BN254_OK = 0,
  BN254_E_BAD_ARG = -1,
  BN254_E_NO_DEVICE = -2,   /* no usable HIP device / kernel image: the product never falls back to the CPU */
  BN254_E_HIP = -3,         /* a HIP runtime call failed; bn254_last_error() has the text */
  BN254_E_VK = -4,          /* the verifying key bytes do not load: whatever load_groth16_verifying_key_from_bytes / load_plonk_verifying_key_from_bytes turn into a panic
                               (short buffer, flag 0b00, no square root; groth16/converter.rs:28-89, plonk/converter.rs:18-119).  Status byte equivalent: BN254_ERR_MALFORMED.
                               A key that LOADS is never refused: no K points at all (every proof: loader error, else BN254_ERR_INPUT_LEN), G2 elements outside the r-torsion
                               (the loaders are "unchecked", converter.rs:113-133: computed on, as the reference does) */
  BN254_E_NOMEM = -5
};

/* ---- verifying-key interpretation ---------------------------------------------------------
 * BN254_VK_REFERENCE reproduces the reference's actual input->output function: compressed G2 roots ordered by c0 only
 * (as the pinned `bn` does), beta negated on load (groth16/converter.rs:79) and the literal equation of
 * groth16/verify.rs:70-77.  BN254_VK_GNARK uses gnark-exact G2 decompression and gnark's equation
 * e(A,B) = e(alpha,beta) e(L,gamma) e(C,delta).  The two agree on every proof for verifying keys whose beta2, gamma2 have
 * y.c0 / y.c1 in different halves of [0,p) and delta2 in the same half (the SP1 key is presumably of this kind). */
enum { BN254_VK_REFERENCE = 0, BN254_VK_GNARK = 1 };

/* ---- per-call option flags of the batch entry points ------------------------------------------------------------
 * BN254_FLAG_STRICT_SCALARS  public inputs >= r are answered with BN254_ERR_NOT_MEMBER instead of being used modulo r.  The
 *     default (flag clear) is the reference's behaviour: bn::Fr::from_slice stores any 256-bit value and AffineG1 * Fr consumes it
 *     bit by bit, so x and x + r verify alike (examples/script/src/main.rs:204-213).  In the reference's
 *     typed API a range-checked Fr would fail at construction, before verify() is entered, so the strict error takes precedence
 *     over every proof error.
 * BN254_FLAG_RLC  random-linear-combination batch mode (the reference batches the same way inside KZG,
 *     plonk/kzg.rs:149-187): proofs are checked in groups with fresh random weights r_i (128 bits of entropy each),
 *         prod_i e(r_i A_i, B_i) * e(sum r_i L_i, gamma') * e(sum r_i C_i, delta') * e(-(sum r_i) alpha, beta') == 1,
 *     one variable-argument Miller loop per proof and one final exponentiation per group; proofs of a group that fails are
 *     re-verified by the exact path, so the status bytes are those of the exact path except with probability <= 2^-120 per batch
 *     (a false ACCEPT).  Loader errors (member / curve / subgroup) are always exact.  The weights come from ChaCha20 keyed by
 *     getrandom(2) per call.  The call synchronises the stream once (to learn which groups failed).
 *     The mode is a longer pipeline than the exact path and pays from about 200 000 proofs (2.0 x at 2^20): below that the flag is
 *     ignored (bn254_set_rlc_params, or BN254_RLC_MIN_BATCH in the environment when the library is loaded, moves the threshold; never below 64).
 *     Adaptive: an RLC pass costs about half an exact pass and every proof of a failed group pays the exact pass on top, so per
 *     (key, device) the share of proofs that fell back is tracked, and while it is above 0.45 the flag is ignored (the exact path
 *     runs: same status bytes) except for one measuring RLC pass every 8 calls.  bn254_set_rlc_params(-1, 0, -1) (or
 *     BN254_RLC_ADAPTIVE=0 in the environment at load time) switches this off; bn254_groth16_rlc_state reports the tracked share (-1: no RLC pass yet) and the number of bypassed calls. */
enum { BN254_FLAG_STRICT_SCALARS = 1u, BN254_FLAG_RLC = 2u };

typedef struct bn254_g16_pvk bn254_g16_pvk;

/* Parse + decompress a gnark Groth16 verifying key ONCE (replaces the per-call load_groth16_verifying_key_from_bytes,
 * groth16/converter.rs:28-89, and the per-call pairing(alpha, beta), groth16/verify.rs:70): decompression, e(alpha,beta),
 * Miller-loop line tables for the two fixed G2 arguments.  Host work; no GPU needed: 3 ms for a 2-input key, 4 ms for 16 inputs, 14 ms for 1024.
 * The fixed-base tables for vk.K (13-bit windows, 13 MB per input; keys with more than 16 inputs: comb tables, 655 KB per input) are NOT built here: each device builds its
 * own copy from the key's K points on first use (bn254_groth16_reserve, or the first batch; csrc/bn254_k_comb.hip: 2 ms for 2 inputs, 17 ms for 1024), and the host keeps
 * 72 bytes per input (until round 5 the host built them: 9 ms for 2 inputs, 0.18 s for 16, 2.2 s for 1024 on 8 threads, and held the copy).  LIMIT a caller must still plan
 * for: 671 MB of DEVICE memory per 1024-input key and device (+ 226 MB of scratch during the construction); keep the handle, do not prepare per call.
 * BN254_TABLES_HOST=1 keeps the host construction. */
int bn254_groth16_vk_prepare(const uint8_t* vk, size_t vk_len, unsigned mode, bn254_g16_pvk** out);
void bn254_groth16_vk_free(bn254_g16_pvk* pvk);
/* number of public inputs the key expects (len(vk.K) - 1); SIZE_MAX for a key without K points: no input count satisfies groth16/verify.rs:54 */
size_t bn254_groth16_vk_num_public(const bn254_g16_pvk* pvk);

/* verify_batch on host buffers.  proofs: n records of proof_stride bytes (>= 256; bytes beyond 256 -- gnark's commitment
 * count / commitments / PoK -- are ignored exactly as in groth16/converter.rs:15-25).  public_inputs: n * n_public * 32 bytes,
 * big-endian, NOT range-checked and used modulo r exactly like bn::Fr::from_slice + AffineG1 * Fr.
 * status: n bytes.  device: HIP device ordinal. */
int bn254_groth16_verify_batch(const bn254_g16_pvk* pvk, const uint8_t* proofs, size_t proof_stride,
                               const uint8_t* public_inputs, size_t n_public, size_t n, uint8_t* status, int device, unsigned flags);
/* Same over several GPUs of the node: bit d of device_mask selects HIP device d.  The batch is cut into contiguous shards, one
 * host thread per device drives its shard through bn254_groth16_verify_batch, and the status bytes land in the caller's buffer
 * (the gather of SURVEY.md section 8(e) done by the host threads; a multi-process job gathers with RCCL instead, see bench.py). */
int bn254_groth16_verify_batch_multi(const bn254_g16_pvk* pvk, const uint8_t* proofs, size_t proof_stride,
                                     const uint8_t* public_inputs, size_t n_public, size_t n, uint8_t* status,
                                     uint64_t device_mask, unsigned flags);
/* The shard plan bn254_groth16_verify_batch_multi follows (host arithmetic, no GPU): devices[k] = the k-th set bit of device_mask, its shard
 * the contiguous range [first[k], first[k] + count[k]) of the batch -- balanced, the first n % w shards one proof longer, the partition of
 * SURVEY.md section 8(e) (2^20 proofs over 8 GPUs = 131 072 each).  device_count = number of devices the caller has (a set bit at or
 * above it is BN254_E_BAD_ARG). */
int bn254_shard_plan(size_t n, uint64_t device_mask, int device_count, int devices[64], size_t first[64], size_t count[64], int* n_shards);

/* The gather of a multi-PROCESS job (one process per GPU, rank r verifying shard r of bn254_shard_plan(n, ranks 0..world-1)): ONE ncclAllGather of the
 * status bytes on hip_stream, every rank ends with the full n-byte vector in d_full (device memory).  nccl_comm: the caller's ncclComm_t (RCCL; the
 * library does not link RCCL: ncclAllGather is looked up in the process, then in librccl.so).  d_local: this rank's shard statuses (device memory).
 * d_scratch: world * ceil(n / world) bytes of device memory, needed only when n is not a multiple of world (ragged shards travel as padded blocks).
 * bench.py's ranks do the same through torch.distributed (snark-bn254-verifier_amd/sharding.py); this entry is for a Rust / C host. */
int bn254_status_all_gather(void* nccl_comm, int world, int rank, const void* d_local, size_t n, void* d_full, void* d_scratch, void* hip_stream);

/* Same, with proofs / public_inputs / status already resident in the memory of `device` (the bench path: inputs in
 * HBM when the timed region starts).  hip_stream is a hipStream_t (NULL = default stream); the call only enqueues
 * work and returns, so the caller synchronises the stream before reading status.  Use bn254_groth16_reserve() first to
 * keep the call free of allocations (graph capture). */
int bn254_groth16_verify_batch_device(const bn254_g16_pvk* pvk, const void* d_proofs, size_t proof_stride,
                                      const void* d_public_inputs, size_t n_public, size_t n, void* d_status,
                                      int device, void* hip_stream, unsigned flags);
/* pre-allocate the per-device workspace for batches of up to n proofs and upload the key's tables */
int bn254_groth16_reserve(const bn254_g16_pvk* pvk, size_t n, int device);

/* Groth16Verifier::verify (lib.rs:44-49) as one call: one proof, one status byte, vk given as bytes on every call like the
 * reference.  The prepared form of the last four keys (exact byte match, per mode) is kept, so only the first call with a key pays
 * its preparation (about 6.5 ms of an 8.5 ms call; 2 ms afterwards: profiles/r05_new_key_cost.txt); BN254_KEY_CACHE=0 in the environment switches the cache off, BN254_KEY_CACHE=N (1 .. 64) keeps the last N keys (default 4).
 * bn254_plonk_verify does the same.  Runs on the GPU (device 0). */
int bn254_groth16_verify(const uint8_t* proof, size_t proof_len, const uint8_t* vk, size_t vk_len,
                         const uint8_t* public_inputs, size_t n_public, unsigned mode, uint8_t* status);

/* Raw gnark proof writer: A (64) | B (128: x.c1, x.c0, y.c1, y.c0) | C (64) | u32 0 (no commitments) | 64 zero bytes (commitment PoK):
 * the 324-byte form load_groth16_proof_from_bytes reads (groth16/converter.rs:14-26; SP1 fixtures carry exactly this). */
#define BN254_GROTH16_RAW_PROOF_LEN 324
int bn254_groth16_proof_write_raw(const uint8_t a[64], const uint8_t b[128], const uint8_t c[64], uint8_t out[BN254_GROTH16_RAW_PROOF_LEN]);

/* ---- PlonK (gnark / SP1 format), BASELINE configs[3] --------------------------------------------------------------
 * Replaces PlonkVerifier::verify (verifier/src/lib.rs:69-73) = load_plonk_proof_from_bytes (plonk/converter.rs:121-178) +
 * load_plonk_verifying_key_from_bytes (plonk/converter.rs:18-119, hoisted into vk_prepare) + verify_plonk
 * (plonk/verify.rs:46-317: Fiat-Shamir transcripts transcript.rs:15-108, BSB22 hash_to_field.rs:9-122, kzg::fold_proof and
 * kzg::batch_verify_multi_points plonk/kzg.rs:87-190).  Everything per proof runs on the GPU: the transcripts, the hash-to-field and the scalar-field
 * arithmetic as one-proof-per-lane kernels (csrc/bn254_k_plonk.hip, compiled from the same source as the host-thread stages that BN254_PLONK_HOST=1 still
 * selects), every group operation (24 G1 scalar multiplications and the two-pair pairing check per proof) as before.
 * Status bytes: BN254_ACCEPT or an error code; PlonK never returns BN254_REJECT (plonk/verify.rs:316).  Each proof occupies
 * proof_stride bytes (>= its length: 904 for the SP1 circuits); public inputs are n_public x 32 big-endian bytes per proof.
 * Threads: a prepared key may be used from several host threads at once.  Each call takes one of the key's eight per-device contexts (stream, device
 * buffers, pinned staging) per sub-batch of its plan (bn254_set_plonk_params) and a call that finds too few free waits; up to eight batches of 4096 are therefore in
 * flight on one key, which is how a verifier that always has requests pending should drive it: a single batch of that size is a chain of latency-bound
 * launches (1.06 M proofs/s), two in flight give 1.2-1.3 M proofs/s, four 1.4-1.6 M; calls of 65 536 proofs and more run at 2.5-3.1 M proofs/s. */
typedef struct bn254_plonk_pvk bn254_plonk_pvk;
int bn254_plonk_vk_prepare(const uint8_t* vk, size_t vk_len, bn254_plonk_pvk** out);
void bn254_plonk_vk_free(bn254_plonk_pvk* pvk);
size_t bn254_plonk_vk_num_public(const bn254_plonk_pvk* pvk);
int bn254_plonk_verify_batch(const bn254_plonk_pvk* pvk, const uint8_t* proofs, size_t proof_stride, const uint8_t* public_inputs,
                             size_t n_public, size_t n, uint8_t* status, int device);
/* The same with flags.  BN254_FLAG_RLC (round 4): the pairing checks of a pass are batched across proofs -- every proof's two points carry a random 128-bit
 * weight (drawn per call, folded into the scalars of the multi-scalar multiplication at no group cost), the weighted points of the 64 proofs of a wavefront are
 * added and ONE pairing check runs per group; the proofs of a group that fails are then checked one by one, so the status bytes are those of the exact path
 * except that a forged proof is accepted with probability ~2^-127 (weights are odd 128-bit values; the same kind of batching the reference applies to a proof's two openings, plonk/kzg.rs:149-187).
 * Honoured from 8192 proofs per pass (BN254_PLONK_RLC_MIN); below, the one remaining pairing is the same latency-bound launch and the flag changes nothing.  (The
 * diagnostic host-thread stages of BN254_PLONK_HOST=1 ignore the flag.)  Any other flag bit is refused with BN254_E_BAD_ARG. */
int bn254_plonk_verify_batch_flags(const bn254_plonk_pvk* pvk, const uint8_t* proofs, size_t proof_stride, const uint8_t* public_inputs,
                                   size_t n_public, size_t n, uint8_t* status, int device, unsigned flags);
/* The same three entry shapes as Groth16 (north_star: one verify_batch surface for both verifiers):
 * _device  proofs, public inputs and status bytes resident in the memory of `device`.  Unlike the Groth16 entry this one is host-synchronous: it first waits for the work
 *          already enqueued on hip_stream (whatever still writes the inputs), runs the passes on the key's own context streams and returns when the status bytes are in
 *          d_status (a PlonK batch is several passes on several contexts driven by host threads, and BN254_FLAG_RLC has to read a counter back between two stages).
 * _multi   several GPUs of the node: the contiguous shards of bn254_shard_plan, one host thread per device through the host-buffer entry.
 * reserve  allocates NOW what a batch of up to n proofs needs on `device` (contexts of the plan, their buffers; proof_stride > 0: also the pinned staging of the host-buffer
 *          entry for records of that stride), so that the batch itself neither allocates nor frees.  Footprint per context, measured for the SP1 key shape (about 10 KB per proof of capacity plus
 *          1.8 KB per proof and variable MSM term): 0.16 GB for passes of up to 5040 proofs, 1.7 GB for 65 536, 3.5 GB for 131 072, 6.9 GB for 262 144; a batch above
 *          65 536 proofs uses up to eight contexts of its pass size (two for 262 144 proofs); beside them the window tables of the key's points, 131 MB for the reference's key
 *          (13 MB per point, built on the device at the key's first use).  bn254_plonk_footprint reports what a key holds on a device right now. */
int bn254_plonk_verify_batch_device(const bn254_plonk_pvk* pvk, const void* d_proofs, size_t proof_stride, const void* d_public_inputs, size_t n_public, size_t n,
                                    void* d_status, int device, void* hip_stream, unsigned flags);
int bn254_plonk_verify_batch_multi(const bn254_plonk_pvk* pvk, const uint8_t* proofs, size_t proof_stride, const uint8_t* public_inputs, size_t n_public, size_t n,
                                   uint8_t* status, uint64_t device_mask, unsigned flags);
int bn254_plonk_reserve(const bn254_plonk_pvk* pvk, size_t n, size_t proof_stride, int device);
int bn254_plonk_footprint(const bn254_plonk_pvk* pvk, int device, size_t* bytes, int* contexts);
int bn254_plonk_verify(const uint8_t* proof, size_t proof_len, const uint8_t* vk, size_t vk_len, const uint8_t* public_inputs,
                       size_t n_public, uint8_t* status);
/* Knobs of the PlonK batch plan (process-wide, atomic; -1 leaves a knob alone; initial values from BN254_PLONK_PIECE / _WORKERS / _BIG_FROM / _BIG_PIECE, read once at
 * load time): below big_from proofs a batch is up to `workers` chains of passes of at most `piece` proofs side by side (latency-bound launches), from big_from on
 * few passes of up to big_piece <= 262144 proofs (throughput-bound launches).  big_from = 0 (the default) selects the plan measured on the MI355X: chains up to
 * ~9000 proofs, one pass up to ~20 000, two passes side by side up to ~40 000, one pass up to 65 536, passes of big_piece (default 131 072) on up to eight contexts beyond.
 * Same status bytes whatever the plan. */
void bn254_set_plonk_params(long piece, int workers, long big_from, long big_piece);
/* Durations (ms) of the first sub-batch of the bn254_plonk_verify_batch that finished last on `device`, from HIP events on the sub-batch's stream:
 *   [0] host: staging copy into pinned memory (with BN254_PLONK_HOST=1: stage 1 on host threads)      [1] k_plonk_stage1
 *   [2] k_g1_msm_rows of the linearised-polynomial digest   [3] its k_g1_sum_affine                   [4] k_plonk_stage2 (BN254_PLONK_HOST=1: host stage 2)
 *   [5] k_g1_msm_rows of the KZG check (P0 and P1)          [6] their k_g1_sum_affine                 [7] the pairing check     [8] the sub-batch, host wall time
 * lanes: lanes (rows x items rounded up to 64) of the two k_g1_msm_rows launches. */
#define BN254_PLONK_NUM_TIMINGS 9
int bn254_plonk_last_timing(const bn254_plonk_pvk* pvk, int device, float ms[BN254_PLONK_NUM_TIMINGS], size_t lanes[2]);

/* ---- gnark / SP1 formats, both directions (host only) ------------------------------------------------------------------
 * Point codecs of verifier/src/converter.rs:23-153.  compress: uncompressed big-endian coordinates (G1: x | y; G2: x.c1 | x.c0 |
 * y.c1 | y.c0) -> gnark compressed form (flag 0b10 / 0b11 = lexicographically smallest / largest y in the top two bits).
 * decompress: the inverse; `checked` selects compressed_x_to_g{1,2}_point (converter.rs:46,91: curve and, for G2, r-torsion
 * checks) over the unchecked variants (converter.rs:62,113) the key loaders use; mode = BN254_VK_REFERENCE / BN254_VK_GNARK
 * picks the reading of the G2 root order.  *status: BN254_ACCEPT, BN254_ERR_MALFORMED (flag 0b00, no square root),
 * BN254_ERR_NOT_ON_CURVE, BN254_ERR_NOT_IN_SUBGROUP.
 * bn254_sp1_fixture_parse: the SP1 v2.0.0 SP1ProofWithPublicValues files of examples/binaries/ (bincode) -> variant (2 PlonK,
 * 3 Groth16), raw gnark proof bytes, the two public inputs as 32-byte big-endian values, and the vkey hash: exactly what
 * examples/script/src/main.rs:115-138 feeds to the verifiers. */
int bn254_g1_compress(const uint8_t xy[64], uint8_t out[32]);
int bn254_g2_compress(const uint8_t xy[128], uint8_t out[64]);
int bn254_g1_decompress(const uint8_t in[32], uint8_t out[64], int checked, uint8_t* status);
int bn254_g2_decompress(const uint8_t in[64], uint8_t out[128], unsigned mode, int checked, uint8_t* status);
int bn254_sp1_fixture_parse(const uint8_t* buf, size_t len, int* variant, uint8_t* raw_proof, size_t raw_cap, size_t* raw_len,
                            uint8_t public_inputs[64], uint8_t vkey_hash[32]);

/* ---- measurement support ------------------------------------------------------------------------------------------
 * A sub-batch above COOP12_MAX_PROOFS runs as about 120 kernel launches: k_g16_prepare, k_vm_init, the whole Miller loop as ONE k_miller_run (or a few, g16_launch_form),
 * k_g16_subgroup, one launch per Fp12-level operation of the final exponentiation (k_f12_mul x 60, k_f12_cyclo_sqr_n x 39, ...) and k_g16_compare; up to
 * COOP12_MAX_PROOFS as four (k_g16_prepare, the cooperative kernel, k_g16_subgroup, k_g16_compare).  When profiling is enabled, verify_batch_device records HIP
 * events on the launch stream (a) at the four phase boundaries (prepare | subgroup | Miller loop | final exponentiation) and
 * (b) around every launch whose kernel kind is selected by bn254_set_profile_kernels (bit i = kind i, default all).
 * After the stream has been synchronised bn254_groth16_last_kernel_ms returns the phase durations and
 * bn254_groth16_kernel_profile the number of launches and the summed duration per kernel kind, together with the number of
 * proofs each launch covered (the first sub-batch when the batch is split over concurrent streams, BN254_STREAMS).
 * bn254_set_profiling: 0 off; 1 the event pairs of (b) are those of the LAST batch; 2 they accumulate over every batch enqueued since the last call of one of the
 * two setters (up to 1024 launches per sub-batch stream, further ones are not recorded), so that a caller timing back-to-back batches reads them once, after its
 * final synchronisation, instead of waiting for each batch. */
#define BN254_G16_NUM_KERNELS 4   /* phases */
void bn254_set_profiling(int enabled);
void bn254_set_profile_kernels(unsigned mask);
int bn254_groth16_last_kernel_ms(const bn254_g16_pvk* pvk, int device, float ms[BN254_G16_NUM_KERNELS]);
int bn254_groth16_rlc_state(const bn254_g16_pvk* pvk, int device, float* fallback_share, unsigned* bypassed_calls);   /* BN254_FLAG_RLC, adaptive use */
/* A large batch runs as two sub-batches on two streams; the HIP runtime maps the streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4) and
 * streams that share a queue run one after the other.  The library does not touch the environment (GPU_MAX_HW_QUEUES=8 is a deployment setting);
 * it measures: a two-stream batch of a (key, device) is bracketed with events, a later call reads them.  overlap = sum of the two sub-batches' durations /
 * their union (~2 side by side, ~1 one after the other; -1 not measured yet).  One measurement decides nothing (another tenant's kernels, a tool that serialises dispatches):
 * three in a row must read "one after the other" before single_stream = 1 (batches that fit one launch then run as one sub-batch), one that reads "side by side" settles it
 * the other way; on single_stream every 256th batch runs two sub-batches again and is measured, so a transient cause does not pin the key to the slower plan.  The one-line
 * explanation is kept per (key, device): this call copies it to the calling thread's bn254_last_diagnostic(). */
int bn254_groth16_stream_overlap(const bn254_g16_pvk* pvk, int device, float* overlap, int* single_stream);
const char* bn254_last_diagnostic(void);
/* Knobs of BN254_FLAG_RLC (process-wide, atomics; a negative argument leaves that knob alone): the batch size from which the flag is honoured
 * (default 200 000, never below 64), the adaptive bypass on / off, and the lanes a launch part must keep for its proofs to share Miller-loop
 * accumulators (default 65536).  The environment variables BN254_RLC_MIN_BATCH / BN254_RLC_ADAPTIVE / BN254_RLC_SHARE_MIN_LANES give the
 * initial values and are read once, when the library is loaded. */
void bn254_set_rlc_params(long min_batch, int adaptive, long share_min_lanes);
const char* bn254_groth16_kernel_name(int i);                 /* phase names */
int bn254_groth16_num_kernel_kinds(void);
const char* bn254_groth16_kernel_kind_name(int i);
int bn254_groth16_kernel_profile(const bn254_g16_pvk* pvk, int device, unsigned launches[], float total_ms[], size_t* proofs_per_launch);
/* Same over the first TWO sub-batches (two streams side by side), plus union_ms[kind]: the length of the union of the launch intervals of that kind
 * on a common time base.  Work of all the launches / union = the rate the GPU delivered while that kernel kind ran, whether the two streams' launches
 * overlapped (union = about one launch) or ran one after the other (union = the sum). */
int bn254_groth16_kernel_profile_all(const bn254_g16_pvk* pvk, int device, unsigned launches[], float total_ms[], float union_ms[], size_t* proofs_per_launch);

/* ---- synthetic gnark-format workload generator (bench / tests; host threads, no GPU) --------------------------------
 * Deterministic (SplitMix64 seed).  Writes a gnark-compressed verifying key (292 + 32 (n_public+1) + 4 + 128 bytes), n
 * proofs (256 bytes each, A | B | C uncompressed) that satisfy gnark's equation, their public inputs, and the status the
 * verifier must return.  If invalid_every > 0 every invalid_every-th proof is corrupted, cycling through: public input
 * + 1 (REJECT), C + G1 (REJECT), A.y + 1 (NOT_ON_CURVE), B replaced by a twist point outside G2 (NOT_IN_SUBGROUP),
 * A.x >= p (NOT_MEMBER).  agree bit 0: the key is sampled so that BN254_VK_REFERENCE and BN254_VK_GNARK agree on it; bit 1:
 * every proof with index = 3 (mod 7) gets a last public input that makes L = K0 + sum x_i K_i the identity (still a valid proof). */
size_t bn254_synth_groth16_vk_len(size_t n_public);
int bn254_synth_groth16(uint64_t seed, size_t n_public, size_t n, int invalid_every, int agree, int threads,
                        uint8_t* vk_out, uint8_t* proofs_out, uint8_t* inputs_out, uint8_t* expected_status_out);
/* proofs [first, first + n) of the same stream (proof i depends on (seed, i) only), written to positions 0 .. n-1: a rank of a sharded job
 * generates just its own contiguous shard; the key is the same for every range */
int bn254_synth_groth16_range(uint64_t seed, size_t n_public, size_t first, size_t n, int invalid_every, int agree, int threads,
                              uint8_t* vk_out, uint8_t* proofs_out, uint8_t* inputs_out, uint8_t* expected_status_out);

/* ---- probes of the device arithmetic, used by the GPU parity tests (tests/test_gpu_*.py) ---------------------------
 * Each runs one lane per item on `device` and copies the result back.  Fp12 layout: 12 x 32 bytes in tower order
 * c0.c0.c0, c0.c0.c1, c0.c1.c0, ... c1.c2.c1; G1: x | y; G2: x.c1 | x.c0 | y.c1 | y.c0 (gnark order). */
/* measurement probe: lane-level v_mad_u64_u32 per second of `device` (four wavefronts per SIMD, launches of about 2 ms, 16 independent chains per lane): the VALU peak of THIS box */
int bn254_dbg_valu_peak(int device, double* mads_per_s);
/* the same kernel back to back for about ms_target (<= 2000) milliseconds, timed as one interval: the rate the box SUSTAINS over the length of the path's long kernels */
int bn254_dbg_valu_peak_sustained(int device, double ms_target, double* mads_per_s);
int bn254_dbg_fp_mul(const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n, int device);                 /* n x 32 B each */
int bn254_dbg_fp12_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n, int device);        /* 0 mul 1 sqr 2 inv 3 cyclo_sqr(after easy part) 4 frob1 */
int bn254_dbg_pairing(const uint8_t* g1, const uint8_t* g2, uint8_t* out_gt, size_t n, int device);          /* e(P_i, Q_i), n x 384 B */
int bn254_dbg_g2_subgroup(const uint8_t* g2, uint8_t* out_flags, size_t n, int device);                       /* 1 = in G2 (gnark's psi relation, one kernel) */
int bn254_dbg_g2_subgroup_ate(const uint8_t* g1, const uint8_t* g2, uint8_t* out_flags, size_t n, int device); /* 1 = in G2: the product's test, from the Miller loop's final point (g1: any G1 points) */

/* stage 1 of the PlonK path as the DEVICE runs it (csrc/bn254_k_plonk.hip), for n proofs: zeta -- the last of the four chained Fiat-Shamir challenges
 * (plonk/verify.rs:62-95), 32-byte big-endian, canonical -- and the stage's status per proof (BN254_ACCEPT: alive; else the error it decided) */
int bn254_dbg_plonk_stage1(const bn254_plonk_pvk* pvk, const uint8_t* proofs, size_t proof_stride, const uint8_t* public_inputs, size_t n_public, size_t n,
                           uint8_t* zeta_out, uint8_t* status_out, int device);

/* host-only probe of the GLV scalar decomposition the PlonK MSMs use: k = (-1)^neg1 k1 + (-1)^neg2 k2 lambda (mod r), k1, k2 < 2^127 */
int bn254_dbg_glv_decompose(const uint8_t k32[32], uint8_t k1_16[16], uint8_t k2_16[16], int* neg1, int* neg2);

/* host-only probes of the PlonK batch plan (sub-batches side by side, proofs per sub-batch, proofs per pass) and of the MSM launches (csrc/bn254_msm.h): the row plan of
 * the stage-1 / stage-2 launch for a key with n_qcp commitments and n proofs under a lane budget (0 = the library's) -- rows, rows that use window-table scratch, the
 * scratch lanes that launch needs, its longest row in the planner's cost units, rows and fixed terms per sum, optionally the rows themselves (MSM_MAX_ROWS = 32 rows x 9 ints:
 * variable term (-1: none, or a joint row), pos_lo, pos_hi, unit term, sum, first scratch slot, fixed windows [lo, hi), and for a JOINT row -- several variable terms
 * walked together over all 128 positions, large launches -- the bit mask of its terms); and the scratch lanes a context of `capacity` proofs allocates for launches
 * of n_var variable terms.  tests/test_capi_cpu.py: need <= allocation for every n <= capacity. */
int bn254_dbg_plonk_plan(size_t n, size_t piece, int max_workers, int* workers, size_t* per_worker, size_t* per_pass);
/* host-only probe of the Groth16 plan (csrc/bn254_g16_plan.h: the functions the library itself allocates and enqueues by): a key with key_inputs public inputs
 * (comb != 0: comb tables), a context reserved for `reserved` proofs, a batch of n proofs with n_public inputs each.  alloc = {workspace bytes, partial-sum bytes,
 * digit bytes, proofs per launch of the wide MSM}; out: 8 values per launch {chunk, first proof in the chunk, proofs, stream slot (-1 = the caller's stream),
 * form (0 lane kernels, 1 cooperative, 2 latency mode), Miller steps per launch, first workspace byte, one past its last}.  tests/test_capi_cpu.py walks batch
 * sizes against reservations: every launch inside the allocation, concurrent launches disjoint, the batch covered exactly once. */
int bn254_dbg_g16_plan(size_t key_inputs, int comb, size_t reserved, size_t n, size_t n_public, int n_streams, int single_stream, uint64_t alloc[4], uint64_t* out,
                       int max_launches, int* n_launches);
/* ... and of BN254_FLAG_RLC's group status bytes: what the launch parts of a chunk of m proofs address (need) against what a context whose RLC buffers were sized
 * for `reserved` proofs holds (alloc) */
int bn254_dbg_g16_rlc_plan(size_t reserved, size_t m, int n_streams, int log2_group, int log2_share, size_t min_lanes, uint64_t* need, uint64_t* alloc);
size_t bn254_dbg_plonk_scratch_lanes(size_t capacity, int n_var);
/* ... and the projective points (rows x items) the row buffer of a context of `capacity` proofs holds for launches of that key shape and stage (stage 3: the weighted
 * second launch of BN254_FLAG_RLC): n_rows x n of every launch over n <= capacity items must fit (tests/test_msm_rows.py) */
size_t bn254_dbg_plonk_part_points(size_t capacity, int n_qcp, int stage);
int bn254_dbg_plonk_msm_plan(int n_qcp, int stage, size_t n, size_t lane_budget, int* n_rows, int* n_var_rows, size_t* scratch_lanes, int* chain, int sum_rows[2],
                             int fixed_terms[2], int* rows_out);

/* host-only probe of the modular inversion of the PlonK stages (which = 0: the constant-time form the stages use, 1: the Fermat form, 2: the classic
 * shift-and-subtract binary GCD; field 0: Fr, 1: Fp); 32-byte big-endian in / out */
int bn254_dbg_fr_inverse(const uint8_t in32[32], uint8_t out32[32], int which, int field);
/* host-only probe of the two Montgomery product forms of the PlonK stages (form 32: 8 x 32-bit words, what the device runs; 64: 4 x 64-bit limbs, what the host
 * runs): n products of 32-byte big-endian values (reduced first), out = canonical big-endian a * b mod m (field 0: Fr, 1: Fp) */
int bn254_dbg_fr_mul(const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n, int form, int field);

/* host-only probe of the comb tables used for keys with many public inputs (csrc/bn254_host.hpp::build_comb_table): x * P computed from P's table
 * and the column digits of the 256-bit big-endian x, as the kernels do; out64 = uncompressed point, all zero for the identity */
int bn254_dbg_comb_mul(const uint8_t p64[64], const uint8_t x32[32], uint8_t out64[64]);
/* prepared keys the single-proof entries keep (BN254_KEY_CACHE in the environment: unset 4, 0 off, N up to 64) */
int bn254_dbg_key_cache_slots(void);
/* the fixed-base tables `device` built for a key (csrc/bn254_k_comb.hip) against host arithmetic, as field values; *mismatches = entries that differ.  Comb tables (more
 * than 16 public inputs): every entry of the first `inputs` inputs against bn254_host.hpp::build_comb_table.  13-bit window tables (up to 16 inputs): every window's first,
 * middle and last entries and a pseudo-random sample of every input, each against d 2^(13 w) K by double-and-add. */
int bn254_dbg_comb_table_compare(const bn254_g16_pvk* pvk, int device, int inputs, size_t* mismatches);
int bn254_dbg_plonk_table_compare(const bn254_plonk_pvk* pvk, int device, size_t* mismatches);   /* the window tables of a PlonK key's points (csrc/bn254_fw.h): every window's first, middle and last entries and a pseudo-random sample */

/* Revision of this header's binary interface: bumped whenever a function changes its arguments, an array argument its length or a slot its meaning (5: this round --
 * BN254_PLONK_NUM_TIMINGS has been 9 since revision 4, bn254_dbg_plonk_msm_plan writes 9 ints per row).  A binding compares it with the value it was generated for. */
#define BN254_ABI_VERSION 5
int bn254_abi_version(void);
const char* bn254_status_string(int status_byte);
const char* bn254_last_error(void);
const char* bn254_version(void);

#ifdef __cplusplus
}
#endif
#endif
