/*
 * oracle/ -- CPU restatement of the reference's Groth16 / PlonK BN254 verification path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under snark-bn254-verifier_amd/ (the product) includes, links or
 * calls this code; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * What it restates (file:line relative to /root/reference):
 *   verifier/src/lib.rs:44-49,69-74            Groth16Verifier::verify / PlonkVerifier::verify
 *   verifier/src/groth16/verify.rs:53-78       prepare_inputs, verify_groth16
 *   verifier/src/groth16/converter.rs:14-89    gnark proof / vk loaders
 *   verifier/src/converter.rs:23-153           gnark point codecs
 *   verifier/src/plonk/{verify,kzg,converter}.rs, transcript.rs, hash_to_field.rs   (PlonK path)
 * The field / curve / pairing arithmetic is NOT in the reference tree: it lives in the git dependency
 * substrate-bn 0.7.0 (sp1-patches/bn, branch patch-v0.7.0, rev 3c53d2561492f26b9428c1d37d134031d0156152,
 * reference Cargo.lock:405-407).  It is restated here from its published algorithm (the zcash/libff
 * alt_bn128 design: flipped ate Miller loop over the NAF of 6u+2 with (ell_0, ell_VW, ell_VV) line
 * coefficients, mul_by_024, and the exp_by_neg_z final-exponentiation chain).
 *
 * Parity pinning: the 4 PlonK fixtures of examples/binaries + the PlonK vk recovered from
 * examples/program/elf/plonk are end-to-end known-answer tests for this arithmetic (tests/test_oracle_*.py).
 * Groth16 verify() results are NOT pinned by the reference's own files (its vk is absent from the tree);
 * only parse-level facts of the 4 Groth16 fixtures are.  See DESIGN.md "Oracle".
 */
#ifndef BN254_ORACLE_H
#define BN254_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes shared with include/bn254_verify.h (kept numerically identical; tests assert it) ---- */
enum {
  ORC_REJECT = 0,
  ORC_ACCEPT = 1,
  ORC_ERR_NOT_MEMBER = 2,      /* bn::FieldError::NotMember     (coordinate >= p)               */
  ORC_ERR_NOT_ON_CURVE = 3,    /* bn::GroupError::NotOnCurve                                      */
  ORC_ERR_NOT_IN_SUBGROUP = 4, /* bn::GroupError::NotInSubgroup (G2 only)                         */
  ORC_ERR_INPUT_LEN = 5,       /* Groth16Error::PrepareInputsFailed / PlonK InvalidWitness        */
  ORC_ERR_MALFORMED = 6,       /* everything the reference turns into a panic (short buffer, bad flag, no sqrt) */
  ORC_ERR_OPENING_MISMATCH = 7,/* PlonK Error::OpeningPolyMismatch                                */
  ORC_ERR_PAIRING_FAILED = 8,  /* PlonK Error::PairingCheckFailed                                 */
  ORC_ERR_BSB22_MISMATCH = 9,  /* PlonK Error::Bsb22CommitmentMismatch                            */
  ORC_ERR_INVERSE = 10         /* PlonK Error::InverseNotFound                                    */
};

/* vk interpretation modes */
enum {
  ORC_MODE_REFERENCE = 0, /* literal: G2 roots ordered by c0 only, beta negated on load, equation of groth16/verify.rs:70-77 */
  ORC_MODE_GNARK = 1      /* gnark-exact G2 decompression + e(A,B) = e(alpha,beta) e(L,gamma) e(C,delta) */
};

typedef struct { uint64_t l[4]; } u256;
typedef u256 fp;  /* Montgomery form mod p */
typedef u256 fr;  /* Montgomery form mod r */
typedef struct { fp c0, c1; } fp2;
typedef struct { fp2 c0, c1, c2; } fp6;
typedef struct { fp6 c0, c1; } fp12;
typedef struct { fp x, y; int inf; } g1a;
typedef struct { fp x, y, z; } g1j;
typedef struct { fp2 x, y; int inf; } g2a;
typedef struct { fp2 x, y, z; } g2j;

typedef struct {
  u256 m;       /* modulus */
  uint64_t inv; /* -m^-1 mod 2^64 */
  u256 r1;      /* 2^256 mod m  (Montgomery one) */
  u256 r2;      /* 2^512 mod m */
} fctx;

extern fctx FP, FR;
void orc_init(void); /* idempotent; every exported entry point calls it */

/* counters (Fp Montgomery multiplications incl. squarings) -- used to size the work model in DESIGN.md */
extern __thread uint64_t orc_fp_mul_count;

/* ---- u256 / prime field ---- */
int u256_cmp(const u256* a, const u256* b);
int u256_is_zero(const u256* a);
void u256_from_be(u256* o, const uint8_t* b32);
void u256_to_be(uint8_t* b32, const u256* a);
int u256_bit(const u256* a, int i);
void f_add(const fctx* F, u256* o, const u256* a, const u256* b);
void f_sub(const fctx* F, u256* o, const u256* a, const u256* b);
void f_neg(const fctx* F, u256* o, const u256* a);
void f_mul(const fctx* F, u256* o, const u256* a, const u256* b);
void f_sqr(const fctx* F, u256* o, const u256* a);
void f_to_mont(const fctx* F, u256* o, const u256* a);   /* a canonical (< m) -> Montgomery */
void f_from_mont(const fctx* F, u256* o, const u256* a); /* Montgomery -> canonical */
void f_pow(const fctx* F, u256* o, const u256* a, const u256* e /* plain integer */);
int f_inv(const fctx* F, u256* o, const u256* a);        /* 0 if a == 0 */
void f_reduce_be(const fctx* F, u256* o_mont, const uint8_t* be, size_t n); /* big-endian bytes mod m -> Montgomery */
int fp_sqrt(fp* o, const fp* a);                          /* 1 if a is a square */

/* ---- tower ---- */
void fp2_add(fp2* o, const fp2* a, const fp2* b);
void fp2_sub(fp2* o, const fp2* a, const fp2* b);
void fp2_neg(fp2* o, const fp2* a);
void fp2_conj(fp2* o, const fp2* a);
void fp2_mul(fp2* o, const fp2* a, const fp2* b);
void fp2_sqr(fp2* o, const fp2* a);
void fp2_mul_fp(fp2* o, const fp2* a, const fp* b);
void fp2_mul_xi(fp2* o, const fp2* a); /* times 9+i */
int fp2_inv(fp2* o, const fp2* a);
int fp2_sqrt(fp2* o, const fp2* a);
int fp2_eq(const fp2* a, const fp2* b);
int fp2_is_zero(const fp2* a);
void fp6_mul(fp6* o, const fp6* a, const fp6* b);
void fp12_one(fp12* o);
void fp12_mul(fp12* o, const fp12* a, const fp12* b);
void fp12_sqr(fp12* o, const fp12* a);
int fp12_inv(fp12* o, const fp12* a);
void fp12_conj(fp12* o, const fp12* a); /* unitary inverse */
void fp12_frob(fp12* o, const fp12* a, int power);
void fp12_mul_by_024(fp12* o, const fp12* a, const fp2* ell_0, const fp2* ell_vw, const fp2* ell_vv);
void fp12_cyclo_sqr(fp12* o, const fp12* a);
int fp12_eq(const fp12* a, const fp12* b);
int fp12_is_one(const fp12* a);

/* ---- groups ---- */
void g1_from_affine(g1j* o, const g1a* a);
void g1_to_affine(g1a* o, const g1j* a);
void g1_double(g1j* o, const g1j* a);
void g1_add(g1j* o, const g1j* a, const g1j* b);
void g1_mul(g1j* o, const g1j* a, const u256* k /* plain 256-bit integer, used bit by bit, NOT reduced */);
void g1_neg_affine(g1a* o, const g1a* a);
int g1_on_curve(const fp* x, const fp* y);
void g2_from_affine(g2j* o, const g2a* a);
void g2_to_affine(g2a* o, const g2j* a);
void g2_double(g2j* o, const g2j* a);
void g2_add(g2j* o, const g2j* a, const g2j* b);
void g2_mul(g2j* o, const g2j* a, const u256* k);
void g2_neg_affine(g2a* o, const g2a* a);
int g2_on_curve(const fp2* x, const fp2* y);
int g2_in_subgroup_naive(const g2a* q); /* [r-1]Q + Q == O, as bn's AffineG2::new */
void g1_generator(g1a* o);
void g2_generator(g2a* o);

/* ---- pairing (bn::pairing / bn::pairing_batch) ---- */
void miller_loop_batch(fp12* f, const g1a* ps, const g2a* qs, int n); /* pairs with an infinity operand are skipped */
void final_exponentiation(fp12* o, const fp12* f);
void final_exponentiation_plain(fp12* o, const fp12* f); /* easy part + square-and-multiply by (p^4-p^2+1)/r: cross-check only */
void pairing_batch(fp12* o, const g1a* ps, const g2a* qs, int n);

/* ---- gnark codecs (verifier/src/converter.rs) ---- */
int dec_g1_uncompressed(g1a* o, const uint8_t* b64);              /* converter.rs:78-88   -> status or ORC_ACCEPT */
int dec_g2_uncompressed(g2a* o, const uint8_t* b128);             /* converter.rs:135-153 */
int dec_g1_compressed_unchecked(g1a* o, const uint8_t* b32);      /* converter.rs:62-76   */
int dec_g2_compressed_unchecked(g2a* o, const uint8_t* b64, int mode); /* converter.rs:113-133 */
void enc_g1_uncompressed(uint8_t* b64, const g1a* p);
void enc_g2_uncompressed(uint8_t* b128, const g2a* p);
void enc_g1_compressed(uint8_t* b32, const g1a* p);               /* gnark encoder (the reference only decodes) */
void enc_g2_compressed(uint8_t* b64, const g2a* p);

/* ---- sha256 ---- */
typedef struct { uint32_t h[8]; uint8_t buf[64]; uint64_t len; } sha256_ctx;
void sha256_init(sha256_ctx* c);
void sha256_update(sha256_ctx* c, const uint8_t* d, size_t n);
void sha256_final(sha256_ctx* c, uint8_t out[32]);

/* ================= exported test / baseline API (byte level; all field values 32-byte big-endian) ========== */

/* Groth16Verifier::verify (lib.rs:44-49) in reference-faithful form: vk re-parsed and e(alpha,beta) recomputed
 * on every call, naive subgroup check, sequential double-and-add prepare_inputs, 3-way pairing_batch. */
int orc_groth16_verify(const uint8_t* proof, size_t proof_len, const uint8_t* vk, size_t vk_len,
                       const uint8_t* inputs /* n_inputs x 32 B BE, unreduced */, size_t n_inputs, int mode);
/* loop of the above over a batch; returns 0 */
int orc_groth16_verify_many(const uint8_t* proofs, size_t proof_stride, const uint8_t* vk, size_t vk_len,
                            const uint8_t* inputs, size_t n_inputs, size_t n, int mode, uint8_t* status);
/* batch mode of the same algorithm: vk loaded once, pairing(alpha, beta) computed once */
int orc_groth16_verify_many_prepared(const uint8_t* proofs, size_t proof_stride, const uint8_t* vk, size_t vk_len,
                                     const uint8_t* inputs, size_t n_inputs, size_t n, int mode, uint8_t* status);
const char* orc_build_flags(void);   /* compiler + flags of this build, for the cpu_baseline line */
/* PlonkVerifier::verify (lib.rs:69-74).  lambda32 = the KZG batching scalar (the reference draws it from OsRng,
 * plonk/kzg.rs:149-154); pass NULL for a fixed non-trivial constant. */
int orc_plonk_verify(const uint8_t* proof, size_t proof_len, const uint8_t* vk, size_t vk_len,
                     const uint8_t* inputs, size_t n_inputs, const uint8_t* lambda32);
/* stage goldens of the PlonK transcript: writes gamma,beta,alpha,zeta digests (4 x 32 B)
 * and the 48-byte hash_to_field output of the first BSB22 commitment */
int orc_plonk_pairing_inputs(const uint8_t* proof, size_t proof_len, const uint8_t* vk, size_t vk_len,
                             const uint8_t* inputs, size_t n_inputs, uint8_t* out384 /* P0 | P1 | Q0 | Q1, gnark uncompressed */);
int orc_plonk_pairing_inputs_lam(const uint8_t* proof, size_t proof_len, const uint8_t* vk, size_t vk_len, const uint8_t* inputs, size_t n_inputs,
                                 const uint8_t* lambda32, uint8_t* out384);
int orc_plonk_stage_digests(const uint8_t* proof, size_t proof_len, const uint8_t* vk, size_t vk_len,
                            const uint8_t* inputs, size_t n_inputs, uint8_t* out176);

/* small arithmetic probes used by the unit tests and by the GPU parity tests */
void orc_fp_op(int op /*0 add 1 sub 2 mul 3 inv 4 sqrt 5 neg*/, uint8_t* o32, const uint8_t* a32, const uint8_t* b32, int field /*0=Fp 1=Fr*/);
void orc_fp2_op(int op /*0 add 1 sub 2 mul 3 inv 4 sqrt 5 sqr*/, uint8_t* o64, const uint8_t* a64, const uint8_t* b64); /* (c0,c1) order */
void orc_fp12_op(int op /*0 mul 1 sqr 2 inv 3 frob1 4 frob2 5 frob3 6 cyclo_sqr 7 conj*/, uint8_t* o384, const uint8_t* a384, const uint8_t* b384);
int orc_g1_scalar_mul(uint8_t* o64, const uint8_t* p64, const uint8_t* k32);  /* gnark uncompressed; returns 0 if result is infinity (o zeroed) */
int orc_g1_add_bytes(uint8_t* o64, const uint8_t* p64, const uint8_t* q64);
int orc_g2_scalar_mul(uint8_t* o128, const uint8_t* p128, const uint8_t* k32);
int orc_g2_add_bytes(uint8_t* o128, const uint8_t* p128, const uint8_t* q128);
void orc_g1_gen(uint8_t* o64);
void orc_g2_gen(uint8_t* o128);
int orc_g2_subgroup_check(const uint8_t* p128); /* 1 in subgroup, 0 not, <0 not on curve */
/* Fp12 byte layout of the probes: 12 x 32 B, order c0.c0.c0, c0.c0.c1, c0.c1.c0, ... c1.c2.c1 (tower order) */
void orc_miller_loop(uint8_t* o384, const uint8_t* g1s /*n x 64*/, const uint8_t* g2s /*n x 128*/, int n);
void orc_final_exp(uint8_t* o384, const uint8_t* f384, int plain);
void orc_pairing_bytes(uint8_t* o384, const uint8_t* g1s, const uint8_t* g2s, int n);
int orc_decompress_g1(uint8_t* o64, const uint8_t* b32);
int orc_decompress_g2(uint8_t* o128, const uint8_t* b64, int mode);
void orc_compress_g1(uint8_t* o32, const uint8_t* b64);
void orc_compress_g2(uint8_t* o64, const uint8_t* b128);
void orc_sha256(uint8_t* o32, const uint8_t* d, size_t n);
void orc_expand_msg_xmd(uint8_t* out, size_t len, const uint8_t* msg, size_t msg_len, const uint8_t* dst, size_t dst_len);
uint64_t orc_get_fp_mul_count(void);   /* of the calling thread */
void orc_set_threads(int n);            /* OpenMP team size of orc_groth16_verify_many */
void orc_reset_fp_mul_count(void);

#ifdef __cplusplus
}
#endif
#endif
