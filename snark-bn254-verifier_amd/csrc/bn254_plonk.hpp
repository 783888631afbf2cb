// bn254_plonk.hpp -- host side of the PlonK batch verifier (BASELINE configs[3]; reference verifier/src/plonk/{verify,kzg,
// converter,proof}.rs, transcript.rs, hash_to_field.rs).
//
// Split of the work per proof:
//   once per key (host)   the loader, the transcript prefix (the window tables of the key's points, plonk_table_point, are built on each device: bn254_k_comb.hip)
//   per proof, DEVICE     everything: parsing, the Fiat-Shamir transcripts (SHA-256), BSB22 hash-to-field, the Fr arithmetic of the linearisation and the GLV
//                         decomposition (PlonkStage1 / PlonkStage2 below, one proof per lane in csrc/bn254_k_plonk.hip), then every group operation: the two
//                         multi-scalar multiplications as rows (bn254_msm.h, k_g1_msm_rows) and the two-pair pairing check (bn254_kernels.hip / bn254_coop12.hip)
// The same stage code compiles for the host: the library runs it there only under BN254_PLONK_HOST=1 (a debugging aid) and the sanitizer harness of
// tests/hostsan does.  The second transcript hashes the first MSM's result, so a pass is stage 1 -> MSM -> stage 2 -> MSM -> pairing, all on one stream.
// Nothing here is shared with oracle/: this is product code.
#pragma once
#include <cstring>
#include <string>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include "bn254_host.hpp"
#include "bn254_msm.h"

// The per-proof stages (parsing, transcripts, Fr arithmetic, GLV decomposition) also run ON THE DEVICE, one proof per lane (csrc/bn254_k_plonk.hip,
// round 3): that translation unit defines BN254_PLONK_DEVICE_TU before it includes this header, which turns PL_HD into __host__ __device__ and routes
// fr_ctx() / fp64_ctx() to copies of the constants in device memory.  Every other includer sees plain host code, as before.
#if defined(BN254_PLONK_DEVICE_TU) && defined(__HIPCC__)
#define PL_HD __host__ __device__ inline
#define PL_DEVICE_PASS defined(__HIP_DEVICE_COMPILE__)
#else
#define PL_HD inline
#endif

namespace bn254host {

// Diagnostics build only (make EXTRA=-DBN254_PLONK_MARKS, tools/plonk_stage_marks.py): the first lane of a launch stamps the 100 MHz wall clock at
// the marked points of the stages, so that a run tells where one lane's chain spends its time.  Not compiled into the product.
#if defined(BN254_PLONK_MARKS) && defined(BN254_PLONK_DEVICE_TU) && defined(__HIPCC__)
__device__ unsigned long long g_plonk_marks[32];
#endif
#if defined(BN254_PLONK_MARKS) && defined(BN254_PLONK_DEVICE_TU) && defined(__HIP_DEVICE_COMPILE__)
#define PL_MARK(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_plonk_marks[k] = wall_clock64(); } while (0)
#define PL_MARK_H(k) do { if (blockIdx.x == 0 && threadIdx.x == 64) g_plonk_marks[k] = wall_clock64(); } while (0)   /* first lane of the helper wavefront */
#else
#define PL_MARK(k) ((void)0)
#define PL_MARK_H(k) ((void)0)
#endif

#if defined(BN254_PLONK_MARKS)
// ... and every SHA-256 compression of the first lane (device) / of the last proof (host): the 16 message words as read and the 8 state words after
#if defined(BN254_PLONK_DEVICE_TU) && defined(__HIPCC__)
__device__ uint32_t g_plonk_sha_dump[32][24];
__device__ uint32_t g_plonk_sha_n;
#endif
inline uint32_t g_plonk_sha_dump_host[32][24];
inline uint32_t g_plonk_sha_n_host;
#endif

#define PL_HELPER_SHA_STRIDE 68     /* bytes: a 64-byte block + one dword, so that the helper lanes' same-offset accesses fall on different banks */
#if defined(BN254_PLONK_DEVICE_TU) && defined(__HIP_DEVICE_COMPILE__)
// Device side: a lane's pending SHA-256 block lives in LDS, not in its private memory (a byte buffer indexed by a run-time fill level would be
// scratch: a vector-memory round trip per byte, with one wavefront per SIMD to hide it).  The kernels of bn254_k_plonk.hip lay their dynamic LDS out as
//   [0] u32 lane stride in bytes | 16 + lane * stride: [0, 64) this lane's SHA block, [64, ...) its proof bytes, then its public inputs | 64 SHA blocks of the helper lanes
// and a lane hashes with ONE Sha256 object at a time (the transcripts are sequential), so the slot needs no owner.
// The slot is addressed by its LDS offset, not through an `extern __shared__` declaration: the functions below may be compiled out of line, and
// dynamic LDS is only nameable from the kernel itself.  The two kernels declare no static LDS, so their dynamic LDS starts at offset 0.
__device__ __forceinline__ uint8_t* pl_lane_lds() {
  typedef __attribute__((address_space(3))) uint8_t lds_u8;
  typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
  const uint32_t stride = *(lds_cu32*)(uintptr_t)0;
  // lanes 64..127 of a workgroup (the helper wavefront of k_plonk_stage1) hash in blocks of their own behind the 64 slots
  lds_u8* p = threadIdx.x < 64u ? (lds_u8*)(uintptr_t)(16u + threadIdx.x * stride) : (lds_u8*)(uintptr_t)(16u + 64u * stride + (threadIdx.x - 64u) * (uint32_t)PL_HELPER_SHA_STRIDE);
  return (uint8_t*)p;
}
#define PL_SHA_BUF() pl_lane_lds()
#else
#define PL_SHA_BUF() buf
#endif

// ---------------------------------------------------------------- SHA-256 (FIPS 180-4), transcript.rs / hash_to_field.rs use sha2
struct Sha256 {
  uint32_t h[8]; uint8_t buf[64]; uint64_t len; size_t fill;
  // continue from a saved state (the key-side prefix of a transcript): on the device the pending bytes move into this lane's LDS slot
  PL_HD void adopt(const Sha256& src) {
#if defined(BN254_PLONK_DEVICE_TU) && defined(__HIP_DEVICE_COMPILE__)
    uint8_t* b = PL_SHA_BUF();
    for (size_t q = 0; q < src.fill; q++) b[q] = src.buf[q];
#else
    (void)src;
#endif
  }
  PL_HD Sha256() { reset(); }
  PL_HD void reset() {
    const uint32_t iv[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    memcpy(h, iv, sizeof h); len = 0; fill = 0;
  }
  PL_HD static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
#if defined(__x86_64__)
  // the same compression function on the SHA extensions (every x86 host an MI355X box has carries them; checked at run time): the five
  // transcripts of a proof are ~30 blocks, a quarter of the host time of stage 1 with the portable code below
  static bool have_shani() { static const bool v = __builtin_cpu_supports("sha") && __builtin_cpu_supports("sse4.1") && __builtin_cpu_supports("ssse3"); return v; }
  __attribute__((target("sha,sse4.1,ssse3"))) void block_shani(const uint8_t* p) {
    static const uint32_t K[64] = {
      0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u,
      0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
      0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u,
      0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
      0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u,
      0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    const __m128i shuf = _mm_set_epi64x(0x0c0d0e0f08090a0bll, 0x0405060700010203ll);
    __m128i tmp = _mm_loadu_si128((const __m128i*)&h[0]);        // d c b a
    __m128i st1 = _mm_loadu_si128((const __m128i*)&h[4]);        // h g f e
    tmp = _mm_shuffle_epi32(tmp, 0xB1);                           // c d a b
    st1 = _mm_shuffle_epi32(st1, 0x1B);                           // e f g h
    __m128i st0 = _mm_alignr_epi8(tmp, st1, 8);                   // a b e f
    st1 = _mm_blend_epi16(st1, tmp, 0xF0);                        // c d g h
    const __m128i save0 = st0, save1 = st1;
    __m128i m[4];
    for (int i = 0; i < 4; i++) m[i] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 16 * i)), shuf);
    for (int r = 0; r < 16; r++) {
      __m128i w;
      if (r < 4) w = m[r];
      else {
        // w[r] from the four previous message quadruples (FIPS 180-4 schedule through sha256msg1 / sha256msg2)
        __m128i t = _mm_sha256msg1_epu32(m[(r - 4) & 3], m[(r - 3) & 3]);
        t = _mm_add_epi32(t, _mm_alignr_epi8(m[(r - 1) & 3], m[(r - 2) & 3], 4));
        w = _mm_sha256msg2_epu32(t, m[(r - 1) & 3]);
        m[r & 3] = w;
      }
      __m128i wk = _mm_add_epi32(w, _mm_loadu_si128((const __m128i*)&K[4 * r]));
      st1 = _mm_sha256rnds2_epu32(st1, st0, wk);
      wk = _mm_shuffle_epi32(wk, 0x0E);
      st0 = _mm_sha256rnds2_epu32(st0, st1, wk);
    }
    st0 = _mm_add_epi32(st0, save0); st1 = _mm_add_epi32(st1, save1);
    tmp = _mm_shuffle_epi32(st0, 0x1B);                           // f e b a
    st1 = _mm_shuffle_epi32(st1, 0xB1);                           // d c h g
    st0 = _mm_blend_epi16(tmp, st1, 0xF0);                        // d c b a
    st1 = _mm_alignr_epi8(st1, tmp, 8);                           // h g f e
    _mm_storeu_si128((__m128i*)&h[0], st0);
    _mm_storeu_si128((__m128i*)&h[4], st1);
  }
#endif
  PL_HD void block(const uint8_t* p) {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    if (have_shani()) { block_shani(p); return; }
#endif
    static const uint32_t K[64] = {
      0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u,
      0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
      0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u,
      0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
      0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u,
      0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    // a rolling 16-word schedule, fully unrolled: every index is static, so the words stay in registers on the device
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
      uint32_t wi;
      if (i < 16) wi = w[i];
      else {
        const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
        const uint32_t s0 = rotr(w15, 7) ^ rotr(w15, 18) ^ (w15 >> 3), s1 = rotr(w2, 17) ^ rotr(w2, 19) ^ (w2 >> 10);
        wi = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        w[i & 15] = wi;
      }
      uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + wi;
      uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
#if defined(BN254_PLONK_MARKS) && defined(BN254_PLONK_SHA_DUMP)
    {
#if defined(BN254_PLONK_DEVICE_TU) && defined(__HIP_DEVICE_COMPILE__)
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint32_t j = g_plonk_sha_n;
        if (j < 32) { for (int i = 0; i < 16; i++) g_plonk_sha_dump[j][i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3]; for (int i = 0; i < 8; i++) g_plonk_sha_dump[j][16 + i] = h[i]; }
        g_plonk_sha_n = j + 1;
      }
#else
      const uint32_t j = g_plonk_sha_n_host;
      if (j < 32) { for (int i = 0; i < 16; i++) g_plonk_sha_dump_host[j][i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3]; for (int i = 0; i < 8; i++) g_plonk_sha_dump_host[j][16 + i] = h[i]; }
      g_plonk_sha_n_host = j + 1;
#endif
    }
#endif
  }
  PL_HD void update(const void* data, size_t n) {
    const uint8_t* p = (const uint8_t*)data; len += n;
    while (n) {
      size_t k = 64 - fill < n ? 64 - fill : n;
      uint8_t* bb = PL_SHA_BUF();
      for (size_t q = 0; q < k; q++) bb[fill + q] = p[q];
      fill += k; p += k; n -= k;
      if (fill == 64) { block(bb); fill = 0; }
    }
  }
  PL_HD void finish(uint8_t out[32]) {
    uint64_t bits = len * 8; uint8_t pad = 0x80; update(&pad, 1); uint8_t z = 0;
    while (fill != 56) update(&z, 1);
    uint8_t lb[8]; for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
    update(lb, 8);
    for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i]; }
  }
};

// ---------------------------------------------------------------- Fr in Montgomery form (4 x 64), host only
struct FrM { uint64_t l[4]; };
struct FrCtx {
  uint64_t m[4], inv; FrM one, r2, shift256;
  explicit FrCtx(const uint32_t* words = BN_R_WORDS) {   // the same arithmetic serves the base field (fp64_ctx below)
    for (int i = 0; i < 4; i++) m[i] = (uint64_t)words[2 * i] | ((uint64_t)words[2 * i + 1] << 32);
    uint64_t x = 1; for (int i = 0; i < 6; i++) x *= 2 - m[0] * x;   // m[0]^-1 mod 2^64 (Newton)
    inv = ~x + 1;                                                    // -m^-1
    FrM t = {{1, 0, 0, 0}};
    for (int i = 0; i < 256; i++) t = dbl_mod(t);
    one = t;                                                         // 2^256 mod r
    for (int i = 0; i < 256; i++) t = dbl_mod(t);
    r2 = t;                                                          // 2^512 mod r
    shift256 = mul(one, r2);                                         // Montgomery form of 2^256: (2^256 mod r) * R
  }
  PL_HD bool geq_m(const FrM& a) const { for (int i = 3; i >= 0; i--) { if (a.l[i] > m[i]) return true; if (a.l[i] < m[i]) return false; } return true; }
  PL_HD FrM sub_m(const FrM& a) const { FrM r; uint64_t br = 0; for (int i = 0; i < 4; i++) { unsigned __int128 d = (unsigned __int128)a.l[i] - m[i] - br; r.l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; } return r; }
  PL_HD FrM dbl_mod(const FrM& a) const {
    FrM r; uint64_t c = 0;
    for (int i = 0; i < 4; i++) { uint64_t v = a.l[i]; r.l[i] = (v << 1) | c; c = v >> 63; }
    if (c || geq_m(r)) r = sub_m(r);
    return r;
  }
  PL_HD FrM add(const FrM& a, const FrM& b) const {
    FrM r; unsigned __int128 c = 0;
    for (int i = 0; i < 4; i++) { c += (unsigned __int128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    if (c || geq_m(r)) r = sub_m(r);
    return r;
  }
  PL_HD FrM neg(const FrM& a) const {
    if (!(a.l[0] | a.l[1] | a.l[2] | a.l[3])) return a;
    FrM r; uint64_t br = 0; for (int i = 0; i < 4; i++) { unsigned __int128 d = (unsigned __int128)m[i] - a.l[i] - br; r.l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; } return r;
  }
  PL_HD FrM sub(const FrM& a, const FrM& b) const { return add(a, neg(b)); }
  // CIOS Montgomery product.  Two forms of the same function: 4 x 64-bit limbs through unsigned __int128 (what a CPU wants), and 8 x 32-bit words (what
  // gfx950 has: its widest multiply is v_mad_u64_u32, 32 x 32 + 64 -> 64; the 64-bit form compiles to ~890 instructions there, this one to about a third).
  // The device stages use the 32-bit form, the host the 64-bit one; BN254_FR_MUL_FORM=32 / 64 forces either everywhere (tests/cpp, tools/fr_mul_probe.hip:
  // bit-exact against each other on both sides, any a < 2^256 and b < m).
  PL_HD FrM mul(const FrM& a, const FrM& b) const {
#if (defined(__HIP_DEVICE_COMPILE__) && !(defined(BN254_FR_MUL_FORM) && BN254_FR_MUL_FORM == 64)) || (defined(BN254_FR_MUL_FORM) && BN254_FR_MUL_FORM == 32)
    return mul_w32(a, b);
#else
    return mul_w64(a, b);
#endif
  }
  // On the device the product is ONE out-of-line function (BN254_FR_MUL_INLINE=0 leaves the choice to the compiler, 1 forces inlining: diagnostics): the stages
  // call it ~250 times, inlined copies make k_plonk_stage1 0.6 MB of code, and the one build in which the compiler inlined it next to the byte shuffles of a
  // freshly written digest computed a wrong challenge (from_be32 below carries the second half of the workaround).
#if defined(BN254_FR_MUL_INLINE) && BN254_FR_MUL_INLINE == 1
  __attribute__((always_inline))
#elif defined(__HIP_DEVICE_COMPILE__) && !(defined(BN254_FR_MUL_INLINE) && BN254_FR_MUL_INLINE == 0)
  __attribute__((noinline))
#endif
  PL_HD FrM mul_w32(const FrM& a, const FrM& b) const {
    uint32_t aw[8], bw[8], mw[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      aw[2 * i] = (uint32_t)a.l[i]; aw[2 * i + 1] = (uint32_t)(a.l[i] >> 32); bw[2 * i] = (uint32_t)b.l[i]; bw[2 * i + 1] = (uint32_t)(b.l[i] >> 32);
      mw[2 * i] = (uint32_t)m[i]; mw[2 * i + 1] = (uint32_t)(m[i] >> 32);
    }
    const uint32_t ninv = (uint32_t)inv;                     // -1 / m mod 2^32
    uint32_t t[10];
#pragma unroll
    for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint64_t c = 0;
#pragma unroll
      for (int j = 0; j < 8; j++) { const uint64_t s = (uint64_t)aw[j] * bw[i] + t[j] + c; t[j] = (uint32_t)s; c = s >> 32; }   // <= (2^32 - 1)^2 + 2 (2^32 - 1) = 2^64 - 1
      const uint64_t s8 = (uint64_t)t[8] + c; t[8] = (uint32_t)s8; t[9] = (uint32_t)(s8 >> 32);
      const uint32_t q = t[0] * ninv;
      c = ((uint64_t)q * mw[0] + t[0]) >> 32;
#pragma unroll
      for (int j = 1; j < 8; j++) { const uint64_t s = (uint64_t)q * mw[j] + t[j] + c; t[j - 1] = (uint32_t)s; c = s >> 32; }
      const uint64_t s9 = (uint64_t)t[8] + c; t[7] = (uint32_t)s9; t[8] = t[9] + (uint32_t)(s9 >> 32);
    }
    FrM r;
#pragma unroll
    for (int i = 0; i < 4; i++) r.l[i] = (uint64_t)t[2 * i] | ((uint64_t)t[2 * i + 1] << 32);
    if (t[8] || geq_m(r)) r = sub_m(r);
    return r;
  }
  PL_HD FrM mul_w64(const FrM& a, const FrM& b) const {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
      unsigned __int128 c = 0;
      for (int j = 0; j < 4; j++) { c += (unsigned __int128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
      c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
      uint64_t q = t[0] * inv;
      c = (unsigned __int128)q * m[0] + t[0]; c >>= 64;
      for (int j = 1; j < 4; j++) { c += (unsigned __int128)q * m[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
      c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    FrM r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || geq_m(r)) r = sub_m(r);
    return r;
  }
  PL_HD FrM from_u64(uint64_t v) const { FrM t = {{v, 0, 0, 0}}; return mul(t, r2); }
  PL_HD FrM from_canon(const FrM& a) const { return mul(a, r2); }          // a < r
  PL_HD FrM to_canon(const FrM& a) const { FrM o = {{1, 0, 0, 0}}; return mul(a, o); }
  PL_HD bool is_zero(const FrM& a) const { return !(a.l[0] | a.l[1] | a.l[2] | a.l[3]); }
  PL_HD bool eq(const FrM& a, const FrM& b) const { return a.l[0] == b.l[0] && a.l[1] == b.l[1] && a.l[2] == b.l[2] && a.l[3] == b.l[3]; }
  PL_HD FrM pow_u64(const FrM& a, uint64_t e) const {
    FrM r = one, b = a;
    while (e) { if (e & 1) r = mul(r, b); b = mul(b, b); e >>= 1; }
    return r;
  }
  PL_HD FrM inverse_fermat(const FrM& a) const {  // a^(r-2); 0 -> 0
    uint64_t e[4] = {m[0] - 2, m[1], m[2], m[3]};
    FrM r = one;
    for (int i = 255; i >= 0; i--) { r = mul(r, r); if ((e[i / 64] >> (i % 64)) & 1) r = mul(r, a); }
    return r;
  }
  // Inverse in CONSTANT TIME: the binary extended GCD with approximated operands of bn254_fp.h::fp_inv (Pornin 2020, k = 30: 18 rounds of 29 inner
  // steps on 60-bit stand-ins, then one multiply-accumulate pass over 9 digits of 29 bits for (a, b) and one for (u, v) with a Montgomery digit retired),
  // here for the modulus of this context.  Invariants a = u y / C, b = v y / C (mod m) from (a, u, b, v) = (y, C, m, 0); with y the stored (Montgomery)
  // value and C = 2^512 mod m the result v = C / y is the Montgomery form of the inverse.  0 -> 0.  Fixed trip counts and no data-dependent branch: on the
  // device the 64 proofs of a wavefront stay in step (the classic shift-and-subtract form below, whose loops run a data-dependent number of times, took
  // 416 us of the 844 us of k_plonk_stage1 -- every lane waiting for the slowest of its wavefront at each of ~1000 diverging loop heads).
  PL_HD FrM inverse(const FrM& x) const {
    const int LB = 29; const uint32_t MASK = (1u << 29) - 1;
    auto digits = [&](int32_t d[9], const uint64_t l[4]) {   // 256-bit value -> 9 unsigned digits of 29 bits
#pragma unroll
      for (int i = 0; i < 9; i++) {
        const int bit = LB * i, w = bit >> 6, sh = bit & 63;
        uint64_t v = l[w] >> sh;
        if (sh > 64 - LB && w + 1 < 4) v |= l[w + 1] << (64 - sh);
        d[i] = (int32_t)((uint32_t)v & MASK);
      }
    };
    int32_t a[9], b[9], u[9], v[9], md[9];
    digits(a, x.l); digits(md, m); digits(u, r2.l);
#pragma unroll
    for (int i = 0; i < 9; i++) { b[i] = md[i]; v[i] = 0; }
    const uint32_t ninv = (uint32_t)inv & MASK;                       // -1 / m mod 2^29
    auto sext29 = [](uint32_t t) { return (int32_t)(t << 3) >> 3; };
#pragma unroll 1
    for (int round = 0; round < 18; round++) {
      int32_t ah = 0, am = 0, al = 0, bh = 0, bm = 0, bl = 0, low = 0;
      bool found = false;
#pragma unroll
      for (int i = 8; i >= 2; i--) {
        const bool take = !found && ((a[i] | b[i]) != 0);
        ah = take ? a[i] : ah; am = take ? a[i - 1] : am; al = take ? a[i - 2] : al;
        bh = take ? b[i] : bh; bm = take ? b[i - 1] : bm; bl = take ? b[i - 2] : bl;
        low = take ? (i == 2 ? 1 : 0) : low;
        found = found || take;
      }
      const uint64_t hiA = ((uint64_t)(uint32_t)ah << LB) | (uint32_t)am, hiB = ((uint64_t)(uint32_t)bh << LB) | (uint32_t)bm;
      const int len = 64 - bn_clz64(hiA | hiB | 1);
      const int sh = len - 31;
      const uint64_t topA = sh >= 0 ? (hiA >> (sh & 63)) : ((hiA << 1) | ((uint32_t)al >> 28));
      const uint64_t topB = sh >= 0 ? (hiB >> (sh & 63)) : ((hiB << 1) | ((uint32_t)bl >> 28));
      const uint64_t lowA = ((uint64_t)(uint32_t)a[1] << LB) | (uint32_t)a[0], lowB = ((uint64_t)(uint32_t)b[1] << LB) | (uint32_t)b[0];
      const bool exact3 = found && low != 0 && len <= 33;
      uint64_t A = !found ? lowA : exact3 ? ((hiA << LB) | (uint32_t)al) : ((topA << LB) | (uint32_t)a[0]);
      uint64_t B = !found ? lowB : exact3 ? ((hiB << LB) | (uint32_t)bl) : ((topB << LB) | (uint32_t)b[0]);
      int32_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
#pragma unroll 1
      for (int j = 0; j < LB; j++) {
        const bool odd = (A & 1) != 0;
        const bool swap = odd && A < B;
        const uint64_t tA = swap ? B : A, tB = swap ? A : B;
        const int32_t tf0 = swap ? f1 : f0, tg0 = swap ? g1 : g0, tf1 = swap ? f0 : f1, tg1 = swap ? g0 : g1;
        A = (tA - (odd ? tB : 0)) >> 1; B = tB;
        f0 = tf0 - (odd ? tf1 : 0); g0 = tg0 - (odd ? tg1 : 0);
        f1 = (int32_t)((uint32_t)tf1 << 1); g1 = (int32_t)((uint32_t)tg1 << 1);   // (as unsigned: shifting a negative value left is undefined before C++20)
      }
      int32_t na[9], nb[9];
      {
        int64_t ca = (int64_t)a[0] * f0 + (int64_t)b[0] * g0, cb = (int64_t)a[0] * f1 + (int64_t)b[0] * g1;
        ca >>= LB; cb >>= LB;
#pragma unroll
        for (int i = 1; i < 9; i++) {
          ca += (int64_t)a[i] * f0 + (int64_t)b[i] * g0; cb += (int64_t)a[i] * f1 + (int64_t)b[i] * g1;
          na[i - 1] = (int32_t)((uint32_t)ca & MASK); nb[i - 1] = (int32_t)((uint32_t)cb & MASK);
          ca >>= LB; cb >>= LB;
        }
        na[8] = (int32_t)ca; nb[8] = (int32_t)cb;
      }
      const bool nega = na[8] < 0, negb = nb[8] < 0;
      {
        int32_t ba = 0, bb = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
          const int32_t ta = -na[i] - ba, tb = -nb[i] - bb;
          const int32_t da = i < 8 ? (ta & (int32_t)MASK) : ta, db = i < 8 ? (tb & (int32_t)MASK) : tb;
          ba = i < 8 ? ((ta >> LB) & 1) : 0; bb = i < 8 ? ((tb >> LB) & 1) : 0;
          a[i] = nega ? da : na[i]; b[i] = negb ? db : nb[i];
        }
      }
      f0 = nega ? -f0 : f0; g0 = nega ? -g0 : g0; f1 = negb ? -f1 : f1; g1 = negb ? -g1 : g1;
      {
        int64_t cu = (int64_t)u[0] * f0 + (int64_t)v[0] * g0, cv = (int64_t)u[0] * f1 + (int64_t)v[0] * g1;
        const int32_t qu = sext29((uint32_t)cu * ninv), qv = sext29((uint32_t)cv * ninv);
        cu += (int64_t)qu * md[0]; cv += (int64_t)qv * md[0];
        cu >>= LB; cv >>= LB;
        int32_t nu[9], nv[9];
#pragma unroll
        for (int i = 1; i < 9; i++) {
          cu += (int64_t)u[i] * f0 + (int64_t)v[i] * g0 + (int64_t)qu * md[i];
          cv += (int64_t)u[i] * f1 + (int64_t)v[i] * g1 + (int64_t)qv * md[i];
          nu[i - 1] = (int32_t)((uint32_t)cu & MASK); nv[i - 1] = (int32_t)((uint32_t)cv & MASK);
          cu >>= LB; cv >>= LB;
        }
        nu[8] = (int32_t)cu; nv[8] = (int32_t)cv;
#pragma unroll
        for (int i = 0; i < 9; i++) { u[i] = nu[i]; v[i] = nv[i]; }
      }
    }
    // |v| < 12 m (fp_inv's bound: the cofactors grow by less than m / 2 + |f| + |g| digits' worth per round): w = v + 16 m in (4 m, 28 m), then
    // conditional subtractions of 16 m, 8 m, 4 m, 2 m, m bring it into [0, m)
    int64_t w[9];
    {
      int64_t c = 0;
#pragma unroll
      for (int i = 0; i < 9; i++) { c += (int64_t)v[i] + 16 * (int64_t)md[i]; w[i] = i < 8 ? (int64_t)((uint64_t)c & MASK) : c; c = i < 8 ? (c >> LB) : 0; }
    }
#pragma unroll
    for (int k = 16; k >= 1; k >>= 1) {
      int64_t t[9], c = 0;
#pragma unroll
      for (int i = 0; i < 9; i++) { c += w[i] - (int64_t)k * md[i]; t[i] = i < 8 ? (int64_t)((uint64_t)c & MASK) : c; c = i < 8 ? (c >> LB) : 0; }
      const bool ge = t[8] >= 0;
#pragma unroll
      for (int i = 0; i < 9; i++) w[i] = ge ? t[i] : w[i];
    }
    FrM r = {{0, 0, 0, 0}};
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int bit = LB * i, wd = bit >> 6, sh = bit & 63;
      r.l[wd] |= (uint64_t)w[i] << sh;
      if (sh > 64 - LB && wd + 1 < 4) r.l[wd + 1] |= (uint64_t)w[i] >> (64 - sh);
    }
    return r;
  }
  // The classic binary extended Euclidean algorithm on the canonical value (HAC 14.61 for an odd modulus): kept as the cross-check of inverse()
  // (tests/test_capi_cpu.py::test_fr_inverse_binary_gcd).  Its loops run a data-dependent number of times: never on the device.
  PL_HD FrM inverse_bgcd(const FrM& a) const {
    FrM u = to_canon(a), v = {{m[0], m[1], m[2], m[3]}}, x1 = {{1, 0, 0, 0}}, x2 = {{0, 0, 0, 0}};
    if (is_zero(u)) return u;
    auto is_one = [](const FrM& t) { return t.l[0] == 1 && !(t.l[1] | t.l[2] | t.l[3]); };
    auto shr1 = [](FrM& t, uint64_t top) { for (int i = 0; i < 3; i++) t.l[i] = (t.l[i] >> 1) | (t.l[i + 1] << 63); t.l[3] = (t.l[3] >> 1) | (top << 63); };
    auto half_mod = [&](FrM& x) {            // x / 2 mod m
      if (x.l[0] & 1) { unsigned __int128 c = 0; for (int i = 0; i < 4; i++) { c += (unsigned __int128)x.l[i] + m[i]; x.l[i] = (uint64_t)c; c >>= 64; } shr1(x, (uint64_t)c); }
      else shr1(x, 0);
    };
    auto ge = [](const FrM& p, const FrM& q) { for (int i = 3; i >= 0; i--) { if (p.l[i] > q.l[i]) return true; if (p.l[i] < q.l[i]) return false; } return true; };
    auto sub_into = [](FrM& p, const FrM& q) { uint64_t br = 0; for (int i = 0; i < 4; i++) { unsigned __int128 d = (unsigned __int128)p.l[i] - q.l[i] - br; p.l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; } };
    for (int guard = 0; guard < 1024 && !is_one(u) && !is_one(v); guard++) {
      while (!(u.l[0] & 1)) { shr1(u, 0); half_mod(x1); }
      while (!(v.l[0] & 1)) { shr1(v, 0); half_mod(x2); }
      if (ge(u, v)) { sub_into(u, v); x1 = sub_canon(x1, x2); } else { sub_into(v, u); x2 = sub_canon(x2, x1); }
    }
    return from_canon(is_one(u) ? x1 : x2);
  }
  PL_HD FrM sub_canon(const FrM& a, const FrM& b) const {   // a - b mod m on canonical values
    FrM r; uint64_t br = 0;
    for (int i = 0; i < 4; i++) { unsigned __int128 d = (unsigned __int128)a.l[i] - b.l[i] - br; r.l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
    if (br) { unsigned __int128 c = 0; for (int i = 0; i < 4; i++) { c += (unsigned __int128)r.l[i] + m[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
    return r;
  }
  // big-endian bytes, reduced mod r (Fr::from_slice stores the raw value and every later operation is mod r; the challenges and
  // hash_to_field reduce explicitly).  One Montgomery product per 32-byte block: mul(raw, r2) is exact for any raw < 2^256.
  PL_HD FrM from_be32(const uint8_t* b) const {
    FrM raw;
    for (int i = 0; i < 4; i++) { uint64_t v = 0; for (int j = 0; j < 8; j++) v = v << 8 | b[(3 - i) * 8 + j]; raw.l[i] = v; }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BN254_FR_NO_BARRIER)
    // the limbs as plain register values before the product: keeps the byte shuffles of a digest that was just written out (Sha256::finish) from being fused
    // with the multiplication's word splitting ("a wrong challenge on the device")
    for (int i = 0; i < 4; i++) asm volatile("" : "+v"(raw.l[i]));
#endif
    return mul(raw, r2);
  }
  PL_HD FrM from_be_reduce(const uint8_t* b, size_t n) const {
    if (n == 32) return from_be32(b);
    uint8_t pad[32];
    FrM acc = {{0, 0, 0, 0}};
    // Horner over 32-byte blocks from the most significant end; the first block may be short
    size_t first = n % 32 ? n % 32 : 32;
    memset(pad, 0, 32); memcpy(pad + 32 - first, b, first);
    acc = from_be32(pad);
    for (size_t off = first; off < n; off += 32) acc = add(mul(acc, shift256), from_be32(b + off));
    return acc;
  }
  PL_HD void to_be(uint8_t out[32], const FrM& a) const {
    FrM c = to_canon(a);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) out[(3 - i) * 8 + j] = (uint8_t)(c.l[i] >> (56 - 8 * j));
  }
  PL_HD void to_words(uint32_t w[8], const FrM& a) const { FrM c = to_canon(a); for (int i = 0; i < 4; i++) { w[2 * i] = (uint32_t)c.l[i]; w[2 * i + 1] = (uint32_t)(c.l[i] >> 32); } }
};
#if defined(BN254_PLONK_DEVICE_TU) && defined(__HIPCC__)
// device copies of the field constants (raw storage: the constructors run on the host; bn254_k_plonk.hip uploads them once per device)
__device__ uint64_t g_plonk_fr_raw[(sizeof(FrCtx) + 7) / 8];
#endif
PL_HD const FrCtx& fr_ctx() {
#if defined(BN254_PLONK_DEVICE_TU) && defined(__HIP_DEVICE_COMPILE__)
  return *reinterpret_cast<const FrCtx*>(g_plonk_fr_raw);
#else
  static const FrCtx c; return c;
#endif
}
// Fp on 4 x 64-bit limbs for the host-side checks of proof points: a CIOS product costs a third of the 9 x 29-bit digit product the kernels'
// representation needs on a CPU.  k261: 2^(261 + 256) mod p, so that mul(x, k261) = x 2^261 mod p -- the digit form's Montgomery factor.
struct Fp64Ctx {
  FrCtx F; FrM three, k261;
  Fp64Ctx() : F(BN_P_WORDS) {
    three = F.from_u64(3);
    FrM t = F.one;                                         // 2^256 mod p
    for (int i = 0; i < 261; i++) t = F.dbl_mod(t);        // 2^517 mod p
    k261 = t;
  }
};
#if defined(BN254_PLONK_DEVICE_TU) && defined(__HIPCC__)
__device__ uint64_t g_plonk_fp64_raw[(sizeof(Fp64Ctx) + 7) / 8];
#endif
PL_HD const Fp64Ctx& fp64_ctx() {
#if defined(BN254_PLONK_DEVICE_TU) && defined(__HIP_DEVICE_COMPILE__)
  return *reinterpret_cast<const Fp64Ctx*>(g_plonk_fp64_raw);
#else
  static const Fp64Ctx c; return c;
#endif
}

// Diagnostics build only (BN254_PLONK_MARKS): intermediate values of the stages, of the first lane on the device and of the last proof on the host, so that
// tools/plonk_stage_dump.py can say at which value the two sides part.
#if defined(BN254_PLONK_MARKS)
#if defined(BN254_PLONK_DEVICE_TU) && defined(__HIPCC__)
__device__ FrM g_plonk_dump[64];
#endif
inline FrM g_plonk_dump_host[64];
#if defined(BN254_PLONK_DEVICE_TU) && defined(__HIP_DEVICE_COMPILE__)
#define PL_DUMP(k, v) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_plonk_dump[k] = (v); } while (0)
#else
#define PL_DUMP(k, v) do { g_plonk_dump_host[k] = (v); } while (0)
#endif
#else
#define PL_DUMP(k, v) ((void)0)
#endif

// ---------------------------------------------------------------- key and proof (plonk/converter.rs:18-178, proof.rs)
enum { PLONK_MAX_QCP = 8, PLONK_MAX_CLAIMED = 16 };
struct PlonkKey {
  uint64_t size, nb_public; FrM size_inv, generator, coset_shift;
  G1Aff s[3], ql, qr, qm, qo, qk, qcp[PLONK_MAX_QCP]; uint32_t n_qcp;
  G1Aff kzg_g1; G2Aff kzg_g2[2];
  uint64_t cci[PLONK_MAX_QCP]; uint64_t n_cci;
  FrM wpow[PLONK_MAX_QCP];             // generator^(nb_public + cci[i]): the evaluation point of the i-th BSB22 Lagrange term
  uint8_t enc[8 + PLONK_MAX_QCP][64];  // uncompressed encodings of s1..3, ql, qr, qm, qo, qk, qcp[]: what the transcript binds
  Sha256 gamma_mid;                    // SHA-256 state after "gamma" and those encodings: the key-side prefix of every proof's first challenge
};
PL_HD uint32_t pl_be32(const uint8_t* b) { return (uint32_t)b[0] << 24 | (uint32_t)b[1] << 16 | (uint32_t)b[2] << 8 | b[3]; }
PL_HD uint64_t be64(const uint8_t* b) { uint64_t v = 0; for (int i = 0; i < 8; i++) v = v << 8 | b[i]; return v; }
// plonk/converter.rs:18-119.  G1 points: unchecked decompression (converter.rs:62-76); G2: converter.rs:113-133 in the reference's
// reading of the root order; 33 788 bytes of precomputed lines are skipped (converter.rs:58).
inline int parse_plonk_vk(PlonkKey& vk, const uint8_t* b, size_t n) {
  const FrCtx& F = fr_ctx();
  if (n < 372) return DEC_MALFORMED;
  vk.size = be64(b);
  vk.size_inv = F.from_be_reduce(b + 8, 32);
  vk.generator = F.from_be_reduce(b + 40, 32);
  vk.nb_public = be64(b + 72);
  vk.coset_shift = F.from_be_reduce(b + 80, 32);
  G1Aff* pts[8] = {&vk.s[0], &vk.s[1], &vk.s[2], &vk.ql, &vk.qr, &vk.qm, &vk.qo, &vk.qk};
  for (int i = 0; i < 8; i++) { if (dec_g1_compressed(*pts[i], b + 112 + 32 * i) != DEC_OK) return DEC_MALFORMED; enc_g1_uncompressed(vk.enc[i], *pts[i]); }
  vk.n_qcp = be32(b + 368);
  if (vk.n_qcp > PLONK_MAX_QCP) return DEC_MALFORMED;
  size_t off = 372;
  if (n < off + 32 * (size_t)vk.n_qcp + 160 + 33788 + 8) return DEC_MALFORMED;
  for (uint32_t i = 0; i < vk.n_qcp; i++, off += 32) { if (dec_g1_compressed(vk.qcp[i], b + off) != DEC_OK) return DEC_MALFORMED; enc_g1_uncompressed(vk.enc[8 + i], vk.qcp[i]); }
  if (dec_g1_compressed(vk.kzg_g1, b + off) != DEC_OK) return DEC_MALFORMED;
  if (dec_g2_compressed(vk.kzg_g2[0], b + off + 32, 0) != DEC_OK) return DEC_MALFORMED;
  if (dec_g2_compressed(vk.kzg_g2[1], b + off + 96, 0) != DEC_OK) return DEC_MALFORMED;
  off += 160 + 33788;
  vk.n_cci = be64(b + off); off += 8;
  if (vk.n_cci > PLONK_MAX_QCP || n < off + 8 * vk.n_cci) return DEC_MALFORMED;
  for (uint64_t i = 0; i < vk.n_cci; i++, off += 8) { vk.cci[i] = be64(b + off); vk.wpow[i] = F.pow_u64(vk.generator, vk.nb_public + vk.cci[i]); }
  vk.gamma_mid.reset();
  vk.gamma_mid.update("gamma", 5);
  for (uint32_t i = 0; i < 8 + vk.n_qcp; i++) vk.gamma_mid.update(vk.enc[i], 64);
  return DEC_OK;
}
struct PlonkProof {
  G1Aff lro[3], z, h[3], batch_h, zs_h, bsb[PLONK_MAX_QCP];
  const uint8_t* raw;                       // the proof bytes (the transcript binds the encodings as they came)
  size_t off_claimed, off_zs_h, off_bsb;
  FrM claimed[PLONK_MAX_CLAIMED], zs_value; uint32_t n_claimed, n_bsb;
};
// status codes as in include/bn254_verify.h
enum { PL_OK = 1, PL_NOT_MEMBER = 2, PL_NOT_ON_CURVE = 3, PL_INPUT_LEN = 5, PL_MALFORMED = 6, PL_OPENING = 7, PL_PAIRING = 8, PL_BSB22 = 9, PL_INVERSE = 10 };
// converter.rs:78-88: two field members (>= p rejected), then the curve equation
PL_HD int dec_g1_uncompressed_checked(G1Aff& o, const uint8_t* b) {
  const Fp64Ctx& C = fp64_ctx();
  const FrCtx& F = C.F;
  FrM x, y;
  for (int i = 0; i < 4; i++) { x.l[i] = be64(b + (3 - i) * 8); y.l[i] = be64(b + 32 + (3 - i) * 8); }
  if (F.geq_m(x) || F.geq_m(y)) return PL_NOT_MEMBER;
  // y^2 == x^3 + 3 on 64-bit limbs (the all-zero encoding is not on the curve, as in bn254_curve.h::g1_on_curve)
  const FrM xm = F.from_canon(x), ym = F.from_canon(y);
  const bool on = F.eq(F.mul(ym, ym), F.add(F.mul(F.mul(xm, xm), xm), C.three));
  // the kernels' form: x 2^261 mod p as balanced 29-bit digits
  uint32_t wx[8], wy[8];
  const FrM dx = F.mul(x, C.k261), dy = F.mul(y, C.k261);
  for (int i = 0; i < 4; i++) { wx[2 * i] = (uint32_t)dx.l[i]; wx[2 * i + 1] = (uint32_t)(dx.l[i] >> 32); wy[2 * i] = (uint32_t)dy.l[i]; wy[2 * i + 1] = (uint32_t)(dy.l[i] >> 32); }
  o.x = fp_from_words_raw(wx); o.y = fp_from_words_raw(wy);
  return on ? PL_OK : PL_NOT_ON_CURVE;
}
// plonk/converter.rs:121-178
PL_HD int parse_plonk_proof(PlonkProof& p, const uint8_t* b, size_t n) {
  const FrCtx& F = fr_ctx();
  int st;
  if (n < 516) return PL_MALFORMED;
  p.raw = b;
  G1Aff* pts[8] = {&p.lro[0], &p.lro[1], &p.lro[2], &p.z, &p.h[0], &p.h[1], &p.h[2], &p.batch_h};
  for (int i = 0; i < 8; i++) if ((st = dec_g1_uncompressed_checked(*pts[i], b + 64 * i)) != PL_OK) return st;
  p.n_claimed = pl_be32(b + 512);
  if (p.n_claimed > PLONK_MAX_CLAIMED) return PL_MALFORMED;
  size_t off = 516;
  if (n < off + 32 * (size_t)p.n_claimed + 100) return PL_MALFORMED;
  p.off_claimed = off;
  for (uint32_t i = 0; i < p.n_claimed; i++, off += 32) p.claimed[i] = F.from_be_reduce(b + off, 32);
  p.off_zs_h = off;
  if ((st = dec_g1_uncompressed_checked(p.zs_h, b + off)) != PL_OK) return st;
  p.zs_value = F.from_be_reduce(b + off + 64, 32);
  p.n_bsb = pl_be32(b + off + 96);
  if (p.n_bsb > PLONK_MAX_QCP) return PL_MALFORMED;
  off += 100;
  if (n < off + 64 * (size_t)p.n_bsb) return PL_MALFORMED;
  p.off_bsb = off;
  for (uint32_t i = 0; i < p.n_bsb; i++, off += 64) if ((st = dec_g1_uncompressed_checked(p.bsb[i], b + off)) != PL_OK) return st;
  return PL_OK;
}

// The length fields of a proof alone -- where its claimed values, its second opening and its commitments sit -- with parse_plonk_proof's bounds checks and none of its
// point work: what the transcripts need.  false: the proof is malformed (parse_plonk_proof then fails too, with that or an earlier point error).
struct PlonkLayout { size_t off_claimed, off_zs_h, off_bsb; uint32_t n_claimed, n_bsb; };
PL_HD bool plonk_proof_layout(PlonkLayout& l, const uint8_t* b, size_t n) {
  if (n < 516) return false;
  l.n_claimed = pl_be32(b + 512);
  if (l.n_claimed > PLONK_MAX_CLAIMED) return false;
  size_t off = 516;
  if (n < off + 32 * (size_t)l.n_claimed + 100) return false;
  l.off_claimed = off;
  off += 32 * (size_t)l.n_claimed;
  l.off_zs_h = off;
  l.n_bsb = pl_be32(b + off + 96);
  if (l.n_bsb > PLONK_MAX_QCP) return false;
  off += 100;
  if (n < off + 64 * (size_t)l.n_bsb) return false;
  l.off_bsb = off;
  return true;
}

// transcript.rs:15-108: challenge = SHA-256(name | digest of the previous challenge (position > 0) | bindings in order)
struct Challenge {
  Sha256 h;
  PL_HD Challenge(const char* name, size_t name_len, const uint8_t* prev) { h.update(name, name_len); if (prev) h.update(prev, 32); }
  PL_HD explicit Challenge(const Sha256& mid) : h(mid) { h.adopt(mid); }     // continue from a saved state (name and key-side bindings already absorbed)
  PL_HD void bind(const void* d, size_t n) { h.update(d, n); }
  PL_HD FrM finish(uint8_t digest[32]) { h.finish(digest); return fr_ctx().from_be_reduce(digest, 32); }
};
// hash_to_field.rs:45-97: RFC 9380 expand_message_xmd(SHA-256), 48 bytes, DST "BSB22-Plonk", reduced mod r
PL_HD FrM bsb22_hash_to_field(const uint8_t g1_uncompressed[64]) {
  const char dst[] = "BSB22-Plonk";
  const uint8_t dl = 11;
  uint8_t b0[32], b1[32], b2[32], z[64] = {0}, lib[3] = {0, 48, 0}, idx = 1, out[48], sx[32];
  Sha256 h; h.update(z, 64); h.update(g1_uncompressed, 64); h.update(lib, 3); h.update(dst, dl); h.update(&dl, 1); h.finish(b0);
  h.reset(); h.update(b0, 32); h.update(&idx, 1); h.update(dst, dl); h.update(&dl, 1); h.finish(b1);
  for (int j = 0; j < 32; j++) sx[j] = b0[j] ^ b1[j];
  idx = 2;
  h.reset(); h.update(sx, 32); h.update(&idx, 1); h.update(dst, dl); h.update(&dl, 1); h.finish(b2);
  for (int j = 0; j < 32; j++) out[j] = b1[j];
  for (int j = 0; j < 16; j++) out[32 + j] = b2[j];
  return fr_ctx().from_be_reduce(out, 48);
}

// ---------------------------------------------------------------- per-proof state carried between the two host stages
struct MsmTerm { int32_t pt[18]; uint32_t k[8]; };   // affine point (Montgomery digits) and the GLV halves |k1|, |k2| of the scalar: what k_g1_scalar_mul reads
static_assert(sizeof(MsmTerm) == 104, "term layout");
// GLV decomposition of a scalar for G1: k = s1 k1 + s2 k2 lambda (mod r), k1, k2 < 2^127, lambda the eigenvalue of phi(x, y) = (beta x, y)
// (bn254_rlc.h).  Lattice basis (a1, b1), (a2, b2) of {(x, y): x + y lambda = 0 mod r} from the extended Euclidean algorithm on (r, lambda):
//   a1 = b2 = 0x89d3256894d213e3,  b1 = -0x6f4d8248eeb859fc8211bbeb7d4f1128,  a2 = 0x6f4d8248eeb859fd0be4e1541221250b,  a1 b2 - a2 b1 = r
// c1 = floor(k g1 / 2^256), c2 = floor(k g2 / 2^256) with g1 = floor(2^256 b2 / r), g2 = floor(-2^256 b1 / r);  k1 = k - c1 a1 - c2 a2,
// k2 = -c1 b1 - c2 b2.  (tests/test_capi_cpu.py::test_glv_decomposition checks the identity and the bounds through bn254_dbg_glv_decompose.)
struct Glv { uint64_t k1[2], k2[2]; bool neg1, neg2; };
PL_HD Glv glv_decompose(const FrM& kc /* canonical, < r */) {
  typedef unsigned __int128 u128;
  const uint64_t A1 = 0x89d3256894d213e3ull;
  const uint64_t B1[2] = {0x8211bbeb7d4f1128ull, 0x6f4d8248eeb859fcull};        // |b1|
  const uint64_t A2[2] = {0x0be4e1541221250bull, 0x6f4d8248eeb859fdull};
  const uint64_t G1[2] = {0xd91d232ec7e0b3d7ull, 0x2ull};
  const uint64_t G2[3] = {0x7a7bd9d4391eb18dull, 0x4ccef014a773d2cfull, 0x2ull};
  auto mul_hi256 = [&](const uint64_t* g, int gn, uint64_t out[3]) {   // floor(k * g / 2^256), at most 3 limbs
    uint64_t prod[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
      u128 c = 0;
      for (int j = 0; j < gn; j++) { c += (u128)kc.l[i] * g[j] + prod[i + j]; prod[i + j] = (uint64_t)c; c >>= 64; }
      prod[i + gn] += (uint64_t)c;
    }
    out[0] = prod[4]; out[1] = prod[5]; out[2] = prod[6];
  };
  uint64_t c1[3], c2[3];
  mul_hi256(G1, 2, c1); mul_hi256(G2, 3, c2);     // c1 < 2^65, c2 < 2^129: the third limbs are 0 or tiny; products below keep 5 limbs
  // 5-limb two's-complement accumulators
  auto mac = [](uint64_t acc[5], const uint64_t* x, int xn, const uint64_t* y, int yn, bool subtract) {
    uint64_t p[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < xn; i++) {
      u128 c = 0;
      for (int j = 0; j < yn && i + j < 5; j++) { c += (u128)x[i] * y[j] + p[i + j]; p[i + j] = (uint64_t)c; c >>= 64; }
      if (i + yn < 5) p[i + yn] += (uint64_t)c;
    }
    if (subtract) { uint64_t br = 0; for (int i = 0; i < 5; i++) { u128 d = (u128)acc[i] - p[i] - br; acc[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; } }
    else { u128 c = 0; for (int i = 0; i < 5; i++) { c += (u128)acc[i] + p[i]; acc[i] = (uint64_t)c; c >>= 64; } }
  };
  uint64_t k1[5] = {kc.l[0], kc.l[1], kc.l[2], kc.l[3], 0}, k2[5] = {0, 0, 0, 0, 0};
  const uint64_t a1[1] = {A1};
  mac(k1, c1, 3, a1, 1, true); mac(k1, c2, 3, A2, 2, true);
  mac(k2, c1, 3, B1, 2, false); mac(k2, c2, 3, a1, 1, true);        // -c1 b1 = +c1 |b1|;  -c2 b2 with b2 = a1
  auto finish = [](uint64_t v[5], uint64_t out[2], bool& neg) {
    neg = (v[4] >> 63) != 0;
    if (neg) { u128 c = 1; for (int i = 0; i < 5; i++) { c += (u128)(~v[i]); v[i] = (uint64_t)c; c >>= 64; } }
    out[0] = v[0]; out[1] = v[1];
  };
  Glv g;
  finish(k1, g.k1, g.neg1); finish(k2, g.k2, g.neg2);
  return g;
}
// flag: bit 1 / bit 2 = sign of k1 / k2 (bit 0, set by the caller, marks an identity point)
PL_HD void put_term(MsmTerm& t, const G1Aff& p, const FrM& k, uint8_t* flag) {
  { const Fp cx = fp_reduce(fp_norm(p.x)), cy = fp_reduce(fp_norm(p.y)); for (int i = 0; i < BN_NL; i++) { t.pt[i] = cx.v[i]; t.pt[BN_NL + i] = cy.v[i]; } }
  const Glv g = glv_decompose(fr_ctx().to_canon(k));
  t.k[0] = (uint32_t)g.k1[0]; t.k[1] = (uint32_t)(g.k1[0] >> 32); t.k[2] = (uint32_t)g.k1[1]; t.k[3] = (uint32_t)(g.k1[1] >> 32);
  t.k[4] = (uint32_t)g.k2[0]; t.k[5] = (uint32_t)(g.k2[0] >> 32); t.k[6] = (uint32_t)g.k2[1]; t.k[7] = (uint32_t)(g.k2[1] >> 32);
  *flag = (uint8_t)((*flag & 1) | (g.neg1 ? 2 : 0) | (g.neg2 ? 4 : 0));
}
// a key-side (fixed-base) term: only its scalar travels, as 8 canonical words -- its MSM_FW_BITS-bit digits index the key's window table of that point (bn254_msm.h, bn254_fw.h)
PL_HD void put_fixed(MsmTerm& t, const FrM& k, uint8_t* flag) {
  for (int i = 0; i < 2 * BN_NL; i++) t.pt[i] = 0;
  fr_ctx().to_words(t.k, k);
  *flag = 0;
}
// a term with the scalar +-1: the point itself (flag bit 1: negated)
PL_HD void put_unit(MsmTerm& t, const G1Aff& p, bool neg, uint8_t* flag) {
  { const Fp cx = fp_reduce(fp_norm(p.x)), cy = fp_reduce(fp_norm(p.y)); for (int i = 0; i < BN_NL; i++) { t.pt[i] = cx.v[i]; t.pt[BN_NL + i] = cy.v[i]; } }
  for (int i = 0; i < 8; i++) t.k[i] = 0;
  *flag = (uint8_t)((*flag & 1) | (neg ? 2 : 0));
}
struct PlonkWork {
  int status;                 // PL_OK while the proof is still alive, else the final status
  PlonkProof pr;
  FrM zeta, lin_opening;      // lin_opening: the value stage 1 computed for the opening of the linearised polynomial (what claimed[0] must equal); the self-test builds a proof that passes from it
  FrM lambda;
  // k_plonk_stage1 runs a proof on two lanes of two wavefronts: what the helper lane hands to the chain lane (the parsed proof travels in `pr`)
  int parse_status;
  FrM h2f[PLONK_MAX_QCP];     // hash_to_field of the BSB22 commitments
};
enum { PLONK_STAGE1_TERMS_BASE = 10 };  // + n_bsb
PL_HD int plonk_stage1_terms(const PlonkKey& vk) { return PLONK_STAGE1_TERMS_BASE + (int)vk.n_qcp; }
PL_HD int plonk_stage2_terms(const PlonkKey& vk) { return 10 + (int)vk.n_qcp; }   // lin_digest, lro x3, s1, s2, qcp.., z, kzg_g1, batch_h, zs_h
// The key's points that enter the MSMs with per-proof scalars get window tables (bn254_fw.h: 20 windows of 13 bits, 13 MB each), built on each device at the key's first use:
// table numbers, and which terms of the two MSM launches are variable-base (proof points), fixed-base (key points) or a bare point (scalar -1).
enum { PLONK_TAB_QL = 0, PLONK_TAB_QR, PLONK_TAB_QM, PLONK_TAB_QO, PLONK_TAB_QK, PLONK_TAB_S3, PLONK_TAB_S1, PLONK_TAB_S2, PLONK_TAB_KZG_G1, PLONK_TAB_QCP0 };
inline int plonk_num_tables(const PlonkKey& vk) { return (int)PLONK_TAB_QCP0 + (int)vk.n_qcp; }
inline const G1Aff& plonk_table_point(const PlonkKey& vk, int t) {
  switch (t) {
    case PLONK_TAB_QL: return vk.ql; case PLONK_TAB_QR: return vk.qr; case PLONK_TAB_QM: return vk.qm; case PLONK_TAB_QO: return vk.qo; case PLONK_TAB_QK: return vk.qk;
    case PLONK_TAB_S3: return vk.s[2]; case PLONK_TAB_S1: return vk.s[0]; case PLONK_TAB_S2: return vk.s[1]; case PLONK_TAB_KZG_G1: return vk.kzg_g1;
    default: return vk.qcp[t - PLONK_TAB_QCP0];
  }
}
// stage 1 (term order of PlonkStage1::b): bsb[0..q) | ql qr qm qo qk s3 | z h0 h1 h2
inline void plonk_msm1_shape(const PlonkKey& vk, MsmShape& sh) {
  memset(&sh, 0, sizeof sh);
  const int q = (int)vk.n_qcp;
  sh.n_sums = 1;
  for (int i = 0; i < q; i++) sh.var_term[0][sh.n_var[0]++] = (int8_t)i;
  for (int i = 0; i < 6; i++) { sh.fixed_term[0][sh.n_fixed[0]] = (int8_t)(q + i); sh.fixed_tab[0][sh.n_fixed[0]++] = (int8_t)(PLONK_TAB_QL + i); }
  for (int i = 0; i < 4; i++) sh.var_term[0][sh.n_var[0]++] = (int8_t)(q + 6 + i);
}
// stage 2 (term order of plonk_stage2): P0 = lin l r o | s1 s2 qcp[0..q) | z | kzg_g1 | batch_h zs_h ;  P1 = batch_h (scalar -1) | zs_h
// joint = the pairing checks of a pass are batched across proofs (BN254_FLAG_RLC): every scalar of both sums carries the proof's random weight, so the -1 of
// batch_h is a scalar like the others and the term a variable one
inline void plonk_msm2_shape(const PlonkKey& vk, MsmShape& sh, bool joint = false) {
  memset(&sh, 0, sizeof sh);
  const int q = (int)vk.n_qcp;
  sh.n_sums = 2;
  for (int i = 0; i < 4; i++) sh.var_term[0][sh.n_var[0]++] = (int8_t)i;
  sh.fixed_term[0][sh.n_fixed[0]] = 4; sh.fixed_tab[0][sh.n_fixed[0]++] = PLONK_TAB_S1;
  sh.fixed_term[0][sh.n_fixed[0]] = 5; sh.fixed_tab[0][sh.n_fixed[0]++] = PLONK_TAB_S2;
  for (int i = 0; i < q; i++) { sh.fixed_term[0][sh.n_fixed[0]] = (int8_t)(6 + i); sh.fixed_tab[0][sh.n_fixed[0]++] = (int8_t)(PLONK_TAB_QCP0 + i); }
  sh.var_term[0][sh.n_var[0]++] = (int8_t)(6 + q);
  sh.fixed_term[0][sh.n_fixed[0]] = (int8_t)(7 + q); sh.fixed_tab[0][sh.n_fixed[0]++] = PLONK_TAB_KZG_G1;
  sh.var_term[0][sh.n_var[0]++] = (int8_t)(8 + q); sh.var_term[0][sh.n_var[0]++] = (int8_t)(9 + q);
  if (joint) sh.var_term[1][sh.n_var[1]++] = (int8_t)(10 + q);
  else sh.unit_term[1][sh.n_unit[1]++] = (int8_t)(10 + q);
  sh.var_term[1][sh.n_var[1]++] = (int8_t)(11 + q);
}

// Stage 1 (plonk/verify.rs:46-284): everything up to the scalars of the linearised polynomial digest.  On a failed check the
// proof's final status is returned and its terms are left zeroed.
// The stage has ONE field inversion (of the product of its denominators, Montgomery's trick), which is a third of its host time; it is
// therefore written in two halves around it -- a() up to the product `acc`, b(1 / acc) from there -- so that a batch can invert the
// products of many proofs with a single inversion (bn254_capi.hip::plonk_run); plonk_stage1() below runs both halves for one proof.
struct PlonkStage1 {
#if defined(__HIP_DEVICE_COMPILE__)
  enum { MAXIN = 8 };    // a lane keeps the stage's arrays in its private memory: the batched inversion covers 8 public inputs, further ones invert singly
#else
  enum { MAXIN = 64 };
#endif
  enum { MAXDEN = 2 + MAXIN + PLONK_MAX_QCP };
  const PlonkKey* vkp; const uint8_t* proof; const uint8_t* inputs; size_t n_inputs; PlonkWork* wkp;
  FrM alpha, beta, gamma, zeta, zeta_n, zh_zeta, acc;
  FrM den[MAXDEN], pre[MAXDEN]; bool zero[MAXDEN]; int nden; size_t n_in;
  // a = parse + counts + chain.  The device runs the three on two lanes: parse_plonk_proof (the curve checks of the nine points) beside the chain of transcripts,
  // which needs the proof's bytes and its layout only (bn254_k_plonk.hip::k_plonk_stage1)
  PL_HD int a(const PlonkKey& vk, const uint8_t* proof_, size_t proof_len, const uint8_t* inputs_, size_t n_inputs_, PlonkWork& wk);
  PL_HD static int counts(const PlonkKey& vk, uint32_t n_bsb, uint32_t n_claimed, size_t n_inputs_);
  PL_HD int chain(const PlonkKey& vk, const uint8_t* proof_, const uint8_t* inputs_, size_t n_inputs_, PlonkWork& wk, size_t off_bsb, uint32_t n_bsb);
  // h2f: the hash_to_field values of the commitments when another lane computed them (nullptr: computed here)
  PL_HD int b(const FrM& acc_inv, MsmTerm* terms /* plonk_stage1_terms(vk) */, uint8_t* tflags /* one per term */, const FrM* h2f = nullptr);
};
PL_HD int PlonkStage1::counts(const PlonkKey& vk, uint32_t n_bsb, uint32_t n_claimed, size_t n_inputs_) {
  if (n_bsb != vk.n_qcp) return PL_BSB22;                                 // verify.rs:52-54
  if (n_inputs_ != vk.nb_public) return PL_INPUT_LEN;                     // verify.rs:57-59 (InvalidWitness)
  if (n_claimed != 6 + vk.n_qcp || vk.n_cci != vk.n_qcp) return PL_MALFORMED;     // index panics in the reference
  return PL_OK;
}
PL_HD int PlonkStage1::a(const PlonkKey& vk, const uint8_t* proof_, size_t proof_len, const uint8_t* inputs_, size_t n_inputs_, PlonkWork& wk) {
  PlonkProof& pr = wk.pr;
  PL_MARK(1);
  int st = parse_plonk_proof(pr, proof_, proof_len);                      // lib.rs:70
  PL_MARK(2);
  if (st != PL_OK) return st;
  if ((st = counts(vk, pr.n_bsb, pr.n_claimed, n_inputs_)) != PL_OK) return st;
  return chain(vk, proof_, inputs_, n_inputs_, wk, pr.off_bsb, pr.n_bsb);
}
PL_HD int PlonkStage1::chain(const PlonkKey& vk, const uint8_t* proof_, const uint8_t* inputs_, size_t n_inputs_, PlonkWork& wk, size_t off_bsb, uint32_t n_bsb) {
  const FrCtx& F = fr_ctx();
  vkp = &vk; proof = proof_; inputs = inputs_; n_inputs = n_inputs_; wkp = &wk;
  const FrM one = F.one;
  // Fiat-Shamir (verify.rs:62-95, 319-362)
  uint8_t dg[32], db[32], da[32], dz[32];
  Challenge cg(vk.gamma_mid);                                             // "gamma" | s1..3, ql, qr, qm, qo, qk, qcp[] (parse_plonk_vk)
  cg.bind(inputs, 32 * n_inputs);                                         // the public inputs as stored (raw big-endian)
  cg.bind(proof, 192);                                                    // l, r, o
  gamma = cg.finish(dg);
  PL_MARK(3);
  Challenge cb("beta", 4, dg); beta = cb.finish(db);
  Challenge ca("alpha", 5, db);
  ca.bind(proof + off_bsb, 64 * (size_t)n_bsb);
  ca.bind(proof + 192, 64);                                               // z
  alpha = ca.finish(da);
  Challenge cz("zeta", 4, da);
  cz.bind(proof + 256, 192);                                              // h0, h1, h2
  zeta = cz.finish(dz);
  PL_MARK(4);
  PL_DUMP(0, gamma); PL_DUMP(1, beta); PL_DUMP(2, alpha); PL_DUMP(3, zeta);
#if defined(BN254_PLONK_MARKS)
  { FrM t; memcpy(&t, dg, 32); PL_DUMP(27, t); memcpy(&t, db, 32); PL_DUMP(28, t); memcpy(&t, da, 32); PL_DUMP(29, t); memcpy(&t, dz, 32); PL_DUMP(30, t); }
#endif
  wk.zeta = zeta;
  // verify.rs:97-107
  zeta_n = F.pow_u64(zeta, vk.size);
  zh_zeta = F.sub(zeta_n, one);
  FrM zm1 = F.sub(zeta, one);
  if (F.is_zero(zm1)) return PL_INVERSE;
  // every inversion of the proof in one (Montgomery's trick): zeta - 1, zeta - omega^i (public inputs), zeta - omega^(n_pub + cci)
  // (BSB22).  A zero denominator keeps the reference's behaviour: batch_invert leaves zeros alone (verify.rs:377,389) and the
  // BSB22 division by zero gives zero.
  nden = 0;
  den[nden++] = zm1;
  n_in = n_inputs <= (size_t)MAXIN ? n_inputs : (size_t)MAXIN;   // larger public-input counts fall back to per-term inversions below
  {
    FrM accw = one;
    for (size_t i = 0; i < n_in; i++) { den[nden++] = F.sub(zeta, accw); accw = F.mul(accw, vk.generator); }
  }
  for (uint64_t i = 0; i < vk.n_cci; i++) den[nden++] = F.sub(zeta, vk.wpow[i]);
  acc = one;
  for (int i = 0; i < nden; i++) { zero[i] = F.is_zero(den[i]); pre[i] = acc; if (!zero[i]) acc = F.mul(acc, den[i]); }
  PL_MARK(5);
  PL_DUMP(4, zeta_n); PL_DUMP(5, zh_zeta); PL_DUMP(6, acc); PL_DUMP(7, den[0]); PL_DUMP(8, den[1]); PL_DUMP(9, den[2]); PL_DUMP(10, den[3]); PL_DUMP(11, pre[3]);
  return PL_OK;
}
PL_HD int PlonkStage1::b(const FrM& acc_inv, MsmTerm* terms, uint8_t* tflags, const FrM* h2f) {
  const FrCtx& F = fr_ctx();
  const PlonkKey& vk = *vkp; PlonkWork& wk = *wkp; PlonkProof& pr = wk.pr;
  const FrM one = F.one;
  FrM inv[MAXDEN];
  PL_MARK(6);
  {
    FrM ai = acc_inv;
    for (int i = nden - 1; i >= 0; i--) {
      if (zero[i]) { inv[i] = den[i]; continue; }
      inv[i] = F.mul(ai, pre[i]); ai = F.mul(ai, den[i]);
    }
  }
  PL_DUMP(12, acc_inv); PL_DUMP(13, inv[0]); PL_DUMP(14, inv[1]); PL_DUMP(15, inv[2]); PL_DUMP(16, inv[3]);
  FrM lagrange_one = F.mul(F.mul(inv[0], zh_zeta), vk.size_inv);
  PL_DUMP(17, lagrange_one);
  // verify.rs:109-137: PI = sum_i L_i(zeta) w_i
  FrM pi = {{0, 0, 0, 0}}, accw = one;
  const FrM zs = F.mul(zh_zeta, vk.size_inv);
  for (size_t i = 0; i < n_inputs; i++) {
    FrM iv = i < n_in ? inv[1 + i] : (F.is_zero(F.sub(zeta, accw)) ? F.sub(zeta, accw) : F.inverse(F.sub(zeta, accw)));
    FrM x = F.mul(F.mul(F.mul(zs, iv), accw), F.from_be32(inputs + 32 * i));
    accw = F.mul(accw, vk.generator);
    pi = F.add(pi, x);
  }
  // verify.rs:139-163: BSB22 commitments enter the public-input polynomial through hash_to_field
  PL_MARK(7);
  PL_DUMP(18, pi);
  for (uint64_t i = 0; i < vk.n_cci; i++) {
    FrM hashed = h2f ? h2f[i] : bsb22_hash_to_field(proof + pr.off_bsb + 64 * i);
    FrM lag = F.mul(F.mul(F.mul(zs, vk.wpow[i]), inv[1 + n_in + i]), hashed);
    pi = F.add(pi, lag);
  }
  // verify.rs:165-214: the constant term of the linearised polynomial must equal the claimed opening
  PL_MARK(8);
  PL_DUMP(19, pi);
  const FrM &l = pr.claimed[1], &r = pr.claimed[2], &o = pr.claimed[3], &s1 = pr.claimed[4], &s2 = pr.claimed[5], &zu = pr.zs_value;
  FrM a2l1 = F.mul(F.mul(lagrange_one, alpha), alpha);
  FrM cl = F.add(F.add(F.mul(beta, s1), gamma), l);
  cl = F.mul(cl, F.add(F.add(F.mul(beta, s2), gamma), r));
  cl = F.mul(cl, F.add(o, gamma));
  cl = F.mul(F.mul(cl, alpha), zu);
  PL_DUMP(20, a2l1); PL_DUMP(21, cl);
  cl = F.neg(F.add(F.sub(cl, a2l1), pi));
  wk.lin_opening = cl;
  PL_DUMP(22, cl); PL_DUMP(23, pr.claimed[0]); PL_DUMP(24, pr.claimed[1]); PL_DUMP(25, pr.claimed[5]); PL_DUMP(26, pr.zs_value);
  {
    // Fr == compares stored words: a claimed value that is not reduced (>= r) can never equal the reduced left-hand side
    FrM raw;
    const uint8_t* cb0 = proof + pr.off_claimed;
    for (int i = 0; i < 4; i++) { uint64_t v = 0; for (int j = 0; j < 8; j++) v = v << 8 | cb0[(3 - i) * 8 + j]; raw.l[i] = v; }
    if (F.geq_m(raw) || !F.eq(cl, pr.claimed[0])) return PL_OPENING;
  }
  // verify.rs:216-250
  FrM t1;
  FrM _s1 = F.add(F.add(F.mul(beta, s1), l), gamma);
  t1 = F.add(F.add(F.mul(beta, s2), r), gamma);
  _s1 = F.mul(F.mul(F.mul(F.mul(_s1, t1), beta), alpha), zu);
  FrM _s2 = F.add(F.add(F.mul(beta, zeta), gamma), l);
  FrM u = F.mul(beta, vk.coset_shift);
  _s2 = F.mul(_s2, F.add(F.add(F.mul(u, zeta), gamma), r));
  FrM u2 = F.mul(u, vk.coset_shift);
  _s2 = F.mul(_s2, F.add(F.add(F.mul(u2, zeta), gamma), o));
  _s2 = F.neg(F.mul(_s2, alpha));
  FrM coeff_z = F.add(a2l1, _s2);
  FrM rl = F.mul(l, r);
  FrM zn2 = F.mul(zeta_n, F.mul(zeta, zeta));                              // zeta^(n+2)
  FrM zn2sq = F.mul(zn2, zn2);
  zn2 = F.neg(F.mul(zn2, zh_zeta));
  zn2sq = F.neg(F.mul(zn2sq, zh_zeta));
  FrM zh = F.neg(zh_zeta);
  // verify.rs:252-284: the MSM of the linearised polynomial digest
  PL_MARK(9);
  int np = 0;
  auto put = [&](const G1Aff& p, const FrM& k) { tflags[np] = 0; put_term(terms[np], p, k, &tflags[np]); np++; };
  auto fix = [&](const FrM& k) { put_fixed(terms[np], k, &tflags[np]); np++; };          // key-side points: fixed-base tables (plonk_msm1_shape)
  for (uint32_t i = 0; i < pr.n_bsb; i++) put(pr.bsb[i], pr.claimed[6 + i]);
  fix(l); fix(r); fix(rl); fix(o);                                                         // ql, qr, qm, qo
  fix(one); fix(_s1); put(pr.z, coeff_z);                                                  // qk, s3
  put(pr.h[0], zh); put(pr.h[1], zn2); put(pr.h[2], zn2sq);
  PL_MARK(10);
  return PL_OK;
}
PL_HD int plonk_stage1(const PlonkKey& vk, const uint8_t* proof, size_t proof_len, const uint8_t* inputs, size_t n_inputs,
                        PlonkWork& wk, MsmTerm* terms /* plonk_stage1_terms(vk) */, uint8_t* tflags /* one per term */) {
  PlonkStage1 s;
  int st = s.a(vk, proof, proof_len, inputs, n_inputs, wk);
  if (st != PL_OK) return st;
  return s.b(fr_ctx().inverse(s.acc), terms, tflags);
}

// Stage 2 (plonk/verify.rs:286-303, kzg.rs:46-190): fold the opening proofs at zeta and at zeta * omega.  lin_digest: the stage-1
// MSM result as 16 little-endian words (x | y) and its identity flag.  Writes the terms of
//   P0 = sum_i gamma^i D_i + lambda Z - fe G_kzg + zeta H_batch + lambda zeta omega H_zs      (plonk_stage2_terms(vk) terms)
//   P1 = -(H_batch + lambda H_zs)                                                              (2 terms)
// for the check e(P0, g2[0]) e(P1, g2[1]) == 1 (kzg.rs:175-187).  t1 must be t0 + plonk_stage2_terms(vk): one flag array (t0_inf) covers both.
// weight != nullptr (BN254_FLAG_RLC): every scalar of P0 and P1 is multiplied by *weight, a random non-zero value of the call -- the pass then checks
//   e(sum_i w_i P0_i, g2[0]) e(sum_i w_i P1_i, g2[1]) == 1  over groups of proofs (GT has prime order: a group passes iff each of its proofs does, up to 2^-128).
PL_HD void plonk_stage2(const PlonkKey& vk, const uint8_t* proof, const PlonkWork& wk, const uint32_t lin_words[16], bool lin_inf,
                         MsmTerm* t0, uint8_t* t0_inf, MsmTerm* t1, const FrM* weight = nullptr) {
  const FrCtx& F = fr_ctx();
  const PlonkProof& pr = wk.pr;
  const int nd = 6 + (int)vk.n_qcp;
  // the digests in folding order: linearised polynomial, l, r, o, s1, s2, qcp...
  uint8_t lin_enc[64];
  G1Aff lin;
  if (lin_inf) { for (int j = 0; j < 64; j++) lin_enc[j] = 0; lin.x = fp_zero(); lin.y = fp_zero(); }
  else {
    // the MSM result arrives as canonical little-endian words: its big-endian encoding is a byte shuffle, its digit form one product each
    lin.x = fp_from_words(lin_words); lin.y = fp_from_words(lin_words + 8);
    words_to_be(lin_enc, lin_words); words_to_be(lin_enc + 32, lin_words + 8);
  }
  // kzg.rs:46-72: a fresh transcript for the folding challenge
  uint8_t b32[32], dgam[32];
  Challenge cg("gamma", 5, nullptr);
  F.to_be(b32, wk.zeta); cg.bind(b32, 32);
  cg.bind(lin_enc, 64);
  cg.bind(proof, 192);                                      // l, r, o
  cg.bind(vk.enc[0], 64); cg.bind(vk.enc[1], 64);           // s1, s2
  for (uint32_t i = 0; i < vk.n_qcp; i++) cg.bind(vk.enc[8 + i], 64);
  cg.bind(proof + pr.off_claimed, 32 * (size_t)nd);         // the claimed values as stored
  cg.bind(proof + pr.off_zs_h + 64, 32);                    // z(zeta omega) as stored
  FrM kgamma = cg.finish(dgam);
  PL_MARK(17);
  FrM gi[PLONK_MAX_QCP + 6];
  gi[0] = F.one;
  for (int i = 1; i < nd; i++) gi[i] = F.mul(gi[i - 1], kgamma);
  FrM folded_eval = {{0, 0, 0, 0}};
  for (int i = 0; i < nd; i++) folded_eval = F.add(folded_eval, F.mul(pr.claimed[i], gi[i]));
  // kzg.rs:128-190: batch the two opening points with lambda
  const FrM lam = wk.lambda;
  FrM fe = F.add(folded_eval, F.mul(pr.zs_value, lam));
  FrM shifted = F.mul(wk.zeta, vk.generator);
  int np = 0;
  PL_MARK(18);
  for (int j = 0; j < plonk_stage2_terms(vk) + 2; j++) t0_inf[j] = 0;
  auto wt = [&](const FrM& k) -> FrM { return weight ? F.mul(k, *weight) : k; };
  auto put = [&](const G1Aff& p, const FrM& k) { put_term(t0[np], p, wt(k), &t0_inf[np]); np++; };
  auto fix = [&](const FrM& k) { put_fixed(t0[np], wt(k), &t0_inf[np]); np++; };          // key-side points: fixed-base tables (plonk_msm2_shape)
  t0_inf[np] = lin_inf ? 1 : 0; put(lin, gi[0]);
  put(pr.lro[0], gi[1]); put(pr.lro[1], gi[2]); put(pr.lro[2], gi[3]);
  fix(gi[4]); fix(gi[5]);                                                                  // s1, s2
  for (uint32_t i = 0; i < vk.n_qcp; i++) fix(gi[6 + i]);                                  // qcp[i]
  put(pr.z, lam);
  fix(F.neg(fe));                                                                          // kzg_g1
  put(pr.batch_h, wk.zeta);
  put(pr.zs_h, F.mul(lam, shifted));
  // P1 = -(H_batch + lambda H_zs): the two terms follow P0's (t1 = t0 + plonk_stage2_terms(vk), their flags likewise); the first is the bare point, negated
  if (weight) put_term(t1[0], pr.batch_h, F.neg(*weight), &t0_inf[np]);
  else put_unit(t1[0], pr.batch_h, true, &t0_inf[np]);
  put_term(t1[1], pr.zs_h, wt(F.neg(lam)), &t0_inf[np + 1]);
  PL_MARK(19);
}

}  // namespace bn254host
