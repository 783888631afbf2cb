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
#define BN254_ST_PENDING 0x80  // internal: no error so far (low 7 bits: deferred error of point C)

#define MSM_ENTRY_DWORDS 20    // affine G1 point, 2 x 9 limbs + 2 pad: 80-byte entries, 16-byte aligned
#define G16_WS_ELEMS 23        // Fp elements per proof in the SoA workspace: A 2, B 4, C 2, L 3, f 12

struct G16LaunchArgs {
  const uint8_t* proofs; size_t stride;
  const uint8_t* inputs; int n_public;
  size_t n;
  int32_t* ws;              // G16_WS_ELEMS * 9 * n dwords
  uint8_t* status;          // n bytes
  const int32_t* msm_tab;   // n_public * 32 * 255 entries of MSM_ENTRY_DWORDS
  const int32_t* k0;        // 18 dwords: affine K[0]
  const int32_t* gtab;      // BN_ATE_STEPS * 36 dwords: lines of the G2 argument paired with L
  const int32_t* dtab;      // same for the one paired with C
  const int32_t* target;    // 108 dwords: the GT element the product must equal
  int inputs_match_key;     // n_public + 1 == len(vk.K)
};
hipError_t bn254_launch_g16(const G16LaunchArgs& a, hipStream_t s, hipEvent_t* ev);
hipError_t bn254_launch_dbg_fp_mul(const uint8_t* a, const uint8_t* b, uint8_t* o, size_t n, hipStream_t s);
hipError_t bn254_launch_dbg_fp12_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* o, size_t n, hipStream_t s);
hipError_t bn254_launch_dbg_pairing(const uint8_t* g1, const uint8_t* g2, uint8_t* o, size_t n, hipStream_t s);
hipError_t bn254_launch_dbg_g2_subgroup(const uint8_t* g2, uint8_t* o, size_t n, hipStream_t s);
