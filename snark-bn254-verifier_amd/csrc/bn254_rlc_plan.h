// bn254_rlc_plan.h -- the fold plan of the RLC batch mode (bn254_rlc.h): which lanes are folded into which, hence which proofs form a group.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define BN_PLAN_HD __host__ __device__ inline
#else
#define BN_PLAN_HD inline
#endif
namespace bn254 {
// fold plan: lane counts after each round (cur -> ceil(cur / 2): lane j < cur - half takes lane j + half), down to about n >> log2_group lanes
// The first `pre` rounds undo the proofs-per-lane layout of the shared-accumulator Miller loop (proof q of lane j sits at index j + q m,
// m = lanes): their halves are m 2^k, and they fold only the per-proof values (C', t_j), the accumulator f being shared already.
struct RlcPlan { int rounds, pre; uint32_t half[28]; uint32_t groups, lanes; };
inline RlcPlan rlc_plan(uint32_t n, int log2_group, int log2_share = 0) {
  RlcPlan p; p.rounds = 0; p.pre = 0;
  const uint32_t share = 1u << log2_share;
  const uint32_t m = (n + share - 1) / share;
  p.lanes = m;
  for (int k = log2_share - 1; k >= 0; k--) { p.half[p.rounds++] = m << k; p.pre++; }
  uint32_t target = n >> log2_group; if (target < 1) target = 1;
  uint32_t cur = m;
  while (cur > target && p.rounds < 28) { uint32_t h = (cur + 1) / 2; p.half[p.rounds++] = h; cur = h; }
  p.groups = cur;
  return p;
}
BN_PLAN_HD uint32_t rlc_group_of(uint32_t i, const RlcPlan& p) {
  for (int k = 0; k < p.rounds; k++) if (i >= p.half[k]) i -= p.half[k];
  return i;
}

}  // namespace bn254
