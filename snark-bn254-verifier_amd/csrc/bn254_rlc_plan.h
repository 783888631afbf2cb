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
struct RlcPlan { int rounds; uint32_t half[24]; uint32_t groups; };
inline RlcPlan rlc_plan(uint32_t n, int log2_group) {
  RlcPlan p; p.rounds = 0;
  uint32_t target = n >> log2_group; if (target < 1) target = 1;
  uint32_t cur = n;
  while (cur > target && p.rounds < 24) { uint32_t h = (cur + 1) / 2; p.half[p.rounds++] = h; cur = h; }
  p.groups = cur;
  return p;
}
BN_PLAN_HD uint32_t rlc_group_of(uint32_t i, const RlcPlan& p) {
  for (int k = 0; k < p.rounds; k++) if (i >= p.half[k]) i -= p.half[k];
  return i;
}

}  // namespace bn254
