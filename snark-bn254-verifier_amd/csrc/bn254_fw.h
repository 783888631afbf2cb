// bn254_fw.h -- the width of the fixed-base windows of the PlonK MSMs: shared by the row plans (bn254_msm.h, also compiled for the host tests), the kernels and the
// table construction (bn254_k_comb.hip).  No other dependency.
#pragma once
// Window tables of the key points of the PlonK MSMs (bn254_msm.h rows, built by bn254_k_comb.hip form 2): MSM_FW_WINDOWS windows of MSM_FW_BITS bits, 2^bits - 1 multiples
// each.  The bases are batch-constant and a large call makes 2^18 x 200 of these additions, so wider windows pay: byte windows (rounds 3-4) are 32 additions per term and
// 0.65 MB per point, 13 bits 20 additions and 13 MB, 16 bits 16 additions and 84 MB (and 30 ms to build at a key's first use): profiles/r05_plonk_fixed_windows.txt.
#ifndef MSM_FW_BITS
#define MSM_FW_BITS 13
#endif
#define MSM_FW_WINDOWS ((256 + MSM_FW_BITS - 1) / MSM_FW_BITS)
#define MSM_FW_ENTRIES ((1u << MSM_FW_BITS) - 1u)
