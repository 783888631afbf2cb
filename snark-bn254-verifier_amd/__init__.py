"""MI355X-native batch BN254 verifier: Python-side access to the C ABI (include/bn254_verify.h).

The product is the shared library built from csrc/ (HIP kernels + C ABI); this package only loads it with ctypes so that
tests, __graft_entry__ and bench.py can call the same entry points a Rust/C host would bind.  There is no Python or CPU
implementation of verification here: if the library or a GPU is missing, calls raise.
"""
from .binding import (  # noqa: F401
    ACCEPT, REJECT, ERR_NOT_MEMBER, ERR_NOT_ON_CURVE, ERR_NOT_IN_SUBGROUP, ERR_INPUT_LEN, ERR_MALFORMED,
    VK_REFERENCE, VK_GNARK, Bn254Error, Groth16Verifier, PreparedVk, build, lib, lib_path, synth_groth16, kernel_kinds,
    set_profile_kernels, PreparedPlonkVk, PlonkVerifier, FLAG_STRICT_SCALARS, FLAG_RLC, RAW_PROOF_LEN, proof_write_raw, shard_plan, set_rlc_params, set_plonk_params,
)
