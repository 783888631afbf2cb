"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE: imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product package."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
# BN254_ORACLE_LIB: another build of the same sources (tests/test_sanitizers.py: liboracle_san.so, AddressSanitizer + UBSan)
LIB_PATH = os.environ.get("BN254_ORACLE_LIB") or os.path.join(HERE, "liboracle.so")

REJECT, ACCEPT, ERR_NOT_MEMBER, ERR_NOT_ON_CURVE, ERR_NOT_IN_SUBGROUP, ERR_INPUT_LEN, ERR_MALFORMED = range(7)
ERR_OPENING_MISMATCH, ERR_PAIRING_FAILED, ERR_BSB22_MISMATCH, ERR_INVERSE = 7, 8, 9, 10
MODE_REFERENCE, MODE_GNARK = 0, 1

P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, os.path.basename(LIB_PATH)])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_get_fp_mul_count.restype = C.c_uint64
    return _lib


def _buf(n):
    return (C.c_uint8 * n)()


def _b(x):
    return bytes(x)


def be32(v):
    return int(v).to_bytes(32, "big")


def fp_op(op, a, b=0, field=0):
    o = _buf(32)
    lib().orc_fp_op(op, o, be32(a), be32(b), field)
    return int.from_bytes(_b(o), "big")


def fp2_op(op, a, b=(0, 0)):
    o = _buf(64)
    lib().orc_fp2_op(op, o, be32(a[0]) + be32(a[1]), be32(b[0]) + be32(b[1]))
    r = _b(o)
    return int.from_bytes(r[:32], "big"), int.from_bytes(r[32:], "big")


def fp12_op(op, a, b=None):
    o = _buf(384)
    lib().orc_fp12_op(op, o, a, b if b is not None else bytes(384))
    return _b(o)


def g1_gen():
    o = _buf(64); lib().orc_g1_gen(o); return _b(o)


def g2_gen():
    o = _buf(128); lib().orc_g2_gen(o); return _b(o)


def g1_mul(p, k):
    o = _buf(64); lib().orc_g1_scalar_mul(o, p, be32(k)); return _b(o)


def g1_add(p, q):
    o = _buf(64); lib().orc_g1_add_bytes(o, p, q); return _b(o)


def g2_mul(p, k):
    o = _buf(128); lib().orc_g2_scalar_mul(o, p, be32(k)); return _b(o)


def g2_add(p, q):
    o = _buf(128); lib().orc_g2_add_bytes(o, p, q); return _b(o)


def g2_subgroup_check(p):
    return lib().orc_g2_subgroup_check(p)


def miller_loop(g1s, g2s):
    n = len(g1s) // 64
    o = _buf(384); lib().orc_miller_loop(o, g1s, g2s, n); return _b(o)


def final_exp(f, plain=False):
    o = _buf(384); lib().orc_final_exp(o, f, 1 if plain else 0); return _b(o)


def pairing(g1s, g2s):
    n = len(g1s) // 64
    o = _buf(384); lib().orc_pairing_bytes(o, g1s, g2s, n); return _b(o)


def decompress_g1(b):
    o = _buf(64); st = lib().orc_decompress_g1(o, b); return st, _b(o)


def decompress_g2(b, mode=MODE_GNARK):
    o = _buf(128); st = lib().orc_decompress_g2(o, b, mode); return st, _b(o)


def compress_g1(b):
    o = _buf(32); lib().orc_compress_g1(o, b); return _b(o)


def compress_g2(b):
    o = _buf(64); lib().orc_compress_g2(o, b); return _b(o)


def sha256(d):
    o = _buf(32); lib().orc_sha256(o, d, C.c_size_t(len(d))); return _b(o)


def expand_msg_xmd(msg, dst, n):
    o = _buf(n); lib().orc_expand_msg_xmd(o, C.c_size_t(n), msg, C.c_size_t(len(msg)), dst, C.c_size_t(len(dst))); return _b(o)


def groth16_verify(proof, vk, inputs, mode=MODE_REFERENCE):
    """inputs: list of ints (or 32-byte strings)"""
    ib = b"".join(i if isinstance(i, (bytes, bytearray)) else be32(i) for i in inputs)
    return lib().orc_groth16_verify(proof, C.c_size_t(len(proof)), vk, C.c_size_t(len(vk)), ib, C.c_size_t(len(inputs)), mode)


def groth16_verify_many(proofs, stride, vk, inputs_bytes, n_inputs, n, mode=MODE_REFERENCE):
    st = _buf(n)
    lib().orc_groth16_verify_many(proofs, C.c_size_t(stride), vk, C.c_size_t(len(vk)), inputs_bytes, C.c_size_t(n_inputs), C.c_size_t(n), mode, st)
    return _b(st)


def groth16_verify_many_prepared(proofs, stride, vk, inputs_bytes, n_inputs, n, mode=MODE_REFERENCE):
    """Batch mode of the same algorithm: key loaded once, pairing(alpha, beta) hoisted (second CPU row)."""
    st = _buf(n)
    lib().orc_groth16_verify_many_prepared(proofs, C.c_size_t(stride), vk, C.c_size_t(len(vk)), inputs_bytes, C.c_size_t(n_inputs), C.c_size_t(n), mode, st)
    return _b(st)


def build_flags():
    """Compiler and flags the oracle was built with (reported in bench.py's cpu_baseline line)."""
    lib().orc_build_flags.restype = C.c_char_p
    return lib().orc_build_flags().decode()


def plonk_verify(proof, vk, inputs, lam=None):
    ib = b"".join(i if isinstance(i, (bytes, bytearray)) else be32(i) for i in inputs)
    return lib().orc_plonk_verify(proof, C.c_size_t(len(proof)), vk, C.c_size_t(len(vk)), ib, C.c_size_t(len(inputs)), be32(lam) if lam is not None else None)


def plonk_stage_digests(proof, vk, inputs):
    ib = b"".join(be32(i) for i in inputs)
    o = _buf(176)
    st = lib().orc_plonk_stage_digests(proof, C.c_size_t(len(proof)), vk, C.c_size_t(len(vk)), ib, C.c_size_t(len(inputs)), o)
    r = _b(o)
    return st, {"gamma": r[0:32], "beta": r[32:64], "alpha": r[64:96], "zeta": r[96:128], "h2f": r[128:176]}


def plonk_pairing_inputs(proof, vk, inputs):
    """(status, [P0, P1], [Q0, Q1]): the operands of the final KZG pairing check of a PlonK proof (plonk/kzg.rs:175-187)."""
    ib = b"".join(be32(i) for i in inputs)
    o = _buf(384)
    st = lib().orc_plonk_pairing_inputs(proof, C.c_size_t(len(proof)), vk, C.c_size_t(len(vk)), ib, C.c_size_t(len(inputs)), o)
    r = _b(o)
    return st, [r[0:64], r[64:128]], [r[128:256], r[256:384]]


def plonk_pairing_inputs_lam(proof, vk, inputs, lam):
    """Same with an explicit KZG batching scalar."""
    ib = b"".join(be32(i) for i in inputs)
    o = _buf(384)
    st = lib().orc_plonk_pairing_inputs_lam(proof, C.c_size_t(len(proof)), vk, C.c_size_t(len(vk)), ib, C.c_size_t(len(inputs)), be32(lam), o)
    r = _b(o)
    return st, [r[0:64], r[64:128]], [r[128:256], r[256:384]]


def set_threads(n):
    lib().orc_set_threads(int(n))


def fp_mul_count(reset=False):
    v = lib().orc_get_fp_mul_count()
    if reset:
        lib().orc_reset_fp_mul_count()
    return v
