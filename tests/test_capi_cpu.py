"""CPU-side checks of the product library: it loads, exports every symbol include/bn254_verify.h declares, shares the
oracle's status codes, prepares keys on the host, refuses to verify without a GPU (no CPU fallback), and its synthetic
workload generator emits gnark bytes that the ORACLE judges exactly as the generator predicts."""
import ctypes as C
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


def test_exports_match_header(pkg):
    hdr = open(os.path.join(ROOT, "include", "bn254_verify.h")).read()
    names = set(re.findall(r"\b(bn254_[a-z0-9_]+)\s*\(", hdr))
    names -= {"bn254_g16_pvk"}
    assert len(names) >= 18
    L = pkg.lib()
    for n in sorted(names):
        assert hasattr(L, n), "missing export " + n
    assert L.bn254_version().startswith(b"bn254-verify-amd")


def test_header_is_plain_c_and_links(pkg, tmp_path):
    """The boundary is a C ABI: include/bn254_verify.h compiles as C99 (-pedantic, no C++), a C program links against the library, and the library built here answers
    the header's BN254_ABI_VERSION (what binding.py and the Rust crate check before they call anything else)."""
    import subprocess
    src = tmp_path / "use_header.c"
    src.write_text('#include "bn254_verify.h"\n#include <stdio.h>\nint main(void) { int dev[64], ns = 0; size_t first[64], count[64];\n'
                   '  if (bn254_shard_plan(10, 0x7, 8, dev, first, count, &ns) != 0 || ns != 3) return 2;\n'
                   '  printf("%d %s %d %d\\n", bn254_abi_version(), bn254_version(), (int)first[1], (int)(first[1] + count[1]));\n'
                   '  return bn254_abi_version() == BN254_ABI_VERSION ? 0 : 1; }\n')
    libdir = os.path.join(ROOT, "snark-bn254-verifier_amd")
    exe = tmp_path / "use_header"
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-L", libdir, "-lbn254_verify_amd",
                    "-Wl,-rpath," + libdir, "-o", str(exe)], check=True, capture_output=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    ver, name, lo, hi = r.stdout.split()[0], r.stdout.split()[1], r.stdout.split()[-2], r.stdout.split()[-1]
    hdr = open(os.path.join(ROOT, "include", "bn254_verify.h")).read()
    assert int(ver) == int(re.search(r"#define BN254_ABI_VERSION (\d+)", hdr).group(1)) and name.startswith("bn254-verify-amd")
    sharding = __import__("importlib").import_module("snark-bn254-verifier_amd.sharding")
    assert (int(lo), int(hi)) == sharding.shard_bounds(10, 3, 1)


def test_key_cache_capacity_from_the_environment(pkg):
    """The single-proof entries keep the last prepared keys by exact bytes: 4 by default, BN254_KEY_CACHE=0 none, =N that many (capped at 64) -- read once per process."""
    import subprocess, sys
    code = ("import ctypes as C, os; L = C.CDLL(os.path.join(%r, 'snark-bn254-verifier_amd', 'libbn254_verify_amd.so')); print(L.bn254_dbg_key_cache_slots())" % ROOT)
    for env, want in ((None, 4), ("0", 0), ("1", 1), ("9", 9), ("1000", 64), ("-3", 0)):
        e = dict(os.environ); e.pop("BN254_KEY_CACHE", None)
        if env is not None:
            e["BN254_KEY_CACHE"] = env
        r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
        assert r.returncode == 0 and int(r.stdout.strip()) == want, (env, r.stdout, r.stderr[-500:])


def test_status_codes_shared_with_oracle(pkg, O):
    hdr = open(os.path.join(ROOT, "include", "bn254_verify.h")).read()
    ohdr = open(os.path.join(ROOT, "oracle", "oracle.h")).read()
    for name in ("REJECT", "ACCEPT", "ERR_NOT_MEMBER", "ERR_NOT_ON_CURVE", "ERR_NOT_IN_SUBGROUP", "ERR_INPUT_LEN", "ERR_MALFORMED",
                 "ERR_OPENING_MISMATCH", "ERR_PAIRING_FAILED", "ERR_BSB22_MISMATCH", "ERR_INVERSE"):
        a = int(re.search(r"BN254_%s = (\d+)" % name, hdr).group(1))
        b = int(re.search(r"ORC_%s = (\d+)" % name, ohdr).group(1))
        assert a == b == getattr(O, name)
    assert (pkg.ACCEPT, pkg.REJECT, pkg.ERR_MALFORMED) == (O.ACCEPT, O.REJECT, O.ERR_MALFORMED)


def test_synth_workload_judged_by_oracle(pkg, O):
    n = 20
    vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540001, 2, n, invalid_every=2, agree=True, threads=4)
    assert len(vk) == 520 and len(proofs) == 256 * n and len(inputs) == 64 * n
    assert set(exp) == {O.ACCEPT, O.REJECT, O.ERR_NOT_ON_CURVE, O.ERR_NOT_IN_SUBGROUP, O.ERR_NOT_MEMBER}
    # on an agreement-set key the reference-literal and the gnark reading of the key coincide on every proof
    for mode in (O.MODE_REFERENCE, O.MODE_GNARK):
        assert O.groth16_verify_many(proofs, 256, vk, inputs, 2, n, mode) == exp
    # deterministic
    assert pkg.synth_groth16(0xB2540001, 2, n, invalid_every=2, agree=True, threads=2) == (vk, proofs, inputs, exp)


def test_mode_disagreement_outside_agreement_set(pkg, O):
    """for most gnark keys the reference's literal equation rejects valid proofs."""
    seen_disagree = False
    for seed in range(1, 12):
        vk, proofs, inputs, exp = pkg.synth_groth16(seed, 1, 2, invalid_every=0, agree=False, threads=2)
        g = O.groth16_verify_many(proofs, 256, vk, inputs, 1, 2, O.MODE_GNARK)
        r = O.groth16_verify_many(proofs, 256, vk, inputs, 1, 2, O.MODE_REFERENCE)
        assert g == bytes([O.ACCEPT] * 2)
        if r != g:
            seen_disagree = True
            assert r == bytes([O.REJECT] * 2)
            break
    assert seen_disagree


def test_vk_prepare_on_host(pkg):
    vk, _, _, _ = pkg.synth_groth16(3, 2, 0, invalid_every=0, agree=True, threads=1)
    for mode in (pkg.VK_REFERENCE, pkg.VK_GNARK):
        p = pkg.PreparedVk(vk, mode)
        assert p.n_public == 2
        p.close()
    L = pkg.lib()
    h = C.c_void_p()
    assert L.bn254_groth16_vk_prepare(vk[:100], 100, 0, C.byref(h)) == -4          # short buffer (slice panic in the reference)
    bad = bytearray(vk); bad[0] &= 0x3F                                           # flag 0b00 (constants.rs:24 panics)
    assert L.bn254_groth16_vk_prepare(bytes(bad), len(bad), 0, C.byref(h)) == -4
    bad = bytearray(vk); bad[0] = 0x40; bad[1:32] = bytes(31)                    # compressed G1 infinity: x = 0, 3 is a non-residue
    assert L.bn254_groth16_vk_prepare(bytes(bad), len(bad), 0, C.byref(h)) == -4
    bad = bytearray(vk); bad[288:292] = (1000).to_bytes(4, "big")                # nK runs past the buffer
    assert L.bn254_groth16_vk_prepare(bytes(bad), len(bad), 0, C.byref(h)) == -4
    assert L.bn254_groth16_vk_prepare(vk, len(vk), 7, C.byref(h)) == -1           # bad mode


@pytest.mark.skipif(not _no_gpu(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback(pkg):
    """The product path must fail loudly, not route through any CPU implementation."""
    vk, proofs, inputs, _ = pkg.synth_groth16(3, 2, 2, invalid_every=0, agree=True, threads=1)
    p = pkg.PreparedVk(vk)
    with pytest.raises(pkg.Bn254Error) as ei:
        p.verify_batch(proofs, inputs)
    assert "-2" in str(ei.value) or "-3" in str(ei.value)
    with pytest.raises(pkg.Bn254Error):
        pkg.Groth16Verifier.verify(proofs[:256], vk, [1, 2])


@pytest.mark.skipif(not _no_gpu(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback_plonk(pkg, fixtures):
    """Same for the PlonK path: the key parses on the host, verification needs the GPU."""
    fx, vk = fixtures
    f = next(v for v in fx.values() if v["variant"] == "plonk")
    p = pkg.PreparedPlonkVk(vk)
    assert p.n_public == 2
    with pytest.raises(pkg.Bn254Error) as ei:
        p.verify_batch(bytes.fromhex(f["raw_proof"]), b"".join(int(x).to_bytes(32, "big") for x in f["public_inputs"]))
    assert "-2" in str(ei.value) or "-3" in str(ei.value)


def test_product_does_not_reference_oracle():
    pk = os.path.join(ROOT, "snark-bn254-verifier_amd")
    for dp, _, fns in os.walk(pk):
        if "build" in dp:
            continue
        for fn in fns:
            if fn.endswith((".py", ".h", ".hpp", ".hip", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert not re.search(r'#include\s*[<"][^">]*oracle|liboracle|^\s*(from|import)\s+oracle', txt, re.M), fn


def test_point_codecs_both_directions(pkg, O):
    """gnark compressed <-> uncompressed through the C ABI against the oracle's codecs (converter.rs:23-153), both root-order
    readings for G2, checked variants on points outside the subgroup, malformed flags."""
    import ctypes as C, random
    L = pkg.lib()
    rng = random.Random(5)
    R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    g1, g2 = O.g1_gen(), O.g2_gen()
    st = C.c_uint8(0)
    for _ in range(8):
        p = O.g1_mul(g1, rng.randrange(1, R)); q = O.g2_mul(g2, rng.randrange(1, R))
        c1 = (C.c_uint8 * 32)(); c2 = (C.c_uint8 * 64)(); u1 = (C.c_uint8 * 64)(); u2 = (C.c_uint8 * 128)()
        assert L.bn254_g1_compress(p, c1) == 0 and bytes(c1) == O.compress_g1(p)
        assert L.bn254_g2_compress(q, c2) == 0 and bytes(c2) == O.compress_g2(q)
        for checked in (0, 1):
            assert L.bn254_g1_decompress(bytes(c1), u1, checked, C.byref(st)) == 0 and st.value == pkg.ACCEPT and bytes(u1) == p
            assert L.bn254_g2_decompress(bytes(c2), u2, pkg.VK_GNARK, checked, C.byref(st)) == 0 and st.value == pkg.ACCEPT and bytes(u2) == q
        # the reference's reading of the root order agrees with the oracle's mode-0 decoder (it may return -q)
        assert L.bn254_g2_decompress(bytes(c2), u2, pkg.VK_REFERENCE, 0, C.byref(st)) == 0 and st.value == pkg.ACCEPT
        ok, ref = O.decompress_g2(bytes(c2), O.MODE_REFERENCE)
        assert ok == O.ACCEPT and bytes(u2) == ref
    bad = bytearray(O.compress_g1(g1)); bad[0] &= 0x3f                      # flag 0b00
    u1 = (C.c_uint8 * 64)()
    assert L.bn254_g1_decompress(bytes(bad), u1, 1, C.byref(st)) == 0 and st.value == pkg.ERR_MALFORMED


def test_sp1_fixture_reader(pkg, fixtures):
    """bn254_sp1_fixture_parse on the reference's 8 fixture files gives the fields tests/golden/fixtures.json was built from."""
    import ctypes as C, os
    L = pkg.lib()
    fx, _vk = fixtures
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sp1")
    for name, f in fx.items():
        buf = open(os.path.join(d, name + "_proof.bin"), "rb").read()
        variant = C.c_int(0); raw = (C.c_uint8 * 2048)(); raw_len = C.c_size_t(0); pis = (C.c_uint8 * 64)(); h = (C.c_uint8 * 32)()
        assert L.bn254_sp1_fixture_parse(buf, C.c_size_t(len(buf)), C.byref(variant), raw, C.c_size_t(2048), C.byref(raw_len), pis, h) == 0
        assert variant.value == (2 if f["variant"] == "plonk" else 3)
        assert bytes(raw)[:raw_len.value] == bytes.fromhex(f["raw_proof"])
        assert bytes(pis) == b"".join(int(x).to_bytes(32, "big") for x in f["public_inputs"])
        assert bytes(h).hex() == f["vkey_hash"]
    assert L.bn254_sp1_fixture_parse(b"\x03\x00", C.c_size_t(2), C.byref(variant), raw, C.c_size_t(2048), C.byref(raw_len), pis, h) != 0


def test_raw_proof_writer_reproduces_reference_fixtures(pkg, fixtures):
    """bn254_groth16_proof_write_raw(A, B, C) rebuilds, byte for byte, the 324-byte raw proofs of the reference's Groth16 fixtures
    (examples/binaries/*_groth16_proof.bin; layout read by groth16/converter.rs:14-26)."""
    fx, _ = fixtures
    seen = 0
    for name, f in sorted(fx.items()):
        if f["variant"] != "groth16":
            continue
        raw = bytes.fromhex(f["raw_proof"])
        assert len(raw) == pkg.RAW_PROOF_LEN
        assert pkg.proof_write_raw(raw[0:64], raw[64:192], raw[192:256]) == raw
        seen += 1
    assert seen == 4


def test_synth_identity_public_input_point_judged_by_oracle(pkg, O):
    """Generator option: proofs whose public-input point L is the identity (the last input cancels K0 + the rest).  bn::pairing_batch skips a
    pair with an identity operand; the oracle must accept these proofs, in both key modes, and they must differ from ordinary ones only in
    their inputs."""
    n = 16
    vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540077, 3, n, invalid_every=0, agree=True, threads=4, l_identity=True)
    vk0, proofs0, inputs0, exp0 = pkg.synth_groth16(0xB2540077, 3, n, invalid_every=0, agree=True, threads=4)
    assert vk == vk0 and exp == exp0 == bytes([pkg.ACCEPT]) * n
    for i in range(n):
        same = inputs[96 * i:96 * i + 96] == inputs0[96 * i:96 * i + 96]
        assert same == (i % 7 != 3)
        assert inputs[96 * i:96 * i + 64] == inputs0[96 * i:96 * i + 64]          # only the last input differs
    for mode in (O.MODE_REFERENCE, O.MODE_GNARK):
        assert O.groth16_verify_many(proofs, 256, vk, inputs, 3, n, mode) == exp


def test_comb_table_multiplication_vs_oracle(pkg, O):
    """x * P from the comb table of P and the kernels' column digits (host probe) equals the oracle's scalar multiplication, for scalars that are
    raw 256-bit values (used modulo r like bn::Fr), all-ones, single bits at tooth and word boundaries, zero."""
    import random
    L = pkg.lib()
    R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    rng = random.Random(3)
    g = O.g1_gen()
    for p in (g, O.g1_mul(g, rng.randrange(1, R))):
        xs = [0, 1, 2, R - 1, R, R + 5, (1 << 256) - 1, (1 << 255), (1 << 240) + (1 << 20), (1 << 19) | (1 << 39) | (1 << 259 - 20)]
        xs += [1 << b for b in (19, 20, 31, 32, 63, 64, 199, 200, 239, 240, 255)] + [rng.getrandbits(256) for _ in range(6)]
        for x in xs:
            out = (C.c_uint8 * 64)()
            assert L.bn254_dbg_comb_mul(p, x.to_bytes(32, "big"), out) == 0
            k = x % R
            assert bytes(out) == (O.g1_mul(p, k) if k else bytes(64)), hex(x)


def test_comb_table_constants_never_vanish():
    """The comb tables of keys with many public inputs (bn254_host.hpp::build_comb_table) hold the sums of 2^(cols * i) K over every non-empty
    set of teeth; K has order r, so an entry is the identity only if that sum of powers is a multiple of r: none is.  The parameters are read
    from the header (13 teeth x 20 columns >= 256 bits: 8191 sums, all below 2^241 < r)."""
    import re
    hdr = open(os.path.join(ROOT, "snark-bn254-verifier_amd", "csrc", "bn254_kernels.h")).read()
    teeth = int(re.search(r"#define G16_COMB_TEETH (\d+)", hdr).group(1)); cols = int(re.search(r"#define G16_COMB_COLS (\d+)", hdr).group(1))
    R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    assert teeth * cols >= 256 and (teeth, cols) == (13, 20)
    assert all(sum(((idx >> i) & 1) << (cols * i) for i in range(teeth)) % R for idx in range(1, 1 << teeth))


def test_no_constant_kzg_batching_scalar_in_the_product():
    """The PlonK path keys its batching scalars with getrandom(2) on every call (a ChaCha20 key and nonce, expanded per proof); no literal
    scalar may come back."""
    import re
    src = open(os.path.join(ROOT, "snark-bn254-verifier_amd", "csrc", "bn254_capi.hip")).read()
    body = src[src.index("static int plonk_run("):src.index("int bn254_plonk_verify(const uint8_t* proof")]
    assert "getrandom(seed" in body and "chacha20_block4(lw" in body and "from_be_reduce((const uint8_t*)lw, 48)" in body
    assert not re.search(r"lambda\s*=\s*fr_ctx\(\)\.from_u64", src)


def test_glv_decomposition(pkg):
    """The scalar decomposition behind the PlonK MSMs (bn254_plonk.hpp::glv_decompose): s1 k1 + s2 k2 lambda == k (mod r), k1, k2 < 2^127."""
    import random
    L = pkg.lib()
    R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    LAM = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd
    rng = random.Random(11)
    cases = [0, 1, 2, R - 1, R - 2, LAM, LAM + 1, R // 2, (1 << 128) - 1, 1 << 127, (1 << 256) - 1] + [rng.randrange(R) for _ in range(3000)]
    k1 = (C.c_uint8 * 16)(); k2 = (C.c_uint8 * 16)(); n1 = C.c_int(); n2 = C.c_int()
    for k in cases:
        assert L.bn254_dbg_glv_decompose(int(k).to_bytes(32, "big"), k1, k2, C.byref(n1), C.byref(n2)) == 0
        a, b = int.from_bytes(bytes(k1), "big"), int.from_bytes(bytes(k2), "big")
        assert a < (1 << 127) and b < (1 << 127), hex(k)
        assert ((-a if n1.value else a) + (-b if n2.value else b) * LAM - k) % R == 0, hex(k)


def test_fr_inverse_binary_gcd(pkg):
    """The inversion of the PlonK stages (constant-time binary GCD with approximated operands on 9 x 29-bit digits, shared by the host path and the device
    kernels) against Python's pow, against the Fermat form and against the classic shift-and-subtract form it replaced on the device, in Fr and in Fp; 0 -> 0."""
    import random
    L = pkg.lib()
    R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    rng = random.Random(12)
    for field, mod in ((0, R), (1, P)):
        vals = [0, 1, 2, mod - 1, mod - 2, (mod + 1) // 2, 1 << 255, (1 << 256) - 1] + [rng.randrange(1 << 256) for _ in range(400)] + [rng.randrange(1, 1 << k) for k in (3, 17, 64, 65, 128, 200)]
        for v in vals:
            o1 = (C.c_uint8 * 32)(); o2 = (C.c_uint8 * 32)(); o3 = (C.c_uint8 * 32)()
            assert L.bn254_dbg_fr_inverse(v.to_bytes(32, "big"), o1, 0, field) == 0 and L.bn254_dbg_fr_inverse(v.to_bytes(32, "big"), o2, 1, field) == 0
            assert L.bn254_dbg_fr_inverse(v.to_bytes(32, "big"), o3, 2, field) == 0
            want = pow(v % mod, -1, mod) if v % mod else 0
            assert int.from_bytes(bytes(o1), "big") == want == int.from_bytes(bytes(o2), "big") == int.from_bytes(bytes(o3), "big"), (field, hex(v))


def test_fr_product_forms_agree(pkg):
    """The 8 x 32-bit-word Montgomery product the device stages run and the 4 x 64-bit one of the host, in Fr and in Fp, against Python integers: 20 000 random pairs
    and every pair of edge values (0, 1, m - 1, m, 2^256 - 1, words of all ones / a single bit) -- the two forms must be the same function."""
    import random
    L = pkg.lib()
    L.bn254_dbg_fr_mul.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_uint8), C.c_size_t, C.c_int, C.c_int]
    R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    rng = random.Random(77)
    for field, mod in ((0, R), (1, P)):
        edge = [0, 1, 2, mod - 1, mod, mod + 1, (1 << 256) - 1, 1 << 255, (1 << 32) - 1, 1 << 32, (1 << 64) - 1, 1 << 64, (1 << 224) - 1, 0xffffffff << 96,
                sum(0xffffffff << (64 * i) for i in range(4)), sum(0xffffffff << (64 * i + 32) for i in range(4))] + [1 << (32 * k) for k in range(8)] + [(1 << (32 * k)) - 1 for k in range(1, 8)]
        pairs = [(x, y) for x in edge for y in edge] + [(rng.randrange(1 << 256), rng.randrange(1 << 256)) for _ in range(20000)]
        a = b"".join(x.to_bytes(32, "big") for x, _ in pairs)
        b = b"".join(y.to_bytes(32, "big") for _, y in pairs)
        outs = []
        for form in (32, 64):
            o = (C.c_uint8 * (32 * len(pairs)))()
            assert L.bn254_dbg_fr_mul(a, b, o, len(pairs), form, field) == 0
            outs.append(bytes(o))
        assert outs[0] == outs[1], "the two product forms differ (field %d)" % field
        # value check (the probe converts to Montgomery form, multiplies in the form under test and converts back)
        for i in list(range(len(edge) ** 2)) + list(range(len(pairs) - 300, len(pairs))):
            x, y = pairs[i]
            got = int.from_bytes(outs[0][32 * i:32 * i + 32], "big")
            assert got == (x % mod) * (y % mod) % mod, (field, hex(x), hex(y))


def test_kernel_mads_record_matches_the_built_library(tmp_path):
    """profiles/kernel_mads.json (tools/count_mads.py: multiply-adds per proof and launch from the gfx950 code objects, what every VALU fraction of the bench line is
    made of) is the count of the library as BUILT -- same figures, same list of loops the model prices once -- and the figures the rooflines rest on are in their known
    ranges (round 4: a restructured loop was silently priced as straight-line code and a config's fraction read 0.07 instead of 0.5 until a bench line looked wrong)."""
    import subprocess, sys
    out = tmp_path / "kernel_mads.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "count_mads.py")], env=dict(os.environ, COUNT_MADS_OUT=str(out)), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    new = json.load(open(out))["kernels"]
    rec = json.load(open(os.path.join(ROOT, "profiles", "kernel_mads.json")))["kernels"]
    assert set(new) == set(rec), sorted(set(new) ^ set(rec))
    for k, e in new.items():
        assert len(e.get("unmodelled") or []) == len(rec[k].get("unmodelled") or []), (k, e.get("unmodelled"))      # a loop the model does not know shows up here first
        assert abs(e["mads_per_proof_launch"] - rec[k]["mads_per_proof_launch"]) <= 1e-6 * max(1.0, rec[k]["mads_per_proof_launch"]), (k, e["mads_per_proof_launch"], rec[k]["mads_per_proof_launch"])
    for k, lo, hi in (("k_miller_run", 2.4e6, 2.7e6), ("k_g16_msm_partial_comb", 3.1e7, 3.6e7), ("k_coop12_miller_g16", 5.3e6, 5.9e6), ("k_coop12_miller_fixed", 4.2e6, 4.8e6),
                      ("k_f12_mul", 7.5e3, 9.0e3), ("k_g1_msm_rows", 2.6e5, 3.0e5), ("k_miller_run_fixed2", 1.3e6, 1.6e6), ("k_g16_prepare", 0.6e5, 1.2e5)):
        assert lo <= new[k]["mads_per_proof_launch"] <= hi, (k, new[k]["mads_per_proof_launch"])


def test_plonk_plan(pkg):
    """The PlonK batch plan (sub-batches side by side, balanced passes of at most `piece` proofs) covers the batch exactly.  (The sizing of the window-table scratch
    against every launch a context can see -- the round-3 heap overflow -- is tests/test_msm_rows.py::test_plonk_context_scratch_holds_every_launch.)"""
    import ctypes as C
    L = pkg.lib()
    L.bn254_dbg_plonk_plan.argtypes = [C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    for n in list(range(1, 300)) + [4096, 5039, 5040, 5041, 6144, 10080, 10081, 20000, 40320, 40321, 65536, 100000, 131072, 262144, 1000003]:
        for piece, mw in ((5040, 8), (5040, 4), (700, 3), (65536, 1)):
            w, per, ps = C.c_int(), C.c_size_t(), C.c_size_t()
            assert L.bn254_dbg_plonk_plan(n, piece, mw, C.byref(w), C.byref(per), C.byref(ps)) == 0
            w, per, ps = w.value, per.value, ps.value
            assert 1 <= w <= mw and 1 <= ps <= piece and ps <= per
            assert w * per >= n > (w - 1) * per - per          # contiguous sub-batches of `per` proofs cover [0, n), the last one may be short (or empty for tiny n)
            assert w == 1 or n > piece                         # one sub-batch while a single pass holds the batch
            passes = -(-per // ps)
            assert passes * ps >= per and (passes - 1) * ps < per and passes == -(-per // piece)   # no more passes than the piece size forces, all of one size


def _g16_plan(L, key_inputs, comb, reserved, n, n_public, n_streams, single):
    import ctypes as C
    L.bn254_dbg_g16_plan.argtypes = [C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int, C.POINTER(C.c_int)]
    alloc = (C.c_uint64 * 4)(); out = (C.c_uint64 * (8 * 64))(); k = C.c_int()
    assert L.bn254_dbg_g16_plan(key_inputs, comb, reserved, n, n_public, n_streams, single, alloc, out, 64, C.byref(k)) == 0
    assert k.value <= 64
    rows = [[int(out[8 * i + j]) for j in range(8)] for i in range(k.value)]
    for r in rows:
        if r[3] >= 1 << 63:
            r[3] -= 1 << 64
    return [int(x) for x in alloc], rows


def test_groth16_plan_fits_every_reservation(pkg):
    """no Groth16 launch can outrun its buffers.  The library allocates (ensure_dev) and enqueues (g16_enqueue_exact, bn254_launch_g16) by the
    pure functions of csrc/bn254_g16_plan.h; bn254_dbg_g16_plan exposes them.  For keys with 2 / 16 / 17 / 40 / 1024 inputs, reservations from 1 proof to beyond one
    workspace chunk, batches at and below the reservation (a context an earlier, larger call has grown), matching and wrong input counts, 1 / 2 / 4 sub-batch streams
    and the serialised-streams fallback: the launches cover the batch exactly once; every launch addresses workspace inside the allocation and at most
    G16_MAX_LAUNCH proofs (32-bit buffer offsets); launches that run side by side use disjoint workspace; a wide-MSM launch fits the partial-sum and digit buffers;
    the cooperative form only takes a whole small batch."""
    import random
    L = pkg.lib()
    rng = random.Random(77)
    WS, MAXL, MAXB, COOP, WIDE = 4680, 786432, 1 << 20, 30720, 65536
    sizes = [1, 2, 255, 256, 257, 4096, 16384, 16385, 30720, 30721, 65536, 65537, 65552, 131072, 262144, 524288, 786432, 786433, (1 << 20) - 1, 1 << 20, (1 << 20) + 777, 2500000]
    for key_inputs, comb in ((2, 0), (16, 0), (17, 1), (17, 0), (40, 1), (1024, 1)):
        for reserved in sizes + [rng.randrange(1, 1 << 21) for _ in range(6)]:
            for n in {reserved, max(1, reserved // 2), max(1, reserved - 1), 1, min(reserved, 70000), min(reserved, 65537)}:
                if key_inputs == 1024 and n > 200000:
                    continue
                for n_public in (key_inputs, key_inputs - 1):
                    for n_streams, single in ((2, 0), (1, 0), (4, 0), (2, 1)):
                        alloc, rows = _g16_plan(L, key_inputs, comb, reserved, n, n_public, n_streams, single)
                        assert alloc[0] == min((reserved + 255) // 256 * 256, MAXB) * WS
                        covered = 0
                        by_chunk = {}
                        for chunk, first, count, slot, form, steps, lo, hi in rows:
                            assert count >= 1 and count <= MAXL and hi <= alloc[0] and lo == first * WS and hi == (first + count) * WS, (key_inputs, reserved, n, rows)
                            assert hi - lo < 0xfffffffc      # a launch addresses its own part of the workspace with 32-bit buffer offsets
                            by_chunk.setdefault(chunk, []).append((first, count, slot, form))
                            covered += count
                            wide = n_public == key_inputs and key_inputs > 16
                            if wide:
                                assert slot == -1 and count <= alloc[3] and count <= WIDE
                                chunks = (key_inputs + 15) // 16
                                assert chunks * 27 * 4 * count <= alloc[1] and (not comb or 20 * key_inputs * 2 * count <= alloc[2])
                            if form == 1:
                                assert count <= COOP and (wide or n_public <= 16)
                            if form == 2:
                                assert count <= 16384
                            assert steps in (11, 22, 44, 88)
                        assert covered == n
                        for chunk, parts in by_chunk.items():
                            parts.sort()
                            pos = 0
                            for first, count, slot, form in parts:
                                assert first == pos            # contiguous, disjoint: parts that run side by side never share workspace
                                pos += count
                            assert pos == min(MAXB, n - chunk * MAXB)
                            assert all(f == 0 for _, _, _, f in parts) or len(parts) == 1      # the cooperative form and the latency mode only take a whole chunk
                            slots = [s for _, _, s, _ in parts]
                            if len(parts) > 1 and slots[0] >= 0:
                                assert len(set(slots)) == min(len(parts), 4) and single == 0 and n_streams > 1


def test_groth16_rlc_group_buffer_fits(pkg):
    """The same for BN254_FLAG_RLC's group status bytes: what the launch parts of a chunk address against what rlc_ensure allocated for the reservation, over
    group sizes 2 .. 2^10, sharing 1 .. 8 proofs per lane, 1 .. 4 streams."""
    import ctypes as C
    import random
    L = pkg.lib()
    L.bn254_dbg_g16_rlc_plan.argtypes = [C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    rng = random.Random(78)
    need, alloc = C.c_uint64(), C.c_uint64()
    for m in [64, 65, 255, 256, 257, 4096, 32767, 32768, 65536, 65537, 200000, 262144, 786432, 786433, (1 << 20) - 1, 1 << 20] + [rng.randrange(64, 1 << 20) for _ in range(30)]:
        for n_streams in (1, 2, 3, 4):
            for lg in (1, 2, 5, 10):
                for ls in (0, 1, 3):
                    for min_lanes in (1, 65536):
                        assert L.bn254_dbg_g16_rlc_plan(m, m, n_streams, lg, ls, min_lanes, C.byref(need), C.byref(alloc)) == 0
                        assert need.value <= alloc.value, (m, n_streams, lg, ls, min_lanes, need.value, alloc.value)
