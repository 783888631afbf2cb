"""The C++ host mirror of the crate's surface (include/bn254_verify.hpp: Groth16Verifier / PlonkVerifier ::verify, ::verify_batch over the C ABI),
compiled with g++ and driven as a C++ host would.  CPU: status-byte mapping, loader panics, "no device".  GPU: verdicts against the generator's
expected statuses (Groth16) and the oracle (PlonK fixtures and mutations)."""
import json
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "snark-bn254-verifier_amd")


@pytest.fixture(scope="module")
def mirror_check(pkg, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cpp") / "mirror_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "mirror_check.cpp"),
                           "-o", out, "-L", PKG, "-l:libbn254_verify_amd.so", "-Wl,-rpath," + PKG, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"])
    return out


def test_cpp_mirror_mapping_and_panics(mirror_check):
    r = subprocess.run([mirror_check, "cpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mirror_check cpu ok" in r.stdout


@pytest.mark.gpu
def test_cpp_mirror_on_device(mirror_check, O, fixtures, tmp_path):
    fx, vk = fixtures
    base = [(bytes.fromhex(f["raw_proof"]), [int(x) for x in f["public_inputs"]]) for f in fx.values() if f["variant"] == "plonk"]
    rng = random.Random(21)
    cases = []
    for i in range(12):
        p, q = base[i % len(base)]
        p = bytearray(p); q = list(q)
        if i >= 4:
            kind = i % 4
            if kind == 0: q[rng.randrange(len(q))] ^= 1 << rng.randrange(200)                 # a wrong public input
            elif kind == 1: p[516 + rng.randrange(32 * 6)] ^= 1 << rng.randrange(8)            # a claimed value
            elif kind == 2: p[rng.randrange(512)] ^= 1 << rng.randrange(8)                      # a commitment coordinate
            else: p[516 + 32 * 6 + 64 + rng.randrange(32)] ^= 1 << rng.randrange(8)              # the claimed value at zeta * omega
        cases.append((bytes(p), q))
    open(tmp_path / "plonk_vk.bin", "wb").write(vk)
    for i, (p, q) in enumerate(cases):
        open(tmp_path / ("proof_%d.bin" % i), "wb").write(p)
        open(tmp_path / ("inputs_%d.bin" % i), "wb").write(b"".join(int(x).to_bytes(32, "big") for x in q))
    r = subprocess.run([mirror_check, "gpu", str(tmp_path)], capture_output=True, text=True, timeout=600, env=dict(os.environ, BN254_RLC_MIN_BATCH="64"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mirror_check gpu ok" in r.stdout
    lines = {l.split()[0]: l.split()[1:] for l in r.stdout.splitlines() if l.startswith("plonk_batch")}
    got = [int(x) for x in lines["plonk_batch"]]
    want = [O.plonk_verify(p, vk, q) for p, q in cases]
    assert got == want
    assert want[:4] == [O.ACCEPT] * 4 and len(set(want)) >= 3


def _build_gather_check(tmp_path_factory, mock):
    out = str(tmp_path_factory.mktemp("cppg") / ("gather_check_mock" if mock else "gather_check_rccl"))
    cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "gather_check.cpp"), "-o", out,
           "-L", PKG, "-l:libbn254_verify_amd.so", "-Wl,-rpath," + PKG]
    cmd += ["-DGATHER_MOCK", "-rdynamic"] if mock else ["-lrccl"]
    subprocess.check_call(cmd)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("mock", [True, False])
def test_status_all_gather_cpp_host(pkg, tmp_path_factory, mock):
    """bn254_status_all_gather, the gather of a multi-process job for a C / Rust host: with an in-process stand-in for ncclAllGather that plays 2 ... 8
    ranks (equal and ragged shards, first and last rank), and with a real one-rank RCCL communicator."""
    exe = _build_gather_check(tmp_path_factory, mock)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "gather_check ok" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_status_all_gather_rejects_bad_arguments(pkg):
    import ctypes as C
    L = pkg.lib()
    L.bn254_status_all_gather.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    assert L.bn254_status_all_gather(None, 2, 0, None, 10, None, None, None) == -1          # no communicator
    assert L.bn254_status_all_gather(C.c_void_p(1), 2, 5, C.c_void_p(1), 10, C.c_void_p(1), None, None) == -1   # rank outside the world
    assert L.bn254_status_all_gather(C.c_void_p(1), 2, 0, None, 0, None, None, None) == 0   # nothing to gather
