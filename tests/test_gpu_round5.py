"""GPU tests of the round-5 additions, all through the C ABI: PlonK passes above 65 536 proofs (up to 2^18 per pass) on two key shapes, the device-resident and the
multi-device PlonK entries, bn254_plonk_reserve / bn254_plonk_footprint, the known-answer self-test the library runs on a key's first use of a device."""
import ctypes as C
import os
import random

import pytest

pytestmark = pytest.mark.gpu


def be(v):
    return int(v).to_bytes(32, "big")


def _plonk_cases(O, fixtures, rng, per_fixture):
    from test_gpu_parity import _plonk_cases as f
    return f(O, fixtures, rng, per_fixture)


def _expected(O, vk, cases, npub=2):
    return bytes(O.plonk_verify(c[0], vk, [int.from_bytes(c[1][32 * k:32 * k + 32], "big") for k in range(npub)]) for c in cases)


def _tile(cases, exp, n, stride=904):
    k = len(cases)
    reps, tail = divmod(n, k)
    pb, ib = b"".join(c[0].ljust(stride, b"\0") for c in cases), b"".join(c[1] for c in cases)
    isz = len(cases[0][1])
    return pb * reps + pb[:stride * tail], ib * reps + ib[:isz * tail], exp * reps + exp[:tail]


def test_plonk_passes_above_65536_on_two_key_shapes(pkg, O, fixtures):
    """160 000 proofs through passes of MORE than 65 536 proofs (round 4 capped a pass there: one wavefront per SIMD for the pairing stage): the default plan (two passes of
    80 000 on two contexts), ONE pass of 160 000, and passes of 65 536 as before -- the SP1 key (statuses: the oracle's on the fixtures and their mutations, every class
    present) and a key reshaped to two commitments and three public inputs (every proof fails, with the oracle's status).  The exact path and BN254_FLAG_RLC."""
    from test_gpu_round4 import _reshape_plonk_key, _reshape_plonk_proof
    fx, vk = fixtures
    rng = random.Random(51)
    cases, _ = _plonk_cases(O, fixtures, rng, 6)
    exp = _expected(O, vk, cases)
    assert len(set(exp)) >= 4 and exp.count(bytes([pkg.ACCEPT])) >= 4
    n = 160000
    try:
        pvk = pkg.PreparedPlonkVk(vk)
        p, q, want = _tile(cases, exp, n)
        for plan in (dict(big_from=0), dict(piece=5040, workers=1, big_from=1, big_piece=262144), dict(piece=5040, workers=4, big_from=1, big_piece=65536)):
            pkg.set_plonk_params(**plan)
            assert pvk.verify_batch(p, q, n) == want, plan
        pkg.set_plonk_params(piece=5040, workers=8, big_from=0, big_piece=131072)
        assert pvk.verify_batch(p, q, n, flags=pkg.FLAG_RLC) == want
        bytes_held, ctxs = pvk.footprint()
        assert ctxs >= 2 and bytes_held > 2 * 160000 * 6000
        pvk.close()
        # another key shape: two commitments, three public inputs
        key = _reshape_plonk_key(vk, 2, 3)
        base = [(bytes.fromhex(f["raw_proof"]), [int(x) for x in f["public_inputs"]]) for f in fx.values() if f["variant"] == "plonk"]
        c2 = []
        for proof, pis in base:
            for pq in (2, 1, 0):
                c2.append((_reshape_plonk_proof(proof, pq), b"".join(be(x) for x in (pis + [7])[:3])))
        want2 = _expected(O, key, c2, 3)
        assert pkg.ACCEPT not in want2 and len(set(want2)) >= 2
        stride = max(len(c[0]) for c in c2)
        pvk2 = pkg.PreparedPlonkVk(key)
        p, q, want = _tile(c2, want2, n, stride)
        for plan in (dict(big_from=0), dict(piece=5040, workers=1, big_from=1, big_piece=262144)):
            pkg.set_plonk_params(**plan)
            assert pvk2.verify_batch(p, q, n, proof_stride=stride, n_public=3) == want, plan
        pvk2.close()
    finally:
        pkg.set_plonk_params(piece=5040, workers=8, big_from=0, big_piece=131072)


def test_plonk_device_resident_entry(pkg, O, fixtures):
    """bn254_plonk_verify_batch_device: proofs, inputs and status bytes in device memory (torch tensors here, raw pointers across the ABI) -- chains of small passes, one
    pass, passes above 65 536 -- against the host-buffer entry and the oracle's statuses; an odd stride; BN254_FLAG_RLC; the inputs still being written by the caller's
    stream when the call is made (the entry waits for that stream)."""
    import torch
    rng = random.Random(52)
    cases, vk = _plonk_cases(O, fixtures, rng, 6)
    exp = _expected(O, vk, cases)
    pvk = pkg.PreparedPlonkVk(vk)
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream(device=dev)
    if os.environ.get("BN254_PLONK_HOST") == "1":
        # the diagnostic build of the stages on host threads reads the proofs on the host: the resident entry refuses it, by name (tools/gpu_variants.sh runs the suite this way)
        d = torch.zeros(904 + 64 + 1, dtype=torch.uint8, device=dev)
        with pytest.raises(Exception, match="BN254_PLONK_HOST"):
            pvk.verify_batch_device(d.data_ptr(), d.data_ptr() + 904, d.data_ptr() + 968, 1, proof_stride=904, stream=side.cuda_stream)
        return
    for n, stride, flags in ((1, 904, 0), (300, 904, 0), (7000, 904, 0), (30000, 905, 0), (70000, 904, 0), (140000, 904, pkg.FLAG_RLC)):
        p, q, want = _tile(cases, exp, n, stride)
        hp = torch.frombuffer(bytearray(p), dtype=torch.uint8).pin_memory(); hq = torch.frombuffer(bytearray(q), dtype=torch.uint8).pin_memory()
        d_status = torch.full((n,), 0xEE, dtype=torch.uint8, device=dev)
        with torch.cuda.stream(side):
            d_p = hp.to(dev, non_blocking=True); d_q = hq.to(dev, non_blocking=True)          # still in flight on `side` when the entry is called
            pvk.verify_batch_device(d_p.data_ptr(), d_q.data_ptr(), d_status.data_ptr(), n, proof_stride=stride, stream=side.cuda_stream, flags=flags)
        got = bytes(d_status.cpu().numpy().tobytes())                                           # the entry is host-synchronous: no further wait needed
        assert got == want, (n, stride, [(i, got[i], want[i]) for i in range(n) if got[i] != want[i]][:6])
        if n <= 7000:
            assert pvk.verify_batch(p, q, n, proof_stride=stride) == want
    pvk.close()


def test_plonk_multi_entry_and_reserve(pkg, O, fixtures):
    """bn254_plonk_verify_batch_multi on every visible device in turn and on all at once (the driver's box has one: the one-shard path; tests/hostsan runs the threaded
    branch under a fake eight-device runtime), and bn254_plonk_reserve: after it, a batch of the reserved size leaves the footprint where it was."""
    import torch
    rng = random.Random(53)
    cases, vk = _plonk_cases(O, fixtures, rng, 5)
    exp = _expected(O, vk, cases)
    n = 9000
    p, q, want = _tile(cases, exp, n)
    pvk = pkg.PreparedPlonkVk(vk)
    assert pvk.footprint() == (0, 0)
    pvk.reserve(n, proof_stride=904)
    held = pvk.footprint()
    assert held[0] > 0 and held[1] >= 1
    cnt = torch.cuda.device_count()
    for d in range(cnt):
        assert pvk.verify_batch_multi(p, q, 1 << d) == want, d
    assert pvk.verify_batch_multi(p, q, (1 << cnt) - 1, flags=pkg.FLAG_RLC) == want
    assert pvk.verify_batch(p, q, n) == want
    assert pvk.footprint() == held                       # nothing grew: the reservation covered the batch
    with pytest.raises(pkg.Bn254Error):
        pvk.verify_batch_multi(p, q, 1 << cnt)           # a device that does not exist
    pvk.close()


def test_plonk_self_test_guards_the_device_stages(pkg, fixtures):
    """The library checks k_plonk_stage1 on the device against the host's compile of the same source before a key is first used there (a known-answer test on a synthetic
    proof made of the key's own points).  It passes on the shipped build for the SP1 key and for reshaped keys (0 and 2 commitments; 1 and 3 public inputs) -- a failure
    would surface as BN254_E_HIP from the first batch, with the differing value in bn254_last_error()."""
    from test_gpu_round4 import _reshape_plonk_key
    fx, vk = fixtures
    f = next(v for v in fx.values() if v["variant"] == "plonk")
    proof = bytes.fromhex(f["raw_proof"])
    for n_qcp, nb_public in ((1, None), (0, None), (2, 3), (0, 1)):
        key = vk if (n_qcp, nb_public) == (1, None) else _reshape_plonk_key(vk, n_qcp, nb_public)
        npub = 2 if nb_public is None else nb_public
        pvk = pkg.PreparedPlonkVk(key)
        st = pvk.verify_batch(proof, bytes(32 * npub), 1, n_public=npub)          # the first use of the key on the device runs the self-test
        assert len(st) == 1 and st[0] in (pkg.ACCEPT, 5, 6, 7, 9), st
        pvk.close()
    assert pkg.lib().bn254_abi_version() == 5


def test_tables_built_on_device_match_the_host_construction(pkg, fixtures):
    """The fixed-base tables of a key -- byte windows (32 x 255 entries per point) for keys with up to 16 public inputs and for the points of a PlonK key, comb tables (8192
    entries per point) above 16 inputs -- are built by the device that uses them (csrc/bn254_k_comb.hip) from the key's points; the host keeps no copy.  Entry by entry, as field
    values, they must be what bn254_host.hpp::build_window_table / build_comb_table make: a 2-input and a 16-input key, a 17-input key (the smallest comb key), a 300-input
    key (two construction passes), the reference's PlonK key (all its table points).  The statuses of batches on such keys against the oracle are the rest of the suite."""
    if os.environ.get("BN254_TABLES_HOST", "0") != "0" or os.environ.get("BN254_COMB_HOST", "0") != "0":
        pytest.skip("the host construction is in use (tools/gpu_variants.sh runs the suite this way): nothing was built on the device")
    L = pkg.lib()
    L.bn254_dbg_comb_table_compare.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_size_t)]
    L.bn254_dbg_plonk_table_compare.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]
    for n_public in (2, 16, 17, 300):
        vk = pkg.synth_groth16(0xC0B0 + n_public, n_public, 1, invalid_every=0, agree=True, threads=4)[0]
        pvk = pkg.PreparedVk(vk)
        bad = C.c_size_t(12345)
        assert L.bn254_dbg_comb_table_compare(pvk._h, 0, n_public, C.byref(bad)) == 0, L.bn254_last_error()
        assert bad.value == 0, (n_public, bad.value)
    ppvk = pkg.PreparedPlonkVk(fixtures[1])
    bad = C.c_size_t(12345)
    assert L.bn254_dbg_plonk_table_compare(ppvk._h, 0, C.byref(bad)) == 0, L.bn254_last_error()
    assert bad.value == 0, bad.value


def test_bench_plonk_mode_prints_the_contract_line():
    """`python bench.py --plonk` (one PlonK call sharded like the Groth16 batch; here 2^12 proofs on one GPU): ONE JSON line with the fields of the bench contract, the PlonK
    metric name, a roofline and a CPU baseline; its own checks (every status byte against the workload, the first 16 against the oracle, the gathered vector) have passed
    when it prints."""
    import json, subprocess, sys
    if os.environ.get("BN254_PLONK_HOST") == "1":
        pytest.skip("the diagnostic host-thread stages refuse the device-resident entry the bench times")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--plonk", "--batch-log2", "12", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].startswith("PlonK verifies/sec") and d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 1e5 and d["config"]["global_batch"] == 4096
    assert d["roofline"]["bound"] == "valu" and 0 < d["roofline"]["frac"] < 1 and d["cpu_baseline"]["kind"] == "port"
