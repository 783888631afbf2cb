"""GPU tests of the round-4 additions, all through the C ABI: the shard sizes of the 2- and 8-GPU runs with oracle samples (44 / 22 Miller steps per launch), the
multi-device entry on every visible device, PlonK batch plans against each other (chains of passes, one large pass, the lane kernels of the two-pair pairing check),
PlonK key shapes other than the SP1 circuit's, a proof stride that needs 115 KB of LDS per workgroup, the stream-overlap report."""
import ctypes as C
import os
import random

import pytest

pytestmark = pytest.mark.gpu


def be(v):
    return int(v).to_bytes(32, "big")


@pytest.fixture(scope="module")
def L(pkg):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU; the product has no CPU fallback"
    return pkg.lib()


def _oracle_sample(O, vk, proofs, inputs, n_public, idx):
    sz = 32 * n_public
    sp = b"".join(proofs[256 * j:256 * j + 256] for j in idx); si = b"".join(inputs[sz * j:sz * j + sz] for j in idx)
    return O.groth16_verify_many(sp, 256, vk, si, n_public, len(idx), O.MODE_REFERENCE)


@pytest.mark.parametrize("log2n", [19, 17])
def test_shard_sized_batches_with_oracle_samples(pkg, O, L, log2n):
    """The per-GPU shards of BASELINE configs[2] as whole batches: 2^19 (the 2-GPU shard: two sub-batches of 2^18, k_miller_run with 44 steps per launch -- a form
    no other test takes) and 2^17 (the 8-GPU shard: two sub-batches of 2^16, 11 steps per launch); the generator's statuses everywhere, the oracle on 64 proofs
    spread over the batch with all five failure classes among them."""
    n = 1 << log2n
    vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540100 + log2n, 2, n, invalid_every=16, agree=True, threads=16)
    pvk = pkg.PreparedVk(vk)
    st = pvk.verify_batch(proofs, inputs)
    assert st == exp and st.count(bytes([pkg.ACCEPT])) == n - n // 16
    idx = sorted(set(list(range(0, n, n // 44)) + [16 * q + 15 for q in range(0, n // 16, n // 16 // 20)] + [n - 1, n // 2 - 1, n // 2]))[:64]
    assert _oracle_sample(O, vk, proofs, inputs, 2, idx) == bytes(st[j] for j in idx)
    assert len(set(st[j] for j in idx)) == 5
    pvk.close()


def test_multi_device_entry_on_each_visible_device(pkg, O, L):
    """bn254_groth16_verify_batch_multi with a one-bit mask for EVERY visible device (on the driver's one-GPU box: device 0; on a node: each of its GPUs in turn), then
    with all of them at once: the statuses of the single-device entry."""
    import torch
    n = 3000
    vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540110, 2, n, invalid_every=8, agree=True, threads=8)
    pvk = pkg.PreparedVk(vk)
    cnt = torch.cuda.device_count()
    assert cnt >= 1
    for d in range(cnt):
        assert pvk.verify_batch_multi(proofs, inputs, 1 << d) == exp, d
    assert pvk.verify_batch_multi(proofs, inputs, (1 << cnt) - 1) == exp
    assert [x[1:] for x in pkg.shard_plan(n, (1 << cnt) - 1, cnt)][0][0] == 0
    pvk.close()


def _plonk_cases(O, fixtures, rng, per_fixture):
    from test_gpu_parity import _plonk_cases as f
    return f(O, fixtures, rng, per_fixture)


def test_plonk_plans_agree_and_the_lane_pairing_path(pkg, O, fixtures):
    """The PlonK batch plans against each other through bn254_set_plonk_params: eight chains of 5040-proof passes, the measured default, ONE pass of the whole batch --
    at 41 472 and 53 000 proofs that pass is above the cooperative kernel's range, so the pairing check runs on the lane kernels (k_miller_run_fixed2: the whole two-pair Miller
    loop in one launch, then the final-exponentiation program) and the MSM launches take their unsplit row form.  Same status bytes, and those are the oracle's."""
    rng = random.Random(41)
    cases, vk = _plonk_cases(O, fixtures, rng, 5)
    exp = bytes(O.plonk_verify(c[0], vk, [int.from_bytes(c[1][:32], "big"), int.from_bytes(c[1][32:], "big")]) for c in cases)
    assert len(set(exp)) >= 3
    pb, ib = b"".join(c[0] for c in cases), b"".join(c[1] for c in cases)
    k = len(cases)
    pvk = pkg.PreparedPlonkVk(vk)
    try:
        for n in (12000, 41472, 53000):     # 53 000 in one pass: the MSM launches walk the variable terms of a sum in JOINT rows
            reps, tail = divmod(n, k)
            p, q, want = pb * reps + pb[:904 * tail], ib * reps + ib[:64 * tail], exp * reps + exp[:tail]
            for plan in (dict(piece=5040, workers=8, big_from=1 << 30), dict(big_from=0), dict(piece=5040, workers=1, big_from=1, big_piece=65536),
                         dict(piece=5040, workers=2, big_from=1, big_piece=16384)):
                pkg.set_plonk_params(**plan)
                assert pvk.verify_batch(p, q, n) == want, (n, plan)
    finally:
        pkg.set_plonk_params(piece=5040, workers=8, big_from=0, big_piece=131072)
        pvk.close()


def test_plonk_rlc_flag_gives_the_exact_statuses(pkg, O, fixtures):
    """BN254_FLAG_RLC on the PlonK entry: the pairing checks of a pass batched over the 64 proofs of a wavefront (weights folded into the MSM scalars, one check per
    group, the exact check on the groups that fail).  Statuses must equal the exact path's and the oracle's: all-valid batches (no fallback), batches with every
    status class incl. proofs that fail ONLY the pairing check (tests/kzg_forgery.py: valid points, wrong opening) sprinkled so that some groups fail and some pass,
    sizes on both sides of the threshold below which the flag is ignored, a pass with joint MSM rows (53 000), several passes in flight (150 000)."""
    from kzg_forgery import forge, KNOWN_LAMBDA
    fx, vk = fixtures
    rng = random.Random(47)
    cases, _ = _plonk_cases(O, fixtures, rng, 4)
    f = fx["fibonacci_plonk"]
    proof = bytes.fromhex(f["raw_proof"]); pis = [int(x) for x in f["public_inputs"]]
    tampered, forged = forge(O, proof, vk, pis, KNOWN_LAMBDA)
    ib0 = b"".join(be(x) for x in pis)
    valid = [c for c in cases if O.plonk_verify(c[0], vk, [int.from_bytes(c[1][:32], "big"), int.from_bytes(c[1][32:], "big")]) == pkg.ACCEPT]
    assert len(valid) >= 4
    mixed = cases + [(forged, ib0), (tampered, ib0)]
    exp_of = {id(c): O.plonk_verify(c[0], vk, [int.from_bytes(c[1][:32], "big"), int.from_bytes(c[1][32:], "big")]) for c in mixed}
    # the two proofs of tests/kzg_forgery.py fail the pairing check under every batching scalar but the one they were built for (the oracle's default: there the
    # forged one verifies -- tests/test_gpu_round2.py::test_plonk_rejects_the_known_lambda_forgery has the story); the product draws its own
    exp_of[id(mixed[-1])] = 8; exp_of[id(mixed[-2])] = 8
    assert len(set(exp_of.values())) >= 4
    pvk = pkg.PreparedPlonkVk(vk)
    try:
        for n in (8191, 8192, 12345, 53000, 150000):
            # (a) all valid
            sel = [valid[i % len(valid)] for i in range(n)]
            p, q = b"".join(c[0] for c in sel), b"".join(c[1] for c in sel)
            assert pvk.verify_batch(p, q, n, flags=pkg.FLAG_RLC) == bytes([pkg.ACCEPT]) * n, n
            # (b) mostly valid, every class somewhere, pairing-only failures in a few groups (and two of them in one group)
            sel = list(sel)
            for k in range(0, n, 997):
                sel[k] = mixed[(k // 997) % len(mixed)]
            sel[5] = mixed[-1]; sel[6] = mixed[-2]; sel[n - 1] = mixed[-1]
            early = [c for c in mixed if exp_of[id(c)] not in (pkg.ACCEPT, 8)]            # decided before the pairing check
            for k in range(128, 192):                                                    # a whole group (one wavefront) without a pending proof
                sel[k] = early[k % len(early)]
            for k in range(256, 320):                                                    # ... and a group whose every proof fails only the pairing check
                sel[k] = mixed[-1 - (k & 1)]
            p, q = b"".join(c[0] for c in sel), b"".join(c[1] for c in sel)
            want = bytes(exp_of.get(id(c), pkg.ACCEPT) for c in sel)
            assert want.count(bytes([8])) >= 3
            got = pvk.verify_batch(p, q, n, flags=pkg.FLAG_RLC)
            assert got == want, (n, [(i, got[i], want[i]) for i in range(n) if got[i] != want[i]][:8])
            assert pvk.verify_batch(p, q, n) == want, n
    finally:
        pvk.close()


def _reshape_plonk_key(vk, n_qcp, nb_public=None):
    """The SP1 key with another number of BSB22 commitments (the commitment point and its constraint index dropped or repeated) and, optionally, another public-input
    count: layout of plonk/converter.rs:18-119."""
    vk = bytearray(vk)
    old = int.from_bytes(vk[368:372], "big")
    assert old == 1
    qcp = bytes(vk[372:404])
    rest = bytes(vk[404:404 + 160 + 33788])
    cci_off = 404 + 160 + 33788
    assert int.from_bytes(vk[cci_off:cci_off + 8], "big") == 1
    cci = bytes(vk[cci_off + 8:cci_off + 16])
    head = bytearray(vk[:368])
    if nb_public is not None:
        head[72:80] = int(nb_public).to_bytes(8, "big")
    out = bytes(head) + int(n_qcp).to_bytes(4, "big") + qcp * n_qcp + rest + int(n_qcp).to_bytes(8, "big")
    for i in range(n_qcp):
        out += (int.from_bytes(cci, "big") + 3 * i).to_bytes(8, "big")
    return out


def _reshape_plonk_proof(proof, n_qcp):
    """The fixture's proof with n_qcp commitments and 6 + n_qcp claimed values (plonk/converter.rs:121-178)."""
    assert int.from_bytes(proof[512:516], "big") == 7
    claimed = proof[516:516 + 32 * 7]
    tail_off = 516 + 32 * 7
    zs = proof[tail_off:tail_off + 96]
    assert int.from_bytes(proof[tail_off + 96:tail_off + 100], "big") == 1
    bsb = proof[tail_off + 100:tail_off + 164]
    cl = claimed[:32 * 6] + claimed[32 * 6:] * n_qcp
    return proof[:512] + (6 + n_qcp).to_bytes(4, "big") + cl + zs + int(n_qcp).to_bytes(4, "big") + bsb * n_qcp


def test_plonk_other_key_shapes_vs_oracle(pkg, O, fixtures):
    """PlonK parity on key shapes other than the SP1 circuit's (one BSB22 commitment, two public inputs): the reference's key rewritten to 0 and to 2 commitments and to
    3 public inputs, the proofs rewritten to match or deliberately not -- no prover is needed, because every such proof must FAIL, and with exactly the oracle's status
    (Bsb22CommitmentMismatch / InvalidWitness / OpeningPolyMismatch / the loader's errors: plonk/verify.rs:46-60, plonk/converter.rs:60-119) on the device-stage path,
    alone, in a batch of 257 and in one of 5041 (two passes)."""
    fx, vk = fixtures
    base = [(bytes.fromhex(f["raw_proof"]), [int(x) for x in f["public_inputs"]]) for f in fx.values() if f["variant"] == "plonk"]
    for n_qcp, nb_public in ((0, None), (2, None), (1, 3), (2, 3), (0, 1)):
        key = _reshape_plonk_key(vk, n_qcp, nb_public)
        npub = 2 if nb_public is None else nb_public
        cases = []
        for proof, pis in base:
            ins = (pis + [7, 9])[:npub]
            for pq in sorted({n_qcp, 1, 0, 2}):
                cases.append((_reshape_plonk_proof(proof, pq), ins))
        want = bytes(O.plonk_verify(p, key, ins) for p, ins in cases)
        assert pkg.ACCEPT not in want, "a reshaped key must not accept the SP1 proofs"
        assert len(set(want)) >= 2, sorted(set(want))
        stride = max(len(p) for p, _ in cases)
        pvk = pkg.PreparedPlonkVk(key)
        assert pvk.n_public == npub
        for total in (1, 257, 5041, 49200):        # 49 200: joint rows over this key shape's terms
            reps = -(-total // len(cases))
            sel = (cases * reps)[:total]
            pb = b"".join(p.ljust(stride, b"\0") for p, _ in sel)
            ib = b"".join(b"".join(be(x) for x in ins) for _, ins in sel)
            st = pvk.verify_batch(pb, ib, total, proof_stride=stride, n_public=npub)
            assert st == (want * reps)[:total], (n_qcp, nb_public, total)
        # the wrong input count for this key: InvalidWitness after the loader errors
        p0, ins0 = cases[0]
        assert pkg.PlonkVerifier.verify(p0, key, ins0 + [5]) == O.plonk_verify(p0, key, ins0 + [5])
        pvk.close()


def test_plonk_large_proof_stride_lds(pkg, O, fixtures):
    """A caller's proof stride of 1700 bytes: the stages stage 1664 bytes of every proof in LDS (114 KB per workgroup, above the 64 KB a launch gets without asking:
    hipFuncAttributeMaxDynamicSharedMemorySize), and the lanes' SHA slots are addressed by LDS offset (bn254_plonk.hpp::pl_lane_lds, guarded by a trap if the dynamic
    LDS of the kernel ever stops starting at offset 0).  Statuses: the oracle's."""
    rng = random.Random(43)
    cases, vk = _plonk_cases(O, fixtures, rng, 3)
    exp = bytes(O.plonk_verify(c[0], vk, [int.from_bytes(c[1][:32], "big"), int.from_bytes(c[1][32:], "big")]) for c in cases)
    pvk = pkg.PreparedPlonkVk(vk)
    for stride in (1700, 1664, 905):
        pb = b"".join(c[0].ljust(stride, b"\xa5") for c in cases) * 9
        ib = b"".join(c[1] for c in cases) * 9
        assert pvk.verify_batch(pb, ib, 9 * len(cases), proof_stride=stride) == exp * 9, stride
    pvk.close()


def test_stream_overlap_is_measured_not_assumed(pkg, L):
    """The library no longer touches GPU_MAX_HW_QUEUES: the first batch that runs two sub-batch streams is bracketed with events and a later call reads the overlap
    (bn254_groth16_stream_overlap).  Here (tests/conftest.py asks for eight hardware queues) the two streams run side by side; a process whose streams share a queue
    gets overlap ~ 1, one sub-batch per launch from then on and a line in bn254_last_diagnostic."""
    import time
    n = 1 << 17
    vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540120, 2, n, invalid_every=16, agree=True, threads=16)
    pvk = pkg.PreparedVk(vk)
    L.bn254_groth16_stream_overlap.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.bn254_last_diagnostic.restype = C.c_char_p
    ov, single = C.c_float(), C.c_int()
    assert L.bn254_groth16_stream_overlap(pvk.handle, 0, C.byref(ov), C.byref(single)) == 0 and ov.value == -1.0
    for _ in range(5):      # a "one after the other" reading must repeat three times before the plan changes (each batch's events are read by the next call)
        assert pvk.verify_batch(proofs, inputs) == exp
    assert L.bn254_groth16_stream_overlap(pvk.handle, 0, C.byref(ov), C.byref(single)) == 0
    if os.environ.get("BN254_STREAMS") == "1":      # tools/gpu_variants.sh: one sub-batch per launch by configuration, nothing to measure
        assert ov.value == -1.0 and single.value == 0
        pvk.close()
        return
    assert 0.9 <= ov.value <= 2.1, ov.value
    if ov.value < 1.15:
        assert single.value == 1 and b"hardware queue" in L.bn254_last_diagnostic()
    else:
        assert single.value == 0
    pvk.close()
