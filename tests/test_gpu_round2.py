"""GPU tests of the round-2 additions, all through the C ABI: random-linear-combination batch mode (statuses identical to the exact
path), strict scalars, concurrent callers sharing one prepared key, the multi-device entry, the size just above 65 536."""
import ctypes as C
import os
import threading

import pytest

pytestmark = pytest.mark.gpu
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def be(v):
    return int(v).to_bytes(32, "big")


@pytest.fixture(scope="module")
def L(pkg):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU; the product has no CPU fallback"
    return pkg.lib()


@pytest.fixture(autouse=True)
def _rlc_always(pkg):
    """The RLC tests want the RLC kernels to run on every call: switch the adaptive bypass off (test_rlc_adaptive_bypass switches it on) and
    honour the flag from 64 proofs (by default from 200 000, where the mode pays).  bn254_set_rlc_params, not the environment: the library
    reads its environment once, when it is loaded.  share_min_lanes is LEFT ALONE (-1): tools/gpu_variants.sh runs this suite with
    BN254_RLC_SHARE_MIN_LANES=1 to push shared Miller-loop accumulators through every test, and a fixture that reset it would undo that;
    the tests that need a particular value set and restore it themselves."""
    pkg.set_rlc_params(min_batch=64, adaptive=0, share_min_lanes=-1)
    yield
    pkg.set_rlc_params(min_batch=200000, adaptive=1, share_min_lanes=-1)


@pytest.fixture(scope="module")
def wl(pkg):
    """4133 synthetic proofs (not a multiple of anything), every 8th invalid, cycling through the 5 failure classes."""
    return pkg.synth_groth16(0xB2540011, 2, 4133, invalid_every=8, agree=True, threads=16)


def test_rlc_statuses_identical_to_exact(pkg, O, wl, L):
    """BN254_FLAG_RLC on the 5-class workload: every status byte equals the exact path's, which equals the generator's prediction
    and, on a prefix, the oracle's verdicts.  Groups that contain a REJECT proof go through the exact fallback."""
    vk, proofs, inputs, exp = wl
    n = len(exp)
    for mode, omode in ((pkg.VK_REFERENCE, O.MODE_REFERENCE), (pkg.VK_GNARK, O.MODE_GNARK)):
        pvk = pkg.PreparedVk(vk, mode)
        exact = pvk.verify_batch(proofs, inputs)
        assert exact == exp
        rlc = pvk.verify_batch(proofs, inputs, flags=pkg.FLAG_RLC)
        assert rlc == exact
        m = 64
        assert O.groth16_verify_many(proofs[:256 * m], 256, vk, inputs[:64 * m], 2, m, omode) == rlc[:m]
        # a second call draws fresh weights: same answer
        assert pvk.verify_batch(proofs, inputs, flags=pkg.FLAG_RLC) == exact
        pvk.close()


def test_rlc_adaptive_bypass(pkg, wl, L):
    """With most groups failing (every 8th proof invalid, groups of 32) the flag stops paying: after the first RLC pass has measured the
    fallback share the next calls run the exact path (same status bytes), with a measuring RLC pass every 8 calls; a mostly valid
    workload on the same key brings the mode back."""
    pkg.set_rlc_params(adaptive=1)
    vk, proofs, inputs, exp = wl
    pvk = pkg.PreparedVk(vk)
    assert pvk.rlc_state() == (-1.0, 0)
    for i in range(10):
        assert pvk.verify_batch(proofs, inputs, flags=pkg.FLAG_RLC) == exp
        share, bypassed = pvk.rlc_state()
        assert share > 0.45          # 0.99 with groups of 32, 0.63 with groups of 8 (BN254_RLC_GROUP_LOG2): above the bypass threshold either way
        assert bypassed == (0, 1, 2, 3, 4, 5, 6, 7, 7, 8)[i], (i, bypassed)     # call 0 and call 8 are RLC passes
    # valid proofs only: the measuring passes pull the share down and the mode stays on
    n = len(exp)
    keep = [i for i in range(n) if exp[i] == 1][:2048]
    vp = b"".join(proofs[256 * i:256 * i + 256] for i in keep); vi = b"".join(inputs[64 * i:64 * i + 64] for i in keep)
    seen = pvk.rlc_state()[1]
    for i in range(24):
        assert pvk.verify_batch(vp, vi, len(keep), flags=pkg.FLAG_RLC) == b"\x01" * len(keep)
    share, bypassed = pvk.rlc_state()
    assert share < 0.45 and bypassed - seen <= 14, (share, bypassed - seen)
    pvk.close()


def test_rlc_shared_accumulator_layout(pkg, O, wl, L):
    """The Miller loop of the RLC mode with 2 and 4 proofs per lane (one squaring of f per lane and step): forced on a small batch
    (by default it is used from 2^19 proofs per launch), same status bytes as the exact path."""
    vk, proofs, inputs, exp = wl
    pvk = pkg.PreparedVk(vk)
    pkg.set_rlc_params(share_min_lanes=1)
    try:
        for n in (len(exp), 1000, 257, 67):
            assert pvk.verify_batch(proofs[:256 * n], inputs[:64 * n], n, flags=pkg.FLAG_RLC) == exp[:n], n
    finally:
        pkg.set_rlc_params(share_min_lanes=int(os.environ.get("BN254_RLC_SHARE_MIN_LANES", "65536")))     # what the library read when it was loaded
        pvk.close()


def test_rlc_all_valid_all_invalid_and_sizes(pkg, O, L):
    vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540012, 2, 1500, invalid_every=0, agree=True, threads=16)
    pvk = pkg.PreparedVk(vk)
    assert pvk.verify_batch(proofs, inputs, flags=pkg.FLAG_RLC) == bytes([1]) * 1500 == exp
    # every proof a REJECT (inputs shifted by one proof): every group fails, the fallback decides everything
    sh = inputs[64:] + inputs[:64]
    st = pvk.verify_batch(proofs, sh, flags=pkg.FLAG_RLC)
    assert st == bytes(1500) == pvk.verify_batch(proofs, sh)
    # one bad proof among many; sizes around the group and wave boundaries, and below RLC_MIN_BATCH (exact path)
    for n in (1, 63, 64, 65, 127, 128, 129, 255, 257, 1000):
        bad = bytearray(inputs[:64 * n]); j = n // 2
        bad[64 * j:64 * j + 32] = be(int.from_bytes(bad[64 * j:64 * j + 32], "big") ^ 1)
        want = bytes(0 if i == j else 1 for i in range(n))
        assert pvk.verify_batch(proofs[:256 * n], bytes(bad), n, flags=pkg.FLAG_RLC) == want, n
    # wrong number of public inputs under the flag: exact semantics (INPUT_LEN)
    assert pvk.verify_batch(proofs[:256 * 100], inputs[:32 * 100], 100, n_public=1, flags=pkg.FLAG_RLC) == bytes([pkg.ERR_INPUT_LEN]) * 100
    pvk.close()


def test_rlc_default_threshold_large_batch(pkg, O, L):
    """The flag at its DEFAULT threshold (no override: honoured from 200 000 proofs): 2^18 proofs with a few invalid ones take the RLC kernels
    (the fallback share is reported), and the status bytes are the exact path's, the generator's and -- on a strided sample -- the oracle's."""
    pkg.set_rlc_params(min_batch=200000, adaptive=1)
    n = 1 << 18
    vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540019, 2, n, invalid_every=1024, agree=True, threads=16)
    pvk = pkg.PreparedVk(vk)
    assert pvk.rlc_state()[0] == -1.0
    st = pvk.verify_batch(proofs, inputs, flags=pkg.FLAG_RLC)
    share = pvk.rlc_state()[0]
    assert 0.0 < share < 0.1, share             # an RLC pass ran: 256 invalid proofs send 256 groups of 32 to the fallback (3 %)
    assert st == exp == pvk.verify_batch(proofs, inputs)
    idx = list(range(0, n, 1025))[:200] + list(range(1023, n, 1024 * 37))[:8]
    sp = b"".join(proofs[256 * j:256 * j + 256] for j in idx); si = b"".join(inputs[64 * j:64 * j + 64] for j in idx)
    assert O.groth16_verify_many(sp, 256, vk, si, 2, len(idx), O.MODE_REFERENCE) == bytes(st[j] for j in idx)
    # below the threshold the flag is ignored: no new RLC pass is recorded
    pvk2 = pkg.PreparedVk(vk)
    assert pvk2.verify_batch(proofs[:256 * 4096], inputs[:64 * 4096], 4096, flags=pkg.FLAG_RLC) == exp[:4096]
    assert pvk2.rlc_state()[0] == -1.0
    pvk.close(); pvk2.close()


def test_rlc_more_public_inputs(pkg, O, L):
    for n_public in (1, 5, 8):
        vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540013 + n_public, n_public, 300, invalid_every=7, agree=True, threads=16)
        pvk = pkg.PreparedVk(vk)
        assert pvk.verify_batch(proofs, inputs, flags=pkg.FLAG_RLC) == exp == pvk.verify_batch(proofs, inputs)
        m = 16
        assert O.groth16_verify_many(proofs[:256 * m], 256, vk, inputs[:32 * n_public * m], n_public, m, O.MODE_REFERENCE) == exp[:m]
        pvk.close()


def test_strict_scalars(pkg, O, wl, L):
    """Default: inputs are used modulo r like bn::Fr::from_slice (x + r verifies like x).  BN254_FLAG_STRICT_SCALARS: NOT_MEMBER,
    ahead of every proof error."""
    vk, proofs, inputs, exp = wl
    good = [i for i in range(64) if exp[i] == 1][:3]
    bad_a = next(i for i in range(len(exp)) if exp[i] == 3)
    idx = good + [bad_a]
    pr = b"".join(proofs[256 * i:256 * i + 256] for i in idx)
    ins = [bytearray(inputs[64 * i:64 * i + 64]) for i in idx]
    x0 = int.from_bytes(ins[0][:32], "big"); ins[0][:32] = be(x0 + R)            # >= r, same residue
    x1 = int.from_bytes(ins[1][32:], "big"); ins[1][32:] = be(x1 + R)            # second input
    ins[3][:32] = b"\xff" * 32                                                  # invalid proof AND out-of-range input
    ii = b"".join(bytes(x) for x in ins)
    pvk = pkg.PreparedVk(vk)
    assert pvk.verify_batch(pr, ii, 4) == bytes([1, 1, 1, 3])
    assert pvk.verify_batch(pr, ii, 4) == O.groth16_verify_many(pr, 256, vk, ii, 2, 4, O.MODE_REFERENCE)
    for flags in (pkg.FLAG_STRICT_SCALARS, pkg.FLAG_STRICT_SCALARS | pkg.FLAG_RLC):
        assert pvk.verify_batch(pr, ii, 4, flags=flags) == bytes([2, 2, 1, 2])
    n = 200
    big = bytearray(inputs[:64 * n]); big[64 * 77:64 * 77 + 32] = be(R)         # exactly r
    want = bytearray(exp[:n]); want[77] = 2
    assert pvk.verify_batch(proofs[:256 * n], bytes(big), n, flags=pkg.FLAG_STRICT_SCALARS | pkg.FLAG_RLC) == bytes(want)
    pvk.close()


def test_concurrent_callers_share_one_key(pkg, L):
    """Two host threads and two device streams against ONE prepared key: the library serialises the batches on the key's workspace
    (ADVICE round 1): every caller gets the statuses of its own proofs."""
    import torch
    vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540014, 2, 3000, invalid_every=5, agree=True, threads=16)
    pvk = pkg.PreparedVk(vk)
    a = (proofs[:256 * 1700], inputs[:64 * 1700], exp[:1700])
    b = (proofs[256 * 1700:], inputs[64 * 1700:], exp[1700:])
    out = {}

    def run(name, w, flags):
        for it in range(3):
            out[(name, it)] = pvk.verify_batch(w[0], w[1], flags=flags) == w[2]

    ts = [threading.Thread(target=run, args=("a", a, 0)), threading.Thread(target=run, args=("b", b, pkg.FLAG_RLC))]
    for t in ts: t.start()
    for t in ts: t.join()
    assert len(out) == 6 and all(out.values())
    # two streams, device-resident buffers, no host synchronisation between the enqueues
    dev = torch.device("cuda:0")
    bufs = []
    for w in (a, b):
        bufs.append((torch.frombuffer(bytearray(w[0]), dtype=torch.uint8).to(dev), torch.frombuffer(bytearray(w[1]), dtype=torch.uint8).to(dev),
                     torch.full((len(w[2]),), 0xEE, dtype=torch.uint8, device=dev), torch.cuda.Stream(dev)))
    torch.cuda.synchronize(dev)
    for rep in range(2):
        for (dp, di, ds, st) in bufs:
            pvk.verify_batch_device(dp.data_ptr(), di.data_ptr(), ds.data_ptr(), ds.numel(), 256, 2, 0, st.cuda_stream)
    torch.cuda.synchronize(dev)
    assert bytes(bufs[0][2].cpu().numpy().tobytes()) == a[2] and bytes(bufs[1][2].cpu().numpy().tobytes()) == b[2]
    pvk.close()


def test_multi_device_entry_and_single_verify_loop(pkg, O, wl, L):
    import torch
    vk, proofs, inputs, exp = wl
    pvk = pkg.PreparedVk(vk)
    mask = (1 << torch.cuda.device_count()) - 1
    assert pvk.verify_batch_multi(proofs, inputs, mask) == exp
    assert pvk.verify_batch_multi(proofs, inputs, 1, flags=pkg.FLAG_RLC) == exp
    with pytest.raises(pkg.Bn254Error):
        pvk.verify_batch_multi(proofs, inputs, 1 << 40)
    pvk.close()
    # Groth16Verifier::verify mirror in a loop: prepares and frees a key (streams, events, tables) per call
    for i in range(12):
        assert pkg.Groth16Verifier.verify(proofs[256 * i:256 * i + 256], vk, [inputs[64 * i:64 * i + 32], inputs[64 * i + 32:64 * i + 64]]) == exp[i]


def test_batch_just_above_65536(pkg, L):
    """Regression size of the round-1 dwordx3 layout bug: n = 65 552."""
    n = 65552
    vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540015, 2, n, invalid_every=16, agree=True, threads=16)
    pvk = pkg.PreparedVk(vk)
    assert pvk.verify_batch(proofs, inputs) == exp
    assert pvk.verify_batch(proofs, inputs, flags=pkg.FLAG_RLC) == exp
    pvk.close()


@pytest.mark.parametrize("n", [30720, 30721, 65537])
def test_sizes_around_the_path_boundaries(pkg, O, L, n):
    """The hand-over sizes of the exact path (round 3: cooperative kernels up to 30 720 proofs, one sub-batch of lane kernels up to 65 536, two above):
    the generator's statuses on both sides of each boundary, the oracle on a sample that includes the last proofs."""
    from test_gpu_parity import _mixed_sample, _oracle_sample
    vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540100 + n, 2, n, invalid_every=16, agree=True, threads=16)
    pvk = pkg.PreparedVk(vk)
    st = pvk.verify_batch(proofs, inputs)
    assert st == exp
    idx = _mixed_sample(n, 40, 3) + [n - 2, n - 1]
    assert _oracle_sample(O, vk, proofs, inputs, 2, idx) == bytes(st[j] for j in idx)
    pvk.close()


def test_profile_accumulates_over_batches(pkg, L):
    """bn254_set_profiling(2): the per-launch events of the selected kernel accumulate over back-to-back batches (bench.py reads them once after its
    timed steps); mode 1 keeps the last batch's only; a setter call starts the accumulation over."""
    import torch
    n = 1 << 17
    vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540201, 2, n, invalid_every=16, agree=True, threads=16)
    pvk = pkg.PreparedVk(vk)
    dev = torch.device("cuda", 0)
    dp = torch.frombuffer(bytearray(proofs), dtype=torch.uint8).to(dev); di = torch.frombuffer(bytearray(inputs), dtype=torch.uint8).to(dev)
    ds = torch.zeros(n, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev)
    run = lambda: pvk.verify_batch_device(dp.data_ptr(), di.data_ptr(), ds.data_ptr(), n, 256, 2, 0, st.cuda_stream)
    try:
        L.bn254_set_profiling(1)
        pkg.set_profile_kernels(None)
        run(); torch.cuda.synchronize(dev)
        every, _ = pvk.kernel_profile_all(0)
        dom = max(every, key=lambda k: every[k][1])        # k_miller_run (k_miller_step_dbl under BN254_MILLER_RUN_STEPS=0)
        pkg.set_profile_kernels([dom])
        run(); run(); torch.cuda.synchronize(dev)
        one, per = pvk.kernel_profile_all(0)
        assert per in (n, n // 2, n // 3 + 1, n // 4) and set(one) == {dom}       # two sub-batches unless BN254_STREAMS says otherwise
        L.bn254_set_profiling(2)
        run(); run(); run(); torch.cuda.synchronize(dev)
        three, _ = pvk.kernel_profile_all(0)
        assert three[dom][0] == 3 * one[dom][0]
        assert 2.0 * one[dom][2] < three[dom][2] < 4.5 * one[dom][2]       # union of the launch intervals: three batches' worth
        pkg.set_profile_kernels([dom])                                     # a setter call: start over
        run(); torch.cuda.synchronize(dev)
        again, _ = pvk.kernel_profile_all(0)
        assert again[dom][0] == one[dom][0]
        assert bytes(ds.cpu().numpy().tobytes()) == exp
    finally:
        pkg.set_profile_kernels(None)
        L.bn254_set_profiling(0)
        pvk.close()


def test_plonk_rejects_the_known_lambda_forgery(pkg, O, fixtures, L):
    """The (H + lambda D, H' - D) forgery against round 1's published batching constant (tests/kzg_forgery.py): the oracle run with that
    constant accepts it; the product, which now draws a fresh scalar per proof like the reference (plonk/kzg.rs:149-154), answers
    PairingCheckFailed every time, alone and inside a batch, while the honest proof keeps verifying."""
    from kzg_forgery import forge, KNOWN_LAMBDA
    fx, vk = fixtures
    f = fx["fibonacci_plonk"]
    proof = bytes.fromhex(f["raw_proof"])
    pis = [int(x) for x in f["public_inputs"]]
    tampered, forged = forge(O, proof, vk, pis, KNOWN_LAMBDA)
    assert O.plonk_verify(forged, vk, pis, lam=KNOWN_LAMBDA) == O.ACCEPT
    ib = b"".join(be(x) for x in pis)
    for _ in range(4):
        assert pkg.PlonkVerifier.verify(forged, vk, pis) == 8
    pvk = pkg.PreparedPlonkVk(vk)
    st = pvk.verify_batch((proof + forged + tampered) * 20, ib * 60)
    assert st == bytes([1, 8, 8]) * 20
    pvk.close()


def test_single_verify_key_cache(pkg, O, L):
    """The single-proof entry keeps the prepared form of its last four keys (exact byte match): six keys in rotation (so entries are evicted and
    re-prepared), both readings of the G2 root order under the same bytes, a valid and a rejected proof per key -- the answers never depend on
    what the cache holds."""
    keys = []
    for k in range(6):
        vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540100 + k, 2, 8, invalid_every=2, agree=True, threads=2)
        keys.append((vk, proofs, inputs, exp))
    for rnd in range(3):
        for vk, proofs, inputs, exp in keys:
            for i in (0, 1, 2, 3):
                ins = [int.from_bytes(inputs[64 * i + 32 * j:64 * i + 32 * j + 32], "big") for j in range(2)]
                for mode, omode in ((pkg.VK_REFERENCE, O.MODE_REFERENCE), (pkg.VK_GNARK, O.MODE_GNARK)):
                    got = pkg.Groth16Verifier.verify(proofs[256 * i:256 * i + 256], vk, ins, mode)
                    if rnd == 0:
                        assert got == O.groth16_verify_many(proofs[256 * i:256 * i + 256], 256, vk, inputs[64 * i:64 * i + 64], 2, 1, omode)[0]
                    if mode == pkg.VK_REFERENCE:
                        assert got == exp[i]
    # a key that does not parse is never cached and never confused with a cached one
    bad = bytearray(keys[0][0]); bad[3] ^= 0xFF
    assert pkg.Groth16Verifier.verify(keys[0][1][:256], bytes(bad[:100]), [1, 2]) == pkg.ERR_MALFORMED


def test_public_input_point_at_infinity_on_every_path(pkg, O, L):
    """Valid proofs whose public-input point L is the identity (generator option l_identity): the pair (L, gamma) contributes nothing
    (bn::pairing_batch skips it; the kernels keep f where the line would be multiplied in).  Through the cooperative kernels (4096 proofs),
    the one-proof-per-lane kernels (45 000), the RLC mode and the single-proof entry; sampled against the oracle."""
    for n_public in (1, 2, 5):
        n = 45000 if n_public == 2 else 4096
        vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540300 + n_public, n_public, n, invalid_every=11, agree=True, threads=16, l_identity=True)
        pvk = pkg.PreparedVk(vk)
        assert pvk.verify_batch(proofs, inputs, n) == exp
        assert pvk.verify_batch(proofs[:256 * 4096], inputs[:32 * n_public * 4096], 4096) == exp[:4096]
        assert pvk.verify_batch(proofs, inputs, n, flags=pkg.FLAG_RLC) == exp
        idx = [i for i in range(200) if i % 7 == 3][:12] + [0, 1, 2]
        sp = b"".join(proofs[256 * i:256 * i + 256] for i in idx); si = b"".join(inputs[32 * n_public * i:32 * n_public * (i + 1)] for i in idx)
        assert O.groth16_verify_many(sp, 256, vk, si, n_public, len(idx), O.MODE_REFERENCE) == bytes(exp[i] for i in idx)
        assert exp[3] == pkg.ACCEPT
        ins = [int.from_bytes(inputs[32 * n_public * 3 + 32 * j:32 * n_public * 3 + 32 * j + 32], "big") for j in range(n_public)]
        assert pkg.Groth16Verifier.verify(proofs[256 * 3:256 * 4], vk, ins) == pkg.ACCEPT
        pvk.close()
