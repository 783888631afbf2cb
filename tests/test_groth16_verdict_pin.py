"""Groth16 `verify()` pinned at VERDICT level by an implementation that is independent of both the oracle and the product.

The reference's Groth16 key is not in its repository (examples/script/src/main.rs:178-180), so no reference-held file decides an accept or a
reject.  tests/golden/groth16_verdicts.json holds synthetic gnark-format keys / proofs / inputs and the status bytes that tests/pyref_groth16.py
(pure Python: vk decompression in both root-order readings, prepare_inputs, the literal equation of groth16/verify.rs:70-77, a polynomial-basis
pairing) assigns them; tests/golden/make_groth16_verdicts.py wrote it.  Here:
  * CPU: the oracle must give exactly those status bytes in both modes, for every case; three verdicts are re-derived live with the Python
    restatement (so the fixture cannot drift from its generator); the parse-level pieces are cross-checked against the oracle's codecs.
  * GPU: the product, through the C ABI, must give the same status bytes."""
import json
import os
import random

import pytest

import pyref_groth16 as PY

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def verdicts():
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "groth16_verdicts.json")))
    out = []
    for c in d["cases"]:
        key = d["keys"][c["key"]]
        ib = bytes.fromhex(c["inputs"])
        out.append({"key": c["key"], "vk": bytes.fromhex(key["vk"]), "proof": bytes.fromhex(c["proof"]), "inputs": ib, "n_inputs": len(ib) // 32,
                    "ref": c["verdict_reference"], "gnark": c["verdict_gnark"], "gen": c["generator_expects"]})
    return out


def test_fixture_covers_the_cases_that_matter(verdicts):
    assert len(verdicts) >= 20
    agree = [c for c in verdicts if c["key"] == "agree" and c["gen"] is not None]
    # on the agreement key both readings give the generator's expected status, and every status class is present
    assert all(c["ref"] == c["gnark"] == c["gen"] for c in agree)
    assert {c["ref"] for c in agree} == {PY.REJECT, PY.ACCEPT, PY.ERR_NOT_MEMBER, PY.ERR_NOT_ON_CURVE, PY.ERR_NOT_IN_SUBGROUP}
    assert sum(c["ref"] == PY.ACCEPT for c in verdicts) >= 8
    # outside the agreement set gnark accepts the valid proofs and the reference's literal reading does not
    dis = [c for c in verdicts if c["key"] == "disagree"]
    assert dis and all(c["gnark"] == PY.ACCEPT for c in dis) and all(c["ref"] == PY.REJECT for c in dis)
    assert any(c["ref"] == PY.ERR_INPUT_LEN for c in verdicts) and any(c["ref"] == PY.ERR_MALFORMED for c in verdicts)


def test_oracle_gives_the_independent_verdicts(O, verdicts):
    for k, c in enumerate(verdicts):
        ins = [c["inputs"][32 * j:32 * j + 32] for j in range(c["n_inputs"])]
        assert O.groth16_verify(c["proof"], c["vk"], ins, O.MODE_REFERENCE) == c["ref"], ("reference", k)
        assert O.groth16_verify(c["proof"], c["vk"], ins, O.MODE_GNARK) == c["gnark"], ("gnark", k)


def test_three_verdicts_rederived_live(verdicts):
    """A valid proof under the reference reading, a REJECT (tampered input) and a valid proof of the disagreement key under the gnark
    reading: about 12 s each in pure Python."""
    ints = lambda c: [int.from_bytes(c["inputs"][32 * j:32 * j + 32], "big") for j in range(c["n_inputs"])]
    acc = next(c for c in verdicts if c["key"] == "agree" and c["gen"] == PY.ACCEPT)
    rej = next(c for c in verdicts if c["key"] == "agree" and c["gen"] == PY.REJECT)
    dis = next(c for c in verdicts if c["key"] == "disagree")
    assert PY.verify(acc["proof"], acc["vk"], ints(acc), PY.MODE_REFERENCE) == acc["ref"] == PY.ACCEPT
    assert PY.verify(rej["proof"], rej["vk"], ints(rej), PY.MODE_REFERENCE) == rej["ref"] == PY.REJECT
    assert PY.verify(dis["proof"], dis["vk"], ints(dis), PY.MODE_GNARK) == dis["gnark"] == PY.ACCEPT


def test_python_codecs_against_the_oracle(O, verdicts):
    """The parse-level half of the restatement, piece by piece: G1 / G2 decompression in both modes and prepare_inputs."""
    be = lambda v: int(v).to_bytes(32, "big")
    for name in ("agree", "disagree"):
        vk = next(c["vk"] for c in verdicts if c["key"] == name)
        nk = int.from_bytes(vk[288:292], "big")
        for off in [0, 32, 192] + [292 + 32 * i for i in range(nk)]:
            st, xy = O.decompress_g1(vk[off:off + 32])
            x, y = PY.decompress_g1(vk[off:off + 32])
            assert st == O.ACCEPT and xy == be(x) + be(y)
        for off in (64, 128, 224):
            for pm, om in ((PY.MODE_REFERENCE, O.MODE_REFERENCE), (PY.MODE_GNARK, O.MODE_GNARK)):
                st, q = O.decompress_g2(vk[off:off + 64], om)
                (x0, x1), (y0, y1) = PY.decompress_g2(vk[off:off + 64], pm)
                assert st == O.ACCEPT and q == be(x1) + be(x0) + be(y1) + be(y0)
    c = verdicts[0]
    key = PY.load_vk(c["vk"], PY.MODE_REFERENCE)
    rng = random.Random(5)
    xs = [rng.randrange(1 << 256), rng.randrange(PY.R)]
    L = PY.prepare_inputs(key["k"], xs)
    acc = be(key["k"][0][0]) + be(key["k"][0][1])
    for x, k in zip(xs, key["k"][1:]):
        acc = O.g1_add(acc, O.g1_mul(be(k[0]) + be(k[1]), x))
    assert acc == be(L[0]) + be(L[1])


@pytest.mark.gpu
def test_product_gives_the_independent_verdicts(pkg, verdicts):
    """Single-proof entry (Groth16Verifier::verify) in both modes, and the batch entry per key."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU; the product has no CPU fallback"
    for k, c in enumerate(verdicts):
        ins = [c["inputs"][32 * j:32 * j + 32] for j in range(c["n_inputs"])]
        assert pkg.Groth16Verifier.verify(c["proof"], c["vk"], ins, pkg.VK_REFERENCE) == c["ref"], ("reference", k)
        assert pkg.Groth16Verifier.verify(c["proof"], c["vk"], ins, pkg.VK_GNARK) == c["gnark"], ("gnark", k)
    for name in ("agree", "disagree"):
        cs = [c for c in verdicts if c["key"] == name and c["n_inputs"] == 2 and len(c["proof"]) == 256]
        for mode, field in ((pkg.VK_REFERENCE, "ref"), (pkg.VK_GNARK, "gnark")):
            pvk = pkg.PreparedVk(cs[0]["vk"], mode)
            st = pvk.verify_batch(b"".join(c["proof"] for c in cs), b"".join(c["inputs"] for c in cs))
            assert st == bytes(c[field] for c in cs), (name, field)
            pvk.close()
