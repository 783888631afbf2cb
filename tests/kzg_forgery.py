"""Builds the (H + lambda D, H' - D) forgery against a KNOWN KZG batching scalar (ADVICE high).

plonk/kzg.rs:128-190 checks  e(P0, g2[0]) e(P1, g2[1]) == 1  with
    P0 = sum_i gamma^i D_i + lambda Z - fe G + zeta H + lambda zeta omega H',      P1 = -(H + lambda H')
and the two opening quotients H (offset 448 of the proof) and H' (offset 740) are bound by no transcript.  Replacing
(H, H') by (H + lambda D, H' - D) leaves P1 alone and moves P0 by lambda zeta (1 - omega) D, so a prover who knows lambda can
absorb ANY error E in P0 -- here: a wrong claimed evaluation of the BSB22 selector polynomial, which no earlier check sees --
with D = -E / (lambda zeta (1 - omega)).  Only test code: uses the oracle."""
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
OFF_H, OFF_CLAIMED, OFF_HZ = 448, 516, 740
KNOWN_LAMBDA = 0x9e3779b97f4a7c15   # the constant round 1 shipped


def be(v):
    return int(v).to_bytes(32, "big")


def _neg(p):
    return p[:32] + be((P - int.from_bytes(p[32:], "big")) % P)


def forge(O, proof, vk, pis, lam):
    """Returns (tampered, forged): `tampered` has a wrong claimed value and fails the pairing check under any lambda; `forged`
    additionally shifts the opening quotients so that the check passes under exactly this lambda."""
    tampered = bytearray(proof)
    o = OFF_CLAIMED + 32 * 6                              # claimed value of the first BSB22 selector (qcp) polynomial
    tampered[o:o + 32] = be((int.from_bytes(proof[o:o + 32], "big") + 1) % R)
    tampered = bytes(tampered)
    st0, (p0, p1), _ = O.plonk_pairing_inputs_lam(proof, vk, pis, lam)
    st1, (q0, q1), _ = O.plonk_pairing_inputs_lam(tampered, vk, pis, lam)
    assert st0 == O.ACCEPT and st1 == O.ERR_PAIRING_FAILED and p1 == q1 and p0 != q0
    err = O.g1_add(q0, _neg(p0))                          # E = P0' - P0
    _, dg = O.plonk_stage_digests(proof, vk, pis)
    zeta = int.from_bytes(dg["zeta"], "big") % R
    omega = int.from_bytes(vk[40:72], "big") % R
    k = lam * zeta % R * ((1 - omega) % R) % R
    d = O.g1_mul(_neg(err), pow(k, -1, R))                # D = -E / (lambda zeta (1 - omega))
    h, hz = proof[OFF_H:OFF_H + 64], proof[OFF_HZ:OFF_HZ + 64]
    forged = bytearray(tampered)
    forged[OFF_H:OFF_H + 64] = O.g1_add(h, O.g1_mul(d, lam))
    forged[OFF_HZ:OFF_HZ + 64] = O.g1_add(hz, _neg(d))
    return tampered, bytes(forged)
