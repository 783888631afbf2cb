"""The G1 multi-scalar multiplications of the PlonK path as ROWS (snark-bn254-verifier_amd/csrc/bn254_msm.h) on the CPU: tests/hostsim plans a launch with
msm_plan_build, evaluates every row with the code a lane of k_g1_msm_rows runs (bound tracker on) and adds the rows of each sum; the oracle judges the
sums.  Plus the properties of the plans the library makes for PlonK keys (bn254_dbg_plonk_msm_plan): every term is covered exactly once and every launch a
context can see fits the scratch the context allocates -- the class of the round-3 heap overflow."""
import ctypes as C
import random

import pytest

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def _glv(L, k):
    k1, k2 = (C.c_uint8 * 16)(), (C.c_uint8 * 16)()
    n1, n2 = C.c_int(), C.c_int()
    assert L.bn254_dbg_glv_decompose(int(k % R).to_bytes(32, "big"), k1, k2, C.byref(n1), C.byref(n2)) == 0
    return bytes(k1), bytes(k2), (2 if n1.value else 0) | (4 if n2.value else 0)


def _shape_ints(sums):
    """sums: [(var terms, unit terms, [(fixed term, table)])]"""
    out = [len(sums)]
    for v, u, f in sums:
        out += [len(v), len(u), len(f)]
    for v, u, f in sums:
        out += list(v) + list(u) + [t for t, _ in f] + [b for _, b in f]
    return (C.c_int * len(out))(*out)


def _run(hostsim, L, O, rng, sums, n_terms, n_pad, budget, zero_scalar_terms=(), identity_terms=(), joint_g=0):
    g = O.g1_gen()
    n_tabs = 1 + max([b for _, _, f in sums for _, b in f] + [0])
    bases = [O.g1_mul(g, rng.randrange(1, R)) for _ in range(n_tabs)]
    recs, want = [], []
    kinds = {}
    for s, (v, u, f) in enumerate(sums):
        for t in v: kinds[t] = ("var", s, None)
        for t in u: kinds[t] = ("unit", s, None)
        for t, b in f: kinds[t] = ("fixed", s, b)
    acc = [bytes(64)] * len(sums)

    def add(s, pt):
        if pt == bytes(64):
            return
        acc[s] = pt if acc[s] == bytes(64) else O.g1_add(acc[s], pt)

    for t in range(n_terms):
        kind, s, b = kinds.get(t, ("none", 0, None))
        p = O.g1_mul(g, rng.randrange(1, R))
        k = 0 if t in zero_scalar_terms else rng.choice([rng.randrange(R), rng.randrange(1 << 64), R - 1, 1, rng.randrange(R)])
        if kind == "var":
            k1, k2, fl = _glv(L, k)
            if t in identity_terms:
                recs.append(bytes(64) + k1 + k2 + bytes(32) + bytes([fl | 1]))
            else:
                recs.append(p + k1 + k2 + bytes(32) + bytes([fl]))
                add(s, O.g1_mul(p, k) if k % R else bytes(64))
        elif kind == "unit":
            neg = rng.randrange(2)
            recs.append(p + bytes(32) + bytes(32) + bytes([2 if neg else 0]))
            add(s, O.g1_mul(p, R - 1) if neg else p)
        elif kind == "fixed":
            recs.append(bytes(64) + bytes(32) + (k % R).to_bytes(32, "little") + bytes([0]))
            add(s, O.g1_mul(bases[b], k) if k % R else bytes(64))
        else:
            recs.append(bytes(129))
    out = (C.c_uint8 * (64 * len(sums)))()
    info = (C.c_int * 5)()
    assert hostsim.hs_msm_rows_joint(out, info, _shape_ints(sums), b"".join(recs), b"".join(bases), n_pad, budget, joint_g) == 1
    got = bytes(out)
    for s in range(len(sums)):
        assert got[64 * s:64 * s + 64] == acc[s], (sums, n_pad, budget, s)
    return list(info)


def test_msm_rows_plonk_shapes_vs_oracle(hostsim, pkg, O):
    """The two launches of a PlonK proof (one BSB22 commitment, as the SP1 circuit) in the split form (4096 proofs within one wavefront per SIMD), in the
    unsplit form (a large batch) and with a budget so small that the fixed windows overflow onto the low rows; zero scalars, an identity point, unit terms."""
    L = pkg.lib()
    rng = random.Random(31)
    q = 1
    s1 = [([0, q + 6, q + 7, q + 8, q + 9], [], [(q + i, i) for i in range(6)])]
    s2 = [([0, 1, 2, 3, 6 + q, 8 + q, 9 + q], [], [(4, 6), (5, 7), (6, 9), (7 + q, 8)]), ([11 + q], [10 + q], [])]
    i1 = _run(hostsim, L, O, rng, s1, 10 + q, 4096, 65536)
    assert i1[0] == 11 and i1[1] == 10        # five variable terms split (the planner's split point), the six fixed terms (6 x 16 windows) on one row of their own + windows on the low rows
    i2 = _run(hostsim, L, O, rng, s2, 12 + q, 4096, 65536, zero_scalar_terms=(2,), identity_terms=(0,))
    assert i2[0] == 16 and i2[3] == 14 and i2[4] == 2
    i3 = _run(hostsim, L, O, rng, s2, 12 + q, 65536, 65536)
    assert i3[1] == 8 and i3[0] == 9          # unsplit: one row per variable term, one row of fixed windows
    _run(hostsim, L, O, rng, s1, 10 + q, 64, 64 * 11)      # split with ONE spare row
    _run(hostsim, L, O, rng, s1, 10 + q, 4096, 4096 * 16)


def test_msm_rows_odd_shapes(hostsim, pkg, O):
    """Shapes no PlonK key produces: a sum of fixed terms only, of unit terms only, an empty sum, many fixed terms on one variable term."""
    L = pkg.lib()
    rng = random.Random(32)
    _run(hostsim, L, O, rng, [([], [], [(0, 0), (1, 1)])], 2, 64, 65536)
    _run(hostsim, L, O, rng, [([], [0, 1], [])], 2, 64, 65536)
    _run(hostsim, L, O, rng, [([0], [], []), ([], [], [])], 1, 64, 65536)
    _run(hostsim, L, O, rng, [([0], [1], [(2 + i, i % 3) for i in range(7)])], 9, 64, 128)
    _run(hostsim, L, O, rng, [([0], [1], [(2 + i, i % 3) for i in range(7)])], 9, 64, 65536)


def test_msm_joint_rows_vs_oracle(hostsim, pkg, O):
    """Large launches: up to joint_g variable terms of a sum in ONE row that shares the doublings of a step between them (Straus; bn254_msm.h::msm_joint_eval), the
    fixed windows on rows of their own.  The PlonK shapes with groups of 2 / 4 / 8, zero scalars and identity points inside a group, a sum of one term, odd shapes."""
    L = pkg.lib()
    rng = random.Random(33)
    q = 1
    s1 = [([0, q + 6, q + 7, q + 8, q + 9], [], [(q + i, i) for i in range(6)])]
    s2 = [([0, 1, 2, 3, 6 + q, 8 + q, 9 + q], [], [(4, 6), (5, 7), (6, 9), (7 + q, 8)]), ([11 + q], [10 + q], [])]
    i1 = _run(hostsim, L, O, rng, s1, 10 + q, 65536, 65536, joint_g=4)
    assert i1[1] == 5 and i1[0] <= 5                       # five table slots, two joint rows (3 + 2 terms) + rows of fixed windows
    i2 = _run(hostsim, L, O, rng, s2, 12 + q, 65536, 65536, zero_scalar_terms=(2,), identity_terms=(0, 9 + q), joint_g=4)
    assert i2[1] == 8 and i2[4] == 1                       # seven terms in two joint rows (4 + 3), the second sum's one term a joint row of its own with the unit term
    _run(hostsim, L, O, rng, s2, 12 + q, 65536, 65536, zero_scalar_terms=(1, 3), joint_g=8)
    _run(hostsim, L, O, rng, s2, 12 + q, 65536, 65536, joint_g=2)
    _run(hostsim, L, O, rng, s1, 10 + q, 65536, 65536, identity_terms=(0, q + 6, q + 7, q + 8, q + 9), joint_g=8)      # every point of the group the identity
    _run(hostsim, L, O, rng, [([0, 1, 2], [3], [])], 4, 65536, 65536, joint_g=3)
    _run(hostsim, L, O, rng, [([0], [], [(1, 0)]), ([2, 3], [], [])], 4, 65536, 65536, joint_g=2)
    # a split launch ignores the group size
    assert _run(hostsim, L, O, rng, s1, 10 + q, 4096, 65536, joint_g=4)[0] == 11


def _plan(L, n_qcp, stage, n, budget=0):
    rows, var = C.c_int(), C.c_int()
    scratch = C.c_size_t()
    chain = C.c_int()
    sums, fixed = (C.c_int * 2)(), (C.c_int * 2)()
    desc = (C.c_int * (32 * 9))()
    L.bn254_dbg_plonk_msm_plan.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_int),
                                            C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    assert L.bn254_dbg_plonk_msm_plan(n_qcp, stage, n, budget, C.byref(rows), C.byref(var), C.byref(scratch), C.byref(chain), sums, fixed, desc) == 0
    return rows.value, var.value, scratch.value, chain.value, list(sums), list(fixed), [list(desc[9 * r:9 * r + 9]) for r in range(rows.value)]


def test_plonk_msm_plans_cover_every_term_once(pkg):
    """For every key shape (0..8 commitments), both launches and batch sizes on both sides of every form change: each variable term's 128 joint bit positions are
    covered exactly once, each fixed term's 20 windows (of 13 bits: bn254_fw.h MSM_FW_WINDOWS) exactly once, each unit term once; rows of a sum are contiguous; the split form stays within the budget."""
    L = pkg.lib()
    for q in range(0, 9):
        for stage in (1, 2):
            n_var = (4 + q) if stage == 1 else 8
            n_fixed = [6, 0] if stage == 1 else [3 + q, 0]
            for n in (1, 63, 64, 65, 1000, 2048, 2520, 2521, 4095, 4096, 4097, 5040, 5041, 8192, 49151, 49152, 49153, 65536):
                rows, var, scratch, chain, sums, fixed, desc = _plan(L, q, stage, n)
                n_pad = (n + 63) // 64 * 64
                assert rows <= 32 and fixed == n_fixed and sum(sums) == rows
                split = 2 * n_var * n_pad <= 65536
                assert var == (2 if split else 1) * n_var and scratch == var * n_pad
                if split:
                    assert rows * n_pad <= 65536 or rows == 2 * n_var      # own rows only while the budget has room
                cover = {}
                units = []
                windows = [set(), set()]
                slots = set()
                for r, (vt, lo, hi, ut, s, slot, flo, fhi, jmask) in enumerate(desc):
                    if jmask:                                  # a joint row: its terms over all 128 positions, one table slot each
                        assert vt == -1 and (lo, hi) == (0, 128) and not split
                        for t in range(32):
                            if jmask >> t & 1:
                                cover.setdefault(t, []).append((0, 128))
                                assert slot not in slots
                                slots.add(slot)
                                slot += 1
                    if vt >= 0:
                        assert lo % 2 == 0 and hi % 2 == 0 and lo < hi <= 128
                        cover.setdefault(vt, []).append((lo, hi))
                        assert slot not in slots
                        slots.add(slot)
                    if ut >= 0:
                        units.append(ut)
                    for w in range(flo, fhi):
                        assert w not in windows[s]
                        windows[s].add(w)
                assert len(cover) == n_var
                for v in cover.values():                      # one row, or a low and a high row that meet at an even position
                    v = sorted(v)
                    assert v == [(0, 128)] or (len(v) == 2 and v[0][0] == 0 and v[0][1] == v[1][0] and v[1][1] == 128 and v[0][1] % 2 == 0 and 2 <= v[0][1] <= 126), v
                    assert (len(v) == 2) == split
                assert [len(windows[0]), len(windows[1])] == [20 * n_fixed[0], 0]
                assert units == ([] if stage == 1 else [10 + q])
                assert slots == set(range(var))
                first = [min(r for r, d in enumerate(desc) if d[4] == s) for s in range(2 if stage == 2 else 1)]
                for s, f in enumerate(first):
                    assert all(desc[r][4] == s for r in range(f, f + sums[s]))


def test_plonk_context_scratch_holds_every_launch(pkg):
    """The sizing rule of a PlonK context (bn254_dbg_plonk_scratch_lanes(capacity, variable terms)) against EVERY batch size a context of that capacity can see,
    both launches, every key shape: the launch form follows the batch, the buffer the capacity (rounds 2-3 sized the buffer from the capacity's own form and a
    5000-proof batch on a 5120-proof context wrote 15 MB past the end)."""
    L = pkg.lib()
    L.bn254_dbg_plonk_scratch_lanes.restype = C.c_size_t
    L.bn254_dbg_plonk_scratch_lanes.argtypes = [C.c_size_t, C.c_int]
    L.bn254_dbg_plonk_part_points.restype = C.c_size_t
    L.bn254_dbg_plonk_part_points.argtypes = [C.c_size_t, C.c_int, C.c_int]
    rng = random.Random(9)
    for q in (0, 1, 2, 8):
        n_var = max(4 + q, 8)
        for cap in (256, 512, 2560, 4096, 5120, 8192, 16384, 65536, 131072, 262144):
            have = L.bn254_dbg_plonk_scratch_lanes(cap, n_var)
            # the row buffer (round 5: sized from the key shape's plans instead of 32 rows per proof of capacity -- 0.9 GB at 2^18): rows x items of every launch
            points = [L.bn254_dbg_plonk_part_points(cap, q, stage) for stage in (1, 2)]
            assert all(0 < x <= 32 * cap for x in points)
            sizes = {1, 64, cap, cap - 1, cap // 2, cap // 2 + 1, 2167, 2520, 2521, 4333, 4096, 5000, 5040, 5041, 49151, 49152, 65536, 65537} | {rng.randrange(1, cap + 1) for _ in range(40)}
            sizes |= {x for lim in (65536 // (2 * v) for v in range(1, 14)) for x in (lim - 64, lim - 1, lim, lim + 1, lim + 63, lim + 64, lim + 65)}      # around the split / unsplit hand-over
            for n in sorted(x for x in sizes if 1 <= x <= cap):
                for stage in (1, 2):
                    rows, var, scratch, chain, sums, fixed, desc = _plan(L, q, stage, n)
                    assert scratch <= have, (q, cap, n, stage, scratch, have)
                    assert rows <= 32
                    assert rows * n <= points[stage - 1], (q, cap, n, stage, rows, points)
