#!/usr/bin/env python3
"""Counts the 32x32+64-bit multiply-adds (v_mad_u64_u32 / v_mad_i64_i32) each kernel executes per proof and launch, from the gfx950
code object inside libbn254_verify_amd.so, and writes profiles/kernel_mads.json (read by bench.py for `roofline`).

  python tools/count_mads.py            # after `make -C snark-bn254-verifier_amd/csrc`

Method: `llvm-objdump --offloading` extracts the code object, `llvm-objdump -d` disassembles it; per kernel symbol the instructions
are split at branch targets.  Every BACKWARD branch closes a loop (its body is the address range target..branch) and every FORWARD
conditional branch opens a conditional region (branch..target).  An instruction's weight is the product of the trip counts of the
loops and of the execution probabilities of the conditional regions that contain it; trip counts / probabilities come from MODEL
below (each entry says where the number comes from).  The straight-line kernels that make up > 95 % of the path need no model: their
only branch is the early exit of a wave whose 64 proofs have all failed.  A loop the script does not recognise is counted once and
reported ("!!"), so that a changed kernel shows up instead of silently changing the count."""
import collections
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "snark-bn254-verifier_amd", "libbn254_verify_amd.so")
MAD = re.compile(r"^\s*v_mad_[iu]64_[iu]32\b")
VALU = re.compile(r"^\s*v_")
# multiply-adds of the G1 loop bodies the models recognise by size (bn254_curve.h, round 5: the sum-of-products form of RCB16 -- a complete addition 1765, a mixed
# addition 1580, a doubling 1169; until round 4: 2000 / 1815 / 1243)
R_ADD = (1700, 1800)       # complete projective addition
R_MIXED = (1500, 1650)     # complete mixed addition
R_DBL = (1100, 1250)       # complete doubling
R_STEP = (3950, 4200)      # a two-bit step: two doublings + one addition
def _in(c, r): return r[0] <= c <= r[1]

# exponent bit patterns (snark-bn254-verifier_amd/csrc/bn254_constants.h): fp_pow_bits squares once per bit after the first and multiplies
# on set bits; the multiplication sits in a conditional region inside the loop
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
PM2_BITS = bin(P - 2)[2:]
INV_SQUARINGS = len(PM2_BITS) - 1
INV_MUL_FRACTION = (PM2_BITS.count("1") - 1) / (len(PM2_BITS) - 1)

def extract_code_objects(workdir):
    """One code object per translation unit (kernels, C ABI, the two cooperative files)."""
    lib = os.path.join(workdir, "lib.so")
    shutil.copy(LIB, lib)
    subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", lib], stdout=subprocess.DEVNULL, cwd=workdir)
    cos = sorted(os.path.join(workdir, f) for f in os.listdir(workdir) if "amdgcn" in f)
    if not cos:
        raise SystemExit("no gfx950 code object found in " + LIB)
    return cos


def disassemble(co):
    out = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", co], text=True)
    funcs = collections.OrderedDict()
    cur = None
    for line in out.splitlines():
        m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
        if m:
            cur = []
            funcs[m.group(2)] = cur
            continue
        if cur is None or "//" not in line:
            continue
        text, comment = line.split("//", 1)
        am = re.match(r"\s*([0-9A-Fa-f]+):", comment)
        if not am:
            continue
        addr = int(am.group(1), 16)
        tgt = None
        if re.match(r"^\s*s_c?branch", text):
            tm = re.search(r"<\S+\+0x([0-9a-f]+)>", comment)
            if tm:
                tgt = int(tm.group(1), 16)   # offset from the function start, fixed up below
            elif re.search(r"<\S+>\s*$", comment):
                tgt = 0
        cur.append([addr, text.strip(), tgt])
    for name, ins in funcs.items():
        if not ins:
            continue
        base = ins[0][0]
        for i in ins:
            if i[2] is not None:
                i[2] += base
    return funcs


def short(sym):
    m = re.match(r"_ZN5bn254(\d+)(.*)", sym)
    if not m:
        return sym
    n = int(m.group(1))
    name = m.group(2)[:n]
    return name


# ---- loop models: a loop is recognised by the multiply-adds of one iteration --------------------------------------------------------------------
FP_SQR, FP_MUL = 126, 162            # fp_sqr: 45 + 81; fp_mul: 81 + 81 (bn254_fp.h)
INV_SET_BITS = PM2_BITS.count("1") - 1
INV_GCD_ROUND_MADS = 90               # fp_inv: (a, b) and (u, v) updates of one round: 4 x 18 digit products + 2 x 9 for the Montgomery digit
N_PUBLIC = 2                          # BASELINE configs[2]
TRIPS_BY_KERNEL = {                   # loops whose trip count is a launch parameter: (kernel, mads of one iteration) -> trips, why
    ("k_g16_prepare", "window_outer"): (N_PUBLIC, "one pass per public input"),
    ("k_rlc_group_points", "window_outer"): (N_PUBLIC, "one pass per public input"),
    ("k_g16_msm_partial", "window_outer"): (16, "G16_WIDE_MSM_INPUTS_PER_LANE inputs per lane"),
    ("k_rlc_scale", "scalar_mul"): (64, "64 joint bit positions of the GLV weight k1 + k2 lambda"),
    ("k_g1_scalar_mul", "scalar_mul"): (128, "128 joint bit positions of the GLV halves of a 254-bit scalar"),
    ("k_rlc_scale", "fr_products"): (N_PUBLIC, "one Fr product pair per public input"),
    ("k_f12_cyclo_sqr_n", "cyclo_sqr"): (186 / 39.0, "3 x 62 squarings of exp-by-u in 39 launches (BN_U_W4)"),
}


def loop_groups(ins, mads):
    """Backward branches grouped by their target (one loop, possibly several latches).  -> [(head, [latch...])]"""
    g = collections.OrderedDict()
    for addr, text, tgt in ins:
        if tgt is not None and tgt <= addr:
            g.setdefault(tgt, []).append(addr)
    return [(h, sorted(l)) for h, l in g.items()]


def count_in(mads, lo, hi):
    return sum(1 for a in mads if lo <= a <= hi)


def model_components(name, ins):
    """Kernels whose loops run a launch-dependent number of times (the row kernels of the PlonK MSMs, the two-pair Miller loop): the multiply-adds of each loop
    BODY, recognised by size -- a complete addition ~2000, a mixed addition ~1800, a doubling ~1240, a two-bit step (2 doublings + 1 addition) ~4460 -- so that
    bench.py prices a launch from its plan (bn254_dbg_plonk_msm_plan): rows x (table + steps x step + doublings x dbl + windows x mixed)."""
    mads = [a for a, t, _ in ins if MAD.match(t)]
    groups = [(h, l, count_in(mads, h, l[-1])) for h, l in loop_groups(ins, mads)]
    groups = [g for g in groups if g[2] > 0]
    comp, in_loops = {}, 0
    if name == "k_g1_msm_rows":
        # Two copies of the variable-term code, in address order: the JOINT rows (a loop over the row's terms that holds a table head + the 3 x 3 table loop, then
        # the step loop: a conditional pair of doublings + one addition per term and step) and the single-term rows (table, step loop, trailing doublings); then
        # the mixed additions of the fixed windows.
        steps = sorted((h, c) for h, l, c in groups if _in(c, R_STEP))
        assert len(steps) == 2, ("k_g1_msm_rows: a joint and a single step loop expected", groups)
        joint_tab = [(h, c) for h, l, c in groups if 6600 <= c <= 7800]
        assert len(joint_tab) == 1 and joint_tab[0][0] < steps[0][0], ("k_g1_msm_rows: the joint rows' table loop expected first", groups)
        for h, l, c in groups:
            if (h, c) == steps[0] or (h, c) == joint_tab[0]:
                continue
            key = "step" if (h, c) == steps[1] else "table_add" if _in(c, R_ADD) else "dbl" if _in(c, R_DBL) else "mixed" if _in(c, R_MIXED) else None
            if key == "table_add" and comp.get(key) == c:
                continue                                  # the 3 x 3 table loops (two copies, outer loops that hold nothing but the inner one)
            assert key and key not in comp, ("k_g1_msm_rows: loop layout changed", c, comp)
            comp[key] = c; in_loops += c
        assert set(comp) == {"step", "table_add", "dbl", "mixed"}, comp
        in_loops += steps[0][1] + joint_tab[0][1]
        rest = len(mads) - in_loops
        comp["unit"] = comp["mixed"]                      # the unit term: one mixed addition outside the loops
        comp["table_head"] = rest - comp["unit"]          # multiples of P and phi(P): 2 doublings, 2 mixed additions, beta x
        comp["joint_add"] = steps[0][1] - 2 * comp["dbl"]     # a joint row's step: one pair of doublings + this per term
        assert abs(joint_tab[0][1] - comp["table_add"] - comp["table_head"]) <= 64 and _in(comp["joint_add"], R_ADD), (joint_tab, comp)
        full = comp["table_head"] + 9 * comp["table_add"] + 64 * comp["step"]
        return {"static_mads": len(mads), "components": comp, "mads_per_proof_launch": float(full), "unmodelled": [],
                "model": "row kernel: multiply-adds per loop body (complete addition %(table_add)d x9 for the window table, two-bit step %(step)d, doubling %(dbl)d, mixed addition "
                         "%(mixed)d per fixed-base byte window / unit term; table head %(table_head)d; joint rows: %(joint_add)d per term and step + one pair of doublings per step); "
                         "mads_per_proof_launch = ONE unsplit variable row (table + 64 steps); a "
                         "launch is priced from its plan" % comp}
    if name == "k_g1_sum_affine":
        # two loops hold a complete addition each: the rows of a lane, the butterfly steps over a quad of lanes (four lanes per item for small batches)
        add = [c for h, l, c in groups if _in(c, R_ADD)]
        assert len(add) == 2 and abs(add[0] - add[1]) <= 2, ("k_g1_sum_affine: loop layout changed", groups)
        e = model_kernel("k_g1_sum_affine_inv", ins)      # the inversion's rounds
        base = e["mads_per_proof_launch"] - add[0] - add[1]
        return {"static_mads": len(mads), "components": {"add": add[0], "tail": base}, "mads_per_proof_launch": float(base + 14 * add[0]), "unmodelled": [],
                "model": "sum of a plan's rows: complete addition %d per row + %d (to affine: binary-GCD inversion, two products); mads_per_proof_launch = 14 rows on one lane (with four lanes per item "
                         "each lane adds a quarter of the rows + two butterfly steps and all four run the tail: the algorithmic count is what is reported)" % (add[0], base)}
    if name == "k_miller_run_fixed2":
        assert len(groups) == 1, ("k_miller_run_fixed2: one step loop expected", groups)
        h, l, c = groups[0]
        # ONE copy of each piece: the two line products of a step form the loop body (address range head .. latch), the squaring of f sits outside that range and
        # is entered from the latch when the next step is a doubling step (64 of the 88 steps).  Until round 4 this model read the layout the other way round
        # ("loop body = squaring + two lines, the code behind = a peeled copy of the two lines") and priced the kernel at 908 632 multiply-adds per proof; the two
        # pieces are the straight-line kernels k_f12_sqr (7425) and k_f12_mul_line_fixed2 (11 413), whose static counts need no model, and a pass executes
        # 64 x 7425 + 88 x 11 413 = 1 479 544 of them (profiles/r05_plonk262144_onepass_kernel_stats.csv: 15.2 ms per 262 144 proofs = 0.77 of the peak, as k_miller_run).
        lines, sqr = c, len(mads) - c
        assert 7000 <= sqr <= 8000 and 10500 <= lines <= 12500, ("k_miller_run_fixed2: layout changed", c, sqr, lines)
        total = 64 * sqr + 88 * lines
        return {"static_mads": len(mads), "components": {"sqr": sqr, "two_lines": lines}, "mads_per_proof_launch": float(total), "mads_per_proof_batch": float(total), "per_pass": True,
                "unmodelled": [], "model": "Miller loop of two table-driven pairs in one launch: squaring of f (%d multiply-adds) x64 + two line products (%d) x88" % (sqr, lines)}
    return None


def model_kernel(name, ins):
    mads = [a for a, t, _ in ins if MAD.match(t)]
    valu = sum(1 for _, t, _ in ins if VALU.match(t))
    groups = [(h, l, count_in(mads, h, l[-1])) for h, l in loop_groups(ins, mads)]
    groups = [g for g in groups if g[2] > 0]
    notes, weight_ranges = [], []   # (lo, hi, factor): instructions in [lo, hi] are multiplied by factor (innermost-first, nested factors multiply)
    unmodelled = []
    prepare_windows = None
    if name == "k_g16_prepare":
        # the public-input MSM of the 2-input path: n_public x 20 windows of 13 bits (bn254_fw.h), a table addition unless the digit is zero.  The software-pipelined loop (the next
        # window's table entry in flight during the addition) is laid out with several back edges over ONE copy of the addition: the union of those regions
        # executes n_public x 20 times, however the compiler nests them
        w = [(h, l[-1]) for h, l, c in groups if _in(c, R_MIXED)]
        if w:
            prepare_windows = (min(a for a, _ in w), max(b for _, b in w))
            assert _in(count_in(mads, *prepare_windows), R_MIXED), ("k_g16_prepare: more than one addition in the window loops", w)
            weight_ranges.append((prepare_windows[0], prepare_windows[1], N_PUBLIC * (19.0 * 8191.0 / 8192.0 + 511.0 / 512.0)))
            notes.append("window loop: %d inputs x 20 windows of 13 bits (the top one 9 bits), table addition (%d mads) unless the digit is zero" % (N_PUBLIC, count_in(mads, *prepare_windows)))
    for h, latches, c in groups:
        if prepare_windows and prepare_windows[0] <= h and latches[-1] <= prepare_windows[1]:
            continue
        inner = [g for g in groups if g is not (h, latches, c) and h <= g[0] and g[1][-1] <= latches[-1] and (g[0], g[1][-1]) != (h, latches[-1])]
        own = c - sum(g[2] for g in inner if not any(o is not g and o[0] <= g[0] and g[1][-1] <= o[1][-1] and o in inner for o in inner))
        first = count_in(mads, h, latches[0])
        if len(latches) == 2 and first == FP_SQR and c == FP_SQR + FP_MUL:
            # fp_pow_bits with the exponent p - 2: a squaring per bit, a product on set bits (second latch region)
            weight_ranges.append((h, latches[0], float(INV_SQUARINGS)))
            weight_ranges.append((latches[0] + 1, latches[1], float(INV_SET_BITS)))
            notes.append("Fermat inversion: %d squarings + %d products" % (INV_SQUARINGS, INV_SET_BITS))
        elif name == "k_g16_msm_partial_comb":
            # comb tables (bn254_kernels.hip): 20 columns x 16 inputs per lane; the compiler lays the column loop out as a doubling region followed
            # by the input loop, so the two pieces are weighted directly: 19 doublings (none in the top column), 320 table additions unless the
            # 13-bit column digit is zero
            if 2650 <= c <= 3150:
                # round 4: ONE software-pipelined loop over the 320 (column, input) pairs: a doubling in front of a column's first addition (19 of the 320
                # trips), a table addition unless the 13-bit column digit is zero -- two forward-branched regions inside the loop
                weight_ranges.append((h, latches[-1], 320.0))
                regs = []
                for a, text, tgt in ins:
                    if h <= a <= latches[-1] and tgt is not None and a < tgt <= latches[-1] + 64 and text.startswith("s_cbranch"):
                        m = count_in(mads, a + 1, tgt - 1)
                        if _in(m, R_DBL): regs.append((a + 1, tgt - 1, 19.0 / 320.0, "19 doublings (%d mads each)" % m))
                        elif _in(m, R_MIXED): regs.append((a + 1, tgt - 1, 8191.0 / 8192.0, "320 table additions (%d mads each) unless the column digit is zero" % m))
                assert len(regs) == 2, ("k_g16_msm_partial_comb: a doubling and an addition region expected in the loop", regs)
                for lo_, hi_, f_, note_ in regs:
                    weight_ranges.append((lo_, hi_, f_)); notes.append(note_)
            elif _in(c, R_MIXED):
                weight_ranges.append((h, latches[-1], 320.0 * 8191.0 / 8192.0)); notes.append("320 table additions (%d mads each) unless the column digit is zero" % c)
            elif _in(c, R_DBL) and not any(r[0] <= h and latches[-1] <= r[1] or h <= r[0] and r[1] <= latches[-1] for r in weight_ranges if r[2] == 19.0):
                weight_ranges.append((h, latches[-1], 19.0)); notes.append("19 doublings (%d mads each)" % c)
            elif own > 0:
                unmodelled.append("loop +0x%x..+0x%x (%d mads) of k_g16_msm_partial_comb not recognised" % (h - ins[0][0], latches[-1] - ins[0][0], c))
        elif _in(c, R_MIXED) and not inner:
            weight_ranges.append((h, latches[-1], 32.0 * 255.0 / 256.0))
            notes.append("byte-window loop: 32 windows per scalar, table addition (%d mads) unless the byte is zero" % c)
        elif _in(c, R_MIXED) and inner:
            t, why = TRIPS_BY_KERNEL.get((name, "window_outer"), (1, "UNMODELLED outer window loop"))
            weight_ranges.append((h, latches[-1], float(t))); notes.append("outer window loop x%g: %s" % (t, why))
        elif name == "k_g1_scalar_mul" and _in(c, R_ADD) and not inner:
            weight_ranges.append((h, latches[-1], 9.0)); notes.append("table of the two-bit windows: 9 sums i P1 + j P2 (%d mads each)" % c)
        elif name == "k_g1_scalar_mul" and 40 <= c <= 60 and not inner:
            weight_ranges.append((h, latches[-1], 3.0)); notes.append("table rows i P1, j P2 stored (x3)")
        elif name == "k_g1_scalar_mul" and _in(c, R_DBL) and not inner:
            weight_ranges.append((h, latches[-1], 64.0)); notes.append("64 doublings of the high half (%d mads each; split launch, high lanes only)" % c)
        elif _in(c, R_STEP) or (name in ("k_rlc_scale", "k_g1_scalar_mul") and 2700 <= c <= 3400):
            t, why = TRIPS_BY_KERNEL.get((name, "scalar_mul"), (1, "UNMODELLED scalar multiplication loop"))
            if name == "k_g1_scalar_mul":
                split = any(_in(g[2], R_DBL) and not [x for x in groups if x is not g and g[0] <= x[0] and x[1][-1] <= g[1][-1]] for g in groups)
                w2 = c >= 4300          # two doublings + one addition per step (bn254_rlc.h::g1_mul_glv_w2): two bits of each half
                t = (32 if split else 64) if w2 else (64 if split else 128)
                why = "%s%d steps per lane%s" % ("two-bit windows: " if w2 else "", t, " (split launch: low / high half of the GLV halves)" if split else "")
            weight_ranges.append((h, latches[-1], float(t))); notes.append("2-bit window loop x%g (%d mads per window): %s" % (t, c, why))
        elif c == 224:
            t, why = TRIPS_BY_KERNEL.get((name, "fr_products"), (1, "UNMODELLED"))
            weight_ranges.append((h, latches[-1], float(t))); notes.append("Fr product loop x%g: %s" % (t, why))
        elif name == "k_f12_cyclo_sqr_n":
            t, why = TRIPS_BY_KERNEL[(name, "cyclo_sqr")]
            weight_ranges.append((h, latches[-1], float(t))); notes.append("squaring loop x%.3f (%d mads per squaring): %s" % (t, c, why))
        elif c == INV_GCD_ROUND_MADS and not inner:
            # fp_inv (bn254_fp.h): 18 rounds; per round the 29-step stand-in loop (no multiply-adds: only the other instruction classes see it)
            weight_ranges.append((h, latches[-1], 18.0))
            for h2, l2 in loop_groups(ins, mads):
                if h < h2 and l2[-1] < latches[-1] and count_in(mads, h2, l2[-1]) == 0:
                    weight_ranges.append((h2, l2[-1], 29.0))
            notes.append("binary-GCD inversion: 18 rounds of 29 steps (%d multiply-adds per round)" % c)
        elif own > 0:
            unmodelled.append("loop +0x%x..+0x%x (%d mads, %d of its own) counted once" % (h - ins[0][0], latches[-1] - ins[0][0], c, own))
    total = 0.0
    for a in mads:
        w = 1.0
        for lo, hi, f in weight_ranges:
            if lo <= a <= hi:
                w *= f
        total += w
    # psi / psi^2 regions of the addition steps: forward-conditional regions holding >= one Fp2 product that only 2 of the 23 launches run
    if name == "k_miller_step_add":
        conds = [(addr + 1, tgt) for addr, text, tgt in ins if tgt is not None and tgt > addr and text.startswith("s_cbranch")]
        span = ins[-1][0] - ins[0][0]
        psi = set()
        for lo, hi in conds:
            if (hi - lo) < 0.5 * span:
                psi.update(a for a in mads if lo <= a < hi)
        total -= len(psi) * (1.0 - 2.0 / 23.0)
        notes.append("psi / psi^2 maps (%d mads) run in 2 of the 23 launches" % len(psi))
    return {"static_mads": len(mads), "valu_instructions_static": valu, "mads_per_proof_launch": total, "model": "; ".join(notes) if notes else "straight-line: static count",
            "unmodelled": unmodelled, "weights": weight_ranges}


# ---- cooperative small-batch kernels (bn254_coop12.hip): one launch = public-input MSM + Miller loop + final exponentiation, operations of the
# final exponentiation are out-of-line device functions.  Dynamic count per WAVEFRONT = sum over address ranges of (static count x executions):
# the Miller loop's ranges are recognised by their multiply-add signature, the executions come from the step program (bn254_vm.h /
# bn254_constants.h: 65 doubling steps, 64 of them with the squaring of f, 23 addition steps) and from vm_final_exp_program / vm_exp_u.  The same
# weights applied to (v_mad_64 + v_lshl_add_u64 + v_ashrrev_i64) and to all VALU instructions predict the counters SQ_INSTS_VALU_INT64 and
# SQ_INSTS_VALU of a rocprofv3 --pmc run (tools/gpu_round2.sh), which is how the model is checked: `pmc_check` in the output.
INT64 = re.compile(r"^\s*(v_mad_[iu]64_[iu]32|v_lshl_add_u64|v_ashrrev_i64)\b")     # what SQ_INSTS_VALU_INT64 counts (calibrated on the straight-line kernels)
CLASSES = {"mads": MAD, "int64": INT64, "valu": VALU}


def u_w4_digits():
    src = open(os.path.join(ROOT, "snark-bn254-verifier_amd", "csrc", "bn254_constants.h")).read()
    m = re.search(r"BN_U_W4\[\d+\]\s*=\s*\{([^}]*)\}", src)
    return [int(x) for x in m.group(1).split(",")]


def final_exp_ops():
    """Operation counts of vm_final_exp_program (bn254_vm.h): three vm_exp_u (one squaring + 3 table products + per non-zero digit after the
    first a run of squarings and a product) and the fixed part."""
    d = u_w4_digits()
    nz = sum(1 for x in d[1:] if x != 0)
    return {"inv": 1, "conj": 5, "frob": 4, "mul": 12 + 3 * (3 + nz), "cyclo_calls": 3 + 3 * (1 + nz), "cyclo_squarings": 3 + 3 * (1 + (len(d) - 1))}


def wcount(ins, ranges, rx):
    t = 0.0
    for a, text, _ in ins:
        if rx.match(text):
            w = 1.0
            for lo, hi, f in ranges:
                if lo <= a <= hi:
                    w *= f
            t += w
    return t


def counts(ins, ranges):
    return {k: wcount(ins, ranges, rx) for k, rx in CLASSES.items()}


def add(a, b, f=1.0):
    return {k: a.get(k, 0.0) + f * b[k] for k in b}


def loops_of(ins):
    mads = [a for a, t, _ in ins if MAD.match(t)]
    return [(h, l, count_in(mads, h, l[-1])) for h, l in loop_groups(ins, mads)], mads


def callee_straight(funcs, frag):
    sym = [k for k in funcs if frag in k]
    assert len(sym) == 1, (frag, sym)
    return funcs[sym[0]]


def coop12_callees(funcs, notes):
    """Dynamic counts per call of the out-of-line operations."""
    ops = final_exp_ops()
    out = {}
    for name in ("c12_mul", "c12_conj", "c12_frob"):
        ins = callee_straight(funcs, "%d%sENS" % (len(name), name))
        assert not [1 for a, t, tg in ins if tg is not None and tg <= a and MAD.match("v_mad_u64_u32") and count_in([x for x, tt, _ in ins if MAD.match(tt)], tg, a) > 0], name
        out[name] = counts(ins, [])
    # Granger-Scott squarings: prologue + count x loop body
    ins = callee_straight(funcs, "15c12_cyclo_sqr_nENS")
    lg, _ = loops_of(ins)
    lg = [g for g in lg if g[2] > 0]
    assert len(lg) == 1, "c12_cyclo_sqr_n: one squaring loop expected"
    h, latches, c = lg[0]
    per_call = counts(ins, [(h, latches[-1], 0.0)])
    per_sq = add({}, counts(ins, []), 1.0)
    per_sq = {k: per_sq[k] - per_call[k] for k in per_sq}
    out["cyclo_call"], out["cyclo_sq"] = per_call, per_sq
    notes.append("c12_cyclo_sqr_n: %d multiply-adds per squaring" % per_sq["mads"])
    # inversion: fp12_inv with its Fermat loop (same recognition as k_f12_inv) + the out-of-line fp6 products it calls
    ins = callee_straight(funcs, "7c12_invENS")
    e = model_kernel("c12_inv", ins)
    assert not e["unmodelled"], e["unmodelled"]
    calls = sum(1 for _, t, _ in ins if t.startswith("s_swappc"))
    f6 = counts(callee_straight(funcs, "10fp6_mul_nlE"), [])
    out["c12_inv"] = add(counts(ins, e["weights"]), f6, calls)
    notes.append("c12_inv: %s + %d fp6_mul_nl calls" % (e["model"], calls))
    # public-input MSM: ceil(20 n_public / 12) window additions per lane (the wavefront runs the longest lane), then the tree
    ins = callee_straight(funcs, "20c12_public_input_msmENS")
    lg, _ = loops_of(ins)
    lg = [g for g in lg if g[2] > 1000]
    assert len(lg) == 1, "c12_public_input_msm: one window loop expected"
    trips = -(-20 * N_PUBLIC // 12)
    out["msm"] = counts(ins, [(lg[0][0], lg[0][1][-1], float(trips))])
    notes.append("c12_public_input_msm: window loop x%d (%d multiply-adds per table addition)" % (trips, lg[0][2]))
    return out, ops


def coop12_kernel(funcs, frag, kind, callees, ops, n_pairs=2):
    """kind 'g16': k_coop12_miller_g16; 'fixed': k_coop12_miller_fixed with n_pairs table-driven pairs."""
    ins = callee_straight(funcs, frag)
    lg, mads = loops_of(ins)
    h, latches, c = max(lg, key=lambda g: g[2])              # the Miller loop: the loop holding the most multiply-adds
    lo, hi = h, latches[-1]
    conds = []
    for a, text, tgt in ins:
        if lo <= a <= hi and tgt is not None and a < tgt <= hi and text.startswith("s_cbranch"):
            m = count_in(mads, a + 1, tgt - 1)
            if m >= 100:
                conds.append((a + 1, tgt - 1, m))
    steps = 88.0
    ranges = [(lo, hi, steps)]
    notes = []
    if kind == "g16":
        # two layouts of the same loop: the doubling rounds as the fall-through between two latches, or as a third forward-branched block
        assert (len(latches), len(conds)) in ((2, 2), (1, 3)), ("k_coop12_miller_g16: layout changed", latches, conds)
        if len(conds) == 2:
            (s_lo, s_hi, s_m), (a_lo, a_hi, a_m) = conds
            d_lo, d_hi = latches[0] + 1, latches[1]
            d_m = count_in(mads, d_lo, d_hi)
        else:
            (s_lo, s_hi, s_m), (a_lo, a_hi, a_m), (d_lo, d_hi, d_m) = conds
        assert 700 <= s_m <= 800 and 950 <= a_m <= 1000 and 800 <= d_m <= 900, (s_m, a_m, d_m)
        ranges += [(s_lo, s_hi, 64.0 / steps), (a_lo, a_hi, 23.0 / steps), (d_lo, d_hi, 65.0 / steps)]
        notes.append("Miller loop x88: squaring of f (%d multiply-adds) x64, G2 addition rounds (%d) x23, G2 doubling rounds (%d) x65, three line products (%d) x88"
                     % (s_m, a_m, d_m, count_in(mads, lo, hi) - s_m - a_m - d_m))
    else:
        # sqr (conditional), pair 0, pair 1 (conditional: n_pairs > 1), first latch (taken when n_pairs <= 2), pair 2, second latch
        assert len(latches) == 2 and len(conds) == 2, ("k_coop12_miller_fixed: layout changed", latches, conds)
        (s_lo, s_hi, s_m), (p1_lo, p1_hi, p1_m) = conds
        p2_lo, p2_hi = latches[0] + 1, latches[1]
        p2_m = count_in(mads, p2_lo, p2_hi)
        assert 700 <= s_m <= 800 and 800 <= p1_m <= 900 and p1_m == p2_m, (s_m, p1_m, p2_m)
        ranges += [(s_lo, s_hi, 64.0 / steps), (p1_lo, p1_hi, 1.0 if n_pairs > 1 else 0.0), (p2_lo, p2_hi, 1.0 if n_pairs > 2 else 0.0)]
        notes.append("Miller loop x88: squaring of f (%d multiply-adds) x64, line product of a table-driven pair (%d) x88 for each of the %d pairs"
                     % (s_m, p1_m, n_pairs))
    tot = counts(ins, ranges)
    if kind == "g16":
        tot = add(tot, callees["msm"])
    tot = add(tot, callees["c12_inv"], ops["inv"])
    tot = add(tot, callees["c12_conj"], ops["conj"])
    tot = add(tot, callees["c12_frob"], ops["frob"])
    tot = add(tot, callees["c12_mul"], ops["mul"])
    tot = add(tot, callees["cyclo_call"], ops["cyclo_calls"])
    tot = add(tot, callees["cyclo_sq"], ops["cyclo_squarings"])
    notes.append("final exponentiation: %(inv)d inversion, %(mul)d products, %(cyclo_squarings)d cyclotomic squarings in %(cyclo_calls)d calls, %(frob)d Frobenius maps, %(conj)d conjugations" % ops)
    return tot, notes


def model_coop12(funcs):
    notes = []
    callees, ops = coop12_callees(funcs, notes)
    pmc = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "coop12_pmc_counts.json")))
    except Exception:
        pass
    out = {}
    for name, frag, kind in (("k_coop12_miller_g16", "19k_coop12_miller_g16E", "g16"), ("k_coop12_miller_fixed", "21k_coop12_miller_fixedE", "fixed")):
        tot, n2 = coop12_kernel(funcs, frag, kind, callees, ops)
        e = {"lanes_per_proof": 12, "proofs_per_wavefront": 5,
             "wavefront_instructions": {k: round(v, 1) for k, v in tot.items()},
             "mads_per_proof_launch": tot["mads"] * 12,
             "static_mads": sum(1 for _, t, _ in callee_straight(funcs, frag) if MAD.match(t)),
             "model": "; ".join(n2 + notes) + "; per proof = 12 lanes x the wavefront's count (lanes 60..63 idle)", "unmodelled": [], "symbol": frag}
        if pmc and name in pmc:
            m = pmc[name]
            e["pmc_check"] = {"SQ_INSTS_VALU_INT64_per_wavefront": m["SQ_INSTS_VALU_INT64"], "model_int64": round(tot["int64"], 1),
                              "SQ_INSTS_VALU_per_wavefront": m["SQ_INSTS_VALU"], "model_valu": round(tot["valu"], 1),
                              "int64_error": tot["int64"] / m["SQ_INSTS_VALU_INT64"] - 1.0, "valu_error": tot["valu"] / m["SQ_INSTS_VALU"] - 1.0}
        out[name] = e
    return out


def main():
    with tempfile.TemporaryDirectory() as wd:
        per_co = [disassemble(co) for co in extract_code_objects(wd)]
    funcs = collections.OrderedDict()
    for d in per_co:
        for sym, ins in d.items():
            funcs.setdefault(sym, ins)      # out-of-line helpers that several translation units carry (fp6_mul_nl) are identical copies
    kernels, inst = {}, {}
    for sym, ins in funcs.items():
        name = short(sym)
        if not name.startswith("k_") or not ins:
            continue
        if name.startswith("k_coop"):
            continue                          # cooperative kernels: model_coop12 (call graph + step program)
        if name in ("k_valu_peak", "k_plonk_stage1", "k_plonk_stage2", "k_plonk_dbg_zeta"):
            continue                          # the measurement kernel; the PlonK stages (transcripts + Fr arithmetic on 32-bit words: < 1 % of a proof's multiply-adds, counted by PMC only)
        e = model_components(name, ins) or model_kernel(name, ins)
        e.pop("weights", None)
        e["symbol"] = sym
        inst.setdefault(name, []).append(e)
    for name, es in inst.items():
        if len(es) == 1:
            kernels[name] = es[0]
        elif name == "k_miller_step_dbl":
            # template instances of one kernel kind: <true> runs 64 of the 65 doubling steps, <false> the first (no squaring of f = 1)
            a, b = sorted(es, key=lambda e: -e["static_mads"])
            e = dict(a)
            e["mads_per_proof_launch"] = (64 * a["mads_per_proof_launch"] + b["mads_per_proof_launch"]) / 65.0
            e["model"] = "64 launches with the squaring of f (%d mads) + the first step without (%d)" % (a["static_mads"], b["static_mads"])
            kernels[name] = e
        elif name == "k_g1_scalar_mul":
            # template instances: <false> one lane per term; <true> two lanes per term (the count is the HIGH lane's chain, the launch's duration)
            for e in es:
                kernels[name + ("_split" if "split launch" in e["model"] else "") + ("_w2" if "two-bit windows" in e["model"] else "")] = e
        else:
            for i, e in enumerate(es):
                kernels["%s#%d" % (name, i)] = e
    if "k_miller_run" in kernels and "k_miller_step_dbl" in kernels and "k_miller_step_add" in kernels:
        # the whole Miller loop in one launch (bn254_vm.h::vm_miller_run): the same field operations as the one-launch-per-step kernels -- 64 doubling
        # steps with the squaring of f, the first one without, 23 additions (2 of them with a psi map) -- inside one loop with a wave-uniform branch per
        # step kind; the loads / stores of f between the steps are what is gone, and they carry no multiply-adds.  The sum is CHECKED against the
        # counter: profiles/miller_run_pmc_counts.json holds SQ_INSTS_VALU_INT64 / SQ_INSTS_VALU per wavefront from a rocprofv3 --pmc pass.
        run = kernels["k_miller_run"]
        a, b = sorted(inst["k_miller_step_dbl"], key=lambda e: -e["static_mads"])
        add = kernels["k_miller_step_add"]
        total = 64 * a["mads_per_proof_launch"] + b["mads_per_proof_launch"] + 23 * add["mads_per_proof_launch"]
        run["mads_per_proof_launch"] = total
        run["mads_per_proof_batch"] = total
        run["per_pass"] = True      # the 88 steps run in 1, 2, 4 or 8 launches depending on the sub-batch size: the count is per pass over a sub-batch
        run["unmodelled"] = []
        run["model"] = ("whole Miller loop in one launch = 64 doubling steps with the squaring of f (%d multiply-adds, k_miller_step_dbl<true>) + the first without (%d) + "
                        "23 addition steps (%.1f on average, k_miller_step_add); static multiply-adds of the loop body with both branches: %d"
                        % (a["static_mads"], b["static_mads"], add["mads_per_proof_launch"], run["static_mads"]))
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "miller_run_pmc_counts.json")))
            # per wavefront (64 proofs = one lane each): the multiply-adds are a known share of the 64-bit integer instructions of the step kernels
            run["pmc_check"] = {"SQ_INSTS_VALU_per_wavefront": pmc["SQ_INSTS_VALU"], "SQ_INSTS_VALU_INT64_per_wavefront": pmc["SQ_INSTS_VALU_INT64"],
                                "model_mads_per_lane": total, "note": pmc.get("note", "")}
        except Exception:
            pass
    if "k_f12_cyclo_sqr_n" in kernels:
        e = kernels["k_f12_cyclo_sqr_n"]
        e["mads_per_proof_batch"] = e["mads_per_proof_launch"] * 39     # all 39 launches of a batch together (exact: 186 squarings)
    # keys with many public inputs: one lane per (proof, 16-input chunk); the per-proof figure is for BASELINE configs[4] (1024 inputs: 64 lanes)
    for name in ("k_g16_msm_partial", "k_g16_msm_partial_comb"):
        if name in kernels:
            e = kernels[name]
            e["mads_per_lane"] = e["mads_per_proof_launch"]; e["lanes_per_proof"] = 64
            e["mads_per_proof_launch"] = e["mads_per_lane"] * 64
            e["model"] += "; per proof = 64 lanes (1024 public inputs, 16 per lane)"
    kernels.update(model_coop12(funcs))
    out = {"_note": "v_mad_[iu]64_[iu]32 executed per proof (lane) and launch, from the gfx950 code object of libbn254_verify_amd.so; written by "
                    "tools/count_mads.py (loop trip counts and their sources: the `model` strings; n_public = %d)" % N_PUBLIC,
           "kernels": kernels}
    with open(os.environ.get("COUNT_MADS_OUT") or os.path.join(ROOT, "profiles", "kernel_mads.json"), "w") as f:     # COUNT_MADS_OUT: tests/test_capi_cpu.py compares with the record
        json.dump(out, f, indent=1, sort_keys=True)
    for k, e in sorted(kernels.items(), key=lambda kv: -kv[1]["mads_per_proof_launch"]):
        print("%-24s static %6d  dynamic %10.1f  %s%s" % (k, e["static_mads"], e["mads_per_proof_launch"], e["model"][:110], ("  !! " + "; ".join(e["unmodelled"])) if e["unmodelled"] else ""))


if __name__ == "__main__":
    main()
