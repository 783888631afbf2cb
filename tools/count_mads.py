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

# exponent bit patterns (snark-bn254-verifier_amd/csrc/bn254_constants.h): fp_pow_bits squares once per bit after the first and multiplies
# on set bits; the multiplication sits in a conditional region inside the loop
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
PM2_BITS = bin(P - 2)[2:]
INV_SQUARINGS = len(PM2_BITS) - 1
INV_MUL_FRACTION = (PM2_BITS.count("1") - 1) / (len(PM2_BITS) - 1)

def extract_code_object(workdir):
    lib = os.path.join(workdir, "lib.so")
    shutil.copy(LIB, lib)
    subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", lib], stdout=subprocess.DEVNULL, cwd=workdir)
    for f in os.listdir(workdir):
        if "amdgcn" in f:
            return os.path.join(workdir, f)
    raise SystemExit("no gfx950 code object found in " + LIB)


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


def model_kernel(name, ins):
    mads = [a for a, t, _ in ins if MAD.match(t)]
    valu = sum(1 for _, t, _ in ins if VALU.match(t))
    groups = [(h, l, count_in(mads, h, l[-1])) for h, l in loop_groups(ins, mads)]
    groups = [g for g in groups if g[2] > 0]
    notes, weight_ranges = [], []   # (lo, hi, factor): instructions in [lo, hi] are multiplied by factor (innermost-first, nested factors multiply)
    unmodelled = []
    for h, latches, c in groups:
        inner = [g for g in groups if g is not (h, latches, c) and h <= g[0] and g[1][-1] <= latches[-1] and (g[0], g[1][-1]) != (h, latches[-1])]
        own = c - sum(g[2] for g in inner if not any(o is not g and o[0] <= g[0] and g[1][-1] <= o[1][-1] and o in inner for o in inner))
        first = count_in(mads, h, latches[0])
        if len(latches) == 2 and first == FP_SQR and c == FP_SQR + FP_MUL:
            # fp_pow_bits with the exponent p - 2: a squaring per bit, a product on set bits (second latch region)
            weight_ranges.append((h, latches[0], float(INV_SQUARINGS)))
            weight_ranges.append((latches[0] + 1, latches[1], float(INV_SET_BITS)))
            notes.append("Fermat inversion: %d squarings + %d products" % (INV_SQUARINGS, INV_SET_BITS))
        elif 1700 <= c <= 1900 and not inner:
            weight_ranges.append((h, latches[-1], 32.0 * 255.0 / 256.0))
            notes.append("byte-window loop: 32 windows per scalar, table addition (%d mads) unless the byte is zero" % c)
        elif 1700 <= c <= 1900 and inner:
            t, why = TRIPS_BY_KERNEL.get((name, "window_outer"), (1, "UNMODELLED outer window loop"))
            weight_ranges.append((h, latches[-1], float(t))); notes.append("outer window loop x%g: %s" % (t, why))
        elif 4300 <= c <= 4600 or (name in ("k_rlc_scale", "k_g1_scalar_mul") and 3000 <= c <= 3400):
            t, why = TRIPS_BY_KERNEL.get((name, "scalar_mul"), (1, "UNMODELLED scalar multiplication loop"))
            weight_ranges.append((h, latches[-1], float(t))); notes.append("2-bit window loop x%g (%d mads per window): %s" % (t, c, why))
        elif c == 224:
            t, why = TRIPS_BY_KERNEL.get((name, "fr_products"), (1, "UNMODELLED"))
            weight_ranges.append((h, latches[-1], float(t))); notes.append("Fr product loop x%g: %s" % (t, why))
        elif name == "k_f12_cyclo_sqr_n":
            t, why = TRIPS_BY_KERNEL[(name, "cyclo_sqr")]
            weight_ranges.append((h, latches[-1], float(t))); notes.append("squaring loop x%.3f (%d mads per squaring): %s" % (t, c, why))
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
            "unmodelled": unmodelled}


def main():
    with tempfile.TemporaryDirectory() as wd:
        funcs = disassemble(extract_code_object(wd))
    kernels, inst = {}, {}
    for sym, ins in funcs.items():
        name = short(sym)
        if not name.startswith("k_") or not ins:
            continue
        e = model_kernel(name, ins)
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
        else:
            for i, e in enumerate(es):
                kernels["%s#%d" % (name, i)] = e
    if "k_f12_cyclo_sqr_n" in kernels:
        e = kernels["k_f12_cyclo_sqr_n"]
        e["mads_per_proof_batch"] = e["mads_per_proof_launch"] * 39     # all 39 launches of a batch together (exact: 186 squarings)
    out = {"_note": "v_mad_[iu]64_[iu]32 executed per proof (lane) and launch, from the gfx950 code object of libbn254_verify_amd.so; written by "
                    "tools/count_mads.py (loop trip counts and their sources: the `model` strings; n_public = %d)" % N_PUBLIC,
           "kernels": kernels}
    with open(os.path.join(ROOT, "profiles", "kernel_mads.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for k, e in sorted(kernels.items(), key=lambda kv: -kv[1]["mads_per_proof_launch"]):
        print("%-24s static %6d  dynamic %10.1f  %s%s" % (k, e["static_mads"], e["mads_per_proof_launch"], e["model"][:110], ("  !! " + "; ".join(e["unmodelled"])) if e["unmodelled"] else ""))


if __name__ == "__main__":
    main()
