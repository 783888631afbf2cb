#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/gpu_round3.sh (gpurun_out/r03/) into the tracked summaries under profiles/ (r03_*).
usage: python tools/summarize_r03.py [src_dir_under_repo_root]"""
import collections, csv, glob, json, os, shutil, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r03"
out = os.path.join(root, "profiles")
tag = "r03"


def short(name):
    n = name.split("(")[0].replace("bn254::", "").strip()
    if n.startswith("void "):
        n = n[5:]
    return n.split("<")[0].strip()


def G(pattern):
    r = glob.glob(os.path.join(root, src, pattern), recursive=True)
    return r[0] if r else None


# ---- kernel trace of the bench command: per-kernel calls / total / average
tr = G("prof/**/*kernel_trace.csv")
agg = collections.defaultdict(list)
rows = list(csv.DictReader(open(tr)))
for r in rows:
    agg[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in agg.values())
with open(os.path.join(out, tag + "_kernel_stats.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rlc --no-configs (3 batches of 2^20; 2 sub-batch streams => launches cover 2^19 proofs;\n"
            "# k_miller_run = the whole Miller loop of a sub-batch in ONE launch: the two streams' launches run side by side (about 100 ms each) or, when one stream gets the\n"
            "# GPU first, one after the other (about 60 ms each): see r03_miller_run_timeline.txt)\n")
    f.write("kernel,calls,total_ms,avg_us,min_us,max_us,percent\n")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        f.write("%s,%d,%.3f,%.2f,%.2f,%.2f,%.2f\n" % (k, len(v), sum(v) / 1e6, sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3, 100.0 * sum(v) / tot))
st = G("prof/**/*kernel_stats.csv")
if st:
    shutil.copy(st, os.path.join(out, tag + "_rocprof_kernel_stats_raw.csv"))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
with open(os.path.join(out, tag + "_miller_run_timeline.txt"), "w") as f:
    f.write("# start / end (ms since the first kernel) of the phase-delimiting kernels of the three batches of the kernel trace, per hardware queue (= sub-batch stream)\n")
    for r in rows:
        n = short(r["Kernel_Name"])
        if n in ("k_g16_prepare", "k_miller_run", "k_g16_subgroup", "k_g16_compare"):
            f.write("%-16s queue %s  %9.3f -> %9.3f  (%8.3f ms)\n" % (n, r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6,
                                                                      (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
for name in ("bench.json", "prof_bench.json", "bench_torchrun.json", "bench_steps.json"):
    p = os.path.join(root, src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(out, tag + "_" + name))


# ---- PMC passes (batch 2^18, one stream)
def load(pattern):
    path = G(pattern)
    rows = list(csv.DictReader(open(path)))
    a = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); waves = {}
    first = rows[0]["Counter_Name"]
    for r in rows:
        k = short(r["Kernel_Name"])
        if not k.startswith("k_"):
            continue
        a[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == first:
            cnt[k] += 1
            waves[k] = int(r["Grid_Size"]) // 64
    return a, cnt, waves


sq, cnt, waves = load("pmc_SQ_WAVE_CYCLES/**/*counter_collection.csv")
fs, _, _ = load("pmc_FETCH_SIZE/**/*counter_collection.csv")
ws, _, _ = load("pmc_WRITE_SIZE/**/*counter_collection.csv")
ifs, _, _ = load("pmc_SQ_IFETCH/**/*counter_collection.csv")
ic, _, _ = load("pmc_SQC_ICACHE_REQ/**/*counter_collection.csv")
n = 1 << 18
old = {}
try:
    old = json.load(open(os.path.join(out, "pmc_traffic.json")))
except Exception:
    pass
traffic = {k: v for k, v in old.items() if not k.startswith("_")}
traffic["_note"] = ("HBM bytes per proof and launch of each kernel kind from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, batch 2^18, one stream, bench.py --steps 1 "
                    "--warmup 0); FETCH_SIZE (KB) doubled as MI355X_MICROARCH.md prescribes for gfx950 (checked on k_f12_sqr, whose reads are exactly 432 B/proof), WRITE_SIZE "
                    "(KB) as reported.  Round 3 rows (tools/summarize_r03.py) replace the round 2 rows of the same kernels; k_miller_run: bytes per pass over a sub-batch (the whole Miller loop, however many launches)")
lines = ["kernel,launches,valu_active_frac,any_active_frac,wait_any_frac,wait_inst_frac,valu_insts_per_wave,int64_insts_per_wave,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch,hbm_read_B_per_proof(2xFETCH),hbm_write_B_per_proof,"
         "ifetch_per_wave,icache_req_per_launch,icache_hit_frac,icache_miss_frac(incl_duplicates)"]
for k in sorted(sq, key=lambda k: -sq[k]["SQ_WAVE_CYCLES"]):
    c = sq[k]; wc = c["SQ_WAVE_CYCLES"] or 1; nl = max(cnt[k], 1); w = max(waves.get(k, 1), 1)
    f_ = fs[k]["FETCH_SIZE"] / nl; w_ = ws[k]["WRITE_SIZE"] / nl
    rd = 2 * f_ * 1024 / n; wr = w_ * 1024 / n
    traffic[k] = {"read_bytes_per_proof": round(rd, 1), "write_bytes_per_proof": round(wr, 1)}
    if k == "k_miller_run":
        # the 88 steps run in 1, 2, 4 or 8 launches depending on the sub-batch size: bytes per PASS over a sub-batch (all its launches), as kernel_mads.json counts it
        traffic[k] = {"read_bytes_per_proof": round(rd * nl, 1), "write_bytes_per_proof": round(wr * nl, 1), "per_pass": True, "launches_in_this_run": nl}
    req = ic[k]["SQC_ICACHE_REQ"] or 1
    lines.append("%s,%d,%.3f,%.3f,%.3f,%.3f,%.0f,%.0f,%.0f,%.0f,%.0f,%.0f,%.0f,%.0f,%.4f,%.4f" % (
        k, cnt[k], c["SQ_ACTIVE_INST_VALU"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc,
        c["SQ_INSTS_VALU"] / nl / w, c["SQ_INSTS_VALU_INT64"] / nl / w, f_, w_, rd, wr, ifs[k]["SQ_IFETCH"] / nl / w, req / nl,
        ic[k]["SQC_ICACHE_HITS"] / req, (ic[k]["SQC_ICACHE_MISSES"] + ic[k]["SQC_ICACHE_MISSES_DUPLICATE"]) / req))
open(os.path.join(out, tag + "_pmc_summary.csv"), "w").write(
    "# rocprofv3 --pmc passes (SQ activity | FETCH_SIZE | WRITE_SIZE | SQ_IFETCH | SQC_ICACHE_*), each its own run of: BN254_STREAMS=1 python3 bench.py --steps 1 --warmup 0 "
    "--no-cpu-baseline --no-rlc --no-configs --batch-log2 18 (one batch of 2^18 proofs, one stream)\n" + "\n".join(lines) + "\n")
json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
# VALU instructions per wavefront (= per lane = per proof) and launch of every kernel kind, all of them and the 64-bit integer class (multiply-adds, 64-bit shifts / adds):
# bench.py prices them at the measured issue rates for its `valu_issue_bound` (k_miller_run: per pass over a sub-batch)
valu = {"_note": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 (batch 2^18, one stream): instructions per wavefront and launch; k_miller_run per pass (all its launches)"}
for k in sq:
    nl = max(cnt[k], 1); w = max(waves.get(k, 1), 1)
    per = 1 if k == "k_miller_run" else nl
    valu[k] = {"valu": sq[k]["SQ_INSTS_VALU"] / per / w, "int64": sq[k]["SQ_INSTS_VALU_INT64"] / per / w}
    if k == "k_miller_run":
        valu[k]["per_pass"] = True
json.dump(valu, open(os.path.join(out, "kernel_valu_counts.json"), "w"), indent=1, sort_keys=True)
print("\n".join(lines))
if "k_miller_run" in sq:
    c = sq["k_miller_run"]; w = waves["k_miller_run"]; nl = cnt["k_miller_run"]
    json.dump({"SQ_INSTS_VALU": c["SQ_INSTS_VALU"] / w, "SQ_INSTS_VALU_INT64": c["SQ_INSTS_VALU_INT64"] / w, "wavefronts": w, "launches": nl,
               "note": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 (r03 SQ pass): instructions per wavefront of k_miller_run over the whole Miller loop of 2^18 proofs (all its launches)"},
              open(os.path.join(out, "miller_run_pmc_counts.json"), "w"), indent=1)
# the one-launch-per-step kernels' traffic in the same round (comparison)
fs0, c0, _ = load("pmc_steps_FETCH_SIZE/**/*counter_collection.csv")
ws0, _, _ = load("pmc_steps_WRITE_SIZE/**/*counter_collection.csv")
with open(os.path.join(out, tag + "_miller_traffic_runs_vs_steps.txt"), "w") as f:
    def per_batch(fsx, wsx, cx, kinds):
        t = 0.0
        for k in kinds:
            if k in fsx:
                t += (2 * fsx[k]["FETCH_SIZE"] + wsx[k]["WRITE_SIZE"]) * 1024 / n
        return t
    a = per_batch(fs0, ws0, c0, ("k_miller_step_dbl", "k_miller_step_add"))
    b = per_batch(fs, ws, cnt, ("k_miller_run",))
    whole0 = sum((2 * fs0[k]["FETCH_SIZE"] + ws0[k]["WRITE_SIZE"]) * 1024 / n for k in fs0)
    whole1 = sum((2 * fs[k]["FETCH_SIZE"] + ws[k]["WRITE_SIZE"]) * 1024 / n for k in fs)
    f.write("# HBM counter traffic per proof (FETCH_SIZE x 2 + WRITE_SIZE, batch 2^18, one stream), Miller loop and whole path\n")
    f.write("one launch per step (BN254_MILLER_RUN_STEPS=0): Miller loop %.0f B/proof, whole path %.0f B/proof\n" % (a, whole0))
    f.write("whole loop in one launch (k_miller_run):        Miller loop %.0f B/proof, whole path %.0f B/proof\n" % (b, whole1))
    print(open(f.name).read())
# batch sweep
with open(os.path.join(out, tag + "_batch_sweep.txt"), "w") as f:
    f.write("# bench.py --batch-log2 B --steps 5 --warmup 1 --no-cpu-baseline --no-rlc --no-configs, inputs resident, one MI355X: the strong-scaling shard sizes (2^17 = the 8-GPU shard)\n# batch   proofs/s    ms/batch   ratio to 2^20\n")
    vals = {}
    for lg in (16, 17, 18, 19, 20):
        p = os.path.join(root, src, "sweep_%d.json" % lg)
        if os.path.exists(p):
            d = json.load(open(p)); vals[lg] = d
    for lg, d in vals.items():
        f.write("2^%d  %9.0f  %8.3f   %.3f\n" % (lg, d["value"], d["ms_per_step"], d["value"] / vals[20]["value"] if 20 in vals else 0))
    print(open(f.name).read())
