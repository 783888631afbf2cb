#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/gpu_round.sh (gpurun_out/) into the tracked summaries under profiles/.
usage: python tools/summarize_profiles.py r01_final"""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"      # directory (under the repo root) that holds prof/ and pmc_*/
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")


def short(name):
    n = name.split("(")[0].replace("bn254::", "").strip()
    if n.startswith("void "):
        n = n[5:]
    return n.split("<")[0].strip()   # template instances (k_miller_step_dbl<true> / <false>) count as one kernel kind


# ---- kernel trace: per-kernel calls / total / average (ns) from the raw trace (the --stats file carries the same numbers)
tr = glob.glob(os.path.join(root, src + "/prof/**/*kernel_trace.csv"), recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(tr)):
    agg[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in agg.values())
with open(os.path.join(out, tag + "_kernel_stats.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (3 batches of 2^20; 2 sub-batch streams => launches cover 2^19 proofs)\n")
    f.write("kernel,calls,total_ms,avg_us,min_us,max_us,percent\n")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        f.write("%s,%d,%.3f,%.2f,%.2f,%.2f,%.2f\n" % (k, len(v), sum(v) / 1e6, sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3, 100.0 * sum(v) / tot))
st = glob.glob(os.path.join(root, src + "/prof/**/*kernel_stats.csv"), recursive=True)
if st:
    shutil.copy(st[0], os.path.join(out, tag + "_rocprof_kernel_stats_raw.csv"))
for name in ("bench.json", "prof_bench.json"):
    p = os.path.join(root, src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(out, tag + "_" + name))


# ---- PMC passes (batch 2^18, one stream)
def load(pattern):
    path = glob.glob(os.path.join(root, pattern), recursive=True)[0]
    rows = list(csv.DictReader(open(path)))
    a = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    first = rows[0]["Counter_Name"]
    for r in rows:
        k = short(r["Kernel_Name"])
        if not k.startswith("k_"):
            continue
        a[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == first:
            cnt[k] += 1
    return a, cnt


sq, cnt = load(src + "/pmc_SQ_WAVE_CYCLES/**/*counter_collection.csv")
fs, _ = load(src + "/pmc_FETCH_SIZE/**/*counter_collection.csv")
ws, _ = load(src + "/pmc_WRITE_SIZE/**/*counter_collection.csv")
n = 1 << 18
traffic = {"_note": "HBM bytes per proof and launch of each kernel kind from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, batch 2^18, "
                    "one stream, bench.py --steps 1 --warmup 0); FETCH_SIZE (KB) doubled as MI355X_MICROARCH.md prescribes for gfx950 (checked on "
                    "k_f12_sqr, whose reads are exactly 432 B/proof), WRITE_SIZE (KB) as reported", "batch": n}
lines = ["kernel,launches,valu_active_frac,any_active_frac,wait_any_frac,wait_inst_frac,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch,hbm_read_B_per_proof(2xFETCH),hbm_write_B_per_proof"]
for k in sorted(sq, key=lambda k: -sq[k]["SQ_WAVE_CYCLES"]):
    c = sq[k]; wc = c["SQ_WAVE_CYCLES"] or 1
    f_ = fs[k]["FETCH_SIZE"] / max(cnt[k], 1); w_ = ws[k]["WRITE_SIZE"] / max(cnt[k], 1)
    rd = 2 * f_ * 1024 / n; wr = w_ * 1024 / n
    traffic[k] = {"read_bytes_per_proof": round(rd, 1), "write_bytes_per_proof": round(wr, 1)}
    lines.append("%s,%d,%.3f,%.3f,%.3f,%.3f,%.0f,%.0f,%.0f,%.0f" % (k, cnt[k], c["SQ_ACTIVE_INST_VALU"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc,
                                                                   c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, f_, w_, rd, wr))
open(os.path.join(out, tag + "_pmc_summary.csv"), "w").write("\n".join(lines) + "\n")
json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
print("\n".join(lines))


# ---- the secondary measurements of tools/gpu_round2.sh (each only if its output exists) ---------------------------------------------------------
def kernel_stats(pattern, name, header):
    tr = glob.glob(os.path.join(root, src, pattern), recursive=True)
    if not tr:
        return
    a = collections.defaultdict(list)
    for r in csv.DictReader(open(tr[0])):
        a[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    t = sum(sum(v) for v in a.values())
    with open(os.path.join(out, name), "w") as f:
        f.write("# " + header + "\nkernel,calls,total_ms,avg_us,percent\n")
        for k, v in sorted(a.items(), key=lambda kv: -sum(kv[1])):
            f.write("%s,%d,%.3f,%.1f,%.2f\n" % (k, len(v), sum(v) / 1e6, sum(v) / len(v) / 1e3, 100.0 * sum(v) / t))


rnd = tag.split("_")[0]
kernel_stats("prof_small/**/*kernel_trace.csv", rnd + "_coop12_batch4096_kernel_stats.csv",
             "rocprofv3 --kernel-trace --stats -- python3 bench.py --batch-log2 12 --steps 20 --warmup 2 (22 batches of 4096 Groth16 proofs, cooperative layout, twelve lanes per proof)")
kernel_stats("prof_rlc/**/*kernel_trace.csv", rnd + "_rlc_kernel_stats.csv",
             "rocprofv3 --kernel-trace --stats -- python3 tools/bench_rlc.py --batch-log2 20 --steps 2 --invalid-every 0 (3 exact + 3 RLC passes over 2^20 valid proofs)")
kernel_stats("prof_plonk/**/*kernel_trace.csv", rnd + "_plonk_kernel_stats.csv",
             "rocprofv3 --kernel-trace --stats -- python3 tools/bench_plonk.py --steps 3 (4 batches of 4096 PlonK proofs, cooperative pairing check)")
pm = glob.glob(os.path.join(root, src, "pmc_small/**/*counter_collection.csv"), recursive=True)
if pm:
    a = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.Counter(); waves = {}
    files = pm + glob.glob(os.path.join(root, src, "pmc_small2/**/*counter_collection.csv"), recursive=True)
    for p in files:
        rows = list(csv.DictReader(open(p)))
        first = rows[0]["Counter_Name"]
        for r in rows:
            k = short(r["Kernel_Name"])
            if not k.startswith("k_"):
                continue
            a[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if p == pm[0] and r["Counter_Name"] == first:
                launches[k] += 1
                waves[k] = int(r["Grid_Size"]) // 64
    with open(os.path.join(out, rnd + "_coop12_pmc_summary.csv"), "w") as f:
        f.write("# rocprofv3 --pmc SQ_* -- python3 bench.py --batch-log2 12 --steps 3 --warmup 1 (batch 4096, cooperative layout, twelve lanes per proof): "
                "per-kernel SQ activity as fractions of wave cycles; two counter passes\n")
        f.write("kernel,launches,waves,valu_active_frac,lds_active_frac,any_active_frac,wait_any_frac,wait_inst_frac,valu_insts_per_wave,lds_insts_per_wave,lds_bank_conflict_frac\n")
        for k in sorted(a, key=lambda k: -a[k]["SQ_WAVE_CYCLES"]):
            c = a[k]; wc = c["SQ_WAVE_CYCLES"] or 1; n_l = max(launches[k], 1); w = max(waves.get(k, 1), 1)
            f.write("%s,%d,%d,%.3f,%.3f,%.3f,%.3f,%.3f,%.0f,%.0f,%.3f\n" % (
                k, n_l, w, c["SQ_ACTIVE_INST_VALU"] / wc, c["SQ_ACTIVE_INST_LDS"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_WAIT_ANY"] / wc,
                c["SQ_WAIT_INST_ANY"] / wc, c["SQ_INSTS_VALU"] / n_l / w, c["SQ_INSTS_LDS"] / n_l / w,
                c["SQ_LDS_BANK_CONFLICT"] / (c["SQ_LDS_IDX_ACTIVE"] or 1)))
for a_, b_ in (("small_coop12.txt", "_small_batches_coop12.txt"), ("small_coop.txt", "_small_batches_coop.txt"), ("small_lane.txt", "_small_batches_lane.txt"),
               ("rlc.txt", "_rlc_vs_exact.txt"), ("plonk.json", "_plonk_bench.json"), ("cfg5.json", "_cfg5_bench.json")):
    p = os.path.join(root, src, a_)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(out, rnd + b_))
p = os.path.join(root, src, "plonk.json")
if os.path.exists(p):
    try:
        st_ = json.load(open(p)).get("stages_ms")
        if st_:
            open(os.path.join(out, rnd + "_plonk_stage_times.txt"), "w").write(
                "# tools/bench_plonk.py, batch 4096: durations (ms) of the last batch (bn254_plonk_last_timing: host stages and walls by the host clock, kernels by HIP events)\n"
                + "".join("%-28s %.3f\n" % (k, v) for k, v in st_.items()))
    except Exception:
        pass

# 64-bit integer VALU instructions per wavefront of the cooperative kernels (check of tools/count_mads.py's call-graph model)
cc = {}
for d_ in ("pmc_int64_small", "pmc_int64_plonk"):
    for p in glob.glob(os.path.join(root, src, d_ + "/**/*counter_collection.csv"), recursive=True):
        a = collections.defaultdict(lambda: collections.defaultdict(float)); n_ = collections.Counter(); waves = {}
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            if "coop" not in k:
                continue
            a[k][r["Counter_Name"]] += float(r["Counter_Value"]); n_[(k, r["Counter_Name"])] += 1; waves[k] = int(r["Grid_Size"]) // 64
        for k, v in a.items():
            cc[k] = {c: x / n_[(k, c)] / waves[k] for c, x in v.items()}
            cc[k]["wavefronts"] = waves[k]
if cc:
    cc["_note"] = ("rocprofv3 --pmc SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU SQ_WAVES on batch 4096 (bench.py --batch-log2 12; tools/bench_plonk.py): "
                   "instructions per wavefront and launch.  SQ_INSTS_VALU_INT64 = v_mad_[iu]64_[iu]32 + v_lshl_add_u64 + v_ashrrev_i64 (exact on the straight-line kernels)")
    json.dump(cc, open(os.path.join(out, "coop12_pmc_counts.json"), "w"), indent=1, sort_keys=True)

# 1024 public inputs (BASELINE configs[4]): kernel trace and SQ counters
kernel_stats("prof_cfg5/**/*kernel_trace.csv", rnd + "_cfg5_kernel_stats.csv",
             "rocprofv3 --kernel-trace --stats -- python3 bench.py --n-public 1024 --batch-log2 12 --steps 5 --warmup 1 (6 batches of 4096 proofs, 1024 public inputs, comb tables)")
pc = glob.glob(os.path.join(root, src, "pmc_cfg5/**/*counter_collection.csv"), recursive=True)
if pc:
    a = collections.defaultdict(lambda: collections.defaultdict(float)); n_ = collections.Counter(); waves = {}
    for r in csv.DictReader(open(pc[0])):
        k = short(r["Kernel_Name"])
        if not k.startswith("k_"):
            continue
        a[k][r["Counter_Name"]] += float(r["Counter_Value"]); n_[(k, r["Counter_Name"])] += 1; waves[k] = int(r["Grid_Size"]) // 64
    with open(os.path.join(out, rnd + "_cfg5_pmc_summary.csv"), "w") as f:
        f.write("# rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_INT64 "
                "-- python3 bench.py --n-public 1024 --batch-log2 12 --steps 1 --warmup 0\n")
        f.write("kernel,waves,valu_active_frac,wait_any_frac,valu_insts_per_wave,int64_insts_per_wave\n")
        for k, v in sorted(a.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
            wc = v["SQ_WAVE_CYCLES"] or 1
            f.write("%s,%d,%.3f,%.3f,%.0f,%.0f\n" % (k, waves[k], v["SQ_ACTIVE_INST_VALU"] / wc, v["SQ_WAIT_ANY"] / wc,
                                                      v["SQ_INSTS_VALU"] / n_[(k, "SQ_INSTS_VALU")] / waves[k], v["SQ_INSTS_VALU_INT64"] / n_[(k, "SQ_INSTS_VALU_INT64")] / waves[k]))
