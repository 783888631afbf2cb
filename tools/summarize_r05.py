#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/gpu_round5.sh (gpurun_out/r05/) into the tracked summaries under profiles/ (r05_*).
usage: python tools/summarize_r05.py [src_dir_under_repo_root]"""
import collections, csv, glob, json, os, shutil, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r05"
out = os.path.join(root, "profiles")
tag = "r05"


def short(name):
    n = name.split("(")[0].replace("bn254::", "").strip()
    if n.startswith("void "):
        n = n[5:]
    return n.split("<")[0].strip()


def G(pattern):
    r = glob.glob(os.path.join(root, src, pattern), recursive=True)
    return r[0] if r else None


def last_json_line(path):
    d = None
    for l in open(path):
        if l.startswith("{"):
            d = l
    return d


def kernel_stats(trace_glob, dst, header):
    tr = G(trace_glob)
    agg = collections.defaultdict(list)
    rows = list(csv.DictReader(open(tr)))
    for r in rows:
        agg[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in agg.values())
    with open(os.path.join(out, dst), "w") as f:
        f.write(header)
        f.write("kernel,calls,total_ms,avg_us,min_us,max_us,percent\n")
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            f.write("%s,%d,%.3f,%.2f,%.2f,%.2f,%.2f\n" % (k, len(v), sum(v) / 1e6, sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3, 100.0 * sum(v) / tot))
    return rows


# ---- kernel traces
rows = kernel_stats("prof/**/*kernel_trace.csv", tag + "_kernel_stats.csv",
                    "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rlc --no-configs (3 batches of 2^20; 2 sub-batch streams => launches cover\n"
                    "# 2^19 proofs; k_miller_run = the whole Miller loop of a sub-batch in ONE launch; the two streams' launches run side by side or one after the other: r05_miller_run_timeline.txt)\n")
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
kernel_stats("prof_plonk/**/*kernel_trace.csv", tag + "_plonk4096_kernel_stats.csv",
             "# rocprofv3 --kernel-trace --stats -- python3 tools/bench_plonk.py --batch 4096 --steps 10 --warmup 2 --cpu-sample 0 --no-in-flight (PlonK batches of 4096 proofs, one call at a time:\n"
             "# one pass, the MSMs as rows, the pairing check on the cooperative kernel; includes the warm-up batches and the peak probe k_valu_peak)\n")
kernel_stats("prof_plonk64k/**/*kernel_trace.csv", tag + "_plonk65536_kernel_stats.csv",
             "# rocprofv3 --kernel-trace --stats -- python3 tools/bench_plonk.py --batch 65536 --steps 3 --warmup 1 --cpu-sample 0 (PlonK batches of 65 536 proofs: one pass, unsplit rows,\n"
             "# the pairing check on the lane kernels: k_miller_run_fixed2 + the final-exponentiation program)\n")
kernel_stats("prof_plonk256k/**/*kernel_trace.csv", tag + "_plonk262144_kernel_stats.csv",
             "# rocprofv3 --kernel-trace --stats -- python3 tools/bench_plonk.py --batch 262144 --steps 2 --warmup 1 --cpu-sample 0 --no-in-flight (PlonK batches of 262 144 proofs, resident entry and the\n"
             "# host-buffer entry: two passes of 131 072 proofs on two contexts; k_valu_peak = the peak probe)\n")
for name, dst in (("bench.json", "bench.json"), ("plonk262144.json", "plonk262144_bench.json"), ("bench_default.json", "bench_default.json"), ("prof_bench.json", "prof_bench.json"), ("bench_torchrun.json", "bench_torchrun.json"),
                  ("bench_rehearse2.json", "bench_rehearse_2ranks_one_gpu.json"), ("plonk4096.json", "plonk4096_bench.json"), ("plonk65536.json", "plonk65536_bench.json")):
    p = os.path.join(root, src, name)
    if os.path.exists(p):
        l = last_json_line(p)
        if l:
            open(os.path.join(out, tag + "_" + dst), "w").write(l)


# ---- PMC passes
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


def pmc_table(prefix, n, dst, header, traffic_out=None):
    sq, cnt, waves = load(prefix + "_SQ_WAVE_CYCLES/**/*counter_collection.csv")
    fs, _, _ = load(prefix + "_FETCH_SIZE/**/*counter_collection.csv")
    ws, _, _ = load(prefix + "_WRITE_SIZE/**/*counter_collection.csv")
    lines = ["kernel,launches,wavefronts_per_launch,valu_active_frac,any_active_frac,wait_any_frac,wait_inst_frac,valu_insts_per_wave,int64_insts_per_wave,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch,"
             "hbm_read_B_per_proof(2xFETCH),hbm_write_B_per_proof"]
    for k in sorted(sq, key=lambda k: -sq[k]["SQ_WAVE_CYCLES"]):
        if k == "k_valu_peak":
            continue
        c = sq[k]; wc = c["SQ_WAVE_CYCLES"] or 1; nl = max(cnt[k], 1); w = max(waves.get(k, 1), 1)
        f_ = fs[k]["FETCH_SIZE"] / nl; w_ = ws[k]["WRITE_SIZE"] / nl
        rd = 2 * f_ * 1024 / n; wr = w_ * 1024 / n
        if traffic_out is not None:
            traffic_out[k] = {"read_bytes_per_proof": round(rd, 1), "write_bytes_per_proof": round(wr, 1)}
            if k == "k_miller_run":
                traffic_out[k] = {"read_bytes_per_proof": round(rd * nl, 1), "write_bytes_per_proof": round(wr * nl, 1), "per_pass": True, "launches_in_this_run": nl}
        lines.append("%s,%d,%d,%.3f,%.3f,%.3f,%.3f,%.0f,%.0f,%.0f,%.0f,%.0f,%.0f" % (
            k, cnt[k], w, c["SQ_ACTIVE_INST_VALU"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc,
            c["SQ_INSTS_VALU"] / nl / w, c["SQ_INSTS_VALU_INT64"] / nl / w, f_, w_, rd, wr))
    open(os.path.join(out, dst), "w").write(header + "\n".join(lines) + "\n")
    print("\n".join(lines))
    return sq, cnt, waves


old = {}
try:
    old = json.load(open(os.path.join(out, "pmc_traffic.json")))
except Exception:
    pass
traffic = {k: v for k, v in old.items() if not k.startswith("_")}
traffic["_note"] = ("HBM bytes per proof and launch of each kernel kind from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, batch 2^18, one stream, bench.py --steps 1 "
                    "--warmup 0); FETCH_SIZE (KB) doubled as MI355X_MICROARCH.md prescribes for gfx950 (checked on k_f12_sqr, whose reads are exactly 432 B/proof), WRITE_SIZE "
                    "(KB) as reported.  Round 5 rows (tools/summarize_r05.py) replace the earlier rows of the same kernels; k_miller_run: bytes per pass over a sub-batch (the whole Miller loop, however many launches)")
sq, cnt, waves = pmc_table("pmc", 1 << 18, tag + "_pmc_summary.csv",
                           "# rocprofv3 --pmc passes (SQ activity | FETCH_SIZE | WRITE_SIZE), each its own run of: BN254_STREAMS=1 python3 bench.py --steps 1 --warmup 0 "
                           "--no-cpu-baseline --no-rlc --no-configs --batch-log2 18 (one batch of 2^18 proofs, one stream)\n", traffic)
json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
valu = {"_note": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 (batch 2^18, one stream): instructions per wavefront and launch; k_miller_run per pass (all its launches)"}
for k in sq:
    nl = max(cnt[k], 1); w = max(waves.get(k, 1), 1)
    per = 1 if k == "k_miller_run" else nl
    valu[k] = {"valu": sq[k]["SQ_INSTS_VALU"] / per / w, "int64": sq[k]["SQ_INSTS_VALU_INT64"] / per / w}
    if k == "k_miller_run":
        valu[k]["per_pass"] = True
json.dump(valu, open(os.path.join(out, "kernel_valu_counts.json"), "w"), indent=1, sort_keys=True)
if "k_miller_run" in sq:
    c = sq["k_miller_run"]; w = waves["k_miller_run"]; nl = cnt["k_miller_run"]
    json.dump({"SQ_INSTS_VALU": c["SQ_INSTS_VALU"] / w, "SQ_INSTS_VALU_INT64": c["SQ_INSTS_VALU_INT64"] / w, "wavefronts": w, "launches": nl,
               "note": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 (r05 SQ pass): instructions per wavefront of k_miller_run over the whole Miller loop of 2^18 proofs (all its launches)"},
              open(os.path.join(out, "miller_run_pmc_counts.json"), "w"), indent=1)
# PlonK, 4096 proofs per batch (the warm-up batches of bench_plonk.py included: per-launch averages)
plonk_traffic = {"_note": "HBM bytes per PROOF and launch of the PlonK kernels at 4096 proofs per call (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE doubled as "
                          "for the Groth16 rows of pmc_traffic.json); k_g1_msm_rows and k_g1_sum_affine: the average of the pass's two launches (digest, KZG check)"}
pmc_table("pmcp", 4096, tag + "_plonk4096_pmc_summary.csv",
          "# rocprofv3 --pmc passes (SQ activity | FETCH_SIZE | WRITE_SIZE), each its own run of: python3 tools/bench_plonk.py --batch 4096 --steps 2 --warmup 1 --cpu-sample 0 --no-in-flight "
          "(PlonK batches of 4096 proofs; per-launch averages; bytes per PROOF of the batch)\n", plonk_traffic)
json.dump(plonk_traffic, open(os.path.join(out, "pmc_traffic_plonk.json"), "w"), indent=1, sort_keys=True)
