#!/usr/bin/env python3
"""bench.py -- Groth16 batch-verify throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = one pass of the whole hot path (parse + checks + public-input MSM, G2 subgroup test, 3-pair Miller loop, final
exponentiation, status bytes) over one batch of synthetic gnark-format proofs that are ALREADY RESIDENT IN HBM, followed by
the one collective of the path: the all_gather of the accept/reject bytes (RCCL over xGMI; a no-op at N = 1).
Workload at N = 1: BASELINE.json configs[2], batch 2^20, 2 public inputs, 1/16 of the proofs invalid (5 failure classes).
For N > 1 every rank verifies its own 2^20-proof shard of an N * 2^20 batch (weak scaling, no data-path communication).
After the timed region the statuses are compared with the generator's expected statuses: a wrong answer aborts the bench.

One JSON line on rank 0, with `roofline` for the dominant kernel -- the kernel kind with the largest summed duration over a
batch (k_miller_step_dbl, one whole doubling step of the shared Miller loop, on this code) -- whose launches are bracketed by
HIP events on their launch stream INSIDE the timed region, and `cpu_baseline` (the CPU oracle = C port of the reference algorithm,
timed on this box's host cores on a bounded sample; rank 0, N = 1 only).  The last warm-up step brackets every launch of every
kernel kind instead (`kernels_ms`, informational; ~420 event records, outside the timed region).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per proof of the path = 256 B proof + 64 B public inputs in, 1 B status out
ALGO_BYTES_PER_PROOF = 321
# Algorithmic HBM bytes per proof and LAUNCH of each kernel kind = operands read + results written in the per-proof workspace
# (one Fp = 9 x 4 B, Fp2 = 72 B, Fp12 = 432 B; DESIGN.md "Kernels").  The workspace is the data these kernels exist to move:
# an Fp12-level operation cannot keep its 432-byte operands in registers across launches.
KERNEL_ALGO_BYTES = {
    "k_f12_mul": 3 * 432,                    # a, b in; a*b out
    "k_f12_sqr": 2 * 432, "k_f12_cyclo_sqr": 2 * 432, "k_f12_cyclo_sqr_n": 2 * 432, "k_f12_conj": 2 * 432, "k_f12_copy": 2 * 432, "k_f12_frob": 2 * 432, "k_f12_inv": 2 * 432,
    "k_f12_mul_line_fixed": 2 * 432 + 72,        # f in/out, G1 point (the line table entry is wave-uniform: scalar loads)
    "k_f12_mul_line_fixed2": 2 * 432 + 2 * 72,   # f in/out, two G1 points
    "k_miller_step_dbl": 2 * 432 + 2 * 216 + 3 * 72,            # f, T in/out, three G1 points
    "k_miller_step_add": 2 * 432 + 2 * 216 + 144 + 3 * 72,      # + Q in
    "k_miller_sqr_dbl_var": 2 * 432 + 2 * 216 + 72,
    "k_miller_dbl_var": 2 * 432 + 2 * 216 + 72, "k_miller_add_var": 2 * 432 + 2 * 216 + 144 + 72,   # f, T in/out (, Q), G1 point
    "k_g16_prepare": 321 + 10 * 36, "k_g16_subgroup": 144, "k_vm_init": 432 + 216, "k_g16_compare": 432 + 1,
}
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
# exact Fp-multiplication count per proof of THIS implementation (DESIGN.md "Work model"; counted by tests/hostsim)
VALU_PEAK_MAD_PER_S = 35.1e12   # measured v_mad_u64_u32 lane-rate, profiles/r01_ubench_valu.txt (548 G wave-instr/s x 64)


# 32x32+64 multiply-adds (v_mad_i64_i32) per proof and LAUNCH of each kernel kind: static counts from the gfx950 code object for the
# straight-line kernels, loop trip counts applied for the others (prepare: 64 window additions + one Fermat inversion; cyclo_sqr_n:
# per squaring, multiplied by the run length below).  The binding roofline of this path: DESIGN.md section 5.
KERNEL_MADS = {
    "k_miller_step_dbl": 30354, "k_miller_step_add": 25000, "k_f12_mul": 8271, "k_f12_cyclo_sqr": 3657, "k_f12_inv": 16800,
    "k_f12_frob": 810, "k_g16_prepare": 170000, "k_g16_subgroup": 1992,
}
CYCLO_SQUARINGS_IN_RUNS = 186   # 3 exp-by-u x 62 squarings inside the 39 k_f12_cyclo_sqr_n launches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-log2", type=int, default=20, help="proofs per GPU = 2^this (default: BASELINE 2^20)")
    ap.add_argument("--n-public", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=4096, help="proofs timed on the host for cpu_baseline (~30 CPU-seconds)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run (WORLD_SIZE=%d)" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU: the product has no CPU path"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    pkg = importlib.import_module("snark-bn254-verifier_amd")
    sharding = importlib.import_module("snark-bn254-verifier_amd.sharding")
    L = pkg.lib()

    n = 1 << args.batch_log2                      # per-GPU shard
    n_total = n * world
    threads = max(1, min(32, (os.cpu_count() or 8) // max(1, world)))
    t0 = time.time()
    # one verifying key for the whole job (seed fixed), per-rank proofs (the generator derives proof i from seed and index;
    # different ranks use different seeds for the proofs but must share the key, so generate the key from the common seed and
    # offset only the proof stream)
    seed = 0xB2540002
    vk, proofs, inputs, expected = pkg.synth_groth16(seed, args.n_public, n, invalid_every=16, agree=True, threads=threads)
    if world > 1 and rank > 0:
        # same key (same seed), a different slice of the proof stream: rotate this rank's data so shards are not byte-identical
        k = (rank * 7919) % n
        proofs = proofs[256 * k:] + proofs[:256 * k]
        sz = 32 * args.n_public
        inputs = inputs[sz * k:] + inputs[:sz * k]
        expected = expected[k:] + expected[:k]
    gen_s = time.time() - t0

    pvk = pkg.PreparedVk(vk, pkg.VK_REFERENCE)
    pvk.reserve(n, local_rank)
    d_proofs = torch.frombuffer(bytearray(proofs), dtype=torch.uint8).to(dev)
    d_inputs = torch.frombuffer(bytearray(inputs), dtype=torch.uint8).to(dev)
    d_status = torch.zeros(n, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)
    L.bn254_set_profiling(1)

    def step():
        pvk.verify_batch_device(d_proofs.data_ptr(), d_inputs.data_ptr(), d_status.data_ptr(), n, 256, args.n_public, local_rank, stream.cuda_stream)
        return sharding.gather_status(d_status, n_total, world)   # the only collective of the path

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    pkg.set_profile_kernels(None)                 # warm-up: bracket every launch (per-kind breakdown)
    for _ in range(args.warmup):
        step()
    fence()
    breakdown = None
    if args.warmup:
        breakdown, _per = pvk.kernel_profile(local_rank)
        dom = max(breakdown, key=lambda k: breakdown[k][1])
        pkg.set_profile_kernels([dom])            # timed region: events around the dominant kernel's launches only
    prof = {}
    phase_ms = {}
    per_launch = n
    t_start = time.perf_counter()
    full = None
    for _ in range(args.steps):
        full = step()
        # HIP-event durations of this step (events were recorded on the launch stream); reading them waits for the step's
        # last event only, which the next step would have to wait for anyway (same stream)
        kp, per_launch = pvk.kernel_profile(local_rank)
        for k, (cnt, ms) in kp.items():
            c0, m0 = prof.get(k, (0, 0.0))
            prof[k] = (c0 + cnt, m0 + ms)
        for k, v in pvk.last_kernel_ms(local_rank).items():
            phase_ms.setdefault(k, []).append(v)
    fence()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if breakdown is None:
        breakdown = {k: (c // args.steps, m / args.steps) for k, (c, m) in prof.items()}
    dom = max(prof, key=lambda k: prof[k][1])

    # correctness of the timed work
    got = bytes(d_status.cpu().numpy().tobytes())
    assert got == expected, "rank %d: GPU statuses differ from the expected statuses" % rank
    if world > 1:
        lo, hi = sharding.shard_bounds(n_total, world, rank)
        assert bytes(full[lo:hi].cpu().numpy().tobytes()) == expected, "gathered statuses are wrong"
        assert full.numel() == n_total

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = n_total * args.steps / elapsed
        launches, total_ms = prof[dom]
        avg_launch_ms = total_ms / launches
        algo = dict(KERNEL_ALGO_BYTES)
        chunks = (args.n_public + 15) // 16
        # wide-key MSM (configs[4]): scalars + one 80-byte table entry per 8-bit window in, one projective partial sum per chunk out
        algo["k_g16_msm_partial"] = args.n_public * (32 + 32 * 80) + chunks * 108
        algo["k_g16_msm_reduce"] = chunks * 108 + 72
        algo["k_g16_prepare"] = 256 + 10 * 36 + (0 if args.n_public > 16 else args.n_public * (32 + 32 * 80))
        algo_launch = algo[dom] * per_launch
        achieved = algo_launch / (avg_launch_ms * 1e-3) / 1e9
        out = {
            "metric": "Groth16 verifies/sec (2 pub-inputs) at batch=2^20, 1/2/4/8 MI355X" if (args.n_public, args.batch_log2) == (2, 20)
                      else "Groth16 verifies/sec (%d pub-inputs) at batch=2^%d" % (args.n_public, args.batch_log2),
            "value": value, "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[%d]: batch 2^%d Groth16 proofs per GPU, %d public inputs, gnark-format bytes, 1/16 invalid"
                                   % (4 if args.n_public == 1024 else 2, args.batch_log2, args.n_public),
                       "batch_per_gpu": n, "global_batch": n_total, "n_public": args.n_public, "vk_mode": "reference",
                       "parallelism": "independent proof shards x%d + all_gather of status bytes" % world,
                       "streams_per_gpu": int(os.environ.get("BN254_STREAMS", "2")),
                       "gen_seconds": round(gen_s, 1)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": _measured_traffic(dom, per_launch),
                         "algorithmic_bytes_per_launch": algo_launch, "avg_launch_ms": avg_launch_ms,
                         "launches_timed": launches, "proofs_per_launch": per_launch,
                         "note": "HIP events around every launch of this kernel kind inside the timed region; the kernel is 64-bit "
                                 "integer multiply-add (VALU) work, no MFMA: see DESIGN.md for its VALU roofline"},
            "valu_roofline": _valu_roofline(breakdown, value / world, args.n_public),
            "kernels_ms": {k: {"launches": c, "total_ms": round(m, 3)} for k, (c, m) in sorted(breakdown.items(), key=lambda kv: -kv[1][1])},
            "phases_ms": {k: sum(v) / len(v) for k, v in phase_ms.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = _cpu_baseline(args, vk, proofs, inputs, expected)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _valu_roofline(breakdown, proofs_per_s_per_gpu, n_public):
    """Whole-path VALU utilisation of one GPU: multiply-adds per proof (launch counts of the profiled warm-up step x KERNEL_MADS)
    x proofs/s against the measured v_mad peak (profiles/r01_ubench_valu.txt).  Informational, next to the contract's `roofline`."""
    if n_public != 2:
        return None
    mads = sum(cnt * KERNEL_MADS.get(k, 0) for k, (cnt, _ms) in breakdown.items()) + CYCLO_SQUARINGS_IN_RUNS * KERNEL_MADS["k_f12_cyclo_sqr"]
    achieved = mads * proofs_per_s_per_gpu
    return {"bound": "valu (64-bit integer multiply-add issue)", "mads_per_proof": mads, "achieved": achieved / 1e12, "peak": VALU_PEAK_MAD_PER_S / 1e12,
            "unit": "T mad/s", "frac": achieved / VALU_PEAK_MAD_PER_S}


def _measured_traffic(kernel, proofs_per_launch):
    """HBM bytes per launch of the dominant kernel: PMC FETCH_SIZE / WRITE_SIZE per proof from the committed rocprofv3 --pmc
    run (profiles/pmc_traffic.json, corrected as MI355X_MICROARCH.md prescribes) x the proofs one launch covers; or None."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        e = json.load(open(p))[kernel]
        return (e["read_bytes_per_proof"] + e["write_bytes_per_proof"]) * proofs_per_launch
    except Exception:
        return None


def _cpu_baseline(args, vk, proofs, inputs, expected):
    """The oracle (C port of the reference algorithm, reference-faithful: vk re-parsed per call, 4 Miller loops + 2 final
    exponentiations, naive subgroup check) on the host cores of this box, on a prefix of the same workload."""
    from oracle import oracle as O
    O.build(); O.lib()
    m = min(args.cpu_sample if args.n_public <= 16 else 256, len(expected))  # ~30 CPU-seconds either way
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))  # the 1-GPU box's CPU share is 16 cores
    O.set_threads(cores)
    sz = 32 * args.n_public
    O.groth16_verify_many(proofs[:256 * 8], 256, vk, inputs[:sz * 8], args.n_public, 8, O.MODE_REFERENCE)
    t = time.perf_counter()
    st = O.groth16_verify_many(proofs[:256 * m], 256, vk, inputs[:sz * m], args.n_public, m, O.MODE_REFERENCE)
    dt = time.perf_counter() - t
    assert st == expected[:m], "oracle disagrees with the expected statuses"
    return {"value": m / dt, "unit": "proofs/s", "cores": cores, "kind": "port",
            "sample": "first %d proofs of the same batch, %.1f s wall, OpenMP over %d threads; C restatement of the reference algorithm, not the Rust binary" % (m, dt, cores)}


if __name__ == "__main__":
    main()
