#!/usr/bin/env python3
"""bench.py -- Groth16 batch-verify throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1: if the process was started by torch.distributed.run (WORLD_SIZE set) it is one rank of
the job; otherwise bench.py starts the N ranks ITSELF -- `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a
child process, before this process has imported torch or touched a GPU -- relays the child's JSON line and exits with its code.

A step = one pass of the whole hot path (parse + checks + public-input MSM, G2 subgroup test, 3-pair Miller loop, final
exponentiation, status bytes) over one batch of synthetic gnark-format proofs that are ALREADY RESIDENT IN HBM, followed by
the one collective of the path: the all_gather of the accept/reject bytes (RCCL over xGMI; a no-op at N = 1).
Workload: BASELINE.json configs[2], ONE batch of 2^20 proofs, 2 public inputs, 1/16 of the proofs invalid (5 failure classes).
For N > 1 that batch is sharded: rank r verifies the contiguous range sharding.shard_bounds(2^20, N, r) -- 2^19 / 2^18 / 2^17 proofs per
GPU at N = 2 / 4 / 8 -- with no data-path communication: STRONG scaling, `global_batch` stays 2^20.  Every rank
generates only its own range of the proof stream (bn254_synth_groth16_range).  --weak keeps 2^batch-log2 proofs PER GPU instead.
After the timed region the statuses are compared with the generator's expected statuses: a wrong answer aborts the bench.
At N = 1 the line also carries `configs`: the other BASELINE configurations (batch 4096, PlonK 4096, 1024 public inputs x 4096, the
host-buffer entry at 2^20, single-proof latency), each timed in this run with its own roofline and status check.

One JSON line on rank 0:
  roofline      the binding bound of this path, the integer VALU: 32x32+64-bit multiply-adds of the dominant kernel kind (static count
                from the gfx950 code object, profiles/kernel_mads.json written by tools/count_mads.py) x proofs per launch / the
                kernel's average launch duration (HIP events around every launch of that kind INSIDE the timed region, on the
                launch stream), against the measured v_mad_u64_u32 issue peak (profiles/r01_ubench_valu.txt).  `traffic` = HBM
                bytes per launch from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json).
  hbm_roofline  SURVEY.md section 8(d)'s algorithmic bytes (321 B/proof) x proofs/s against the HBM peak: the figure north_star
                asks for, ~1e-4 because the path is arithmetic-bound.
  cpu_baseline  the CPU oracle (C restatement of the reference algorithm) on this box's host cores on a bounded sample;
                rank 0, N = 1 only.  Rows: reference-faithful on all cores (the headline `value`), on one core, and batch mode.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The HIP runtime multiplexes every stream of a process onto a few hardware queues (four by default), and streams that share a queue run one
# after the other.  A rank of this bench has the library's two sub-batch streams, its copy stream and -- under torch.distributed -- RCCL's own
# streams; with eight queues none of them shares (measured with RCCL initialised: the two sub-batch streams overlap 1.75 x instead of not at all,
# 185 against 192 ms per step).  Read by the runtime when it initialises, i.e. before torch is imported; a deployment sets the same variable.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

def algo_bytes_per_proof(n_public):
    """256 B proof + 32 B per public input in, 1 B status out (321 B at 2 inputs, 33 025 B at 1024)."""
    return 256 + 32 * n_public + 1


HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
VALU_PEAK_REFERENCE = 35.1e12   # v_mad_u64_u32 lane-rate of the box of profiles/r01_ubench_valu.txt (548 G wave-instr/s x 64): the guide has no integer multiply-add peak
VALU_PEAK_MAD_PER_S = VALU_PEAK_REFERENCE   # replaced at run time by the rate of THIS box (measure_valu_peak: bn254_dbg_valu_peak, right after the timed region)
VALU_PEAK_MEASURED = None
VALU_PEAK_SUSTAINED = None                  # the same probe kernel back to back for 100 ms, one interval (bn254_dbg_valu_peak_sustained)


def measure_valu_peak(pkg, device=0):
    """The multiply-add issue rate of the box the bench runs on (the library's k_valu_peak: 16 independent v_mad_u64_u32 chains per lane, four wavefronts per SIMD, launches of about 2 ms, best
    of five launches); every VALU fraction of the line is taken against it, the round-1 constant stays beside it as `peak_reference`."""
    global VALU_PEAK_MAD_PER_S, VALU_PEAK_MEASURED, VALU_PEAK_SUSTAINED
    import ctypes as C
    v = C.c_double(0.0)
    L = pkg.lib()
    L.bn254_dbg_valu_peak.argtypes = [C.c_int, C.POINTER(C.c_double)]
    L.bn254_dbg_valu_peak_sustained.argtypes = [C.c_int, C.c_double, C.POINTER(C.c_double)]
    # first the 100 ms interval (the length of the path's long kernels; the GPU is as warm as the timed region left it), then the best 2 ms launch: `peak`
    if L.bn254_dbg_valu_peak_sustained(device, 100.0, C.byref(v)) == 0 and v.value > 1e12:
        VALU_PEAK_SUSTAINED = v.value
    if L.bn254_dbg_valu_peak(device, C.byref(v)) == 0 and v.value > 1e12:
        VALU_PEAK_MEASURED = v.value
        VALU_PEAK_MAD_PER_S = v.value
    return VALU_PEAK_MEASURED


VALU_ISSUE_CEILING = 1024 * 16 * 2.4e9   # 39.3 T: 1024 SIMDs x 64 lanes / 4 cycles per wavefront instruction x 2.4 GHz nominal -- what no box can exceed


def _peak_fields():
    return {"peak": VALU_PEAK_MAD_PER_S / 1e12, "peak_measured": (VALU_PEAK_MEASURED / 1e12) if VALU_PEAK_MEASURED else None,
            "peak_sustained_100ms": (VALU_PEAK_SUSTAINED / 1e12) if VALU_PEAK_SUSTAINED else None, "peak_reference": VALU_PEAK_REFERENCE / 1e12,
            "issue_ceiling": VALU_ISSUE_CEILING / 1e12}


def _fracs(mads_per_s):
    """One achieved rate against the three denominators, so that lines of different rounds and boxes compare: `frac` = the peak measured on THIS box in THIS run,
    `frac_vs_reference` = the round-1 constant (35.1 T: what rounds 1-3 divided by), `frac_vs_issue_ceiling` = 39.3 T (4-cycle issue at the nominal clock)."""
    r = {"frac": mads_per_s / VALU_PEAK_MAD_PER_S, "frac_vs_reference": mads_per_s / VALU_PEAK_REFERENCE, "frac_vs_issue_ceiling": mads_per_s / VALU_ISSUE_CEILING}
    if VALU_PEAK_SUSTAINED:
        r["frac_vs_sustained"] = mads_per_s / VALU_PEAK_SUSTAINED
    return r
METRIC = "Groth16 verifies/sec (2 pub-inputs) at batch=2^20, 1/2/4/8 MI355X"
SEED = 0xB2540002


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-log2", type=int, default=None, help="the (global) batch = 2^this proofs (default: BASELINE 2^20; 2^18 with --plonk), sharded over the GPUs")
    ap.add_argument("--plonk", action="store_true", help="the PlonK path instead (BASELINE configs[3] at scale): one call of 2^batch-log2 proofs sharded over the GPUs like the "
                    "Groth16 batch (contiguous ranges, resident buffers, all_gather of the status bytes); prints its own line, not the headline metric")
    ap.add_argument("--weak", action="store_true", help="weak scaling: 2^batch-log2 proofs PER GPU (the global batch grows with --gpus)")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` block (the other BASELINE configurations, N = 1 only)")
    ap.add_argument("--n-public", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=16384, help="proofs timed on the host per cpu_baseline row")
    ap.add_argument("--no-rlc", action="store_true", help="skip the rlc_mode side measurement (all-valid batch, exact against RLC)")
    ap.add_argument("--rlc", action="store_true", help="time the random-linear-combination batch mode instead of the exact path")
    ap.add_argument("--host-buffers", action="store_true", help="also time the host-buffer entry (PCIe-inclusive), reported beside `value`")
    ap.add_argument("--rehearse-one-gpu", action="store_true", help="REHEARSAL of --gpus N on ONE GPU: N rank processes, all on cuda:0, gloo for the status gather and the "
                    "reductions (RCCL refuses two ranks on one device); the line says so and its value is not a scaling number")
    a = ap.parse_args(argv)
    if a.batch_log2 is None:
        a.batch_log2 = 18 if a.plonk else 20
    return a


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args, argv):
    """Start the N ranks as children of this (GPU-untouched) process; never exec."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    if args.rehearse_one_gpu:
        env["BENCH_REHEARSE_ONE_GPU"] = "1"
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------------------------------------
# workload
# ---------------------------------------------------------------------------------------------------------------------------
class GpuVerifier:
    """The product path on one GPU: inputs resident in HBM, verify_batch_device on the current stream."""

    def __init__(self, args, pkg, vk, proofs, inputs, local_rank):
        import torch
        self.torch = torch
        self.args, self.pkg, self.local_rank = args, pkg, local_rank
        self.n = len(proofs) // 256
        self.dev = torch.device("cuda", local_rank)
        torch.cuda.set_device(local_rank)
        self.pvk = pkg.PreparedVk(vk, pkg.VK_REFERENCE)
        self.pvk.reserve(self.n, local_rank)
        self.d_proofs = torch.frombuffer(bytearray(proofs), dtype=torch.uint8).to(self.dev)
        self.d_inputs = torch.frombuffer(bytearray(inputs), dtype=torch.uint8).to(self.dev)
        # two status buffers, used in turn: the gather of batch k reads one while batch k + 1 already writes the other
        self.d_st = [torch.zeros(self.n, dtype=torch.uint8, device=self.dev) for _ in range(2)]
        self.d_status, self.k = self.d_st[0], 0
        self.stream = torch.cuda.current_stream(self.dev)
        self.side = torch.cuda.Stream(self.dev)     # the status gather runs here, beside the next batch
        self.flags = pkg.FLAG_RLC if args.rlc else 0
        self.gather_done = [None, None]             # event after the gather that last READ status buffer 0 / 1 (on the side stream)
        self.cpu_gather = getattr(args, "cpu_gather", False)   # rehearsal: the collective runs under gloo on host copies of the shard
        pkg.lib().bn254_set_profiling(1)

    def step(self):
        self.d_status = self.d_st[self.k & 1]
        # batch k writes the buffer that gather k - 2 read: the gather runs on a side stream (and, with world > 1, may wait for a slow rank), so the
        # main stream waits for THAT gather before it lets the batch touch the buffer -- transient PENDING bytes must never reach a collective
        ev = self.gather_done[self.k & 1]
        if ev is not None:
            self.stream.wait_event(ev)
        self.k += 1
        self.pvk.verify_batch_device(self.d_proofs.data_ptr(), self.d_inputs.data_ptr(), self.d_status.data_ptr(), self.n, 256,
                                     self.args.n_public, self.local_rank, self.stream.cuda_stream, flags=self.flags)
        return self.d_status

    def gather(self, st, n_total, world):
        """The path's one collective, enqueued on a side stream behind this batch's last kernel: the next batch starts at once (it writes the other
        status buffer).  Returns the gathered vector and the two events that bracket the collective on the side stream."""
        torch = self.torch
        sharding = importlib.import_module("snark-bn254-verifier_amd.sharding")
        done = torch.cuda.Event()
        done.record(self.stream)
        with torch.cuda.stream(self.side):
            self.side.wait_event(done)
            a = torch.cuda.Event(enable_timing=True); a.record(self.side)
            if self.cpu_gather:
                # rehearsal on one GPU: gloo gathers host copies (the copy waits on the side stream for this batch's last kernel)
                full = sharding.gather_status(st.cpu(), n_total, world).to(self.dev)
            else:
                full = sharding.gather_status(st, n_total, world)
            b = torch.cuda.Event(enable_timing=True); b.record(self.side)
            self.gather_done[(self.k - 1) & 1] = b
        return full, (a, b)

    def accumulate_profile(self, on):
        """on: the per-launch events of the selected kernels accumulate over the batches that follow (read once, after the final synchronisation)"""
        self.pkg.lib().bn254_set_profiling(2 if on else 1)

    def sync(self):
        self.torch.cuda.synchronize(self.dev)

    def timer(self):
        ev = self.torch.cuda.Event(enable_timing=True)
        ev.record(self.stream)
        return ev

    def elapsed_ms(self, a, b):
        return a.elapsed_time(b)

    def select_kernels(self, names):
        self.pkg.set_profile_kernels(names)

    def kernel_profile(self):
        return self.pvk.kernel_profile(self.local_rank)

    def kernel_profile_all(self):
        """{kind: (launches, total_ms, union_ms)} over the first two sub-batch streams, proofs per launch"""
        return self.pvk.kernel_profile_all(self.local_rank)

    def phases(self):
        return self.pvk.last_kernel_ms(self.local_rank)

    def status_bytes(self):
        return bytes(self.d_status.cpu().numpy().tobytes())


def run_rank(args, make_verifier, backend, rank, world, local_rank, synth, emit=print):
    """The rank logic of the bench: workload (shared key, rotated shard), warm-up, K timed steps each ending in the status
    all_gather, max-over-ranks timing, correctness check of the local and the gathered statuses, JSON line on rank 0.
    make_verifier(vk, proofs, inputs, local_rank) -> object with step()/sync()/timer()/... (GpuVerifier, or a stand-in in the
    gloo CPU test); synth(seed, n_public, first, n, threads) -> (vk, proofs, inputs, expected) for the proofs [first, first + n) of the
    stream (the key is the same for every range)."""
    import torch
    import torch.distributed as dist
    sharding = importlib.import_module("snark-bn254-verifier_amd.sharding")

    # strong scaling (default): ONE batch of 2^batch_log2 proofs, rank r takes the contiguous range shard_bounds(n_total, world, r);
    # --weak: 2^batch_log2 proofs per rank, rank r = range [r n, (r + 1) n) of the same stream
    n_total = (1 << args.batch_log2) * (world if args.weak else 1)
    lo, hi = sharding.shard_bounds(n_total, world, rank)
    n = hi - lo                                   # this rank's shard
    threads = max(1, min(32, (os.cpu_count() or 8) // max(1, world)))
    t0 = time.time()
    vk, proofs, inputs, expected = synth(SEED, args.n_public, lo, n, threads)
    gen_s = time.time() - t0
    v = make_verifier(vk, proofs, inputs, local_rank)

    def step():
        st = v.step()
        if hasattr(v, "gather"):
            return v.gather(st, n_total, world)              # the only collective of the path, beside the next batch
        t_a = v.timer()
        full = sharding.gather_status(st, n_total, world)
        t_b = v.timer()
        return full, (t_a, t_b)

    grouped = dist.is_available() and dist.is_initialized()

    def fence():
        v.sync()
        if grouped:
            dist.barrier()
        v.sync()

    v.select_kernels(None)                        # warm-up: bracket every launch (per-kind breakdown)
    for _ in range(args.warmup):
        step()
    fence()
    breakdown = None
    if args.warmup:
        breakdown, _per = v.kernel_profile()
        if breakdown:
            dom = max(breakdown, key=lambda k: breakdown[k][1])
            v.select_kernels([dom])               # timed region: events around the dominant kernel's launches only
    prof, phase_ms, gathers = {}, {}, []
    per_launch = n
    if hasattr(v, "accumulate_profile"):
        v.accumulate_profile(True)                # the events of the dominant kernel's launches accumulate over the K steps
    fence()
    t_start = time.perf_counter()
    full = None
    for _ in range(args.steps):
        # nothing in here waits for the GPU: K batches and their gathers are enqueued back to back (a verifier with requests pending)
        full, tg = step()
        gathers.append(tg)
        if os.environ.get("BENCH_SYNC_EACH_STEP") == "1":
            v.sync()                              # experiment: the host waits for every batch (what rounds 1-3a did through the per-step profile read)
    fence()
    elapsed = time.perf_counter() - t_start
    if rank == 0 and getattr(v, "pkg", None) is not None:
        # the multiply-add peak of THIS box, measured while the GPU is as warm as it was in the timed region (an idle GPU reads 10 % low: its clocks take tens of
        # milliseconds of load to settle; at the start of the run the measurement said 29.7 .. 33.0 T, after the timed region it agrees with tools/ubench_valu)
        measure_valu_peak(v.pkg, v.local_rank)
    # HIP-event durations of the dominant kernel's launches over ALL timed steps (both sub-batch streams), phase durations of the last step
    kp, per_launch = v.kernel_profile_all()
    for k, (cnt, ms, un) in kp.items():
        prof[k] = (cnt, ms, un)
    for k, x in v.phases().items():
        phase_ms.setdefault(k, []).append(x)
    if hasattr(v, "accumulate_profile"):
        v.accumulate_profile(False)
    if grouped:
        t = torch.tensor([elapsed], dtype=torch.float64, device=torch.device("cpu") if backend == "gloo" else full.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    gather_ms = sum(v.elapsed_ms(a, b) for a, b in gathers) / max(1, len(gathers))
    breakdown_is_first_subbatch = breakdown is not None       # the warm-up's breakdown counts the launches of the FIRST sub-batch of one batch
    if breakdown is None:
        breakdown = {k: (c // args.steps, m / args.steps) for k, (c, m, _u) in prof.items()}

    # correctness of the timed work: this rank's shard, and its slice of the gathered vector
    got = v.status_bytes()
    assert got == expected, "rank %d: statuses differ from the expected statuses" % rank
    assert full.numel() == n_total, "gathered vector has the wrong length"
    assert bytes(full[lo:hi].cpu().numpy().tobytes()) == expected, "gathered statuses are wrong"

    out = None
    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = n_total * args.steps / elapsed
        headline = (args.n_public, args.batch_log2, bool(args.rlc), bool(args.weak)) == (2, 20, False, False)
        algo = algo_bytes_per_proof(args.n_public)
        out = {
            "metric": METRIC if headline else "Groth16 verifies/sec (%d pub-inputs) at batch=2^%d%s%s" % (args.n_public, args.batch_log2, " per GPU" if args.weak else "", ", RLC batch mode" if args.rlc else ""),
            "value": value, "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
            "dtype": "int64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[%d]: %s of 2^%d Groth16 proofs%s, %d public inputs, gnark-format bytes, 1/16 invalid"
                                   % (4 if args.n_public == 1024 else 2, "batches" if args.weak else "ONE batch", args.batch_log2,
                                      " per GPU" if args.weak else (" sharded over %d GPUs (contiguous ranges)" % world if world > 1 else ""), args.n_public),
                       "batch_per_gpu": n, "global_batch": n_total, "n_public": args.n_public, "vk_mode": "reference",
                       "mode": "rlc (random linear combination, exact fallback)" if args.rlc else "exact",
                       "parallelism": "independent contiguous proof shards x%d + all_gather of status bytes" % world,
                       "gather_ms": gather_ms, "gen_seconds": round(gen_s, 1)},
            "hbm_roofline": {"bound": "hbm", "algorithmic_bytes_per_proof": algo, "achieved": value / world * algo / 1e9,
                             "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": value / world * algo / 1e9 / HBM_PEAK_GBPS,
                             "note": "SURVEY.md 8(d): 256 B proof + 32 B per public input in, 1 B status out; the path is arithmetic-bound"},
            "kernels_ms": {k: {"launches": c, "total_ms": round(m, 3)} for k, (c, m) in sorted(breakdown.items(), key=lambda kv: -kv[1][1])},
            "phases_ms": {k: sum(x) / len(x) for k, x in phase_ms.items()},
        }
        if prof:
            dom = max(prof, key=lambda k: prof[k][1])
            # passes over a sub-batch that the recorded launches make up: launches / launches per pass (from the warm-up's per-batch breakdown of the first sub-batch);
            # equal to steps x sub-batches unless the event pool (1024 launches per sub-batch stream) filled up during a very long run
            per_pass_launches = breakdown.get(dom, (0, 0.0))[0] if breakdown_is_first_subbatch else 0
            passes = prof[dom][0] / per_pass_launches if per_pass_launches else args.steps * (2 if per_launch < n else 1)
            out["roofline"] = _valu_roofline(dom, prof[dom], per_launch, passes=passes)
            out["valu_whole_path"] = _valu_whole_path(breakdown, value / world)
            out["valu_issue_bound"] = _valu_issue_bound(breakdown, ms_per_step, n)
            out["hbm_roofline"]["traffic_bytes_per_proof"] = _traffic_whole_path(breakdown, algo)
        emit(json.dumps(out))
    return out


def _load_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def _valu_roofline(dom, launches_ms, per_launch, passes=None):
    """The binding roofline, for the dominant kernel kind: multiply-adds of all its launches / the time during which it was running, vs the v_mad peak.
    launches_ms = (launches, summed launch durations, union of the launch intervals) over the sub-batch streams of the timed steps: two streams that
    run the kernel side by side give union = about one launch's duration, launches that happen to run one after the other the sum."""
    launches, total_ms, union_ms = launches_ms
    avg_ms = total_ms / max(1, launches)
    mads = (_load_json("kernel_mads.json") or {}).get("kernels", {}).get(dom, {}).get("mads_per_proof_launch")
    traffic = (_load_json("pmc_traffic.json") or {}).get(dom)
    r = {"bound": "valu", "kernel": dom, "unit": "T mad/s", **_peak_fields(), "avg_launch_ms": avg_ms,
         "launches_timed": launches, "proofs_per_launch": per_launch, "union_ms": union_ms, "overlap": total_ms / union_ms if union_ms else None,
         "traffic": (traffic["read_bytes_per_proof"] + traffic["write_bytes_per_proof"]) * per_launch * (passes / max(1, launches) if traffic and traffic.get("per_pass") and passes else 1) if traffic else None,
         "note": "achieved = v_mad_[iu]64_[iu]32 per proof and launch (counted in the gfx950 code object: tools/count_mads.py -> profiles/kernel_mads.json) x proofs per "
                 "launch x launches / union of the launch intervals (HIP events around every launch of the kind on BOTH sub-batch streams inside the timed region, one time "
                 "base); `overlap` = summed launch durations / union (2 = the two streams ran the kernel side by side the whole time); avg_launch_ms is what rocprofv3's "
                 "kernel trace averages; peak = the multiply-add issue rate measured on THIS box right after the timed region, GPU warm (peak_measured; peak_reference = profiles/r01_ubench_valu.txt).  Cooperative kernels: 12 lanes per proof, count from the call-graph "
                 "model checked against SQ_INSTS_VALU_INT64"}
    entry = (_load_json("kernel_mads.json") or {}).get("kernels", {}).get(dom, {})
    if mads and union_ms:
        if entry.get("per_pass") and passes:
            # a kernel whose launches split one fixed piece of work (k_miller_run: the 88 steps of the Miller loop in 1, 2, 4 or 8 launches by sub-batch size):
            # the count is per pass over a sub-batch, whatever the number of launches
            total = entry["mads_per_proof_batch"] * per_launch * passes
            r["passes_timed"] = passes
        else:
            total = mads * per_launch * launches
        ach = total / (union_ms * 1e-3) / 1e12
        r.update({"mads_per_proof_launch": mads, "achieved": ach, **_fracs(ach * 1e12)})
    else:
        r.update({"achieved": None, "frac": None, "frac_vs_reference": None, "frac_vs_issue_ceiling": None})
    return r


def _valu_whole_path(breakdown, proofs_per_s_per_gpu):
    km = (_load_json("kernel_mads.json") or {}).get("kernels")
    if not km:
        return None
    mads, missing = 0, []
    for k, (cnt, _ms) in breakdown.items():
        e = km.get(k)
        if e is None:
            missing.append(k)
            continue
        mads += e.get("mads_per_proof_batch", cnt * e["mads_per_proof_launch"])
    ach = mads * proofs_per_s_per_gpu
    return {"mads_per_proof": mads, "achieved": ach / 1e12, **_peak_fields(), "unit": "T mad/s", **_fracs(ach),
            "kernels_without_count": missing}


def _valu_issue_bound(breakdown, ms_per_batch, proofs_per_batch_per_gpu):
    """How close the batch runs to the issue bound of its own instruction stream: VALU instructions per proof of every kernel kind (counters: profiles/
    kernel_valu_counts.json) x its launches per batch, priced at the measured issue rates of the two instruction classes (profiles/r01_ubench_valu.txt: a 64-bit integer
    instruction -- multiply-add, 64-bit shift / add -- 4.48 cycles per wavefront and SIMD, a plain 32-bit one 2.45), on 1024 SIMDs at 2.4 GHz."""
    t = _load_json("kernel_valu_counts.json")
    if not t:
        return None
    cycles, missing = 0.0, []
    for k, (cnt, _ms) in breakdown.items():
        e = t.get(k)
        if e is None:
            missing.append(k)
            continue
        cycles += (1 if e.get("per_pass") else cnt) * (e["int64"] * 4.48 + (e["valu"] - e["int64"]) * 2.45)
    waves_per_simd = proofs_per_batch_per_gpu / 65536.0          # 1024 SIMDs x 64 lanes
    issue_ms = cycles * waves_per_simd / 2.4e9 * 1e3
    return {"issue_ms_per_batch": issue_ms, "ms_per_batch": ms_per_batch, "frac": issue_ms / ms_per_batch if ms_per_batch else None, "cycles_per_proof": cycles,
            "kernels_without_counters": missing,
            "note": "VALU issue time of the batch's own instruction stream / its wall time (1 = nothing but instruction issue; the multiply-add roofline above is a part of it)"}


def _traffic_whole_path(breakdown, algo):
    """HBM bytes per proof over the whole path: per kernel kind, counter bytes per proof and launch (profiles/pmc_traffic.json: rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes) x its launches per batch."""
    t = _load_json("pmc_traffic.json")
    if not t:
        return None
    total, missing = 0.0, []
    for k, (cnt, _ms) in breakdown.items():
        e = t.get(k)
        if e is None:
            missing.append(k)
            continue
        total += (1 if e.get("per_pass") else cnt) * (e["read_bytes_per_proof"] + e["write_bytes_per_proof"])
    return {"counter_bytes_per_proof": total, "algorithmic_bytes_per_proof": algo, "ratio": total / algo,
            "kernels_without_counters": missing}


def _cpu_baseline(args, vk, proofs, inputs, expected):
    """The oracle (C port of the reference algorithm) on the host cores of this box, on a prefix of the same workload.
    Rows: reference-faithful (vk re-parsed per call, 4 Miller loops + 2 final exponentiations, naive subgroup check) on all cores
    and on one core; batch mode (vk prepared once: lib.rs:45-46 and groth16/verify.rs:70 hoisted) on all cores."""
    from oracle import oracle as O
    O.build(); O.lib()
    sz = 32 * args.n_public
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))  # the 1-GPU box's CPU share is 16 cores
    wide = args.n_public > 16

    def run(m, threads, batch_mode):
        O.set_threads(threads)
        f = O.groth16_verify_many_prepared if batch_mode else O.groth16_verify_many
        f(proofs[:256 * 4], 256, vk, inputs[:sz * 4], args.n_public, 4, O.MODE_REFERENCE)
        t = time.perf_counter()
        st = f(proofs[:256 * m], 256, vk, inputs[:sz * m], args.n_public, m, O.MODE_REFERENCE)
        dt = time.perf_counter() - t
        assert st == expected[:m], "oracle disagrees with the expected statuses"
        return m / dt, dt

    m_all = min(args.cpu_sample if not wide else 128, len(expected))
    m_one = min(max(64, m_all // 16) if not wide else 16, len(expected))
    v_all, dt_all = run(m_all, cores, False)
    v_one, dt_one = run(m_one, 1, False)
    v_bat, dt_bat = run(m_all, cores, True)
    flags = O.build_flags()
    return {"value": v_all, "unit": "proofs/s", "cores": cores, "kind": "port",
            "sample": "first %d proofs of the same batch, %.1f s wall, OpenMP over %d threads; C restatement of the reference algorithm (vk re-parsed and "
                      "e(alpha,beta) recomputed per proof, naive subgroup check), not the Rust binary; %s" % (m_all, dt_all, cores, flags),
            "single_core": {"value": v_one, "unit": "proofs/s", "cores": 1, "sample": "first %d proofs, %.1f s" % (m_one, dt_one)},
            "batch_mode": {"value": v_bat, "unit": "proofs/s", "cores": cores,
                           "sample": "first %d proofs, %.1f s; vk parsed once, e(alpha,beta) hoisted (reference waste at lib.rs:45-46, groth16/verify.rs:70 removed)" % (m_all, dt_bat)}}


def _host_buffer_line(args, pkg, vk, proofs, inputs, expected, local_rank):
    """PCIe-inclusive rate of the host-buffer entry point (never `value`)."""
    pvk = pkg.PreparedVk(vk, pkg.VK_REFERENCE)
    n = len(expected)
    pvk.verify_batch(proofs, inputs, n, 256, args.n_public, local_rank)     # warm-up (allocations, pinned staging)
    t = time.perf_counter()
    reps = 2
    for _ in range(reps):
        st = pvk.verify_batch(proofs, inputs, n, 256, args.n_public, local_rank)
    dt = (time.perf_counter() - t) / reps
    assert st == expected
    pvk.close()
    return {"value": n / dt, "unit": "proofs/s", "ms": dt * 1e3, "note": "bn254_groth16_verify_batch on pageable host buffers: H2D of proofs and inputs, compute, D2H of status"}


def _rlc_line(args, pkg, local_rank):
    """The random-linear-combination batch mode next to the exact path on an ALL-VALID batch of the same shape (the
    headline workload carries 1/16 invalid proofs, which sends most groups of 32 to the exact fallback; tools/bench_rlc.py has the sweep).
    Never `value`: the status bytes are identical by construction, the mode is opt-in (BN254_FLAG_RLC)."""
    import torch
    n = 1 << args.batch_log2
    dev = torch.device("cuda", local_rank)
    vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540020, args.n_public, n, invalid_every=0, agree=True, threads=min(16, os.cpu_count() or 1))
    pvk = pkg.PreparedVk(vk)
    pvk.reserve(n, local_rank)
    dp = torch.frombuffer(bytearray(proofs), dtype=torch.uint8).to(dev)
    di = torch.frombuffer(bytearray(inputs), dtype=torch.uint8).to(dev)
    ds = torch.zeros(n, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev)
    row = {"workload": "2^%d valid proofs, %d public inputs" % (args.batch_log2, args.n_public), "unit": "proofs/s"}
    steps = 2
    for name, flags in (("exact", 0), ("rlc", pkg.FLAG_RLC)):
        for it in range(steps + 1):
            if it == 1:
                torch.cuda.synchronize(dev)
                t = time.perf_counter()
            ds.zero_()
            pvk.verify_batch_device(dp.data_ptr(), di.data_ptr(), ds.data_ptr(), n, 256, args.n_public, local_rank, st.cuda_stream, flags=flags)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t) / steps
        assert bytes(ds.cpu().numpy().tobytes()) == exp, name
        row[name] = n / dt
    row["speedup"] = row["rlc"] / row["exact"]
    pvk.close()
    return row



# ---------------------------------------------------------------------------------------------------------------------------
# `configs`: the other BASELINE configurations, timed in the same run (N = 1, rank 0), each with its status check and roofline
# ---------------------------------------------------------------------------------------------------------------------------
class _Cfg:
    """argparse-like carrier for GpuVerifier"""
    def __init__(self, n_public, rlc=False):
        self.n_public, self.rlc = n_public, rlc


def device_config(pkg, local_rank, label, n_public, n, steps, warmup, seed, oracle_sample):
    """A Groth16 batch through the same device-resident entry point as the headline: synthetic workload (1/16 invalid), inputs resident,
    K timed steps between synchronisations, statuses against the generator's prediction and -- on a strided sample -- against the oracle."""
    import torch
    t0 = time.time()
    vk, proofs, inputs, expected = pkg.synth_groth16(seed, n_public, n, invalid_every=16, agree=True, threads=min(16, os.cpu_count() or 1))
    gen_s = time.time() - t0
    t0 = time.time()
    v = GpuVerifier(_Cfg(n_public), pkg, vk, proofs, inputs, local_rank)
    prep_s = time.time() - t0
    v.select_kernels(None)
    for _ in range(max(1, warmup)):
        v.step()
    v.sync()
    breakdown, per_launch = v.kernel_profile()
    dom = max(breakdown, key=lambda k: breakdown[k][1])
    v.select_kernels([dom])
    v.sync()
    prof = (0, 0.0, 0.0)
    t = time.perf_counter()
    for _ in range(steps):
        v.step()
        kp, per_launch = v.kernel_profile_all()
        if dom in kp:
            prof = (prof[0] + kp[dom][0], prof[1] + kp[dom][1], prof[2] + kp[dom][2])
    v.sync()
    dt = (time.perf_counter() - t) / steps
    got = v.status_bytes()
    assert got == expected, "%s: statuses differ from the expected statuses" % label
    checked = 0
    if oracle_sample:
        from oracle import oracle as O
        O.build(); O.lib(); O.set_threads(min(16, len(os.sched_getaffinity(0))))
        idx = list(range(0, n, max(1, n // oracle_sample)))[:oracle_sample]
        sz = 32 * n_public
        sp = b"".join(proofs[256 * i:256 * i + 256] for i in idx); si = b"".join(inputs[sz * i:sz * i + sz] for i in idx)
        ref = O.groth16_verify_many(sp, 256, vk, si, n_public, len(idx), O.MODE_REFERENCE)
        assert bytes(got[i] for i in idx) == ref, "%s: statuses differ from the oracle" % label
        checked = len(idx)
    algo = algo_bytes_per_proof(n_public)
    value = n / dt
    out = {"workload": label, "value": value, "unit": "proofs/s", "ms_per_step": dt * 1e3, "steps": steps, "batch": n, "n_public": n_public,
           "status_check": "all %d statuses == generator's; %d strided proofs == oracle" % (n, checked),
           "roofline": _valu_roofline(dom, prof, per_launch, passes=steps * (2 if per_launch < n else 1)),
           "valu_whole_path": _valu_whole_path(breakdown, value),
           "hbm_roofline": {"algorithmic_bytes_per_proof": algo, "achieved": value * algo / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": value * algo / 1e9 / HBM_PEAK_GBPS},
           "kernels_ms": {k: {"launches": c, "total_ms": round(m, 3)} for k, (c, m) in sorted(breakdown.items(), key=lambda kv: -kv[1][1])},
           "gen_seconds": round(gen_s, 2), "key_prepare_and_upload_seconds": round(prep_s, 2)}
    v.pvk.close()
    del v
    torch.cuda.empty_cache()
    return out


def plonk_workload(batch):
    """The PlonK batch of BASELINE configs[3]: the reference's 4 fixtures + copies with one flipped public-input bit at every 8th position.
    Returns (vk, proof bytes, input bytes, [proof], [inputs])."""
    import random
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))
    vk = open(os.path.join(ROOT, "tests", "golden", "plonk_vk.bin"), "rb").read()
    base = [(bytes.fromhex(f["raw_proof"]), b"".join(int(x).to_bytes(32, "big") for x in f["public_inputs"])) for f in fx.values() if f["variant"] == "plonk"]
    rng = random.Random(4)
    proofs, inputs = [], []
    for i in range(batch):
        p, q = base[i % len(base)]
        if i % 8 == 7:
            q = bytearray(q); q[rng.randrange(64)] ^= 1 << rng.randrange(8); q = bytes(q)
        proofs.append(p); inputs.append(q)
    return vk, b"".join(proofs), b"".join(inputs), proofs, inputs


def _plonk_accounting(pkg, km, batch, stage_ms):
    """Executed multiply-adds of a PlonK batch's launches from its plans.  Returns the roofline of the dominant kernel (the row kernel of the KZG check), the pairing
    stage's, and the whole path's multiply-adds per proof."""
    import ctypes as C
    L = pkg.lib()
    comp = km.get("k_g1_msm_rows", {}).get("components")
    sumc = km.get("k_g1_sum_affine", {}).get("components")
    out = {"roofline": None, "pairing": None, "whole": None, "plan": None}
    if not comp or not sumc:
        return out
    # the batch plan (sub-batches in flight, proofs per pass): the library's default, reproduced through its probe
    w, per, ps = C.c_int(), C.c_size_t(), C.c_size_t()
    L.bn254_dbg_plonk_plan.argtypes = [C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    if batch <= 9000: piece, mw = 5040, 8
    elif batch <= 20000: piece, mw = batch, 1
    elif batch <= 40000: piece, mw = (batch + 1) // 2, 2
    elif batch <= 65536: piece, mw = batch, 8
    else: piece, mw = min(batch, 131072), 8      # PLONK_BIG_PIECE_DEFAULT (csrc/bn254_capi.hip)
    assert L.bn254_dbg_plonk_plan(batch, piece, mw, C.byref(w), C.byref(per), C.byref(ps)) == 0
    pass_n = min(ps.value, batch)
    L.bn254_dbg_plonk_msm_plan.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_int),
                                            C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]

    def rows(stage):
        nr, nv = C.c_int(), C.c_int(); sc = C.c_size_t(); ch = C.c_int(); sums, fixed = (C.c_int * 2)(), (C.c_int * 2)(); desc = (C.c_int * (32 * 9))()
        assert L.bn254_dbg_plonk_msm_plan(1, stage, pass_n, 0, C.byref(nr), C.byref(nv), C.byref(sc), C.byref(ch), sums, fixed, desc) == 0
        per_row = []
        for r in range(nr.value):
            vt, lo, hi, ut, _s, _slot, flo, fhi, jmask = desc[9 * r:9 * r + 9]
            m = 0.0
            if jmask:
                g = bin(jmask).count("1")      # a joint row: g tables, 64 steps of one pair of doublings + g additions
                m += g * (comp["table_head"] + 9 * comp["table_add"]) + 64 * (2 * comp["dbl"] + g * comp.get("joint_add", comp["step"] - 2 * comp["dbl"]))
            elif vt >= 0:
                m += comp["table_head"] + 9 * comp["table_add"] + (hi - lo) / 2 * comp["step"] + lo * comp["dbl"]
            if ut >= 0:
                m += comp["unit"]
            m += (fhi - flo) * comp["mixed"] * 255.0 / 256.0       # a zero byte skips its table addition
            per_row.append(m)
        return per_row, list(sums)

    r1, s1 = rows(1)
    r2, s2 = rows(2)
    msm = sum(r1) + sum(r2)
    sums = 2 * sumc["tail"] + sumc["add"] * (s1[0] + s2[0] + s2[1]) + sumc["tail"]
    coop = pass_n <= 40960
    if coop:
        pair = km.get("k_coop12_miller_fixed", {}).get("mads_per_proof_launch")
    else:
        fe = sum(km[k].get("mads_per_proof_batch", c * km[k]["mads_per_proof_launch"]) for k, c in
                 (("k_f12_inv", 1), ("k_f12_conj", 5), ("k_f12_frob", 4), ("k_f12_cyclo_sqr", 6), ("k_f12_cyclo_sqr_n", 39), ("k_f12_mul", 60)) if k in km)
        pair = km.get("k_miller_run_fixed2", {}).get("mads_per_proof_launch", 0) + fe
    out["plan"] = {"sub_batches_in_flight": w.value, "proofs_per_pass": pass_n, "digest_rows": len(r1), "kzg_rows": len(r2), "pairing": "cooperative kernel (12 lanes per proof)" if coop else "lane kernels (k_miller_run_fixed2 + the final-exponentiation program)"}
    t2 = stage_ms.get("k_g1_msm_rows_kzg", 0)
    if t2 > 0:
        # one launch of the first sub-batch: its pass_n proofs x the rows' multiply-adds / the launch's duration (other sub-batches may run beside it: a per-launch figure)
        ach = sum(r2) * pass_n / (t2 * 1e-3)
        n_lanes = len(r2) * ((pass_n + 63) // 64 * 64)
        out["roofline"] = {"bound": "valu", "kernel": "k_g1_msm_rows (KZG check: P0 and P1)", "unit": "T mad/s", **_peak_fields(), "achieved": ach / 1e12, **_fracs(ach),
                           "avg_launch_ms": t2, "rows_per_proof": len(r2), "lanes_per_launch": n_lanes, "executed_mads_per_proof": sum(r2), "longest_row_mads": max(r2),
                           # HBM counter bytes of one launch: per-proof figure of the committed rocprofv3 --pmc passes at 4096 proofs per call (profiles/pmc_traffic_plonk.json: the
                           # average of the pass's two row launches) x the proofs of this launch; mostly the lanes' window tables (1.7 KB written once, read once per step)
                           "traffic": (lambda tr: (tr["read_bytes_per_proof"] + tr["write_bytes_per_proof"]) * pass_n if tr else None)((_load_json("pmc_traffic_plonk.json") or {}).get("k_g1_msm_rows")),
                           "note": "achieved = multiply-adds the launched rows EXECUTE (plan rows x loop bodies of the code object) x proofs of the launch / its HIP-event duration; "
                                   "%d lanes = %.2f wavefronts per SIMD, so up to one wavefront per SIMD the launch lasts as long as its longest row (%d multiply-adds)" % (n_lanes, n_lanes / 65536.0, int(max(r2)))}
    tp = stage_ms.get("pairing_check", 0)
    if pair and tp > 0:
        a2 = pair * pass_n / (tp * 1e-3)
        out["pairing"] = {"kernel": "k_coop12_miller_fixed" if coop else "k_miller_run_fixed2 + final exponentiation", "ms": tp, "mads_per_proof": pair, "achieved": a2 / 1e12, **_fracs(a2)}
    if pair:
        out["whole"] = {"mads_per_proof": msm + sums + pair, "msm_rows": msm, "sums": sums, "pairing": pair,
                        "note": "k_plonk_stage1 / k_plonk_stage2 (transcripts, Fr arithmetic on 32-bit words) carry < 1 % of a proof's multiply-adds and are not counted"}
    return out


def plonk_config(pkg, batch=4096, steps=5, warmup=1, cpu_sample=512, in_flight=True):
    """BASELINE configs[3]: PlonK batch (the reference's 4 fixtures + mutated copies, every 8th proof invalid).  `value`: proofs, inputs and status bytes RESIDENT in
    HBM (bn254_plonk_verify_batch_device), like the headline; `host_buffers`: the same batch through the host-buffer entry (pageable memory in, status bytes out: the
    PCIe-inclusive rate, never `value`).  Statuses of the first cpu_sample proofs against the oracle; cpu_baseline = the oracle's PlonK verifier on one core."""
    from oracle import oracle as O
    import torch
    vk, pb, ib, proofs, inputs = plonk_workload(batch)
    pvk = pkg.PreparedPlonkVk(vk)
    dev = torch.device("cuda", torch.cuda.current_device())
    d_p = torch.frombuffer(bytearray(pb), dtype=torch.uint8).to(dev); d_q = torch.frombuffer(bytearray(ib), dtype=torch.uint8).to(dev)
    d_st = torch.full((batch,), 0xEE, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)

    def resident(flags=0):
        pvk.verify_batch_device(d_p.data_ptr(), d_q.data_ptr(), d_st.data_ptr(), batch, proof_stride=904, device=dev.index, stream=stream.cuda_stream, flags=flags)

    # the host-buffer entry first (it also sizes the contexts' staging), then the resident entry: the timed `value`
    for _ in range(warmup):
        st_h = pvk.verify_batch(pb, ib, device=dev.index)
    t = time.perf_counter()
    for _ in range(steps):
        st_h = pvk.verify_batch(pb, ib, device=dev.index)
    dt_host = time.perf_counter() - t
    for _ in range(warmup):
        resident()
    torch.cuda.synchronize(dev)
    t = time.perf_counter()
    for _ in range(steps):
        resident()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t
    st = bytes(d_st.cpu().numpy().tobytes())
    assert st == st_h, "PlonK: the resident and the host-buffer entry disagree"
    # Accounting.  Multiply-adds EXECUTED by the launched form, from the row plans the library made for this batch size (bn254_dbg_plonk_msm_plan) priced with the
    # loop bodies of the code object (tools/count_mads.py -> kernel_mads.json `components`), against the HIP-event durations of the first sub-batch's launches.
    stage_ms, lanes = pvk.last_timing()
    km = (_load_json("kernel_mads.json") or {}).get("kernels", {})
    acct = _plonk_accounting(pkg, km, batch, stage_ms)
    roofline, pairing, whole = acct["roofline"], acct["pairing"], acct["whole"]
    if whole:
        ach = whole["mads_per_proof"] * batch * steps / dt
        whole.update({"achieved": ach / 1e12, **_peak_fields(), "unit": "T mad/s", **_fracs(ach)})
    m = min(cpu_sample, batch)
    t = time.perf_counter()
    ref = bytes(O.plonk_verify(proofs[i], vk, [int.from_bytes(inputs[i][:32], "big"), int.from_bytes(inputs[i][32:], "big")]) for i in range(m))
    cdt = time.perf_counter() - t
    assert st[:m] == ref, "PlonK: GPU statuses differ from the oracle"
    assert st.count(bytes([pkg.ACCEPT])) == batch - batch // 8
    # the same batch with several calls in flight on the one prepared key (host threads: what a verifier service with requests pending does);
    # a single batch of this size is a chain of latency-bound launches, so the aggregate rate rises until the launches of different calls share SIMDs
    import threading
    run_in_flight, in_flight = in_flight, {}
    for k in ((2, 4) if run_in_flight else ()):
        outs = [None] * k
        def work(j, rounds):
            for _ in range(rounds):
                outs[j] = pvk.verify_batch(pb, ib)
        d2 = 0.0
        rounds = max(steps, 8)
        for timed in (False, True):         # the first pass creates the contexts (streams, buffers) the extra calls need
            th = [threading.Thread(target=work, args=(j, rounds if timed else 1)) for j in range(k)]
            t2 = time.perf_counter()
            for x in th: x.start()
            for x in th: x.join()
            d2 = time.perf_counter() - t2
        assert all(o == st for o in outs), "PlonK: statuses differ between concurrent calls"
        in_flight[str(k)] = {"value": k * rounds * batch / d2, "unit": "proofs/s", "ms_per_round": d2 * 1e3 / rounds, "rounds": rounds}
    # BN254_FLAG_RLC on the PlonK entry (honoured from 8192 proofs per pass): the pairing checks of a pass batched over groups of 64 proofs -- same status bytes, checked
    rlc_mode = None
    if batch >= 8192:
        for _ in range(max(1, warmup)):
            resident(pkg.FLAG_RLC)
        t2 = time.perf_counter()
        for _ in range(steps):
            resident(pkg.FLAG_RLC)
        torch.cuda.synchronize(dev)
        d2 = time.perf_counter() - t2
        st2 = bytes(d_st.cpu().numpy().tobytes())
        assert st2 == st, "PlonK: BN254_FLAG_RLC changed a status byte"
        rlc_mode = {"workload": "the same batch with BN254_FLAG_RLC (one pairing check per 64 proofs, exact fallback on groups that fail: none in this batch -- its invalid proofs fail before the pairing check)",
                    "unit": "proofs/s", "exact": batch * steps / dt, "rlc": batch * steps / d2, "speedup": dt / d2, "ms_per_step": d2 * 1e3 / steps}
    held = pvk.footprint(dev.index)
    pvk.close()
    del d_p, d_q, d_st
    return {"workload": "BASELINE configs[3]: PlonK batch %d, 904-byte proofs (the reference's fixtures + mutations), 2 public inputs, 1/8 invalid; proofs, inputs and status bytes resident in HBM" % batch,
            "value": batch * steps / dt, "unit": "proofs/s", "ms_per_step": dt * 1e3 / steps, "steps": steps, "batch": batch,
            "host_buffers": {"value": batch * steps / dt_host, "unit": "proofs/s", "ms_per_step": dt_host * 1e3 / steps,
                             "workload": "the same batch through bn254_plonk_verify_batch (pageable host buffers in, status bytes out; PCIe-inclusive, never `value`)"},
            "context_footprint": {"bytes": held[0], "contexts": held[1]},
            "status_check": "first %d statuses == oracle; %d ACCEPT of %d" % (m, batch - batch // 8, batch),
            "calls_in_flight": in_flight, "rlc_mode": rlc_mode,
            "roofline": roofline, "pairing_check": pairing, "valu_whole_path": whole, "plan": acct["plan"], "stages_ms": {k: round(v, 3) for k, v in stage_ms.items()},
            "hbm_roofline": {"algorithmic_bytes_per_proof": 904 + 64 + 1, "achieved": batch * steps / dt * 969 / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": batch * steps / dt * 969 / 1e9 / HBM_PEAK_GBPS},
            # a rate from fewer than 256 proofs is not quoted: the entry at 4096 proofs carries the baseline of this workload
            "cpu_baseline": {"value": m / cdt, "unit": "proofs/s", "cores": 1, "kind": "port", "sample": "first %d proofs, %.1f s" % (m, cdt)} if m >= 256 else None}


def single_proof_config(pkg):
    """BASELINE configs[0] shape: the reference-shaped entry points, one proof per call (key bytes on every call as lib.rs:44-49; the prepared
    form of the last keys is cached by the library)."""
    vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540042, 2, 4, invalid_every=0, agree=True, threads=2)
    ins = [int.from_bytes(inputs[32 * i:32 * i + 32], "big") for i in range(2)]
    out = {"workload": "BASELINE configs[0]: Groth16Verifier::verify / PlonkVerifier::verify, one proof per call, vk bytes per call"}
    t = time.perf_counter()
    st = pkg.Groth16Verifier.verify(proofs[:256], vk, ins)
    out["groth16_first_call_ms"] = (time.perf_counter() - t) * 1e3
    reps = 20
    t = time.perf_counter()
    for _ in range(reps):
        st = pkg.Groth16Verifier.verify(proofs[:256], vk, ins)
    out["groth16_verify_ms"] = (time.perf_counter() - t) / reps * 1e3
    assert st == pkg.ACCEPT
    bad = bytearray(proofs[:256]); bad[200] ^= 1
    assert pkg.Groth16Verifier.verify(bytes(bad), vk, ins) != pkg.ACCEPT
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))
    pvkb = open(os.path.join(ROOT, "tests", "golden", "plonk_vk.bin"), "rb").read()
    f = [f for f in fx.values() if f["variant"] == "plonk"][0]
    pp, pi = bytes.fromhex(f["raw_proof"]), [int(x) for x in f["public_inputs"]]
    pkg.PlonkVerifier.verify(pp, pvkb, pi)
    t = time.perf_counter()
    for _ in range(reps):
        st = pkg.PlonkVerifier.verify(pp, pvkb, pi)
    out["plonk_verify_ms"] = (time.perf_counter() - t) / reps * 1e3
    assert st == pkg.ACCEPT
    out["status_check"] = "valid proof ACCEPT, tampered proof not; the reference's PlonK fixture ACCEPT"
    return out


def run_rank_plonk(args, pkg, backend, rank, world, local_rank, rehearse):
    """--plonk: ONE PlonK call of 2^batch_log2 proofs (plonk_workload: the reference's fixtures + mutations, every 8th proof invalid) sharded over the ranks by
    sharding.shard_bounds, each rank's shard resident in its GPU's HBM and verified by bn254_plonk_verify_batch_device; the status bytes are gathered like the Groth16
    ones.  Same timing contract as run_rank (barrier + synchronize on both sides, max over ranks, value = proofs of all ranks / time)."""
    import torch
    import torch.distributed as dist
    sharding = importlib.import_module("snark-bn254-verifier_amd.sharding")
    n_total = (1 << args.batch_log2) * (world if args.weak else 1)
    lo, hi = sharding.shard_bounds(n_total, world, rank)
    n = hi - lo
    t0 = time.time()
    vk, pb, ib, proofs, inputs = plonk_workload(n_total)
    gen_s = time.time() - t0
    dev = torch.device("cuda", local_rank)
    pvk = pkg.PreparedPlonkVk(vk)
    pvk.reserve(n, device=local_rank)
    d_p = torch.frombuffer(bytearray(pb[lo * 904:hi * 904]), dtype=torch.uint8).to(dev)
    d_q = torch.frombuffer(bytearray(ib[lo * 64:hi * 64]), dtype=torch.uint8).to(dev)
    d_st = torch.full((max(n, 1),), 0xEE, dtype=torch.uint8, device=dev)[:n]
    stream = torch.cuda.current_stream(dev)
    grouped = dist.is_available() and dist.is_initialized()

    def step():
        if n:
            pvk.verify_batch_device(d_p.data_ptr(), d_q.data_ptr(), d_st.data_ptr(), n, proof_stride=904, device=local_rank, stream=stream.cuda_stream,
                                    flags=pkg.FLAG_RLC if args.rlc else 0)
        if rehearse:
            return sharding.gather_status(d_st.cpu(), n_total, world)
        return sharding.gather_status(d_st, n_total, world)

    def fence():
        torch.cuda.synchronize(dev)
        if grouped:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t_start = time.perf_counter()
    full = None
    for _ in range(args.steps):
        full = step()
    fence()
    elapsed = time.perf_counter() - t_start
    if rank == 0:
        measure_valu_peak(pkg, local_rank)
    if grouped:
        t = torch.tensor([elapsed], dtype=torch.float64, device=torch.device("cpu") if backend == "gloo" else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # correctness of the timed work: the shard's bytes (ACCEPT exactly where the workload left the inputs alone; the first 16 against the oracle), and the gathered vector
    st = bytes(d_st.cpu().numpy().tobytes())
    assert all((st[i - lo] == pkg.ACCEPT) == (i % 8 != 7) for i in range(lo, hi)), "rank %d: PlonK statuses differ from the workload's" % rank
    from oracle import oracle as O
    m = min(16, n)
    assert st[:m] == bytes(O.plonk_verify(proofs[lo + i], vk, [int.from_bytes(inputs[lo + i][:32], "big"), int.from_bytes(inputs[lo + i][32:], "big")]) for i in range(m)), \
        "rank %d: PlonK statuses differ from the oracle" % rank
    assert full.numel() == n_total and bytes(full[lo:hi].cpu().numpy().tobytes()) == st, "gathered statuses are wrong"
    if rank != 0:
        return None
    value = n_total * args.steps / elapsed
    stage_ms, _lanes = pvk.last_timing(local_rank)
    km = (_load_json("kernel_mads.json") or {}).get("kernels", {})
    acct = _plonk_accounting(pkg, km, n, stage_ms)
    whole = acct["whole"]
    if whole:
        ach = whole["mads_per_proof"] * value / world
        whole.update({"achieved": ach / 1e12, **_peak_fields(), "unit": "T mad/s", **_fracs(ach)})
    out = {"metric": "PlonK verifies/sec (2 pub-inputs) at batch=2^%d%s%s" % (args.batch_log2, " per GPU" if args.weak else "", ", RLC batch mode" if args.rlc else ""),
           "value": value, "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps,
           "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
           "config": {"workload": "BASELINE configs[3] at scale: %s of 2^%d PlonK proofs%s (904 bytes; the reference's fixtures + mutations, 1/8 invalid), 2 public inputs, "
                                  "resident in HBM" % ("batches" if args.weak else "ONE batch", args.batch_log2,
                                                       " per GPU" if args.weak else (" sharded over %d GPUs (contiguous ranges)" % world if world > 1 else "")),
                      "batch_per_gpu": n, "global_batch": n_total, "n_public": 2, "mode": "rlc (one pairing check per 64 proofs, exact fallback)" if args.rlc else "exact",
                      "parallelism": "independent contiguous proof shards x%d + all_gather of status bytes" % world, "gen_seconds": round(gen_s, 1),
                      "context_footprint": dict(zip(("bytes", "contexts"), pvk.footprint(local_rank)))},
           "stages_ms": stage_ms, "roofline": acct["roofline"], "pairing_check": acct["pairing"], "valu_whole_path": whole,
           "status_check": "every status byte of every shard == the workload's (ACCEPT unless the inputs were mutated); first 16 of each shard == oracle; gathered vector checked"}
    if world == 1 and not args.no_cpu_baseline:
        m = min(512, n)
        t = time.perf_counter()
        ref = bytes(O.plonk_verify(proofs[i], vk, [int.from_bytes(inputs[i][:32], "big"), int.from_bytes(inputs[i][32:], "big")]) for i in range(m))
        cdt = time.perf_counter() - t
        assert ref == st[:m]
        out["cpu_baseline"] = {"value": m / cdt, "unit": "proofs/s", "cores": 1, "kind": "port", "sample": "first %d proofs of the batch, %.1f s; the oracle's PlonK verifier (C restatement)" % (m, cdt)}
    return out


def other_configs(args, pkg, local_rank, resident_value, keep):
    """The `configs` block of the bench line."""
    cfg = {}
    cfg["batch4096"] = device_config(pkg, local_rank, "BASELINE configs[1]: batch 4096 Groth16 proofs, 2 public inputs, inputs resident", 2, 4096, 20, 2, 0xB2540001, 64)
    cfg["plonk4096"] = plonk_config(pkg, 4096, 5, 1, 512)
    cfg["plonk65536"] = plonk_config(pkg, 65536, 3, 1, 16, in_flight=False)
    cfg["plonk262144"] = plonk_config(pkg, 262144, 3, 1, 16, in_flight=False)
    cfg["groth16_1024x4096"] = device_config(pkg, local_rank, "BASELINE configs[4]: batch 4096 Groth16 proofs, 1024 public inputs, inputs resident", 1024, 4096, 5, 1, 0xB2540004, 16)
    hb = _host_buffer_line(args, pkg, keep["vk"], keep["proofs"], keep["inputs"], keep["expected0"], local_rank)
    hb["workload"] = "the headline batch through bn254_groth16_verify_batch on pageable host buffers (PCIe-inclusive; never `value`)"
    hb["ratio_to_resident"] = hb["value"] / resident_value
    cfg["host_buffers_2^%d" % args.batch_log2] = hb
    cfg["single_proof"] = single_proof_config(pkg)
    return cfg


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, argv))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rehearse = os.environ.get("BENCH_REHEARSE_ONE_GPU") == "1" or (args.rehearse_one_gpu and world > 1)
    if rehearse:
        local_rank = 0                      # every rank drives the ONE device of the box
        args.cpu_gather = True
    assert args.gpus == world, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU: the product has no CPU path"
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run the RCCL group is formed even for one rank, so that the collective path (all_gather of the status bytes,
    # max-reduction of the elapsed time, barriers) is the one that runs
    use_dist = world > 1 or "WORLD_SIZE" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    pkg = importlib.import_module("snark-bn254-verifier_amd")
    keep = {}

    def make(vk, proofs, inputs, lr):
        keep.update(vk=vk, proofs=proofs, inputs=inputs)
        return GpuVerifier(args, pkg, vk, proofs, inputs, lr)

    def synth(seed, n_public, first, n, threads):
        r = pkg.synth_groth16(seed, n_public, n, invalid_every=16, agree=True, threads=threads, first=first)
        keep["expected0"] = r[3]
        return r

    if args.plonk:
        out = run_rank_plonk(args, pkg, "gloo" if rehearse else "nccl", rank, world, local_rank, rehearse)
        if rank == 0:
            if rehearse:
                out["rehearsal"] = "REHEARSAL: %d rank processes on ONE GPU (cuda:0), gloo for the status gather -- not a scaling number" % world
            print(json.dumps(out), flush=True)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return
    lines = []
    out = run_rank(args, make, "gloo" if rehearse else "nccl", rank, world, local_rank, synth, emit=lines.append)
    if rank == 0:
        if rehearse:
            out["rehearsal"] = ("REHEARSAL: %d rank processes on ONE GPU (cuda:0), gloo for the status gather and the reductions -- exercises run_rank end to end with the real "
                                "kernels (range generation, two processes' streams on one device, gathered order against the expected statuses); NOT a scaling number: the "
                                "ranks share one GPU and the collective is not RCCL" % world)
            out["config"]["parallelism"] += " [rehearsal on one GPU]"
        if world == 1:
            headline = (args.n_public, args.batch_log2, bool(args.rlc), bool(args.weak)) == (2, 20, False, False)
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = _cpu_baseline(args, keep["vk"], keep["proofs"], keep["inputs"], keep["expected0"])
            if headline and not args.no_configs:
                out["configs"] = other_configs(args, pkg, local_rank, out["value"], keep)
            elif args.host_buffers:
                out["host_buffers"] = _host_buffer_line(args, pkg, keep["vk"], keep["proofs"], keep["inputs"], keep["expected0"], local_rank)
            if not args.no_rlc and not args.rlc and args.n_public <= 8:
                out["rlc_mode"] = _rlc_line(args, pkg, local_rank)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
