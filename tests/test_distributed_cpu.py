"""world_size-2 (and 3) rehearsal of the multi-GPU path on CPU with the gloo backend: contiguous sharding of the batch and
the single all_gather of status bytes (snark-bn254-verifier_amd/sharding.py, used by bench.py under RCCL)."""
import importlib
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = importlib.import_module("snark-bn254-verifier_amd.sharding")
    lo, hi = sh.shard_bounds(n, world, rank)
    # stand-in for the per-rank GPU result: a deterministic function of the global proof index
    local = torch.tensor([(7 * i + 3) % 5 for i in range(lo, hi)], dtype=torch.uint8)
    dist.barrier()
    full = sh.gather_status(local, n, world)
    if rank == 0:
        q.put(full.tolist())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 1000), (2, 7), (3, 1001)])
def test_shard_and_gather(world, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == [(7 * i + 3) % 5 for i in range(n)]


def test_shard_bounds_cover():
    sh = importlib.import_module("snark-bn254-verifier_amd.sharding")
    for n in (0, 1, 7, 4096, 1 << 20):
        for w in (1, 2, 3, 8):
            b = [sh.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


# ---- bench.py's rank logic under gloo with a stand-in verifier, and its self-launch ------------------------------------------------------------------
class _StandIn:
    """Stand-in for bench.GpuVerifier on a CPU box: 'verifies' a proof by looking its bytes up in the generator's table (so the
    rotation of the shards, the gather and the expected-status check of bench.run_rank are exercised, not the kernels)."""

    def __init__(self, table, proofs):
        import time
        self._t = time
        self.status = torch.tensor([table[proofs[256 * i:256 * i + 256]] for i in range(len(proofs) // 256)], dtype=torch.uint8)

    def step(self): return self.status
    def sync(self): pass
    def timer(self): return self._t.perf_counter()
    def elapsed_ms(self, a, b): return (b - a) * 1e3
    def select_kernels(self, names): pass
    def kernel_profile(self): return {}, self.status.numel()
    def kernel_profile_all(self): return {}, self.status.numel()
    def phases(self): return {}
    def status_bytes(self): return bytes(self.status.numpy().tobytes())


def _bench_worker(rank, world, port, q, weak):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    pkg = importlib.import_module("snark-bn254-verifier_amd")
    args = bench.parse_args(["--gpus", str(world), "--steps", "2", "--warmup", "1", "--batch-log2", "6"] + (["--weak"] if weak else []))
    table, asked = {}, []

    def synth(seed, n_public, first, n, threads):
        asked.append((first, n))
        vk, proofs, inputs, exp = pkg.synth_groth16(seed, n_public, n, invalid_every=4, agree=True, threads=2, first=first)
        for i in range(n):
            table[proofs[256 * i:256 * i + 256]] = exp[i]
        return vk, proofs, inputs, exp

    lines, fulls = [], []

    class Recording(_StandIn):
        pass

    def make(vk, proofs, inputs, lr):
        return Recording(table, proofs)

    # keep the gathered vector of the last step: run_rank checks this rank's slice; the test checks the WHOLE order on rank 0
    sh = importlib.import_module("snark-bn254-verifier_amd.sharding")
    orig = sh.gather_status

    def spy(local, n, w):
        full = orig(local, n, w)
        fulls.append(bytes(full.numpy().tobytes()))
        return full

    sh.gather_status = spy
    bench.run_rank(args, make, "gloo", rank, world, rank, synth, emit=lines.append)
    if rank == 0:
        q.put((lines, asked, fulls[-1]))
    dist.barrier()
    dist.destroy_process_group()


def _run_bench_ranks(world, weak):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q, weak)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_bench_rank_logic_under_gloo_strong_scaling():
    """BASELINE configs[2] / SURVEY.md section 8(e): ONE batch, contiguous shards; `--gpus 2` halves the per-GPU batch."""
    import json
    lines, asked, full = _run_bench_ranks(2, weak=False)
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 64 and out["config"]["batch_per_gpu"] == 32
    assert out["scaling"] == "strong" and out["steps"] == 2 and out["value"] > 0 and "gather_ms" in out["config"]
    assert out["hbm_roofline"]["algorithmic_bytes_per_proof"] == 256 + 32 * 2 + 1
    assert asked == [(0, 32)]                       # rank 0 generated exactly its own contiguous range
    pkg = importlib.import_module("snark-bn254-verifier_amd")
    whole = pkg.synth_groth16(__import__("bench").SEED, 2, 64, invalid_every=4, agree=True, threads=2)[3]
    assert full == whole                            # gathered order = the order of the unsharded batch


def test_bench_rank_logic_under_gloo_weak_scaling():
    import json
    lines, asked, full = _run_bench_ranks(2, weak=True)
    out = json.loads(lines[0])
    assert out["scaling"] == "weak" and out["config"]["global_batch"] == 128 and out["config"]["batch_per_gpu"] == 64
    assert asked == [(0, 64)] and len(full) == 128


def test_strong_scaling_shard_sizes():
    """2^20 over 2 / 4 / 8 GPUs = 2^19 / 2^18 / 2^17 proofs per GPU, contiguous (131 072 at N = 8)."""
    sh = importlib.import_module("snark-bn254-verifier_amd.sharding")
    for w in (1, 2, 4, 8):
        b = [sh.shard_bounds(1 << 20, w, r) for r in range(w)]
        assert [hi - lo for lo, hi in b] == [(1 << 20) // w] * w
        assert [lo for lo, _ in b] == [r * ((1 << 20) // w) for r in range(w)]


def test_verify_batch_multi_planner_matches_the_rank_sharding():
    """bn254_shard_plan (the single-process multi-GPU entry's planner, host arithmetic) cuts the batch exactly like sharding.shard_bounds."""
    pkg = importlib.import_module("snark-bn254-verifier_amd")
    sh = importlib.import_module("snark-bn254-verifier_amd.sharding")
    for n in (0, 1, 7, 4096, (1 << 20) + 777):
        for mask, cnt in ((0b1, 1), (0b11, 2), (0b10110, 8), (0xff, 8)):
            plan = pkg.shard_plan(n, mask, cnt)
            devs = [b for b in range(64) if (mask >> b) & 1]
            assert [d for d, _, _ in plan] == devs
            assert [(f, f + c) for _, f, c in plan] == [sh.shard_bounds(n, len(devs), r) for r in range(len(devs))]
    assert [c for _, _, c in pkg.shard_plan(1 << 20, 0xff, 8)] == [131072] * 8
    with pytest.raises(pkg.Bn254Error):
        pkg.shard_plan(100, 0b100, 2)               # device 2 of a 2-device host
    with pytest.raises(pkg.Bn254Error):
        pkg.shard_plan(100, 0, 8)


def test_synth_range_is_a_slice_of_the_stream():
    pkg = importlib.import_module("snark-bn254-verifier_amd")
    a = pkg.synth_groth16(77, 2, 96, invalid_every=4, agree=True, threads=2)
    b = pkg.synth_groth16(77, 2, 40, invalid_every=4, agree=True, threads=3, first=50)
    assert a[0] == b[0] and a[1][256 * 50:256 * 90] == b[1] and a[2][64 * 50:64 * 90] == b[2] and a[3][50:90] == b[3]


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (how the driver invokes it) starts two ranks itself; on this GPU-less box both
    get as far as the GPU check."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch-log2", "6"],
                       env=env, capture_output=True, text=True, timeout=600)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the GPU run")
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a GPU") >= 2, r.stderr[-2000:]


def test_bench_plonk_mode_arguments_and_shards():
    """`bench.py --plonk` (the PlonK call sharded like the Groth16 batch): 2^18 proofs unless told otherwise, the headline stays 2^20; the ranks' contiguous ranges of
    the PlonK workload tile it exactly (what run_rank_plonk uploads per rank), for ragged counts too."""
    import bench
    assert bench.parse_args([]).batch_log2 == 20 and not bench.parse_args([]).plonk
    a = bench.parse_args(["--plonk", "--gpus", "8"])
    assert a.plonk and a.batch_log2 == 18 and a.gpus == 8
    assert bench.parse_args(["--plonk", "--batch-log2", "12"]).batch_log2 == 12
    sharding = importlib.import_module("snark-bn254-verifier_amd.sharding")
    vk, pb, ib, proofs, inputs = bench.plonk_workload(37)
    for world in (1, 2, 3, 8):
        got_p, got_i = b"", b""
        for r in range(world):
            lo, hi = sharding.shard_bounds(37, world, r)
            got_p += pb[lo * 904:hi * 904]; got_i += ib[lo * 64:hi * 64]
        assert got_p == pb and got_i == ib
    assert all(len(p) == 904 for p in proofs) and sum(inputs[i] != inputs[i % 4] for i in range(37)) == 37 // 8
