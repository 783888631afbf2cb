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
