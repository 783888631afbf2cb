"""Multi-GPU sharding of a proof batch: proofs are independent, so [0, n) is cut into `world`
contiguous ranges, each rank verifies its own range with no data-path communication, and ONE collective gathers the
accept/reject bytes (torch.distributed all_gather; backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in CPU tests)."""
import torch
import torch.distributed as dist


def shard_bounds(n, world, rank):
    """Contiguous, balanced: the first n % world ranks get one extra proof."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_status(local_status, n, world):
    """local_status: uint8 tensor of this rank's shard (device of the backend). Returns the full n-byte vector on every rank."""
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return local_status
    cap = (n + world - 1) // world
    padded = torch.zeros(cap, dtype=torch.uint8, device=local_status.device)
    padded[: local_status.numel()] = local_status
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    out = []
    for r in range(world):
        lo, hi = shard_bounds(n, world, r)
        out.append(parts[r][: hi - lo])
    return torch.cat(out)
