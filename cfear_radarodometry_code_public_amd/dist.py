"""Multi-GPU plumbing (BASELINE configs[3]): independent sequences are sharded across
ranks with no data-path collective; the only exchange is the final throughput reduction
{scans: SUM, seconds: MAX} over torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_sequences(n_total, rank, world):
    """Sequence q -> rank q % world. Returns the global sequence ids of `rank`."""
    return [q for q in range(n_total) if q % world == rank]


def reduce_throughput(scans, seconds, device=None):
    """Whole-job (total scans, max seconds). Works without an initialised process group (world 1)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(scans), float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    s = torch.tensor([float(scans)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return float(s.item()), float(t.item())
