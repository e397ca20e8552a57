"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, "gloo"
in the CPU tests).  Images are independent units: a batch is cut into contiguous per-rank
ranges and no pixel or coefficient ever crosses a GPU.  The only collectives are a 32-byte work descriptor
broadcast from rank 0 and an all-gather of the per-rank result summaries."""
from typing import List, Tuple


def shard_range(base: int, count: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous range [lo, hi) of rank `rank`: sizes differ by at most one, union = [base, base+count)."""
    q, r = divmod(count, world)
    lo = base + rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def broadcast_descriptor(dist, device, base: int = 0, count: int = 0, quality: int = 20, seed: int = 0):
    """Rank 0's {base, count, quality, seed} reaches every rank (4 x int64 = 32 bytes)."""
    import torch
    t = torch.tensor([base, count, quality, seed], dtype=torch.int64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, 0)
    return tuple(int(v) for v in t.tolist())


def gather_summaries(dist, device, bytes_out: int, checksum: int, images_ok: int) -> List[Tuple[int, int, int]]:
    import torch
    mine = torch.tensor([bytes_out, checksum, images_ok], dtype=torch.int64, device=device)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [tuple(int(v) for v in mine.tolist())]
    outs = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, mine)
    return [tuple(int(v) for v in o.tolist()) for o in outs]


def max_over_ranks(dist, device, seconds: float) -> float:
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
