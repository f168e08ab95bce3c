"""Multi-GPU plumbing of the inference path: one process per GPU, images are independent, so a global
batch is SHARDED over ranks with no data-path collective; ranks meet only to time a run (barrier + MAX of
the elapsed times) and, optionally, to gather per-image detections.  Works with backend "nccl" (= RCCL over
xGMI on ROCm) and "gloo" (CPU tests)."""
import time
from typing import Callable, List, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous [start, end) of the items rank owns; sizes differ by at most one, earlier ranks take the extra."""
    if world_size <= 0 or not (0 <= rank < world_size) or n_items < 0:
        raise ValueError("bad shard arguments")
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def timed_steps(step: Callable[[], object], steps: int, sync: Callable[[], None]) -> float:
    """Run `steps` calls of `step` between two fences (sync + barrier + sync) and return the MAX elapsed seconds
    over all ranks -- the bench.py contract."""
    def fence():
        sync()
        if dist.is_available() and dist.is_initialized():
            dist.barrier()
        sync()

    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    return max_over_ranks(elapsed)


def max_over_ranks(value: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_counts(local_counts: torch.Tensor) -> List[torch.Tensor]:
    """All ranks' per-image kept-box counts (evaluation-time metric sync; not on the timed path)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [local_counts]
    sizes = [torch.zeros(1, dtype=torch.int64, device=local_counts.device) for _ in range(dist.get_world_size())]
    dist.all_gather(sizes, torch.tensor([local_counts.numel()], dtype=torch.int64, device=local_counts.device))
    m = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros(m, dtype=local_counts.dtype, device=local_counts.device)
    pad[: local_counts.numel()] = local_counts
    outs = [torch.zeros_like(pad) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, pad)
    return [o[: int(s.item())] for o, s in zip(outs, sizes)]
