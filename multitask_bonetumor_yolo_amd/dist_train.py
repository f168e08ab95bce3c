"""Data-parallel training plumbing: flat gradient buckets with one all-reduce each, the
parameter broadcast at start-up, and the fused AdamW step over flat buckets.

One process per GPU; `torch.distributed` backend "nccl" is RCCL over xGMI on ROCm, "gloo" is used by the CPU tests.  xGMI is
point to point, so a ring all-reduce is bound by one link (~153 GB/s per direction): ~25 MB buckets keep each collective long
enough to reach link bandwidth while leaving several buckets to overlap with the rest of the backward pass; 179 MB of fp32
gradients (44.8 M parameters) are 8 buckets.  The buckets are filled in REVERSE registration order, the order in which a
backward pass produces gradients, so bucket 0 can be reduced while earlier layers are still being differentiated.

`FlatBuckets` is also the storage of the training step's parameter / gradient arenas (`train.make_arena`, kernel layouts, a leading
bucket for parameters the loss never reaches); the step that fills the gradient buckets, all-reduces each one under the rest of the
backward pass and applies the fused optimiser is `trainstep.TrainStep` (DESIGN.md sections 6-7)."""
import ctypes as C
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import _lib as L

BUCKET_BYTES = 25 << 20


class FlatBuckets:
    """fp32 storage for a set of named tensors, packed into flat buckets of about `bucket_bytes` in reverse registration order.

    `views[name]` is a tensor of the registered shape aliasing its slice of a bucket; every slice starts on a 16-byte
    boundary (the kernels' vector accesses)."""

    def __init__(self, named_shapes: Sequence[Tuple[str, Sequence[int]]], device, bucket_bytes: int = BUCKET_BYTES, close_after: Sequence[str] = ()):
        """`close_after`: names after which (in fill order = reverse registration order) the current bucket is closed whatever its
        size -- a group of tensors that must not share a bucket with the rest (parameters that never receive a gradient)."""
        self.views: Dict[str, torch.Tensor] = {}
        self.buckets: List[torch.Tensor] = []
        self.layout: List[List[Tuple[str, int, int]]] = []          # per bucket: (name, offset, numel)
        self.where: Dict[str, Tuple[int, int, int, Tuple[int, ...]]] = {}   # name -> (bucket, offset, numel, shape)
        cur, off = [], 0
        close_after = set(close_after)
        for name, shape in reversed(list(named_shapes)):
            n = 1
            for d in shape:
                n *= int(d)
            cur.append((name, off, n, tuple(shape)))
            off += (n + 3) // 4 * 4
            if off * 4 >= bucket_bytes or name in close_after:
                self._close(cur, off, device)
                cur, off = [], 0
        if cur:
            self._close(cur, off, device)

    def _close(self, items, total, device):
        flat = torch.zeros(total, dtype=torch.float32, device=device)
        self.buckets.append(flat)
        self.layout.append([(n, o, k) for n, o, k, _ in items])
        for name, o, k, shape in items:
            self.views[name] = flat[o:o + k].view(shape)
            self.where[name] = (len(self.buckets) - 1, o, k, shape)

    def snapshot_views(self, names) -> Dict[str, torch.Tensor]:
        """Views (registered shapes) into FRESH copies of the buckets that hold `names`: one flat copy per bucket instead of one per
        tensor.  What an autograd node may hand out: the persistent buckets are overwritten by the next backward pass."""
        copies: Dict[int, torch.Tensor] = {}
        out = {}
        for n in names:
            b, o, k, shape = self.where[n]
            if b not in copies:
                copies[b] = self.buckets[b].clone()
            out[n] = copies[b][o:o + k].view(shape)
        return out

    def zero_(self):
        for b in self.buckets:
            b.zero_()

    def all_reduce_mean(self, stream: Optional[torch.cuda.Stream] = None):
        """Average every bucket over the ranks, one collective per bucket in bucket order (= the order a backward pass
        completes them).  `stream`: a side stream for the collectives (compute keeps running on the current stream); the
        returned work handles must be waited on before the optimiser reads the buckets."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return []
        world = dist.get_world_size()
        works = []
        if stream is not None:
            # the compute stream as seen OUTSIDE the side-stream context (inside it current_stream() is the side stream itself and the
            # wait would be a no-op: the collective could then start before the gradients are written)
            stream.wait_stream(torch.cuda.current_stream(self.buckets[0].device if self.buckets[0].is_cuda else None))
        ctx = torch.cuda.stream(stream) if stream is not None else _null()
        with ctx:
            for b in self.buckets:
                b.div_(world)                                      # pre-scale: SUM of the scaled buckets = mean, no overflow headroom lost
                works.append(dist.all_reduce(b, op=dist.ReduceOp.SUM, async_op=True))
        return works


class _null:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def broadcast_parameters(tensors: Iterable[torch.Tensor], src: int = 0):
    """Rank `src`'s parameters (and buffers) to every rank at start-up, as DDP does at construction."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    for t in tensors:
        dist.broadcast(t.data if isinstance(t, torch.nn.Parameter) else t, src=src)


class FlatAdamW:
    """torch.optim.AdamW over flat buckets with the fused HIP step (csrc/optim.hip): parameters, gradients and both moments
    live in `FlatBuckets` of identical layout.  Defaults = the reference trainer's (running_main_v3.py:732-734)."""

    def __init__(self, params: FlatBuckets, grads: FlatBuckets, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0005):
        assert [len(b) for b in params.buckets] == [len(b) for b in grads.buckets], "parameter and gradient buckets differ in layout"
        self.params, self.grads = params, grads
        self.exp_avg = [torch.zeros_like(b) for b in params.buckets]
        self.exp_avg_sq = [torch.zeros_like(b) for b in params.buckets]
        self.lr, self.betas, self.eps, self.weight_decay, self.steps = lr, betas, eps, weight_decay, 0

    def step(self):
        lib = L.load()
        self.steps += 1
        for p, g, m, v in zip(self.params.buckets, self.grads.buckets, self.exp_avg, self.exp_avg_sq):
            if not p.is_cuda:
                raise RuntimeError("FlatAdamW.step: expected CUDA/HIP buckets on an MI355X (no CPU path)")
            L.check(lib.mtbt_adamw_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), self.lr, self.betas[0], self.betas[1],
                                        self.eps, self.weight_decay, self.steps, None, C.c_void_p(torch.cuda.current_stream(p.device).cuda_stream)),
                    "mtbt_adamw_step")

    def cosine_lr(self, base_lr: float, epoch: int, t_max: int, eta_min_ratio: float = 0.01):
        """CosineAnnealingLR(T_max, eta_min = 0.01 * lr) in closed form (running_main_v3.py:742), applied per epoch."""
        import math
        eta_min = base_lr * eta_min_ratio
        self.lr = eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch / t_max)) / 2
        return self.lr
