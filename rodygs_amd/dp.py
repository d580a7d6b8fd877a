"""Frame-data-parallel plumbing: one process per GPU, every rank renders a different video frame
of the SAME replicated Gaussian set, gradients are summed with ONE all-reduce over one flat fp32 bucket
(RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests), then every rank applies the same fused Adam.

The reference has no distributed code; this is the build's own capability.  Nothing here is a
data-path collective inside the rasterizer: the only exchange is the gradient sum.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
import torch.distributed as dist


class FlatParams:
    """All Gaussian parameters in ONE contiguous fp32 buffer (params, grads, Adam moments alike), so that the
    gradient exchange is a single collective and the optimiser a handful of fused launches.

    ``spec``: ordered {name: (shape, lr)}.  Tensors are views into the flat buffers; ``.grad`` of every parameter
    is a view into ``flat_grad`` (autograd accumulates in place into an existing .grad)."""

    def __init__(self, spec: Dict[str, Tuple[Sequence[int], float]], device, align: int = 64, storage=None):
        """``storage``: optional (flat, flat_grad, exp_avg, exp_avg_sq) float32 buffers of at least this layout's size to lay
        the segments out in, instead of allocating (``FlatStorage``: a trainer that densifies every 100 steps re-uses two
        such sets in turn -- a fresh gigabyte-sized allocation per densification stalls the host for tens of milliseconds).
        Whoever passes them is responsible for their contents; this constructor only clears the gradient buffer and the
        alignment padding of the other three."""
        self.names: List[str] = list(spec)
        self.lr: Dict[str, float] = {k: float(v[1]) for k, v in spec.items()}
        self.shapes = {k: tuple(v[0]) for k, v in spec.items()}
        self.offsets: Dict[str, Tuple[int, int]] = {}
        off = 0
        for k in self.names:
            n = 1
            for s in self.shapes[k]:
                n *= int(s)
            self.offsets[k] = (off, n)
            off += (n + align - 1) // align * align
        self.numel = off
        self.storage = storage
        if storage is None:
            self.flat = torch.zeros(off, dtype=torch.float32, device=device)
            self.flat_grad = torch.zeros(off, dtype=torch.float32, device=device)
            self.exp_avg = torch.zeros(off, dtype=torch.float32, device=device)
            self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=device)
        else:
            if any(b.numel() < off or b.dtype != torch.float32 or not b.is_contiguous() for b in storage.buffers):
                raise ValueError("FlatParams: the storage is smaller than the layout (or not contiguous float32)")
            self.flat, self.flat_grad, self.exp_avg, self.exp_avg_sq = (b[:off] for b in storage.buffers)
            with torch.no_grad():
                self.flat_grad.zero_()
                for k in self.names:             # alignment padding: Adam steps merged neighbours across it
                    o, n = self.offsets[k]
                    end = (o + n + align - 1) // align * align
                    if end > o + n:
                        for b in (self.flat, self.exp_avg, self.exp_avg_sq):
                            b[o + n:end].zero_()
        self.params: Dict[str, torch.Tensor] = {}
        for k in self.names:
            o, n = self.offsets[k]
            p = self.flat[o:o + n].view(self.shapes[k])
            p.requires_grad_(True)
            p.grad = self.flat_grad[o:o + n].view(self.shapes[k])
            self.params[k] = p
        self.step_count = 0

    def __getitem__(self, k):
        return self.params[k]

    def zero_grad(self):
        self.flat_grad.zero_()

    def segment(self, buf: torch.Tensor, k: str) -> torch.Tensor:
        o, n = self.offsets[k]
        return buf[o:o + n]


class FlatStorage:
    """Four float32 buffers of ``capacity`` elements a ``FlatParams`` layout can be placed in (``FlatParams(storage=)``)."""

    def __init__(self, capacity: int, device):
        self.capacity = int(capacity)
        self.buffers = tuple(torch.empty(self.capacity, dtype=torch.float32, device=device) for _ in range(4))


def flat_numel(spec: Dict[str, Tuple[Sequence[int], float]], align: int = 64) -> int:
    """Elements a FlatParams of this spec occupies (the layout rule of ``FlatParams.__init__``)."""
    off = 0
    for shape, _ in spec.values():
        n = 1
        for s_ in shape:
            n *= int(s_)
        off += (n + align - 1) // align * align
    return off


def allreduce_sum_(flat_grad: torch.Tensor, extra: Sequence[torch.Tensor] = ()) -> None:
    """Sum the flat gradient bucket (and any small side tensors, e.g. MLP / pose grads) over all ranks."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    if extra:
        small = torch.cat([t.reshape(-1) for t in extra])
        dist.all_reduce(small, op=dist.ReduceOp.SUM)
        o = 0
        for t in extra:
            n = t.numel()
            t.copy_(small[o:o + n].view_as(t))
            o += n


class BucketedAllReduce:
    """The same sum as ``allreduce_sum_``, issued in pieces so that it overlaps with what is left of the step.

    * ``ready(name)`` -- called from inside backward the moment a segment's gradient is final (the SH-feature
      gradient, 64 % of the bytes, is complete when the per-Gaussian backward kernel has been queued, while the
      activation / deformation / MLP backward still has to run): its all-reduce starts right then.
    * ``finish()`` -- after backward: every segment not sent yet, as maximal contiguous ranges, plus the small
      (MLP + pose) bucket.
    * ``drain()`` -- yields the segment names of each piece as soon as that piece has arrived (stream-ordered wait,
      no host block with RCCL), so the fused Adam of a piece runs while the next piece is still on the wire.
    Collectives are asynchronous ``torch.distributed`` calls: RCCL runs them on its own stream in issue order."""

    def __init__(self, fp: FlatParams, extra: Sequence[torch.Tensor] = ()):
        self.fp = fp
        self.extra = list(extra)
        self._pending: List[Tuple[object, object]] = []
        self._sent: set = set()

    @staticmethod
    def active() -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def ready(self, name: str) -> None:
        if not self.active() or name in self._sent or name not in self.fp.offsets:
            return
        o, n = self.fp.offsets[name]
        work = dist.all_reduce(self.fp.flat_grad[o:o + n], op=dist.ReduceOp.SUM, async_op=True)
        self._pending.append((work, [name]))
        self._sent.add(name)

    def finish(self) -> None:
        if not self.active():
            return
        run: List[str] = []

        def flush():
            if run:
                o0 = self.fp.offsets[run[0]][0]
                o1, n1 = self.fp.offsets[run[-1]]
                work = dist.all_reduce(self.fp.flat_grad[o0:o1 + n1], op=dist.ReduceOp.SUM, async_op=True)
                self._pending.append((work, list(run)))
                del run[:]

        for k in self.fp.names:
            if k in self._sent:
                flush()
            else:
                run.append(k)
        flush()
        for t in self.extra:
            self._pending.append((dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True), None))

    def drain(self):
        pending, self._pending, self._sent = self._pending, [], set()
        for work, names in pending:
            work.wait()
            yield names


def frame_for(step: int, rank: int, world: int, perm: Sequence[int]) -> int:
    """Strided view of the reference's permutation sampler (src/data/dataloader.py:47-71): rank r renders
    perm[(step*world + r) mod T]."""
    return int(perm[(step * world + rank) % len(perm)])
