"""Data-parallel gradient exchange for the DGMR step: one process per GPU, RCCL all-reduce over xGMI.

The reference has no distributed code; Lightning would wrap it in stock DDP.  The step has
two optimisers and three backward passes, so the exchange is explicit here:

* gradients live in ONE flat fp32 buffer per network (generator 204 MiB, discriminator 170 MiB at the paper
  config); each ``param.grad`` is a strided view into it, so the kernels' direct ``param.grad`` accumulation
  fills the buffer and no pack/unpack copies exist;
* after a D backward only D's buffer is reduced, after the G backward only G's — the cross-network gradients
  the reference computes and discards (Q6) never touch the wire;
* the buffer is reduced in a few large chunks (xGMI links are point-to-point, ~153 GB/s each: fewer, larger
  collectives), SUM then a 1/world scale (HIP kernel on GPU tensors);
* the 12 parameters that never receive a gradient simply stay zero in the buffer (a stock DDP reducer would
  stall on them);
* buffers (u/v, BN running statistics) are broadcast from rank 0 once per step, like DDP's broadcast_buffers.

The CPU branch of ``_scale`` exists only so the collective logic can be exercised under gloo in tests.
"""
from __future__ import annotations

from typing import Dict, Iterable, List

import torch
import torch.distributed as dist


class FlatGrads:
    """One flat gradient buffer for a list of parameters; ``p.grad`` become views with p's own strides."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        seen, self.params = set(), []
        for p in params:
            if id(p) not in seen and p.requires_grad:
                seen.add(id(p))
                self.params.append(p)
        total = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        self.flat = torch.zeros(total, device=p0.device, dtype=p0.dtype)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].as_strided(p.shape, p.stride())
            off += n

    def zero_(self):
        self.flat.zero_()
        off = 0
        for p in self.params:  # re-attach in case something dropped the views
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * off:
                p.grad = self.flat[off:off + n].as_strided(p.shape, p.stride())
            off += n


class GradSync:
    def __init__(self, model, process_group=None, chunk_mb: int = 64):
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.gen = FlatGrads(model.generator.parameters())
        self.disc = FlatGrads(model.discriminator.parameters())
        self.chunk = chunk_mb * (1 << 20) // 4
        self.model = model
        self._buffers = [b for _, b in model.named_buffers() if b.is_floating_point() and b.numel() > 0 and b.dim() > 0
                         and not _.endswith("_scratch")]
        # de-duplicate the generator.* aliases
        seen, uniq = set(), []
        for b in self._buffers:
            if id(b) not in seen:
                seen.add(id(b))
                uniq.append(b)
        self._buffers = uniq

    def flat_for(self, which: str) -> FlatGrads:
        return self.gen if which == "g" else self.disc

    def _scale(self, t: torch.Tensor, s: float):
        if t.is_cuda:
            from . import ops

            ops.call("dgmr_axpby", t.data_ptr(), None, t.data_ptr(), float(s), 0.0, t.numel(), ops._stream())
        else:
            t.mul_(s)

    def sync(self, which: str):
        """All-reduce (mean) the gradient buffer of network `which` ('g' or 'd')."""
        if self.world == 1:
            return
        flat = self.flat_for(which).flat
        for o in range(0, flat.numel(), self.chunk):
            piece = flat[o:o + self.chunk]
            dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.pg)
        self._scale(flat, 1.0 / self.world)

    def broadcast_buffers(self, src: int = 0):
        if self.world == 1 or not self._buffers:
            return
        flat = torch.cat([b.reshape(-1) for b in self._buffers])
        dist.broadcast(flat, src=src, group=self.pg)
        off = 0
        with torch.no_grad():
            for b in self._buffers:
                n = b.numel()
                b.copy_(flat[off:off + n].view_as(b))
                off += n

    def broadcast_parameters(self, src: int = 0):
        if self.world == 1:
            return
        for fg in (self.gen, self.disc):
            for p in fg.params:
                dist.broadcast(p.data, src=src, group=self.pg)
        # p.data writes bump neither p._version nor the optimiser's epoch, which key the W W^T / flipped / split-plane caches:
        # any forward that ran before this broadcast (warm-up, smoke, eval) must not leave stale copies behind on ranks != src
        from . import ops

        ops.bump_weights_epoch()
