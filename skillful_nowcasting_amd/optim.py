"""Adam through the fused HIP kernel (torch.optim.Adam semantics as used at dgmr/dgmr.py:292-300)."""
import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    """Adam(lr, betas, eps=1e-8, weight_decay=0, amsgrad=False); one HIP launch per parameter GROUP (dgmr_adam_multi).

    State layout (``step``, ``exp_avg``, ``exp_avg_sq``) matches torch.optim.Adam so optimiser checkpoints
    interchange.  Parameters that received no gradient are skipped, like torch does.
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    flat_grads = None  # set by ddp.GradSync users: gradients then live in one flat buffer that is zeroed, not dropped

    def zero_grad(self, set_to_none: bool = True):
        if self.flat_grads is not None:
            self.flat_grads.zero_()
            return
        super().zero_grad(set_to_none=set_to_none)

    RING = 4  # pinned descriptor tables in flight (see _step_multi)
    multi_tensor = True  # every tensor of a parameter group in ONE launch (dgmr_adam_multi); False: one dgmr_adam launch per tensor

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        written = []
        for group in self.param_groups:
            b1, b2 = group["betas"]
            todo = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                written.append(p)
                ops.require_hip(p, "parameter")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                g = p.grad
                if g.stride() != p.stride():  # kernels index raw storage: bring the gradient to the parameter's layout
                    g = torch.empty_like(p).copy_(g)
                if self.multi_tensor:
                    todo.append((p, g, st))
                else:
                    ops.adam_update(p, g, st["exp_avg"], st["exp_avg_sq"], st["step"], group["lr"], b1, b2, group["eps"])
            if todo:
                self._step_multi(todo, group["lr"], b1, b2, group["eps"])
        ops._core.note_optimizer_step(written)  # (the images of THESE weights are stale; everything else keeps its caches)
        return loss

    def _step_multi(self, todo, lr, b1, b2, eps):
        """One descriptor per tensor (pointers, length, first workgroup, the two step-dependent scalars - torch keeps a step counter per
        parameter), uploaded as one pinned-memory copy, then ONE launch.  The scalars are formed in double and rounded once, like
        dgmr_adam does."""
        import math

        import numpy as np

        from ._lib import ADAM_DESC_DTYPE, load

        chunk = self.__dict__.get("_chunk")
        if chunk is None:
            chunk = self.__dict__["_chunk"] = int(load().dgmr_adam_chunk())
        n = len(todo)
        # The upload is asynchronous and the step never synchronises the host: a pinned table must not be rewritten before the copy
        # that reads it has executed (two step() calls per training step, several parameter groups per call).  A ring of pinned
        # tables, each with the event of its last copy; a slot is reused only after that event.
        nbytes = n * ADAM_DESC_DTYPE.itemsize
        ring = self.__dict__.setdefault("_desc_ring", [])
        turn = self.__dict__["_desc_turn"] = self.__dict__.get("_desc_turn", -1) + 1
        if len(ring) < self.RING:
            ring.append(None)
        slot = turn % len(ring)
        ent = ring[slot]
        if ent is None or ent[0].numel() < nbytes:
            if ent is not None:
                ent[2].synchronize()
            ent = ring[slot] = (torch.empty(nbytes, dtype=torch.uint8).pin_memory(),
                                torch.empty(nbytes, dtype=torch.uint8, device=todo[0][0].device), torch.cuda.Event())
        else:
            ent[2].synchronize()  # (long done in practice: RING - 1 other uploads lie in between)
        host, dev, done = ent
        tab = np.frombuffer(host.numpy(), dtype=ADAM_DESC_DTYPE, count=n)
        block = 0
        for i, (p, g, st) in enumerate(todo):
            numel = p.numel()
            step = st["step"]
            tab[i] = (p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), numel, block,
                      np.float32(lr / (1.0 - b1 ** step)), np.float32(math.sqrt(1.0 - b2 ** step)), 0)
            block += (numel + chunk - 1) // chunk
        dev[:nbytes].copy_(host[:nbytes], non_blocking=True)
        done.record()
        ops.call("dgmr_adam_multi", dev.data_ptr(), n, block, float(b1), float(b2), float(eps), ops._stream())
