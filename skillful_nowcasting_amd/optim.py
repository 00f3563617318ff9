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

    multi_tensor = True  # every tensor of a parameter group in ONE launch (dgmr_adam_multi); False: one dgmr_adam launch per tensor

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            todo = []
            for p in group["params"]:
                if p.grad is None:
                    continue
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
        ops.bump_weights_epoch()
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
        host = self.__dict__.get("_desc_host")
        if host is None or host.numel() < n * ADAM_DESC_DTYPE.itemsize:
            host = self.__dict__["_desc_host"] = torch.empty(n * ADAM_DESC_DTYPE.itemsize, dtype=torch.uint8).pin_memory()
            self.__dict__["_desc_dev"] = torch.empty(n * ADAM_DESC_DTYPE.itemsize, dtype=torch.uint8, device=todo[0][0].device)
        tab = np.frombuffer(host.numpy(), dtype=ADAM_DESC_DTYPE, count=n)
        block = 0
        for i, (p, g, st) in enumerate(todo):
            numel = p.numel()
            step = st["step"]
            tab[i] = (p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), numel, block,
                      np.float32(lr / (1.0 - b1 ** step)), np.float32(math.sqrt(1.0 - b2 ** step)), 0)
            block += (numel + chunk - 1) // chunk
        dev = self.__dict__["_desc_dev"]
        dev[:n * ADAM_DESC_DTYPE.itemsize].copy_(host[:n * ADAM_DESC_DTYPE.itemsize], non_blocking=True)
        ops.call("dgmr_adam_multi", dev.data_ptr(), n, block, float(b1), float(b2), float(eps), ops._stream())
