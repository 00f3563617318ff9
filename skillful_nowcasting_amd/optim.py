"""Adam through the fused HIP kernel (torch.optim.Adam semantics as used at dgmr/dgmr.py:292-300)."""
import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    """Adam(lr, betas, eps=1e-8, weight_decay=0, amsgrad=False); one HIP launch per parameter tensor.

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

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
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
                ops.adam_update(p, g, st["exp_avg"], st["exp_avg_sq"], st["step"], group["lr"], b1, b2, group["eps"])
        ops.bump_weights_epoch()
        return loss
