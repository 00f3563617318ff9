"""Hot-path losses (mirror of dgmr/losses.py:158-206,307-319) on the HIP operators.

The reference's unused loss zoo (SSIM/MS-SSIM/TV/GDL/Focal, losses.py:10-155,209-304,322-378) is not on
the training-step path and is out of scope here.
"""
import torch
import torch.nn as nn

from . import ops


def loss_hinge_disc(score_generated, score_real):
    """Discriminator hinge loss: mean(relu(1 - real)) + mean(relu(1 + generated)) (losses.py:307-313)."""
    return ops.HingeDiscFn.apply(score_generated, score_real)


def loss_hinge_gen(score_generated):
    """Generator hinge loss: -mean(score_generated) (losses.py:316-319)."""
    return ops.MeanFn.apply(score_generated, -1.0)


class GridCellLoss(nn.Module):
    """Grid cell regulariser (losses.py:158-192), including the reference's operator precedence:
    ``||(g - y) * w||_1 / T * H * W`` with ``w = max(y + 1, cap)`` (dgmr/dgmr.py:33)."""

    def __init__(self, weight_fn=None, precip_weight_cap=24.0):
        """`weight_fn(targets, precip_weight_cap)`: dgmr.dgmr.weight_fn (the default of DGMR) is evaluated inside the loss kernel;
        any other callable is called on the targets and its result handed to the kernel as explicit weights.  None constructs, as
        in the reference (losses.py:161-171), and fails at the first call like the reference does: its lambda wrapper is never
        None, so `difference * None` raises TypeError there (losses.py:187-190)."""
        super().__init__()
        self.precip_weight_cap = precip_weight_cap
        self.weight_fn = weight_fn

    def _weights(self, targets):
        if self.weight_fn is None:
            raise TypeError("GridCellLoss was built without a weight_fn: unsupported operand type(s) for *: 'Tensor' and 'NoneType' "
                            "(the reference fails the same way, dgmr/losses.py:171,187-190)")
        if getattr(self.weight_fn, "fused_in_kernel", False):
            return None
        with torch.no_grad():
            return self.weight_fn(targets, self.precip_weight_cap)

    def forward(self, generated_images, targets):
        """`generated_images`: the mean prediction [B,T,C,H,W]."""
        return ops.GridCellFn.apply(generated_images.unsqueeze(0), targets, self.precip_weight_cap, self._weights(targets))

    def forward_stacked(self, stacked_predictions, targets):
        """Mean over the K stacked draws and the loss in one pass: `stacked_predictions` is [K,B,T,C,H,W]."""
        return ops.GridCellFn.apply(stacked_predictions, targets, self.precip_weight_cap, self._weights(targets))


class NowcastingLoss(nn.Module):
    """losses.py:195-206 (constructed by DGMR, never called on the hot path)."""

    def forward(self, x, real_flag):
        if real_flag is True:
            return ops.HingeDiscFn.apply(torch.full_like(x, -1.0), x) - 0.0  # mean(relu(1 - x)) + 0
        return ops.HingeDiscFn.apply(x, torch.full_like(x, 1.0))
