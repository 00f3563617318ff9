"""Stub modules that let the *unmodified* reference import in this container.

The reference (``/root/reference/dgmr``) imports pytorch_lightning, torchvision and pytorch_msssim,
none of which is installed here.  ``install()`` registers minimal stand-ins in
``sys.modules`` so that ``import dgmr`` works; it is used ONLY by ``oracle/gen_golden.py`` and by the
oracle-vs-reference cross-check test, both of which run in the build container where
``/root/reference`` exists.  Test infrastructure, never shipped.
"""
import sys
import types

import torch


def install(reference_root="/root/reference"):
    if "pytorch_lightning" not in sys.modules:
        pl = types.ModuleType("pytorch_lightning")

        class LightningModule(torch.nn.Module):
            def save_hyperparameters(self, *a, **k):
                pass

            def log_dict(self, *a, **k):
                pass

            def manual_backward(self, loss):
                loss.backward()

            def optimizers(self):
                if not hasattr(self, "_opts"):
                    self._opts = self.configure_optimizers()[0]
                return self._opts

        class Trainer:  # pragma: no cover
            def __init__(self, *a, **k):
                pass

        pl.LightningModule = LightningModule
        pl.Trainer = Trainer
        sys.modules["pytorch_lightning"] = pl
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tv.utils = types.ModuleType("torchvision.utils")
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.utils"] = tv.utils
    if "pytorch_msssim" not in sys.modules:
        ms = types.ModuleType("pytorch_msssim")

        class _D(torch.nn.Module):
            def __init__(self, *a, **k):
                super().__init__()

        ms.SSIM = _D
        ms.MS_SSIM = _D
        sys.modules["pytorch_msssim"] = ms
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
