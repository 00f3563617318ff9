"""DGMR LightningModule (mirror of dgmr/dgmr.py) driving the HIP operators.

``training_step`` keeps the reference's manual-optimisation order (2 discriminator passes, then the
generator pass over ``generation_steps`` draws).  With ``strict_reference_semantics=True`` (default) every
OBSERVABLE effect of the reference's step is reproduced — losses, both optimiser updates, and the state that
forwards mutate (spectral-norm u/v, BatchNorm running statistics, the CPU RNG stream), including the second
advance caused by activation checkpointing's recompute and the extra logging forward.
What is NOT executed is gradient work whose result nothing reads: the D-pass back-propagation into the
generator (zeroed by ``g_opt.zero_grad()``) and the G-pass weight gradients of the discriminator (zeroed by
the next ``d_opt.zero_grad()``).  ``strict_reference_semantics=False`` additionally drops the state-only
forwards (checkpoint recompute, logging forward), so buffers advance fewer times than the reference's.
"""
import torch
from huggingface_hub import PyTorchModelHubMixin
from torch.utils.checkpoint import checkpoint

from . import ops
from .common import ContextConditioningStack, LatentConditioningStack
from .discriminators import Discriminator
from .generators import Generator, Sampler
from .losses import GridCellLoss, NowcastingLoss, loss_hinge_disc, loss_hinge_gen
from .optim import FusedAdam

try:  # the reference subclasses pl.LightningModule; Lightning is optional here
    import pytorch_lightning as pl

    _Base = pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # pragma: no cover - depends on the environment
    HAVE_LIGHTNING = False

    class _Base(torch.nn.Module):
        """Minimal stand-in for pl.LightningModule's manual-optimisation surface."""

        automatic_optimization = True

        def save_hyperparameters(self, *args, **kwargs):
            self.hparams = getattr(self, "_hub_mixin_config", {})

        def log_dict(self, metrics, *args, **kwargs):
            self.logged_metrics = metrics  # tensors are kept on device: no host sync in the step

        def manual_backward(self, loss):
            loss.backward()

        def optimizers(self):
            if not hasattr(self, "_optimizers"):
                self._optimizers = self.configure_optimizers()[0]
            return self._optimizers


def weight_fn(y, precip_weight_cap=24.0):
    """w(y) = max(y + 1, cap) (dgmr/dgmr.py:20-33); applied inside the grid-cell loss kernel."""
    return torch.max(y + 1, torch.tensor(precip_weight_cap, device=y.device))


class DGMR(
    _Base,
    PyTorchModelHubMixin,
    library_name="DGMR",
    tags=["nowcasting", "forecasting", "timeseries", "remote-sensing", "gan"],
    repo_url="https://github.com/openclimatefix/skillful_nowcasting",
):
    """Deep Generative Model of Radar (dgmr/dgmr.py:36-327)."""

    def __init__(self, forecast_steps: int = 18, input_channels: int = 1, output_shape: int = 256, gen_lr: float = 5e-5,
                 disc_lr: float = 2e-4, visualize: bool = False, conv_type: str = "standard", num_samples: int = 6,
                 grid_lambda: float = 20.0, beta1: float = 0.0, beta2: float = 0.999, latent_channels: int = 768,
                 context_channels: int = 384, generation_steps: int = 6, precip_weight_cap: float = 24.0,
                 strict_reference_semantics: bool = True):
        super().__init__()
        self.gen_lr = gen_lr
        self.disc_lr = disc_lr
        self.beta1 = beta1
        self.beta2 = beta2
        self.discriminator_loss = NowcastingLoss()
        self.grid_regularizer = GridCellLoss(weight_fn=weight_fn, precip_weight_cap=precip_weight_cap)
        self.grid_lambda = grid_lambda
        self.num_samples = num_samples
        self.visualize = visualize
        self.latent_channels = latent_channels
        self.context_channels = context_channels
        self.input_channels = input_channels
        self.generation_steps = generation_steps
        self.strict_reference_semantics = strict_reference_semantics
        self.conditioning_stack = ContextConditioningStack(input_channels=input_channels, conv_type=conv_type,
                                                           output_channels=self.context_channels)
        self.latent_stack = LatentConditioningStack(shape=(8 * self.input_channels, output_shape // 32, output_shape // 32),
                                                    output_channels=self.latent_channels)
        self.sampler = Sampler(forecast_steps=forecast_steps, latent_channels=self.latent_channels,
                               context_channels=self.context_channels)
        self.generator = Generator(self.conditioning_stack, self.latent_stack, self.sampler)
        self.discriminator = Discriminator(input_channels)
        self.save_hyperparameters()
        self.global_iteration = 0
        self.grad_sync = None  # ddp.GradSync when running data-parallel (attach_data_parallel)
        # Important: This property activates manual optimization.
        self.automatic_optimization = False
        # NB the reference also flips torch.autograd.set_detect_anomaly(True) globally here (dgmr.py:130); that is a
        # debugging aid which changes no value, so it is left to the caller.

    def forward(self, x):
        return self.generator(x)

    # ------------------------------------------------------------------------------------------
    def _generate(self, images):
        if self.strict_reference_semantics:
            return checkpoint(self.forward, images, use_reentrant=False)
        return self.forward(images)

    def _disc_losses(self, images, future_images, predictions):
        generated_sequence = torch.cat([images, predictions], dim=1)
        real_sequence = torch.cat([images, future_images], dim=1)
        concatenated_inputs = torch.cat([real_sequence, generated_sequence], dim=0)
        concatenated_outputs = self.discriminator(concatenated_inputs)
        score_real, score_generated = torch.split(concatenated_outputs, [real_sequence.shape[0], generated_sequence.shape[0]], dim=0)
        score_real_spatial, score_real_temporal = torch.split(score_real, 1, dim=1)
        score_generated_spatial, score_generated_temporal = torch.split(score_generated, 1, dim=1)
        return ops.axpby(loss_hinge_disc(score_generated_spatial, score_real_spatial),
                         loss_hinge_disc(score_generated_temporal, score_real_temporal))

    def _gen_losses(self, images, future_images, predictions):
        grid_cell_reg = self.grid_regularizer.forward_stacked(torch.stack(predictions, dim=0), future_images)
        real_sequence = torch.cat([images, future_images], dim=1)
        generated_scores = []
        for x in predictions:
            g_seq = torch.cat([images, x], dim=1)
            concatenated_inputs = torch.cat([real_sequence, g_seq], dim=0)
            concatenated_outputs = self.discriminator(concatenated_inputs)
            score_real, score_generated = torch.split(concatenated_outputs, [real_sequence.shape[0], g_seq.shape[0]], dim=0)
            generated_scores.append(score_generated)
        generator_disc_loss = loss_hinge_gen(torch.cat(generated_scores, dim=0))
        generator_loss = ops.axpby(generator_disc_loss, grid_cell_reg, 1.0, self.grid_lambda)
        return generator_loss, grid_cell_reg

    def training_step(self, batch, batch_idx):
        """One GAN step (dgmr/dgmr.py:137-218)."""
        images, future_images = batch
        images = images.float()
        future_images = future_images.float()
        self.global_iteration += 1
        g_opt, d_opt = self.optimizers()
        strict = self.strict_reference_semantics
        if self.grad_sync is not None:
            self.grad_sync.broadcast_buffers()
        ##########################
        # Optimize Discriminator #
        ##########################
        for _ in range(2):
            d_opt.zero_grad()
            if strict:
                # reference: predictions = checkpoint(self.forward, images), NOT detached (dgmr.py:150-157).  Its D-loss
                # backward therefore (a) re-runs the generator forward once (checkpoint recompute, same RNG state -> same z),
                # which advances u/v and the BatchNorm running statistics a second time, and (b) back-propagates into generator
                # gradients that g_opt.zero_grad() (dgmr.py:199) discards.  (a) is replayed for its side effects, (b) is dead.
                rng0 = torch.get_rng_state()
                with torch.no_grad():
                    predictions = self.forward(images)
                    rng1 = torch.get_rng_state()
                    torch.set_rng_state(rng0)
                    self.forward(images)
                    torch.set_rng_state(rng1)
            else:
                with torch.no_grad():
                    predictions = self.forward(images)
            discriminator_loss = self._disc_losses(images, future_images, predictions)
            self.manual_backward(discriminator_loss)
            if self.grad_sync is not None:
                self.grad_sync.sync("d")
            d_opt.step()
        ######################
        # Optimize Generator #
        ######################
        predictions = [self._generate(images) for _ in range(self.generation_steps)]
        # D's parameter gradients from this pass are never read (the next d_opt.zero_grad() clears them): not computed
        for p in self.discriminator.parameters():
            p.requires_grad_(False)
        generator_loss, grid_cell_reg = self._gen_losses(images, future_images, predictions)
        g_opt.zero_grad()
        self.manual_backward(generator_loss)
        if self.grad_sync is not None:
            self.grad_sync.sync("g")
        g_opt.step()
        for p in self.discriminator.parameters():
            p.requires_grad_(True)
        self.log_dict({"train/d_loss": discriminator_loss, "train/g_loss": generator_loss, "train/grid_loss": grid_cell_reg},
                      prog_bar=True)
        if strict or self.visualize:
            with torch.no_grad():  # the logging forward (dgmr.py:213): only its side effects on buffers / RNG matter
                generated_images = self(images)
            if self.visualize:
                self.visualize_step(images, future_images, generated_images, self.global_iteration, step="train")
        return {"d_loss": discriminator_loss.detach(), "g_loss": generator_loss.detach(), "grid_loss": grid_cell_reg.detach()}

    def validation_step(self, batch, batch_idx):
        """dgmr/dgmr.py:220-290: the same losses without optimisation."""
        images, future_images = batch
        images = images.float()
        future_images = future_images.float()
        for _ in range(2):
            predictions = self(images)
            discriminator_loss = self._disc_losses(images, future_images, predictions)
        predictions = [self(images) for _ in range(self.generation_steps)]
        generator_loss, grid_cell_reg = self._gen_losses(images, future_images, predictions)
        self.log_dict({"val/d_loss": discriminator_loss, "val/g_loss": generator_loss, "val/grid_loss": grid_cell_reg}, prog_bar=True)
        generated_images = self(images)
        if self.visualize:
            self.visualize_step(images, future_images, generated_images, self.global_iteration, step="val")

    def attach_data_parallel(self, process_group=None, chunk_mb: int = 64):
        """One-process-per-GPU data parallelism: flat gradient buffers + RCCL all-reduce after each backward."""
        from .ddp import GradSync

        self.grad_sync = GradSync(self, process_group, chunk_mb)
        g_opt, d_opt = self.optimizers()
        g_opt.flat_grads = self.grad_sync.gen
        d_opt.flat_grads = self.grad_sync.disc
        self.grad_sync.broadcast_parameters()
        self.grad_sync.broadcast_buffers()
        return self.grad_sync

    def configure_optimizers(self):
        """Two Adam optimisers, lr 5e-5 / 2e-4, betas (0.0, 0.999) by default (dgmr/dgmr.py:292-300)."""
        b1, b2 = self.beta1, self.beta2
        opt_g = FusedAdam(self.generator.parameters(), lr=self.gen_lr, betas=(b1, b2))
        opt_d = FusedAdam(self.discriminator.parameters(), lr=self.disc_lr, betas=(b1, b2))
        return [opt_g, opt_d], []

    def visualize_step(self, x, y, y_hat, batch_idx, step):  # pragma: no cover - needs torchvision + a logger
        """TensorBoard image grids (dgmr/dgmr.py:302-327); off the hot path, requires torchvision."""
        import torchvision

        tensorboard = self.logger.experiment[0]
        for name, seq in (("Input_Image_Stack", x[0]), ("Target_Image", y[0]), ("Generated_Image", y_hat[0])):
            for i, t in enumerate(seq.cpu().detach()):
                grid = torchvision.utils.make_grid([torch.unsqueeze(img, dim=0) for img in t], nrow=self.input_channels)
                tensorboard.add_image(f"{step}/{name}_Frame_{i}", grid, global_step=batch_idx)
