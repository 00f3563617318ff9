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

        @property
        def logger(self):
            """What `visualize_step` writes to (dgmr/dgmr.py:307: `self.logger.experiment[0].add_image(...)`).  Under Lightning this
            is the Trainer's logger; without it, an in-memory recorder with the same `add_image(tag, img, global_step)` surface
            (`model.logger.experiment[0].images`) unless the caller assigns a logger of their own (e.g. an object whose
            `experiment` is `[torch.utils.tensorboard.SummaryWriter(...)]`)."""
            lg = self.__dict__.get("_logger")
            if lg is None:
                lg = self.__dict__["_logger"] = MemoryImageLogger()
            return lg

        @logger.setter
        def logger(self, value):
            self.__dict__["_logger"] = value

        def manual_backward(self, loss):
            loss.backward()

        def optimizers(self):
            if not hasattr(self, "_optimizers"):
                self._optimizers = self.configure_optimizers()[0]
            return self._optimizers

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict: bool = True, weights_only: bool = True, **overrides):
            """Lightning `.ckpt` files (`{"state_dict": ..., "hyper_parameters": ...}`) without Lightning installed: the model is
            built from the stored hyper-parameters (keyword overrides win) and the state dict loaded into it.
            `weights_only=True` (default) unpickles tensors and plain containers only; a checkpoint that carries other Python
            objects needs `weights_only=False`, which executes whatever the pickle contains - only for files you trust."""
            try:
                ckpt = torch.load(checkpoint_path, map_location=map_location or "cpu", weights_only=weights_only)
            except Exception as e:  # pickle.UnpicklingError from the safe unpickler: Lightning's AttributeDict, callback / loop state
                if weights_only and "weights_only" in str(e).lower() or type(e).__name__ == "UnpicklingError":
                    raise RuntimeError(
                        f"{checkpoint_path}: this checkpoint carries Python objects the safe unpickler refuses (older Lightning "
                        "versions store `hyper_parameters` as an AttributeDict, some store callback state). If you trust the file, "
                        f"load it with load_from_checkpoint(..., weights_only=False). Original error: {e}") from e
                raise
            if "state_dict" not in ckpt:
                raise KeyError(f"{checkpoint_path}: not a Lightning checkpoint (no 'state_dict' entry)")
            import inspect

            accepted = set(inspect.signature(cls.__init__).parameters) - {"self"}
            hp = {k: v for k, v in dict(ckpt.get("hyper_parameters", {})).items() if k in accepted}
            hp.update(overrides)
            model = cls(**hp)
            model.load_state_dict(ckpt["state_dict"], strict=strict)
            return model


class MemoryImageLogger:
    """Stand-in for a Lightning logger whose `experiment[0]` is a TensorBoard SummaryWriter: keeps what `add_image` receives."""

    class _Writer:
        def __init__(self):
            self.images = []  # (tag, [3, H, W] CPU tensor, global_step)

        def add_image(self, tag, img_tensor, global_step=None, **kwargs):
            self.images.append((tag, img_tensor.detach().cpu(), global_step))

    def __init__(self):
        self.experiment = [MemoryImageLogger._Writer()]


def weight_fn(y, precip_weight_cap=24.0):
    """w(y) = max(y + 1, cap) (dgmr/dgmr.py:20-33); GridCellLoss evaluates this one inside its kernel."""
    return torch.max(y + 1, torch.tensor(precip_weight_cap, device=y.device))


weight_fn.fused_in_kernel = True  # losses.GridCellLoss: any OTHER weight function is called on the targets instead


class _CheckpointedDraws(torch.autograd.Function):
    """Activation checkpointing of the generator pass's `generation_steps` forwards (dgmr/dgmr.py:176), batched.

    forward: the draws run under no_grad (nothing is kept but the inputs and the latent draws).  backward: the reference's
    `checkpoint(self.forward, images, use_reentrant=False)` re-runs each forward when autograd first needs one of its saved tensors
    - in REVERSE draw order, because the engine processes the latest-created graph first - and back-propagates through the
    recomputed graph.  The recompute advances every spectral-norm u / v and BatchNorm running statistic a second time and its
    (different) sigmas and batch statistics are the ones the gradients see; torch.utils.checkpoint restores the CPU
    RNG state for the recompute, i.e. the same latents are used and the global RNG stream is left untouched.  Here: ONE batched
    recompute whose call sequence is assigned last-draw-first (`reverse=True`), then one backward through it.  Parameter
    gradients are accumulated by the kernels straight into `param.grad`; `anchor` (any generator parameter) only ties this node
    to the graph, `images` receives no gradient (nothing reads it)."""

    @staticmethod
    def forward(ctx, anchor, images, zs, generator, draws):
        ctx.generator, ctx.draws = generator, draws
        ctx.save_for_backward(images, zs)
        with torch.no_grad():
            return generator.forward_draws(images, draws, zs=zs)

    @staticmethod
    def backward(ctx, grad_out):
        images, zs = ctx.saved_tensors
        with torch.enable_grad():
            out = ctx.generator.forward_draws(images.detach(), ctx.draws, reverse=True, zs=zs)
            torch.autograd.backward(out, grad_out)
        return None, None, None, None, None


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
        # debugging aid which changes no value, so it is left to the caller.  Autograd's anomaly mode cannot see the parameter
        # gradients here (kernels write them): `detect_anomaly` is the stand-in - after every backward pass the losses and every
        # parameter gradient of the network just differentiated are scanned for NaN / Inf on the device (dgmr_nonfinite_count) and
        # the first offender is named.  It synchronises the host once per backward pass: opt-in (or DGMR_DETECT_ANOMALY=1).
        import os

        self.detect_anomaly = os.environ.get("DGMR_DETECT_ANOMALY", "0") == "1"

    def forward(self, x):
        return self.generator(x)

    # ------------------------------------------------------------------------------------------
    def _generate(self, images, draws: int, grad: bool):
        """`draws` generator forwards of the reference on `images` as one batched launch set -> [draws * B, T, C, H, W].

        grad=False (discriminator passes, logging): no graph.  grad=True (generator pass): with strict reference semantics the
        forwards are activation-checkpointed like the reference's (dgmr/dgmr.py:176), see _CheckpointedDraws."""
        if not grad:
            with torch.no_grad():
                return self.generator.forward_draws(images, draws)
        if self.strict_reference_semantics:
            zs = torch.cat([self.latent_stack.draw(images) for _ in range(draws)], dim=0)
            anchor = next(p for p in self.generator.parameters() if p.requires_grad)
            return _CheckpointedDraws.apply(anchor, images, zs, self.generator, draws)
        return self.generator.forward_draws(images, draws)

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
        """`predictions`: [K * B, T, C, H, W], the K generator draws stacked draw-major.  The reference scores every draw with its
        own discriminator call on cat(real, draw) (dgmr/dgmr.py:186-193); the K calls run as one batch here, each keeping its own
        frame draw, spectral-norm sigmas and BatchNorm1d statistics (the real half of every call only feeds those statistics)."""
        b = images.shape[0]
        k = predictions.shape[0] // b
        preds = predictions.view(k, b, *predictions.shape[1:])
        grid_cell_reg = self.grid_regularizer.forward_stacked(preds, future_images)
        real_sequence = torch.cat([images, future_images], dim=1)
        g_seq = torch.cat([images.unsqueeze(0).expand(k, *images.shape), preds], dim=2)  # [K, B, 4+T, C, H, W]
        concatenated_inputs = torch.cat([real_sequence.unsqueeze(0).expand(k, *real_sequence.shape).unsqueeze(1), g_seq.unsqueeze(1)],
                                        dim=1)  # [K, (real, generated), B, ...]
        concatenated_outputs = self.discriminator(concatenated_inputs.reshape(2 * k * b, *real_sequence.shape[1:]), calls=k)
        score_generated = concatenated_outputs.view(k, 2, b, *concatenated_outputs.shape[1:])[:, 1]
        generator_disc_loss = loss_hinge_gen(score_generated.reshape(k * b, *concatenated_outputs.shape[1:]))
        generator_loss = ops.axpby(generator_disc_loss, grid_cell_reg, 1.0, self.grid_lambda)
        return generator_loss, grid_cell_reg

    def training_step(self, batch, batch_idx):
        """One GAN step (dgmr/dgmr.py:137-218)."""
        from .nn import SNScope

        # (the generator's spectral-norm sequences of one step repeat from step to step: each is issued one forward ahead, nn.SNScope)
        with SNScope.step(self.generator):
            return self._training_step(batch, batch_idx)

    def _training_step(self, batch, batch_idx):
        from .nn import SNScope

        images, future_images = batch
        images = images.float()
        future_images = future_images.float()
        self.global_iteration += 1
        g_opt, d_opt = self.optimizers()
        strict = self.strict_reference_semantics
        b = images.shape[0]
        if self.grad_sync is not None:
            self.grad_sync.broadcast_buffers()
        ##########################
        # Optimize Discriminator #
        ##########################
        # The discriminator's optimiser step is issued as late as its results are needed: the generator forward that follows a D
        # backward does not read D's parameters, so it runs beside the tail of D's weight gradients (second stream, ops.py) and the
        # step comes after it.  Same operations, same RNG order, same results as "backward; step; forward".
        d_step_pending = False
        d_loss_pending = []

        def finish_d_step():
            ops.join_side_streams()
            if self.grad_sync is not None:
                self.grad_sync.sync("d")
            if self.detect_anomaly:
                # AFTER the exchange: every rank scans the same, fully reduced gradients (a scan between begin() and sync() would read
                # buckets the communication stream is still writing, and a rank that raised alone would leave the others in sync())
                self._check_finite("the discriminator pass", d_loss_pending, self.discriminator)
            d_opt.step()

        for _ in range(2):
            if strict:
                # reference: predictions = checkpoint(self.forward, images), NOT detached (dgmr.py:150-157).  Its D-loss
                # backward therefore (a) re-runs the generator forward once (checkpoint recompute, same RNG state -> same z),
                # which advances u/v and the BatchNorm running statistics a second time, and (b) back-propagates into generator
                # gradients that g_opt.zero_grad() (dgmr.py:199) discards.  (a) is replayed for its side effects - as the second
                # of two "draws" that share one z, in the same batched launches as the first - and (b) is dead.
                z = self.latent_stack.draw(images)
                with torch.no_grad():
                    predictions = self.generator.forward_draws(images, 2, zs=torch.cat([z, z], dim=0))[:b]
            else:
                predictions = self._generate(images, 1, grad=False)
            if d_step_pending:
                finish_d_step()
            d_opt.zero_grad()
            discriminator_loss = self._disc_losses(images, future_images, predictions)
            if self.grad_sync is not None:
                self.grad_sync.begin("d")  # gradient buckets are all-reduced as the backward pass completes them (ddp.py)
            try:
                with ops.defer_side_join():
                    self.manual_backward(discriminator_loss)
            except BaseException:
                if self.grad_sync is not None:
                    self.grad_sync.abort()  # (no stale touch hook for whatever runs backward next)
                raise
            d_loss_pending[:] = [discriminator_loss]
            d_step_pending = True
        ######################
        # Optimize Generator #
        ######################
        predictions = self._generate(images, self.generation_steps, grad=True)
        finish_d_step()
        # D's parameter gradients from this pass are never read (the next d_opt.zero_grad() clears them): not computed
        d_params = [p for p in self.discriminator.parameters() if p.requires_grad]
        for p in d_params:
            p.requires_grad_(False)
        try:
            generator_loss, grid_cell_reg = self._gen_losses(images, future_images, predictions)
            g_opt.zero_grad()
            if self.grad_sync is not None:
                self.grad_sync.begin("g")
            try:
                with ops.defer_wgrads():  # (no-op unless DGMR_WGRAD_DEFER=1: weight gradients beside the ConvGRU backward chains)
                    self.manual_backward(generator_loss)
            except BaseException:
                if self.grad_sync is not None:
                    self.grad_sync.abort()
                raise
            if self.grad_sync is not None:
                self.grad_sync.sync("g")
            if self.detect_anomaly:
                ops.join_side_streams()
                self._check_finite("the generator pass", [generator_loss, grid_cell_reg], self.generator)
            g_opt.step()
            SNScope.weights_changed(self.generator)
        finally:  # an exception (OOM, a refused launch) must not leave the discriminator frozen for a caller that retries
            for p in d_params:
                p.requires_grad_(True)
        self.log_dict({"train/d_loss": discriminator_loss, "train/g_loss": generator_loss, "train/grid_loss": grid_cell_reg},
                      prog_bar=True)
        if strict or self.visualize:
            # the logging forward (dgmr.py:213): only its side effects on buffers / RNG matter
            generated_images = self._generate(images, 1, grad=False)
            if self.visualize:
                self.visualize_step(images, future_images, generated_images, self.global_iteration, step="train")
        return {"d_loss": discriminator_loss.detach(), "g_loss": generator_loss.detach(), "grid_loss": grid_cell_reg.detach()}

    def validation_step(self, batch, batch_idx):
        """dgmr/dgmr.py:220-290: the same losses without optimisation.  Runs under no_grad (Lightning's validation loop does the
        same); train / eval mode is the caller's, exactly as with the reference module."""
        images, future_images = batch
        images = images.float()
        future_images = future_images.float()
        with torch.no_grad():
            for _ in range(2):
                predictions = self._generate(images, 1, grad=False)
                discriminator_loss = self._disc_losses(images, future_images, predictions)
            predictions = self._generate(images, self.generation_steps, grad=False)
            generator_loss, grid_cell_reg = self._gen_losses(images, future_images, predictions)
            self.log_dict({"val/d_loss": discriminator_loss, "val/g_loss": generator_loss, "val/grid_loss": grid_cell_reg},
                          prog_bar=True)
            generated_images = self._generate(images, 1, grad=False)
        if self.visualize:
            self.visualize_step(images, future_images, generated_images, self.global_iteration, step="val")
        return {"d_loss": discriminator_loss.detach(), "g_loss": generator_loss.detach(), "grid_loss": grid_cell_reg.detach()}

    def sample(self, images, num_samples: int = None):
        """Ensemble nowcast: `num_samples` forecasts per input sequence -> [num_samples, B, T, C, H, W] (the reference's usage is a
        Python loop of `model(x)` calls, README.md:73-91; here the context stack runs once and the sampler on all draws at once).
        Call under model.eval() for inference; in train mode it is `num_samples` consecutive train-mode forwards."""
        k = self.num_samples if num_samples is None else num_samples
        out = self._generate(images.float(), k, grad=False)
        return out.view(k, images.shape[0], *out.shape[1:])

    def attach_data_parallel(self, process_group=None, chunk_mb: int = 64, overlap: bool = True, force_exchange: bool = False):
        """One-process-per-GPU data parallelism: flat gradient buffers, RCCL all-reduce of 64 MB buckets launched while the backward
        pass is still running (`overlap`), buffers broadcast from rank 0 once per step."""
        from .ddp import GradSync

        self.grad_sync = GradSync(self, process_group, chunk_mb, overlap, force_exchange)
        g_opt, d_opt = self.optimizers()
        g_opt.flat_grads = self.grad_sync.gen
        d_opt.flat_grads = self.grad_sync.disc
        self.grad_sync.broadcast_parameters()
        self.grad_sync.broadcast_buffers()
        return self.grad_sync

    def _check_finite(self, what: str, losses, module):
        """detect_anomaly: raise if a loss or a parameter gradient of `module` holds NaN / Inf (the reference's
        torch.autograd.set_detect_anomaly(True), dgmr.py:130, raises from inside backward; here the scan follows it)."""
        dev = losses[0].device
        count = torch.zeros(1, device=dev, dtype=torch.int32)
        flat = self.grad_sync.flat_for("g" if module is self.generator else "d").flat if self.grad_sync is not None else None
        tensors = [l.detach().reshape(-1).float().contiguous() for l in losses]
        tensors += [flat] if flat is not None else [p.grad for p in module.parameters() if p.grad is not None]
        for t in tensors:
            ops.call("dgmr_nonfinite_count", t.data_ptr(), t.numel(), count.data_ptr(), ops._stream())
        if int(count.item()) == 0:
            return
        for name, p in module.named_parameters():  # slow path: name the first offender
            if p.grad is not None and not bool(torch.isfinite(p.grad).all()):
                raise RuntimeError(f"detect_anomaly: non-finite gradient in {what}: parameter '{name}' "
                                   f"({int((~torch.isfinite(p.grad)).sum())} of {p.grad.numel()} elements)")
        raise RuntimeError(f"detect_anomaly: non-finite loss in {what}: {[float(l) for l in losses]}")

    def configure_optimizers(self):
        """Two Adam optimisers, lr 5e-5 / 2e-4, betas (0.0, 0.999) by default (dgmr/dgmr.py:292-300)."""
        b1, b2 = self.beta1, self.beta2
        opt_g = FusedAdam(self.generator.parameters(), lr=self.gen_lr, betas=(b1, b2))
        opt_d = FusedAdam(self.discriminator.parameters(), lr=self.disc_lr, betas=(b1, b2))
        return [opt_g, opt_d], []

    def visualize_step(self, x, y, y_hat, batch_idx, step):
        """TensorBoard image grids (dgmr/dgmr.py:302-327): for every INPUT frame index i the input, target and generated frame i of
        the first sample, each as a grid of its channels.  ONE device->host copy of the three stacks (the reference does the same
        three `.cpu()` calls); torchvision is optional - `make_grid` below reproduces `torchvision.utils.make_grid`."""
        try:
            from torchvision.utils import make_grid as grid_fn
        except Exception:
            grid_fn = make_grid
        tensorboard = self.logger.experiment[0]
        images, future_images, generated_images = x[0].detach().cpu(), y[0].detach().cpu(), y_hat[0].detach().cpu()
        for i, t in enumerate(images):  # the reference indexes the target / generated stacks with the INPUT frame index
            for name, frame in (("Input_Image_Stack", t), ("Target_Image", future_images[i]), ("Generated_Image", generated_images[i])):
                grid = grid_fn([torch.unsqueeze(img, dim=0) for img in frame], nrow=self.input_channels)
                tensorboard.add_image(f"{step}/{name}_Frame_{i}", grid, global_step=batch_idx)


def make_grid(tensor, nrow: int = 8, padding: int = 2, pad_value: float = 0.0):
    """torchvision.utils.make_grid for the arguments visualize_step uses (no normalisation): a list of [1, H, W] / [3, H, W] images ->
    one [3, ...] image, `nrow` images per row, `padding` pixels of `pad_value` between them; a single image is returned as is
    (grey images replicated to three channels)."""
    import math

    if isinstance(tensor, (list, tuple)):
        tensor = torch.stack(list(tensor), dim=0)
    if tensor.dim() == 2:
        tensor = tensor.unsqueeze(0)
    if tensor.dim() == 3:
        if tensor.size(0) == 1:
            tensor = torch.cat((tensor, tensor, tensor), 0)
        tensor = tensor.unsqueeze(0)
    if tensor.dim() == 4 and tensor.size(1) == 1:
        tensor = torch.cat((tensor, tensor, tensor), 1)
    if tensor.size(0) == 1:
        return tensor.squeeze(0)
    nmaps = tensor.size(0)
    xmaps = min(nrow, nmaps)
    ymaps = int(math.ceil(float(nmaps) / xmaps))
    height, width = int(tensor.size(2) + padding), int(tensor.size(3) + padding)
    grid = tensor.new_full((tensor.size(1), height * ymaps + padding, width * xmaps + padding), pad_value)
    k = 0
    for yy in range(ymaps):
        for xx in range(xmaps):
            if k >= nmaps:
                break
            grid.narrow(1, yy * height + padding, height - padding).narrow(2, xx * width + padding, width - padding).copy_(tensor[k])
            k += 1
    return grid
