"""Generate tests/golden/*.safetensors by running the UNMODIFIED reference on CPU (build container only).

Usage:  python oracle/gen_golden.py            (needs /root/reference; writes tests/golden/)

The reference ships no numeric golden vectors for the hot path, so these fixtures —
outputs of the reference's own modules on seeded inputs — are what pins ``oracle/dgmr_oracle.py``.
Each file holds: ``sd0.<key>`` the module state_dict BEFORE the call, ``buf1.<key>`` the buffers
AFTER the call (u/v, BN running stats), ``in.*`` inputs, ``out.*`` outputs, ``grad.*`` gradients of
``sum(out * cot)`` w.r.t. inputs / selected parameters, ``cot`` the cotangent.
Test infrastructure: never imported by the product.
"""
import json
import os
import sys

import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import _stubs  # noqa: E402

_stubs.install()
torch.autograd.set_detect_anomaly(False)

from dgmr.common import (  # noqa: E402
    ContextConditioningStack,
    DBlock,
    GBlock,
    LatentConditioningStack,
    LBlock,
    UpsampleGBlock,
)
from dgmr.discriminators import SpatialDiscriminator, TemporalDiscriminator  # noqa: E402
from dgmr.generators import Sampler  # noqa: E402
from dgmr.layers import AttentionLayer, ConvGRU  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def _is_buffer_key(k):
    return k.endswith(("._u", "._v", "running_mean", "running_var", "num_batches_tracked"))


def record(name, module, inputs, call, train=True, grad_params=True, meta=None):
    """Run `call(module, *inputs)` once; store state before/after, outputs and gradients."""
    module.train(train)
    rec = {}
    for k, v in module.state_dict().items():
        rec["sd0." + k] = v.detach().clone()
    ins = []
    for i, x in enumerate(inputs):
        if isinstance(x, torch.Tensor) and x.is_floating_point():
            x = x.clone().requires_grad_(True)
        ins.append(x)
    out = call(module, *ins)
    outs = out if isinstance(out, (tuple, list)) else [out]
    torch.manual_seed(1234)
    cots = [torch.randn_like(o) for o in outs]
    loss = sum((o * c).sum() for o, c in zip(outs, cots))
    loss.backward()
    for i, x in enumerate(ins):
        if isinstance(x, torch.Tensor):
            rec[f"in.{i}"] = x.detach().clone()
            if x.is_floating_point() and x.grad is not None:
                rec[f"grad.in.{i}"] = x.grad.clone()
    for i, (o, c) in enumerate(zip(outs, cots)):
        rec[f"out.{i}"] = o.detach().clone()
        rec[f"cot.{i}"] = c
    if grad_params:
        for k, p in module.named_parameters():
            if p.grad is not None:
                rec["grad.p." + k] = p.grad.clone()
    for k, v in module.state_dict().items():
        if _is_buffer_key(k):
            rec["buf1." + k] = v.detach().clone()
    rec = {k: v.contiguous() for k, v in rec.items()}
    md = {"name": name, "train": str(train)}
    if meta:
        md.update({k: json.dumps(v) for k, v in meta.items()})
    save_file(rec, os.path.join(OUT, name + ".safetensors"), metadata=md)
    nbytes = sum(v.numel() * v.element_size() for v in rec.values())
    print(f"{name:28s} {len(rec):4d} tensors {nbytes / 1e6:7.2f} MB")


def main():
    # ---- DBlock 2-D / 3-D (dgmr/common.py:158-238) ----
    torch.manual_seed(1)
    record("dblock_4_12", DBlock(4, 12), [torch.rand(2, 4, 16, 16) - 0.3], lambda m, x: m(x))
    torch.manual_seed(2)
    record("dblock_12_12_keep", DBlock(12, 12, keep_same_output=True), [torch.rand(2, 12, 8, 8) - 0.3], lambda m, x: m(x))
    torch.manual_seed(3)
    record("dblock_4_8_norelu", DBlock(4, 8, first_relu=False), [torch.rand(2, 4, 8, 8) - 0.3], lambda m, x: m(x))
    torch.manual_seed(4)
    record("dblock3d_4_8_norelu", DBlock(4, 8, conv_type="3d", first_relu=False), [torch.rand(2, 4, 4, 8, 8) - 0.3], lambda m, x: m(x))
    torch.manual_seed(5)
    record("dblock3d_8_16", DBlock(8, 16, conv_type="3d"), [torch.rand(2, 8, 6, 8, 8) - 0.3], lambda m, x: m(x))
    torch.manual_seed(6)
    record("dblock_4_12_eval", DBlock(4, 12), [torch.rand(2, 4, 16, 16) - 0.3], lambda m, x: m(x), train=False)
    # ---- GBlock / UpsampleGBlock (dgmr/common.py:17-155) ----
    torch.manual_seed(7)
    record("gblock_8_8", GBlock(8, 8), [torch.randn(2, 8, 8, 8)], lambda m, x: m(x))
    torch.manual_seed(8)
    record("gblock_8_4", GBlock(8, 4), [torch.randn(2, 8, 8, 8)], lambda m, x: m(x))
    torch.manual_seed(9)
    record("upgblock_8_4", UpsampleGBlock(8, 4), [torch.randn(2, 8, 8, 8)], lambda m, x: m(x))
    torch.manual_seed(10)
    record("gblock_8_8_eval", GBlock(8, 8), [torch.randn(2, 8, 8, 8)], lambda m, x: m(x), train=False)
    # ---- LBlock / Attention (dgmr/common.py:241-300, dgmr/layers/Attention.py) ----
    torch.manual_seed(11)
    record("lblock_8_12", LBlock(8, 12), [torch.randn(1, 8, 4, 4)], lambda m, x: m(x))
    torch.manual_seed(12)
    att = AttentionLayer(32, 32)
    with torch.no_grad():
        att.gamma.fill_(0.7)
    record("attention_32", att, [torch.randn(2, 32, 4, 6)], lambda m, x: m(x))
    # ---- ConvGRU (dgmr/layers/ConvGRU.py) ----
    torch.manual_seed(13)
    record("convgru_8_4_T3", ConvGRU(8 + 4, 4, 3),
           [torch.randn(3, 2, 8, 8, 8), torch.randn(2, 4, 8, 8)], lambda m, xs, h: m(list(xs), h))
    # ---- Context / Latent stacks (dgmr/common.py:303-497) ----
    torch.manual_seed(14)
    record("context_128", ContextConditioningStack(1, 128), [torch.rand(2, 4, 1, 32, 32)], lambda m, x: m(x),
           grad_params=False)
    torch.manual_seed(15)
    lat = LatentConditioningStack((8, 2, 2), 256)
    with torch.no_grad():
        lat.att_block.gamma.fill_(0.5)
    torch.manual_seed(150)
    z = torch.distributions.normal.Normal(torch.Tensor([0.0]), torch.Tensor([1.0])).sample((8, 2, 2))
    z = torch.permute(z, (3, 0, 1, 2)).contiguous()

    def lat_call(m, zz):
        torch.manual_seed(150)  # the reference draws z itself (common.py:481-483); same seed -> same z
        return m(torch.zeros(1))

    record("latent_256", lat, [z], lat_call, grad_params=False)
    # ---- Sampler (dgmr/generators.py:20-182) ----
    torch.manual_seed(16)
    smp = Sampler(forecast_steps=2, latent_channels=64, context_channels=32)
    cond = [torch.randn(2, 4, 16, 16), torch.randn(2, 8, 8, 8), torch.randn(2, 16, 4, 4), torch.randn(2, 32, 2, 2)]
    record("sampler_64_32_T2", smp, cond + [torch.randn(1, 64, 2, 2)],
           lambda m, c0, c1, c2, c3, l: m([c0, c1, c2, c3], l), grad_params=False)
    # ---- Discriminators, reduced depth (dgmr/discriminators.py) ----
    torch.manual_seed(17)
    sd_ = SpatialDiscriminator(input_channels=1, num_timesteps=3, num_layers=1)
    xs = torch.rand(2, 6, 1, 16, 16)
    torch.manual_seed(170)
    idxs = torch.randint(low=0, high=6, size=(3,))

    def sdisc_call(m, x, _idxs):
        torch.manual_seed(170)
        return m(x)

    record("spatial_disc_L1", sd_, [xs, idxs], sdisc_call, grad_params=False)
    torch.manual_seed(18)
    td = TemporalDiscriminator(input_channels=1, num_layers=1)
    record("temporal_disc_L1", td, [torch.rand(2, 8, 1, 32, 32)], lambda m, x: m(x), grad_params=False)
    training_step_golden()
    training_steps_adv_golden()
    validation_step_golden()


TS_KW = dict(forecast_steps=2, input_channels=1, output_shape=128, latent_channels=384, context_channels=192, generation_steps=2)


GRAD_KEYS = ["generator.sampler.conv_1x1.bias", "generator.sampler.bn.weight",
             "generator.sampler.conv_1x1.parametrizations.weight.original",
             "generator.sampler.up_g4.first_conv_3x3.parametrizations.weight.original",
             "generator.sampler.convGRU4.cell.output_conv.parametrizations.weight.original",
             "generator.sampler.gru_conv_1x1_3.parametrizations.weight.original",
             "generator.latent_stack.conv_3x3.parametrizations.weight.original",
             "generator.latent_stack.l_block1.first_conv_3x3.weight",
             "generator.latent_stack.att_block.gamma",
             "generator.conditioning_stack.d1.first_conv_3x3.parametrizations.weight.original",
             "generator.conditioning_stack.conv1.parametrizations.weight.original",
             "discriminator.spatial_discriminator.fc.parametrizations.weight.original",
             "discriminator.spatial_discriminator.bn.weight",
             "discriminator.spatial_discriminator.d1.first_conv_3x3.parametrizations.weight.original",
             "discriminator.spatial_discriminator.intermediate_dblocks.0.conv_1x1.parametrizations.weight.original",
             "discriminator.temporal_discriminator.d1.first_conv_3x3.parametrizations.weight.original",
             "discriminator.temporal_discriminator.d2.last_conv_3x3.bias",
             "discriminator.temporal_discriminator.fc.bias"]


def checksums(sd):
    """Per-tensor (sum, abs-sum, first, last) in float64: compact fingerprint of a ~100 M-parameter state."""
    keys = sorted(sd.keys())
    vals = torch.zeros(len(keys), 4, dtype=torch.float64)
    for i, k in enumerate(keys):
        t = sd[k].detach().double().flatten()
        vals[i, 0], vals[i, 1], vals[i, 2], vals[i, 3] = t.sum(), t.abs().sum(), t[0], t[-1]
    return keys, vals


def training_step_golden():
    """One full `DGMR.training_step` of the unmodified reference on a seeded model (dgmr/dgmr.py:137-218)."""
    from dgmr import DGMR

    torch.manual_seed(42)
    model = DGMR(**TS_KW)
    keys0, cs0 = checksums(model.state_dict())
    torch.manual_seed(43)
    images, future = torch.rand(2, 4, 1, 128, 128), torch.rand(2, 2, 1, 128, 128)
    logged = {}
    model.log_dict = lambda d, **k: logged.update({kk: float(v) for kk, v in d.items()})
    bw = []
    model.manual_backward = lambda loss: (bw.append(float(loss.detach())), loss.backward())
    # snapshot the gradients the optimisers see: D at its first step (the second D pass has zero hinge loss), G at its step
    grads = {}
    g_opt, d_opt = model.optimizers()
    named = dict(model.named_parameters())

    def wrap(opt, prefix, once):
        orig = opt.step

        def step(*a, **k):
            if not (once and any(kk.startswith("grad." + prefix) for kk in grads)):
                for kk in GRAD_KEYS:
                    if kk.startswith(prefix):
                        p = named[kk[len("generator."):]] if kk.startswith("generator.") else named[kk]
                        grads["grad." + kk] = p.grad.detach().clone().contiguous()
            return orig(*a, **k)

        opt.step = step

    wrap(g_opt, "generator.", False)
    wrap(d_opt, "discriminator.", True)
    torch.manual_seed(44)
    model.training_step((images, future), 0)
    sd1 = model.state_dict()
    keys1, cs1 = checksums(sd1)
    assert keys0 == keys1
    rec = {"images": images, "future": future, "cs0": cs0, "cs1": cs1,
           "losses": torch.tensor([logged["train/d_loss"], logged["train/g_loss"], logged["train/grid_loss"]], dtype=torch.float64),
           "backward_losses": torch.tensor(bw, dtype=torch.float64)}  # d pass 1, d pass 2, g
    # a few whole tensors after the step (small ones + slices of big ones) for element-wise checks
    rec.update(grads)
    for k in ["generator.sampler.conv_1x1.bias", "generator.sampler.bn.running_mean", "generator.sampler.bn.weight",
              "generator.latent_stack.conv_3x3.parametrizations.weight.original",
              "generator.conditioning_stack.d1.first_conv_3x3.parametrizations.weight.original",
              "generator.sampler.convGRU4.cell.read_gate_conv.parametrizations.weight.0._u",
              "discriminator.spatial_discriminator.fc.parametrizations.weight.original",
              "discriminator.temporal_discriminator.d1.first_conv_3x3.parametrizations.weight.original",
              "discriminator.spatial_discriminator.bn.running_var"]:
        rec["post." + k] = sd1[k].detach().clone().contiguous()
    save_file(rec, os.path.join(OUT, "training_step.safetensors"),
              metadata={"keys": json.dumps(keys0), "kw": json.dumps(TS_KW), "seeds": "[42, 43, 44]"})
    print("training_step golden: losses", rec["losses"].tolist(), bw)


ADV_HP = dict(grid_lambda=0.0, beta1=0.5, disc_lr=2e-6, gen_lr=5e-6)
ADV_STEPS = 3
ADV_BATCH = 2


def _adv_run():
    """One run of the 3-step adversarial schedule on the reference -> dict of tensors (see training_steps_adv_golden)."""
    from dgmr import DGMR

    torch.manual_seed(42)
    model = DGMR(**TS_KW, **ADV_HP)
    keys0, cs0 = checksums(model.state_dict())
    torch.manual_seed(43)
    images, future = torch.rand(ADV_BATCH, 4, 1, 128, 128), torch.rand(ADV_BATCH, 2, 1, 128, 128)
    logged = []
    model.log_dict = lambda d, **k: logged.append([float(d["train/d_loss"].detach()), float(d["train/g_loss"].detach()),
                                                   float(d["train/grid_loss"].detach())])
    bw = []
    model.manual_backward = lambda loss: (bw.append(float(loss.detach())), loss.backward())
    g_opt, d_opt = model.optimizers()
    named = dict(model.named_parameters())
    grads = {}
    count = {"g": 0, "d": 0}

    def wrap(opt, prefix, tag, want_call):
        orig = opt.step

        def step(*a, **k):
            count[tag] += 1
            if count[tag] == want_call:
                for kk in GRAD_KEYS:
                    if kk.startswith(prefix):
                        p = named[kk[len("generator."):]] if kk.startswith("generator.") else named[kk]
                        grads["grad." + kk] = p.grad.detach().clone().contiguous()
            return orig(*a, **k)

        opt.step = step

    wrap(g_opt, "generator.", "g", ADV_STEPS)            # the generator's gradient at its LAST step
    wrap(d_opt, "discriminator.", "d", 2 * ADV_STEPS)    # the discriminator's at its last (6th) step
    torch.manual_seed(44)
    for i in range(ADV_STEPS):
        model.training_step((images, future), i)
    sd1 = model.state_dict()
    keys1, cs1 = checksums(sd1)
    assert keys0 == keys1
    rec = {"images": images, "future": future, "cs0": cs0, "cs1": cs1,
           "losses": torch.tensor(logged, dtype=torch.float64),           # [step][d, g, grid]
           "backward_losses": torch.tensor(bw, dtype=torch.float64)}      # per step: d pass 1, d pass 2, g
    rec.update(grads)
    for k in ["generator.sampler.conv_1x1.bias", "generator.sampler.bn.running_mean", "generator.sampler.bn.weight",
              "generator.latent_stack.conv_3x3.parametrizations.weight.original",
              "generator.conditioning_stack.d1.first_conv_3x3.parametrizations.weight.original",
              "generator.sampler.convGRU4.cell.read_gate_conv.parametrizations.weight.0._u",
              "generator.sampler.up_g4.first_conv_3x3.parametrizations.weight.0._v",
              "generator.sampler.g1.bn1.running_var",
              "discriminator.spatial_discriminator.fc.parametrizations.weight.original",
              "discriminator.temporal_discriminator.d1.first_conv_3x3.parametrizations.weight.original",
              "discriminator.spatial_discriminator.bn.running_var"]:
        rec["post." + k] = sd1[k].detach().clone().contiguous()
    return rec, keys0


def training_steps_adv_golden():
    """THREE consecutive `DGMR.training_step`s with a visible adversarial path (dgmr/dgmr.py:137-218).

    In the default-hyper-parameter golden above, 20 * grid_cell_reg ~ 1e11 swamps loss_hinge_gen by ten orders of magnitude and the
    second D pass saturates the hinge, so the chain  hinge_gen -> discriminator data gradient -> generator  is invisible there.
    Here grid_lambda = 0 (the generator's gradient is purely adversarial), the learning rates are small enough that every hinge stays
    active over all six D updates, and beta1 = 0.5 with three steps exercises Adam's first moment, both bias corrections (step 2..6
    for D, 2..3 for G) and every weight-derived cache across optimiser updates.

    The reference's OWN reproducibility is recorded next to the values: the same run is repeated with 4 and 1 CPU threads (only the
    fp32 summation order inside torch's kernels changes) and `noise.<key>` = the largest deviation from the stored run, relative to
    the tensor's max magnitude (for `backward_losses`: relative per entry).  After three steps of both networks that band is 1e-3
    ... 2e-2 for the discriminator's gradients and 5e-2 ... 3e-1 for the generator's deep layers (the adversarial gradient through
    batch-statistics BatchNorm on 2-4 samples and ~1e5 ReLU boundaries is chaotic in fp32), so a test can only ask for agreement
    within a multiple of it; the well-conditioned comparison of that chain is tests/test_gpu_adversarial.py (float64 anchor).
    """
    default_threads = torch.get_num_threads()
    rec, keys0 = _adv_run()
    noise = {}
    for threads in (4, 1):
        torch.set_num_threads(threads)
        other, _ = _adv_run()
        for k, r in rec.items():
            if k.startswith("grad.") or k == "backward_losses":
                if k == "backward_losses":
                    dev = ((other[k] - r).abs() / r.abs()).max().item()
                else:
                    dev = (other[k] - r).abs().max().item() / max(r.abs().max().item(), 1e-30)
                noise[k] = max(noise.get(k, 0.0), dev)
    torch.set_num_threads(default_threads)
    for k, v in noise.items():
        rec["noise." + k] = torch.tensor(v, dtype=torch.float64)
        print(f"  reference run-to-run band  {k:100s} {v:.2e}")
    hp = dict(TS_KW)
    hp.update(ADV_HP)
    save_file(rec, os.path.join(OUT, "training_steps_adv.safetensors"),
              metadata={"keys": json.dumps(keys0), "kw": json.dumps(hp), "seeds": "[42, 43, 44]", "steps": str(ADV_STEPS),
                        "threads": str(default_threads)})
    print("training_steps_adv golden: backward losses", rec["backward_losses"].tolist())


def validation_step_golden():
    """`DGMR.validation_step` of the unmodified reference (dgmr/dgmr.py:220-290): the three logged losses, in eval mode under
    no_grad (how Lightning's validation loop calls it) and in train mode (buffers then advance: their fingerprints are stored)."""
    from dgmr import DGMR

    rec = {}
    for mode in ("eval", "train"):
        torch.manual_seed(42)
        model = DGMR(**TS_KW)
        model.train(mode == "train")
        torch.manual_seed(43)
        images, future = torch.rand(2, 4, 1, 128, 128), torch.rand(2, 2, 1, 128, 128)
        logged = {}
        model.log_dict = lambda d, **k: logged.update({kk: float(v) for kk, v in d.items()})
        torch.manual_seed(45)
        with torch.no_grad():
            model.validation_step((images, future), 0)
        keys, cs = checksums(model.state_dict())
        rec[f"{mode}.losses"] = torch.tensor([logged["val/d_loss"], logged["val/g_loss"], logged["val/grid_loss"]], dtype=torch.float64)
        rec[f"{mode}.cs1"] = cs
        print(f"validation_step golden ({mode}):", rec[f"{mode}.losses"].tolist())
    rec["images"], rec["future"] = images, future
    save_file(rec, os.path.join(OUT, "validation_step.safetensors"),
              metadata={"keys": json.dumps(keys), "kw": json.dumps(TS_KW), "seeds": "[42, 43, 45]"})


if __name__ == "__main__":
    if len(sys.argv) > 1:  # regenerate selected fixtures only, e.g. `python oracle/gen_golden.py training_steps_adv_golden`
        for name in sys.argv[1:]:
            globals()[name]()
    else:
        main()
