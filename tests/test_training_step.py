"""The whole `DGMR.training_step` against the golden produced by the unmodified reference.

* CPU (not gpu): the oracle's restatement of the step reproduces the reference (pins oracle.training_step), and
  a seeded construction of our DGMR reproduces the reference's initial state bit-for-bit.
* GPU: our HIP training step reproduces the reference's losses, post-step parameters and buffers.

Tolerances.  Losses: 1e-3 relative (north-star bound).  Gradients: the discriminator's and the generator's last
layer (sampler.bn / sampler.conv_1x1) are well conditioned and are compared at 2e-4 of their max magnitude.  The rest
of the generator's gradient is NOT: it is dominated by the grid-cell term, whose upstream gradient is sign(mean - y) * w
(losses.py:188-192) -- a sum of +-const terms that cancels to ~1/sqrt(#pixels) of its parts, so a single ReLU mask that
flips on fp32 rounding noise moves it by ~0.5 %.  Measured with the CPU oracle alone (same code, same seeds):
8 threads vs 1 thread differ by 1.2e-3 ... 4.3e-3 of max, a 1e-7 relative weight perturbation by up to 8e-3, and the
oracle vs the reference's golden by up to 1.2e-2 (att_block.gamma).  Those tensors are therefore held to 5e-2 of max
AND a cosine similarity >= 0.999 with the reference gradient.  Buffers (u/v, BN running statistics): 1e-3 of the
tensor's max.  Parameters after Adam: with beta1 = 0 the first Adam step is lr * g / (|g| + eps) ~ +-lr, so an
element whose gradient is rounding noise can flip sign (difference 2 * lr); the gradients themselves are compared
at 2e-3 of their max magnitude, and for the stepped parameters we require >= 90 % of the elements to agree to 1e-5 + 1e-4 * |ref| and no element to move by more than 2.1 * lr.
"""
import json
import os

import pytest
import torch

from conftest import GOLDEN, load_golden

LR_MAX = 2e-4


def _golden():
    from safetensors import safe_open

    rec, meta = load_golden("training_step")
    return rec, json.loads(meta["keys"]), json.loads(meta["kw"])


def _checksums(sd, keys):
    vals = torch.zeros(len(keys), 4, dtype=torch.float64)
    for i, k in enumerate(keys):
        t = sd[k].detach().double().flatten().cpu()
        vals[i, 0], vals[i, 1], vals[i, 2], vals[i, 3] = t.sum(), t.abs().sum(), t[0], t[-1]
    return vals


def _is_param(k):
    return not k.endswith(("._u", "._v", "running_mean", "running_var", "num_batches_tracked"))


def _check_post(sd1, rec, keys):
    # element-wise on the stored tensors
    for k, ref in rec.items():
        if not k.startswith("post."):
            continue
        got = sd1[k[5:]].detach().cpu().float()
        if _is_param(k):
            diff = (got - ref).abs()
            bad = (diff > 1e-5 + 1e-4 * ref.abs()).float().mean().item()
            assert bad < 0.10, f"{k}: {bad:.3%} of elements differ"
            assert diff.max().item() <= 2.1 * LR_MAX + 1e-6, f"{k}: max diff {diff.max().item():.3e}"
        else:
            scale = ref.abs().max().item()
            assert (got - ref).abs().max().item() <= 1e-3 * scale + 1e-6, k
    # fingerprints of every tensor in the model.  Parameters: with beta1 = 0 the first Adam step is +-lr per element, so an element
    # whose gradient is rounding noise may land on the other side (2 lr away); allowed for 2 % of a tensor's elements — and for ALL
    # elements of a conv bias that feeds straight into BatchNorm, whose exact gradient is zero (the reference's own update of those
    # is the sign of its rounding noise).
    import re

    noise_bias = re.compile(r"sampler\.(g\d|up_g\d)\.first_conv_3x3\.bias$|sampler\.up_g4\.(last_conv_3x3|conv_1x1)\.bias$")
    cs = _checksums(sd1, keys)
    ref = rec["cs1"]
    for i, k in enumerate(keys):
        n = sd1[k].numel()
        frac = 1.0 if noise_bias.search(k) else 0.02
        tol = (2.5 * LR_MAX * n * frac + 1e-3 * ref[i, 1].item() + 1e-4) if _is_param(k) else (2e-3 * ref[i, 1].item() + 1e-5)
        assert abs(cs[i, 1].item() - ref[i, 1].item()) <= tol, f"{k}: abs-sum {cs[i, 1].item()} vs {ref[i, 1].item()}"


WELL_CONDITIONED = ("grad.discriminator.", "grad.generator.sampler.bn.", "grad.generator.sampler.conv_1x1.")


def _check_grads(grads, rec, well_tol=2e-4, ill_tol=5e-2):
    n = 0
    for k, ref in rec.items():
        if not k.startswith("grad."):
            continue
        assert k[5:] in grads, f"{k}: parameter received no gradient"
        got = grads[k[5:]].detach().cpu().float().reshape(ref.shape)
        scale = ref.abs().max().item()
        err = (got - ref).abs().max().item()
        tol = well_tol if k.startswith(WELL_CONDITIONED) else ill_tol  # see the module docstring
        if k.endswith("att_block.gamma"):
            # a single scalar, i.e. ONE heavily cancelling sum: the CPU oracle alone moves it by 1.2e-2 against the reference's
            # golden; the HIP path (split-K, slab and atomic reductions reorder the sum run to run) lands at 2.6e-2 ... 5.9e-2
            tol = 1.5e-1
        assert err <= tol * scale + 1e-7, f"{k}: grad abs err {err:.3e} at scale {scale:.3e} (tol {tol})"
        if scale > 0:
            cos = torch.nn.functional.cosine_similarity(got.flatten().double(), ref.flatten().double(), dim=0).item()
            assert cos >= 0.999, f"{k}: cosine {cos}"
        n += 1
    assert n >= 10


def _snapshot_grads(opt, named, prefix, store, once):
    orig = opt.step

    def step(*a, **k):
        if not (once and any(kk.startswith(prefix) for kk in store)):
            for kk, p in named.items():
                if kk.startswith(prefix) and p.grad is not None:
                    store[kk] = p.grad.detach().clone()
        return orig(*a, **k)

    opt.step = step


def test_seeded_init_matches_reference():
    import skillful_nowcasting_amd as S

    rec, keys, kw = _golden()
    torch.manual_seed(42)
    model = S.DGMR(**kw)
    sd = model.state_dict()
    assert sorted(sd.keys()) == keys
    assert torch.equal(_checksums(sd, keys), rec["cs0"]), "seeded construction differs from the reference's"


def test_oracle_training_step_matches_reference():
    import skillful_nowcasting_amd as S
    from oracle import dgmr_oracle as O

    rec, keys, kw = _golden()
    torch.manual_seed(42)
    model = S.DGMR(**kw)
    full = model.state_dict()
    sd = {k: v.detach().clone().contiguous() for k, v in full.items() if k.startswith(("generator.", "discriminator."))}
    hp = dict(forecast_steps=kw["forecast_steps"], generation_steps=kw["generation_steps"], grid_lambda=20.0, gen_lr=5e-5,
              disc_lr=2e-4, beta1=0.0, beta2=0.999, precip_weight_cap=24.0, latent_shape=(8, 4, 4), num_spatial_frames=8)
    opt = {"step": {}, "m": {}, "v": {}}
    torch.manual_seed(44)
    d_loss, g_loss, grid = O.training_step(sd, rec["images"], rec["future"], hp, opt)
    ref = rec["losses"].tolist()
    assert abs(d_loss - ref[0]) <= 1e-4 + 1e-3 * abs(ref[0])
    assert abs(g_loss - ref[1]) <= 1e-3 * abs(ref[1])
    assert abs(grid - ref[2]) <= 1e-3 * abs(ref[2])
    sd1 = {k: sd.get(k, sd.get("generator." + k)) for k in keys}
    _check_post(sd1, rec, keys)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f32", "bf16x3", "mixed", "bf16x6"])
def test_hip_training_step_matches_reference(precision):
    """f32: exact-fp32 MFMA kernels.  bf16x3: contractions on the bf16 matrix cores with operands split into two bf16 planes (16-bit
    products).  mixed: bf16x3 with the discriminator forward in bf16x6 (bench.py's default).  bf16x6: three planes, fp32-faithful."""
    import skillful_nowcasting_amd as S

    S.set_precision(precision)
    try:
        # bf16x3 perturbs every product by ~2^-16 instead of 2^-24: more ReLU-mask flips in the cancelling generator gradient
        # (measured up to 5.1e-2 on conditioning_stack.d1, 3.4e-4 on the worst D gradient)
        # mixed = bf16x3 with the discriminator forward in bf16x6 (bench.py's default); bf16x6: fp32-faithful products, f32's bounds
        # pure bf16x3 (the discriminator forward in 16-bit products as well - an A/B mode): its discriminator gradients sit on relu flips
        # of the 4 x 4 maps in front of the heads; round 4 (DBlock tail as one operator: another rounding sequence, same precision)
        # measured 2.7e-3 on temporal_discriminator.d2.last_conv_3x3.bias where round 3 had 3.4e-4 as its worst
        tols = {"f32": (2e-4, 5e-2), "bf16x6": (2e-4, 5e-2), "mixed": (1e-3, 1.5e-1), "bf16x3": (5e-3, 1.5e-1)}
        _hip_training_step(*tols[precision])
    finally:
        S.set_precision("f32")


def _hip_training_step(well_tol, ill_tol):
    import skillful_nowcasting_amd as S

    rec, keys, kw = _golden()
    torch.manual_seed(42)
    model = S.DGMR(**kw).to("cuda")
    bw = []
    orig = model.manual_backward
    model.manual_backward = lambda loss: (bw.append(loss.detach()), orig(loss))
    grads = {}
    named = {("generator." + k if not k.startswith("discriminator.") else k): p for k, p in model.named_parameters()}
    g_opt, d_opt = model.optimizers()
    _snapshot_grads(g_opt, named, "generator.", grads, False)
    _snapshot_grads(d_opt, named, "discriminator.", grads, True)
    torch.manual_seed(44)
    out = model.training_step((rec["images"].cuda(), rec["future"].cuda()), 0)
    torch.cuda.synchronize()
    _check_grads(grads, rec, well_tol, ill_tol)
    ref_bw = rec["backward_losses"].tolist()
    got_bw = [float(x) for x in bw]
    for g, r in zip(got_bw, ref_bw):
        assert abs(g - r) <= 1e-4 + 1e-3 * abs(r), (got_bw, ref_bw)
    ref = rec["losses"].tolist()
    assert abs(float(out["g_loss"]) - ref[1]) <= 1e-3 * abs(ref[1])
    assert abs(float(out["grid_loss"]) - ref[2]) <= 1e-3 * abs(ref[2])
    _check_post(model.state_dict(), rec, keys)
    # the 12 parameters the reference never gives a gradient must be untouched here as well
    dead = [k for k, p in model.named_parameters() if p.grad is None]
    assert len(dead) == 12, dead
