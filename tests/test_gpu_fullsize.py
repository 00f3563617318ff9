"""BASELINE.json's full-size configuration (paper config: 4 -> 18 frames, 256x256, latent 768 / context 384) on the GPU against
the CPU oracle run on the same box: whole-generator and whole-discriminator forward, north-star tolerance 1e-3 relative.

The goldens under tests/golden/ are small by necessity (they are committed); this test closes the gap to the real sizes, where
the T-batched launches, the LDS-window kernel, split-K and the multi-module spectral-norm plan all take their production paths.
"""
import pytest
import torch
import torch.nn.functional as F

from conftest import band_check as _band_check

pytestmark = pytest.mark.gpu

KW = dict(forecast_steps=18, output_shape=256, latent_channels=768, context_channels=384, generation_steps=6)


@pytest.fixture(scope="module")
def setup():
    import skillful_nowcasting_amd as S

    torch.manual_seed(0)
    model = S.DGMR(**KW)
    sd_cpu = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = torch.rand(1, 4, 1, 256, 256)
    y = torch.rand(1, 18, 1, 256, 256)
    return model.to("cuda"), sd_cpu, x, y


@pytest.mark.parametrize("precision,tol", [("f32", 1e-3), ("bf16x3", 1e-3)])
def test_generator_forward_paper_config(setup, precision, tol):
    import skillful_nowcasting_amd as S
    from oracle import dgmr_oracle as O

    model, sd_cpu, x, _ = setup
    sd0 = {k: v.clone() for k, v in sd_cpu.items()}
    model.load_state_dict(sd_cpu)  # both sides start from the same u/v and BatchNorm buffers
    torch.manual_seed(1)
    z = O.draw_latent((8, 8, 8))
    ref = O.generator(sd0, "", x, z, 18, True)
    S.set_precision(precision)
    try:
        torch.manual_seed(1)
        out = model(x.cuda())
        torch.cuda.synchronize()
    finally:
        S.set_precision("f32")
    err = (out.cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert out.shape == (1, 18, 1, 256, 256)
    assert err <= tol, f"{precision}: generator forward rel err {err:.3e}"
    # state moved identically (one power iteration per call, BatchNorm running statistics): a few buffers as witnesses
    sd1 = model.state_dict()
    for k in ("sampler.up_g4.first_conv_3x3.parametrizations.weight.0._u", "sampler.convGRU1.cell.read_gate_conv.parametrizations.weight.0._v",
              "sampler.g2.bn1.running_var", "conditioning_stack.d1.first_conv_3x3.parametrizations.weight.0._u"):
        a, b = sd1[k].cpu(), sd0[k]
        assert (a - b).abs().max().item() <= 2e-3 * b.abs().max().item() + 1e-6, k


def test_discriminator_forward_paper_config(setup):
    from oracle import dgmr_oracle as O

    model, sd_cpu, x, y = setup
    sd0 = {k: v.clone() for k, v in sd_cpu.items()}
    model.load_state_dict(sd_cpu)
    seq = torch.cat([torch.cat([x, y], 1), torch.cat([x, y.flip(1)], 1)], 0)  # [2, 22, 1, 256, 256]
    torch.manual_seed(3)
    idxs = torch.randint(0, 22, (8,)).tolist()
    ref = O.discriminator(sd0, "discriminator.", seq, idxs, True)
    torch.manual_seed(3)
    out = model.discriminator(seq.cuda())
    torch.cuda.synchronize()
    assert out.shape == (2, 2, 1)
    err = (out.cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert err <= 1e-3, f"discriminator forward rel err {err:.3e}"


def test_eval_forward_and_validation_step_small_config():
    """inference (`forward` in eval mode: frozen spectral-norm sigma, BatchNorm running statistics) against
    the oracle's eval path, and `validation_step` (dgmr/dgmr.py:220-290) end to end."""
    import skillful_nowcasting_amd as S
    from oracle import dgmr_oracle as O

    kw = dict(forecast_steps=4, output_shape=128, latent_channels=384, context_channels=192, generation_steps=2)
    torch.manual_seed(5)
    model = S.DGMR(**kw)
    sd_cpu = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to("cuda").eval()
    x = torch.rand(2, 4, 1, 128, 128)
    y = torch.rand(2, 4, 1, 128, 128)
    torch.manual_seed(6)
    z = O.draw_latent((8, 4, 4))
    ref = O.generator(sd_cpu, "", x, z, 4, False)
    torch.manual_seed(6)
    with torch.no_grad():
        out = model(x.cuda())
    err = (out.cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert err <= 1e-3, f"eval-mode generator forward rel err {err:.3e}"
    # eval mode must not move any state
    for k, v in model.state_dict().items():
        assert torch.equal(v.cpu(), sd_cpu[k]), f"{k} changed in eval mode"
    with torch.no_grad():
        model.validation_step((x.cuda(), y.cuda()), 0)
    torch.cuda.synchronize()
    logged = {k: float(v) for k, v in model.logged_metrics.items()} if hasattr(model, "logged_metrics") else {}
    assert set(logged) == {"val/d_loss", "val/g_loss", "val/grid_loss"} and all(v == v for v in logged.values()), logged


# ------------------------------------------------------------------------------------------------------------------------------
# forward + BACKWARD at the benchmarked shapes (reference: tests/test_model.py:227-259,285-306 run the paper-config model forward,
# an MSE loss and backward; here the same on the HIP path, with losses and weight gradients compared against the CPU oracle)
# ------------------------------------------------------------------------------------------------------------------------------
G_GRAD_KEYS = ["sampler.up_g4.first_conv_3x3.parametrizations.weight.original",   # the dominant window kernel's layer (128x128, 96 ch)
               "sampler.up_g4.last_conv_3x3.parametrizations.weight.original",    # 96 -> 48 at 128x128
               "sampler.g1.last_conv_3x3.parametrizations.weight.original",       # 8x8 maps, 768 ch
               "sampler.convGRU1.cell.read_gate_conv.parametrizations.weight.original",  # shared-input ConvGRU, x and h halves
               "sampler.convGRU1.cell.output_conv.parametrizations.weight.original",
               "sampler.convGRU3.cell.update_gate_conv.parametrizations.weight.original",
               "sampler.gru_conv_1x1_2.parametrizations.weight.original",
               "sampler.conv_1x1.parametrizations.weight.original", "sampler.bn.weight", "sampler.g3.bn2.bias",
               "conditioning_stack.d1.first_conv_3x3.parametrizations.weight.original",
               "conditioning_stack.conv4.parametrizations.weight.original",
               "latent_stack.l_block1.first_conv_3x3.weight", "latent_stack.att_block.gamma",
               "latent_stack.conv_3x3.parametrizations.weight.original"]


@pytest.fixture(scope="module")
def oracle_g_fwd_bwd(setup):
    """Oracle: generator forward on x, MSE against y, backward - in float32 (the reference's arithmetic) and in float64 (truth)."""
    from oracle import dgmr_oracle as O

    torch.set_num_threads(min(16, torch.get_num_threads()))  # torch's CPU convs are slowest at this box's default of 128 threads
    _, sd_cpu, x, y = setup
    res = {}
    for dt in (torch.float32, torch.float64):
        sd = {k[len("generator."):]: v.clone().to(dt) if v.is_floating_point() else v.clone() for k, v in sd_cpu.items()
              if k.startswith("generator.")}
        with torch.no_grad():
            sd["latent_stack.att_block.gamma"].fill_(0.3)  # the reference initialises gamma to 0, which hides the attention path
        for k in G_GRAD_KEYS:
            sd[k].requires_grad_(True)
        torch.manual_seed(11)
        z = O.draw_latent((8, 8, 8)).to(dt)
        out = O.generator(sd, "", x.to(dt), z, 18, True)
        loss = torch.nn.functional.mse_loss(out, y.to(dt))
        loss.backward()
        res[dt] = (loss.detach(), {k: sd[k].grad.clone() for k in G_GRAD_KEYS}, out.detach())
    return res


@pytest.mark.parametrize("precision,tol", [("f32", 1e-3), ("bf16x3", 2e-3)])
def test_generator_fwd_bwd_paper_config(setup, oracle_g_fwd_bwd, precision, tol):
    """Paper config, B = 1: forward, MSE, backward (reference: tests/test_model.py:227-259,285-306 does exactly this, asserting
    shapes only).  Output, loss and 15 weight gradients - the dominant window kernel's layer, 8x8-map layers, ConvGRU x / h halves
    incl. the shared-input one, 1x1 convs, BatchNorm parameters, context and latent stacks, the attention gain - against the
    float64 oracle, bound max(tol, 3 x the fp32 oracle's own error), cosine >= 0.9999."""
    import skillful_nowcasting_amd as S

    model, sd_cpu, x, y = setup
    (l32, g32, o32), (l64, g64, o64) = oracle_g_fwd_bwd[torch.float32], oracle_g_fwd_bwd[torch.float64]
    model.load_state_dict(sd_cpu)
    with torch.no_grad():
        model.generator.latent_stack.att_block.gamma.fill_(0.3)
    S.ops.bump_weights_epoch()
    model.train()
    for p in model.parameters():
        p.grad = None
    S.set_precision(precision)
    try:
        torch.manual_seed(11)
        out = model(x.cuda())
        loss = ((out - y.cuda()) ** 2).mean()  # torch elementwise ops: test-side loss, as the reference's own test does
        loss.backward()
        torch.cuda.synchronize()
    finally:
        S.set_precision("f32")
    named = dict(model.generator.named_parameters())
    rows = {"output": (out.detach().cpu(), o32, o64), "loss": (loss.detach().cpu(), l32, l64)}
    for k in G_GRAD_KEYS:
        rows["grad " + k] = (named[k].grad.detach().cpu().float().reshape(g64[k].shape), g32[k], g64[k])
    _band_check("generator fwd + MSE + bwd, paper config", precision, tol, rows)


D_GRAD_KEYS = ["temporal_discriminator.d1.first_conv_3x3.parametrizations.weight.original",   # 3x3x3, 4 -> 48
               "temporal_discriminator.d1.last_conv_3x3.parametrizations.weight.original",    # 3x3x3, 48 -> 48: 31 % of D's FLOPs
               "temporal_discriminator.d2.conv_1x1.parametrizations.weight.original",
               "temporal_discriminator.intermediate_dblocks.1.last_conv_3x3.parametrizations.weight.original",
               "temporal_discriminator.fc.parametrizations.weight.original", "temporal_discriminator.bn.weight",
               "spatial_discriminator.d1.first_conv_3x3.parametrizations.weight.original",
               "spatial_discriminator.intermediate_dblocks.2.last_conv_3x3.parametrizations.weight.original",
               "spatial_discriminator.d6.first_conv_3x3.parametrizations.weight.original",
               "spatial_discriminator.fc.bias"]


def _d_inputs(data):
    torch.manual_seed(31)
    if data == "iid":
        # eight iid uniform-noise sequences = what bench.py feeds (torch.rand frames).  A degenerate batch for the BatchNorm1d in front
        # of the last linear layer: the pooled features of the eight samples agree to 1e-3 of their size (|mean| / batch std = 870
        # median, 2600 at the 90th percentile, measured on the CPU oracle); the normalisation divides by that spread and every
        # gradient of the spatial discriminator inherits the FORWARD's rounding error x ~1e3: 1.7e-1 with 16-bit products (bf16x3,
        # profiles/r02_pytest_gpu_r2p.log), 2.4e-4 for the reference's own fp32.  This is why the bench's default mode ("mixed") runs
        # the discriminator forward in bf16x6; plain bf16x3 is not held to this case.
        seq = torch.rand(8, 22, 1, 256, 256)
    else:
        # eight DISTINCT sequences (per-sample amplitude and smooth structure), as a batch of real and generated radar is: ratio 2.8
        amp = torch.linspace(0.15, 2.5, 8).view(8, 1, 1, 1, 1)
        low = F.interpolate(torch.rand(8 * 22, 1, 8, 8), size=(256, 256), mode="bilinear", align_corners=False).view(8, 22, 1, 256, 256)
        seq = amp * (0.5 * torch.rand(8, 22, 1, 256, 256) + low * torch.rand(8, 1, 1, 1, 1) * 2)
    cot = torch.randn(8, 2, 1)
    return seq, cot


def _oracle_d(setup, data, aligner):
    """Oracle discriminator forward + backward in float32 and float64, on the linear piece the HIP forward was on (KinkAligner)."""
    from oracle import dgmr_oracle as O

    torch.set_num_threads(min(16, torch.get_num_threads()))
    _, sd_cpu, _, _ = setup
    seq, cot = _d_inputs(data)
    torch.manual_seed(3)
    idxs = torch.randint(0, 22, (8,)).tolist()
    ref = {}
    for dt in (torch.float32, torch.float64):
        sd = {k[len("discriminator."):]: v.clone().to(dt) if v.is_floating_point() else v.clone() for k, v in sd_cpu.items()
              if k.startswith("discriminator.")}
        d_keys = [k for k in O.param_keys(sd, "")]  # every parameter: the table then shows WHERE along the backward an error enters
        for k in d_keys:
            sd[k].requires_grad_(True)
        s_ = seq.detach().clone().to(dt).requires_grad_(True)
        with aligner.oracle(O):
            o = O.discriminator(sd, "", s_, idxs, True)
        (o * cot.to(dt)).sum().backward()
        ref[dt] = (o.detach(), {k: sd[k].grad.clone() for k in d_keys if sd[k].grad is not None}, s_.grad.clone())
    return ref


@pytest.mark.parametrize("precision,tol,data", [("f32", 1e-3, "distinct"), ("bf16x3", 2e-3, "distinct"), ("mixed", 2e-3, "distinct"),
                                                ("f32", 1e-3, "iid"), ("mixed", 2e-3, "iid"), ("bf16x6", 1e-3, "iid")])
def test_discriminator_fwd_bwd_paper_config(setup, precision, tol, data):
    """Full-depth discriminator on 4 real + 4 generated 22-frame sequences at 256 x 256 (BatchNorm1d over 8 samples; the step itself
    runs 32): scores, EVERY weight gradient and the gradient with respect to the input frames - the tensor through which
    loss_hinge_gen reaches the generator (dgmr/dgmr.py:186-196).  Float64-anchored band (conftest.band_check) with NO allowance for
    relu flips: the oracle is evaluated on the HIP forward's own relu masks (conftest.KinkAligner), so both sides compute the same
    smooth function.  `iid`: bench.py's own input distribution, in exact f32, in the bench's default mode ("mixed") and in bf16x6."""
    import skillful_nowcasting_amd as S
    from conftest import KinkAligner

    model, sd_cpu, _, _ = setup
    seq, cot = _d_inputs(data)
    model.load_state_dict(sd_cpu)
    S.ops.bump_weights_epoch()
    model.train()
    for p in model.parameters():
        p.grad = None
    seq_dev = seq.detach().clone().cuda().requires_grad_(True)
    S.set_precision(precision)
    try:
        torch.manual_seed(3)
        with KinkAligner(model.discriminator) as aligner:
            out = model.discriminator(seq_dev)
        (out * cot.cuda()).sum().backward()
        torch.cuda.synchronize()
    finally:
        S.set_precision("f32")
    ref = _oracle_d(setup, data, aligner)
    kinks = aligner.report()
    from conftest import _log_band

    msg = (f"relu masks that differ from the oracle's own sign [{precision}, {data}]: "
           + (", ".join(f"{t} ({d.replace('torch.', '')}): {n} elements, |x| <= {r:.1e} of max" for t, d, n, r in kinks) or "none"))
    print("\n" + msg)
    _log_band(msg)
    # a mask may differ from the oracle's sign only where the pre-activation is rounding noise
    lim = {"f32": 2e-5, "bf16x6": 2e-5, "mixed": 2e-5, "bf16x3": 2e-3}[precision]
    assert all(r <= lim for _, d, _, r in kinks if d == "torch.float64"), kinks
    named = dict(model.discriminator.named_parameters())
    (o32, g32, x32), (o64, g64, x64) = ref[torch.float32], ref[torch.float64]
    rows = {"scores": (out.detach().cpu(), o32, o64), "d / d frames": (seq_dev.grad.detach().cpu(), x32, x64)}
    for k in g64:
        if g64[k].abs().max().item() == 0.0:
            continue  # (temporal fc.bias / conv biases in front of nothing: exactly zero on both sides)
        rows["grad " + k] = (named[k].grad.detach().cpu().float().reshape(g64[k].shape), g32[k], g64[k])
    assert all(k in g64 for k in D_GRAD_KEYS)
    _band_check(f"discriminator fwd + bwd, paper config, {data} sequences", precision, tol, rows)


@pytest.mark.parametrize("precision,tol", [("f32", 1e-4), ("bf16x3", 1e-3)])
def test_batched_draws_equal_sequential_forwards_paper_config(setup, precision, tol):
    """Generator.forward_draws(x, K) == K consecutive forward(x) calls (same latents): outputs AND every buffer (u, v, running
    statistics) afterwards, at the paper configuration - the batched generator pass of training_step against the reference's
    Python loop (dgmr/dgmr.py:174-177).  Same arithmetic on both sides, but 3x the rows pick other tile / split-K variants, i.e.
    another fp32 summation order; the 18-step recurrences and B = 1 BatchNorm statistics (64 elements per channel at the first level)
    amplify that rounding noise ~35x: 1e-4 in exact f32, 1e-3 in bf16x3 (measured 2.7e-4)."""
    import skillful_nowcasting_amd as S

    model, sd_cpu, x, _ = setup
    k = 3
    xd = x.cuda()
    S.set_precision(precision)
    try:
        model.load_state_dict(sd_cpu)
        S.ops.bump_weights_epoch()
        model.train()
        torch.manual_seed(21)
        with torch.no_grad():
            seq = torch.cat([model(xd) for _ in range(k)], dim=0)
        sd_seq = {n: v.detach().clone() for n, v in model.state_dict().items()}
        model.load_state_dict(sd_cpu)
        S.ops.bump_weights_epoch()
        torch.manual_seed(21)
        with torch.no_grad():
            bat = model.generator.forward_draws(xd, k)
        torch.cuda.synchronize()
    finally:
        S.set_precision("f32")
    assert bat.shape == seq.shape
    assert (bat - seq).abs().max().item() <= tol * seq.abs().max().item()
    for n, v in model.state_dict().items():
        if n.endswith(("._u", "._v", "running_mean", "running_var", "num_batches_tracked")):
            a, b = v.float(), sd_seq[n].float()
            assert (a - b).abs().max().item() <= tol * b.abs().max().item() + 1e-7, n


# ------------------------------------------------------------------------------------------------------------------------------
# the other BASELINE.json configurations
# ------------------------------------------------------------------------------------------------------------------------------
def test_generator_forward_cfg5_512(setup):
    """BASELINE.json configs[4] (MRMS-shape stress, 512 x 512 crops): generator forward, B = 1, against the oracle."""
    import skillful_nowcasting_amd as S
    from oracle import dgmr_oracle as O

    kw = dict(forecast_steps=18, output_shape=512, latent_channels=768, context_channels=384, generation_steps=6)
    torch.manual_seed(0)
    model = S.DGMR(**kw)
    sd_cpu = {k[len("generator."):]: v.detach().clone() for k, v in model.state_dict().items() if k.startswith("generator.")}
    model = model.to("cuda")
    x = torch.rand(1, 4, 1, 512, 512)
    torch.manual_seed(1)
    z = O.draw_latent((8, 16, 16))
    ref = O.generator({k: v.clone() for k, v in sd_cpu.items()}, "", x, z, 18, True)  # (the oracle advances u / v / BN in its dict)
    for precision in ("f32", "bf16x3"):
        model.load_state_dict({**model.state_dict(), **{"generator." + k: v for k, v in sd_cpu.items()}}, strict=False)
        S.ops.bump_weights_epoch()
        S.set_precision(precision)
        try:
            torch.manual_seed(1)
            with torch.no_grad():
                out = model(x.cuda())
            torch.cuda.synchronize()
        finally:
            S.set_precision("f32")
        assert out.shape == (1, 18, 1, 512, 512)
        err = (out.cpu() - ref).abs().max().item() / ref.abs().max().item()
        assert err <= 1e-3, f"cfg5 {precision}: generator forward rel err {err:.3e}"


@pytest.mark.parametrize("precision,tol", [("f32", 1e-3), ("bf16x3", 2e-3)])
def test_generator_fwd_bwd_cfg5_512(precision, tol):
    """BASELINE.json configs[4] (512 x 512): generator forward, MSE, BACKWARD at B = 1 against the oracle in float32 and float64 (round 4:
    the 512 x 512 configuration had a forward check only, its backward was benched but unchecked).  Same 15 gradients and the same
    float64-anchored band as the paper-configuration test above - the 128 x 128 ... 256 x 256 maps of this configuration put four times
    the tiles through every window / weight-gradient kernel and twice the rows through the ConvGRU steps."""
    import skillful_nowcasting_amd as S
    from oracle import dgmr_oracle as O

    kw = dict(forecast_steps=18, output_shape=512, latent_channels=768, context_channels=384, generation_steps=6)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    torch.manual_seed(0)
    model = S.DGMR(**kw)
    sd_cpu = {k: v.detach().clone() for k, v in model.state_dict().items()}
    torch.manual_seed(3)
    x = torch.rand(1, 4, 1, 512, 512)
    y = torch.rand(1, 18, 1, 512, 512)
    res = _CFG5_ORACLE.get("res")
    if res is None:
        res = {}
        for dt in (torch.float32, torch.float64):
            sd = {k[len("generator."):]: v.clone().to(dt) if v.is_floating_point() else v.clone() for k, v in sd_cpu.items()
                  if k.startswith("generator.")}
            with torch.no_grad():
                sd["latent_stack.att_block.gamma"].fill_(0.3)
            for k in G_GRAD_KEYS:
                sd[k].requires_grad_(True)
            torch.manual_seed(11)
            z = O.draw_latent((8, 16, 16)).to(dt)
            out = O.generator(sd, "", x.to(dt), z, 18, True)
            loss = torch.nn.functional.mse_loss(out, y.to(dt))
            loss.backward()
            res[dt] = (loss.detach(), {k: sd[k].grad.clone() for k in G_GRAD_KEYS}, out.detach())
        _CFG5_ORACLE["res"] = res
    (l32, g32, o32), (l64, g64, o64) = res[torch.float32], res[torch.float64]
    model = model.to("cuda")
    with torch.no_grad():
        model.generator.latent_stack.att_block.gamma.fill_(0.3)
    S.ops.bump_weights_epoch()
    model.train()
    S.set_precision(precision)
    try:
        torch.manual_seed(11)
        out = model(x.cuda())
        loss = ((out - y.cuda()) ** 2).mean()
        loss.backward()
        torch.cuda.synchronize()
    finally:
        S.set_precision("f32")
    named = dict(model.generator.named_parameters())
    rows = {"output": (out.detach().cpu(), o32, o64), "loss": (loss.detach().cpu(), l32, l64)}
    for k in G_GRAD_KEYS:
        rows["grad " + k] = (named[k].grad.detach().cpu().float().reshape(g64[k].shape), g32[k], g64[k])
    # bf16x3 at this size: the context stack's gradients - sums over 4 x 65 k pixels behind 60 layers - land at 12 x (d1.first_conv: 3.9e-2
    # of max) and 17 x (conv4: 1.3e-1 on 328 of 2.65 M elements) the float32 oracle's own distance from float64, against 2 - 8 x at
    # 256 x 256: 16-bit products, not a kernel defect (exact f32 passes the factor-3 band on the same code path) - factor 20 here
    _band_check("generator fwd + MSE + bwd, cfg5 (512 x 512), B = 1", precision, tol, rows, factor=20.0 if precision == "bf16x3" else None)
    del model
    torch.cuda.empty_cache()


_CFG5_ORACLE = {}


def test_cfg2_bf16_forward_and_step():
    """BASELINE.json configs[1]: T = 4, 384 / 192 channels, 256 x 256, plain `bf16` arithmetic (operands rounded to bf16, fp32
    accumulation).  bf16 carries 8 mantissa bits (2^-9 per operand), so the 1e-3 fp32 bound cannot apply to this mode: through ~60
    convolutions, 4 recurrent steps and BatchNorm re-normalisations the forward lands at ~1e-1 rms / ~1.3e-1 of the max magnitude
    at the worst pixel (measured; tests/test_gpu_stages.py lists the growth stage by stage: 3e-3 after the context stack, x1.3 per
    stage, 1.6e-1 at the output of the paper configuration); held to 1.5e-1 rms and 3e-1 max against the fp32 oracle, and one full training step must
    produce finite losses within 5 % of the `f32` mode's on the same seeds.  (The headline mode bf16x3 meets 1e-3, see above.)"""
    import skillful_nowcasting_amd as S
    from oracle import dgmr_oracle as O

    kw = dict(forecast_steps=4, output_shape=256, latent_channels=384, context_channels=192, generation_steps=6)
    torch.manual_seed(0)
    model = S.DGMR(**kw)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sd_cpu = {k[len("generator."):]: v.clone() for k, v in sd0.items() if k.startswith("generator.")}
    model = model.to("cuda")
    x, y = torch.rand(2, 4, 1, 256, 256), torch.rand(2, 4, 1, 256, 256)
    torch.manual_seed(1)
    z = O.draw_latent((8, 8, 8))
    ref = O.generator({k: v.clone() for k, v in sd_cpu.items()}, "", x, z, 4, True)
    losses = {}
    for precision in ("bf16", "f32"):
        model.load_state_dict(sd0)
        S.ops.bump_weights_epoch()
        model._optimizers = model.configure_optimizers()[0]  # fresh Adam state for each mode
        S.set_precision(precision)
        try:
            torch.manual_seed(1)
            with torch.no_grad():
                out = model(x.cuda())
            if precision == "bf16":
                err = (out.cpu() - ref).abs().max().item() / ref.abs().max().item()
                rms = ((out.cpu() - ref).double().pow(2).mean().sqrt() / ref.double().pow(2).mean().sqrt()).item()
                print(f"\ncfg2 bf16 generator forward: max err / max |ref| = {err:.3e}, rms err / rms ref = {rms:.3e}")
                assert rms <= 1.5e-1 and err <= 3e-1, f"cfg2 bf16: generator forward max-rel {err:.3e}, rms-rel {rms:.3e}"
            model.load_state_dict(sd0)
            S.ops.bump_weights_epoch()
            torch.manual_seed(2)
            o = model.training_step((x.cuda(), y.cuda()), 0)
            torch.cuda.synchronize()
        finally:
            S.set_precision("f32")
        losses[precision] = [float(o["d_loss"]), float(o["g_loss"]), float(o["grid_loss"])]
        assert all(v == v and abs(v) != float("inf") for v in losses[precision]), losses
    for a, b in zip(losses["bf16"][1:], losses["f32"][1:]):
        assert abs(a - b) <= 5e-2 * abs(b), losses


def test_bf16x3_tracks_exact_f32_over_three_steps_paper_config():
    """Three consecutive training steps at the PAPER configuration (B = 2), bf16x3 against the exact-f32 kernels (which the tests above
    hold to the float64 oracle at 0.7 ... 1.7 x torch-CPU's own fp32 error) from the same state, seeds and batch: every loss of every
    step within 1e-3 - the multi-step evidence at the benchmarked shapes that the CPU reference is too slow to provide (90 s per step
    at B = 1).  State is NOT compared across the modes beyond the losses: the trajectory itself is chaotic.  Adam's first steps move
    every element by +-lr whatever its gradient's size, so elements whose gradient is rounding noise flip; measured with
    tools/mode_buffers_probe.py: a 4e-6 perturbation in four layers (upsampling convs as phase sums vs as written, same arithmetic
    mode) shows as 1.5e-3 in BatchNorm running statistics / u, v after ONE step and 1e-1 after three, exactly what f32 vs bf16x3
    shows (8e-3 / 2e-1), while f32 vs f32 is bit-identical (the step is deterministic).  Per-step state parity against the reference
    is what tests/test_training_step.py and tests/test_training_steps_adv.py pin, with the reference's own run-to-run band."""
    import skillful_nowcasting_amd as S

    torch.manual_seed(0)
    model = S.DGMR(**KW)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to("cuda")
    torch.manual_seed(5)
    x, y = torch.rand(2, 4, 1, 256, 256).cuda(), torch.rand(2, 18, 1, 256, 256).cuda()
    runs = {}
    for precision in ("f32", "bf16x3"):
        model.load_state_dict(sd0)
        S.ops.bump_weights_epoch()
        model.train()
        model._optimizers = model.configure_optimizers()[0]
        S.set_precision(precision)
        try:
            torch.manual_seed(9)
            losses = []
            for i in range(3):
                o = model.training_step((x, y), i)
                losses.append([float(o["d_loss"]), float(o["g_loss"]), float(o["grid_loss"])])
            torch.cuda.synchronize()
        finally:
            S.set_precision("f32")
        runs[precision] = losses
    l32, l3 = runs["f32"], runs["bf16x3"]
    print("\nlosses per step (d, g, grid)  f32:", l32, " bf16x3:", l3)
    for s32, s3 in zip(l32, l3):
        for a, b in zip(s32, s3):
            assert a == a and b == b
            assert abs(a - b) <= 1e-3 * max(abs(a), 1e-6) + 1e-6, (l32, l3)
