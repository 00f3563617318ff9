"""BASELINE.json's full-size configuration (paper config: 4 -> 18 frames, 256x256, latent 768 / context 384) on the GPU against
the CPU oracle run on the same box: whole-generator and whole-discriminator forward, north-star tolerance 1e-3 relative.

The goldens under tests/golden/ are small by necessity (they are committed); this test closes the gap to the real sizes, where
the T-batched launches, the LDS-window kernel, split-K and the multi-module spectral-norm plan all take their production paths.
"""
import pytest
import torch

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
