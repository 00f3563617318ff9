"""The reference's own acceptance suite - tests/test_model.py:29-306 of openclimatefix/skillful_nowcasting - run against this package
through the `dgmr` import alias, test for test: construct -> torch.rand -> forward -> mse_loss -> backward -> shapes / no NaNs.
The only change is `.cuda()` on modules and inputs (the kernels are HIP-only; a CPU tensor raises, tested at the end).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _alias():
    import skillful_nowcasting_amd as S

    S.install_as("dgmr")


def _imports():
    from dgmr import (DGMR, ContextConditioningStack, Discriminator, Generator, LatentConditioningStack, Sampler,
                      SpatialDiscriminator, TemporalDiscriminator)
    from dgmr.common import DBlock, GBlock
    from dgmr.layers import ConvGRU
    from dgmr.layers.ConvGRU import ConvGRUCell

    return dict(DGMR=DGMR, Generator=Generator, Discriminator=Discriminator, TemporalDiscriminator=TemporalDiscriminator,
                SpatialDiscriminator=SpatialDiscriminator, Sampler=Sampler, LatentConditioningStack=LatentConditioningStack,
                ContextConditioningStack=ContextConditioningStack, ConvGRU=ConvGRU, ConvGRUCell=ConvGRUCell, DBlock=DBlock, GBlock=GBlock)


def _rand(*shape):
    return torch.rand(shape, device=DEV)


def test_dblock():
    model = _imports()["DBlock"](keep_same_output=True).to(DEV)
    out = model(_rand(2, 12, 128, 128))
    F.mse_loss(_rand(2, 12, 128, 128), out).backward()
    assert out.size() == (2, 12, 128, 128)
    assert not torch.isnan(out).any(), "Output included NaNs"


def test_gblock():
    model = _imports()["GBlock"]().to(DEV)
    out = model(_rand(2, 12, 128, 128))
    F.mse_loss(_rand(2, 12, 128, 128), out).backward()
    assert out.size() == (2, 12, 128, 128)
    assert not torch.isnan(out).any(), "Output included NaNs"


def test_conv_gru_cell():
    model = _imports()["ConvGRUCell"](input_channels=768 + 384, output_channels=384, kernel_size=3).to(DEV)
    out, hidden = model(_rand(2, 768, 32, 32), _rand(2, 384, 32, 32))
    F.mse_loss(_rand(2, 384, 32, 32), out).backward()
    assert out.size() == (2, 384, 32, 32)
    assert not torch.isnan(out).any(), "Output included NaNs"


def test_conv_gru():
    model = _imports()["ConvGRU"](input_channels=768 + 384, output_channels=384, kernel_size=3).to(DEV)
    init_states = [_rand(2, 384, 32, 32) for _ in range(4)]
    x = _rand(2, 768, 32, 32)
    hidden_states = [x] * 18
    out = model(hidden_states, init_states[3])
    F.mse_loss(_rand(18, 2, 384, 32, 32), out).backward()
    assert out.size() == (18, 2, 384, 32, 32)
    assert not torch.isnan(out).any(), "Output included NaNs"


def test_latent_conditioning_stack():
    model = _imports()["LatentConditioningStack"]().to(DEV)
    out = model(_rand(2, 4, 1, 128, 128))
    assert out.size() == (1, 768, 8, 8)
    F.mse_loss(_rand(1, 768, 8, 8), out).backward()
    assert not torch.isnan(out).any(), "Output included NaNs"


def test_context_conditioning_stack():
    model = _imports()["ContextConditioningStack"]().to(DEV)
    out = model(_rand(2, 4, 1, 128, 128))
    F.mse_loss(_rand(2, 96, 32, 32), out[0]).backward()
    assert len(out) == 4
    assert out[0].size() == (2, 96, 32, 32)
    assert out[1].size() == (2, 192, 16, 16)
    assert out[2].size() == (2, 384, 8, 8)
    assert out[3].size() == (2, 768, 4, 4)
    assert not any(torch.isnan(out[i]).any() for i in range(len(out))), "Output included NaNs"


def test_temporal_discriminator():
    model = _imports()["TemporalDiscriminator"](input_channels=1).to(DEV)
    out = model(_rand(2, 8, 1, 256, 256))
    assert out.shape == (2, 1, 1)
    F.mse_loss(_rand(2, 1, 1), out).backward()
    assert not torch.isnan(out).any()


def test_spatial_discriminator():
    model = _imports()["SpatialDiscriminator"](input_channels=1).to(DEV)
    out = model(_rand(2, 18, 1, 128, 128))
    assert out.shape == (2, 1, 1)
    F.mse_loss(_rand(2, 1, 1), out).backward()
    assert not torch.isnan(out).any()


def test_discriminator():
    model = _imports()["Discriminator"](input_channels=1).to(DEV)
    out = model(_rand(2, 18, 1, 256, 256))
    assert out.shape == (2, 2, 1)
    F.mse_loss(_rand(2, 2, 1), out).backward()
    assert not torch.isnan(out).any()


def _parts():
    m = _imports()
    conditioning_stack = m["ContextConditioningStack"](input_channels=1, conv_type="standard", output_channels=384)
    latent_stack = m["LatentConditioningStack"](shape=(8, 256 // 32, 256 // 32), output_channels=768)
    sampler = m["Sampler"](forecast_steps=18, latent_channels=768, context_channels=384)
    return m, conditioning_stack.to(DEV), latent_stack.to(DEV), sampler.to(DEV)


def test_sampler():
    """tests/test_model.py:134-224: every sampler sub-module called on its own, the way the reference's forward does."""
    import einops

    _, conditioning_stack, latent_stack, sampler = _parts()
    latent_stack.eval()
    conditioning_stack.eval()
    sampler.eval()
    forecast_steps = 18
    x = _rand(2, 4, 1, 256, 256)

    def ok(hs):
        assert not any(torch.isnan(h).any() for h in hs)

    with torch.no_grad():
        latent_dim = latent_stack(x)
        assert not torch.isnan(latent_dim).any()
        init_states = conditioning_stack(x)
        ok(init_states)
        latent_dim = einops.repeat(latent_dim, "b c h w -> (repeat b) c h w", repeat=init_states[0].shape[0])
        hidden_states = [latent_dim] * forecast_steps
        for lvl, (gru, c11, g, upg) in enumerate((("convGRU1", "gru_conv_1x1", "g1", "up_g1"), ("convGRU2", "gru_conv_1x1_2", "g2", "up_g2"),
                                                  ("convGRU3", "gru_conv_1x1_3", "g3", "up_g3"), ("convGRU4", "gru_conv_1x1_4", "g4", "up_g4"))):
            hidden_states = getattr(sampler, gru)(hidden_states, init_states[3 - lvl])
            ok(hidden_states)
            hidden_states = [getattr(sampler, c11)(h) for h in hidden_states]
            ok(hidden_states)
            hidden_states = [getattr(sampler, g)(h) for h in hidden_states]
            ok(hidden_states)
            hidden_states = [getattr(sampler, upg)(h) for h in hidden_states]
            ok(hidden_states)
        hidden_states = [F.relu(sampler.bn(h)) for h in hidden_states]
        ok(hidden_states)
        hidden_states = [sampler.conv_1x1(h) for h in hidden_states]
        ok(hidden_states)
        hidden_states = [sampler.depth2space(h) for h in hidden_states]
        ok(hidden_states)
        assert hidden_states[0].shape == (2, 1, 256, 256)
        # and the module-by-module walk agrees with the fused forward on the same state (eval mode: no state moves)
        fused = sampler(list(init_states), latent_dim[:1])
        assert (fused - torch.stack(hidden_states, dim=1)).abs().max().item() <= 1e-4 * fused.abs().max().item()


def test_generator():
    m, conditioning_stack, latent_stack, sampler = _parts()
    model = m["Generator"](conditioning_stack=conditioning_stack, latent_stack=latent_stack, sampler=sampler)
    out = model(_rand(2, 4, 1, 256, 256))
    assert out.shape == (2, 18, 1, 256, 256)
    F.mse_loss(_rand(2, 18, 1, 256, 256), out).backward()
    assert not torch.isnan(out).any()


def test_nowcasting_gan_creation():
    model = _imports()["DGMR"](forecast_steps=18, input_channels=1, output_shape=128, latent_channels=768, context_channels=384,
                               num_samples=3).to(DEV)
    model.eval()
    with torch.no_grad():
        out = model(_rand(2, 4, 1, 128, 128))
    assert out.size() == (2, 18, 1, 128, 128)
    assert not torch.isnan(out).any(), "Output included NaNs"


def test_nowcasting_gan_backward():
    """tests/test_model.py:285-306 = BASELINE.json configs[0]."""
    model = _imports()["DGMR"](forecast_steps=4, input_channels=1, output_shape=128, latent_channels=384, context_channels=192,
                               num_samples=3).to(DEV)
    out = model(_rand(2, 4, 1, 128, 128))
    assert out.size() == (2, 4, 1, 128, 128)
    F.mse_loss(_rand(2, 4, 1, 128, 128), out).backward()
    assert not torch.isnan(out).any(), "Output included NaNs"
    grads = [p.grad for p in model.generator.parameters() if p.grad is not None]
    assert len(grads) > 150 and all(torch.isfinite(g).all() for g in grads)


def test_cpu_tensors_are_refused_loudly():
    model = _imports()["DBlock"](keep_same_output=True).to(DEV)
    with pytest.raises(RuntimeError, match="HIP-only"):
        model(torch.rand(2, 12, 16, 16))
