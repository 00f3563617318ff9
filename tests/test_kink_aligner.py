"""conftest.KinkAligner, oracle side (CPU): running the oracle's discriminator with relu replaced by `x * mask` reproduces the plain
run exactly when the masks are the oracle's own signs laid out the way the HIP modules see them (the per-frame calls of a block
stacked frame-major along the batch axis), and follows a FOREIGN mask where one is planted at a pre-activation near zero."""
import torch

from conftest import KinkAligner


def _setup():
    import skillful_nowcasting_amd as S
    from oracle import dgmr_oracle as O

    torch.manual_seed(0)
    disc = S.Discriminator(input_channels=1)
    sd = {k: v.detach().clone().double() if v.is_floating_point() else v.clone() for k, v in disc.state_dict().items()}
    for k in O.param_keys(sd, ""):
        sd[k].requires_grad_(True)
    torch.manual_seed(1)
    x = torch.rand(2, 6, 1, 128, 128, dtype=torch.float64)
    idxs = [0, 3, 5, 1, 2, 4, 0, 3]
    return O, sd, x, idxs


def _run(O, sd, x, idxs):
    for k in sd:
        if sd[k].requires_grad:
            sd[k].grad = None
    sd_run = {k: (v if v.requires_grad else v.clone()) for k, v in sd.items()}  # buffers move in train mode: fresh copies per run
    out = O.discriminator(sd_run, "", x, idxs, True)
    out.sum().backward()
    return out.detach().clone(), {k: v.grad.clone() for k, v in sd.items() if v.requires_grad and v.grad is not None}


def test_own_masks_reproduce_the_plain_run_and_foreign_masks_are_followed():
    O, sd, x, idxs = _setup()
    out0, g0 = _run(O, sd, x, idxs)
    # record the oracle's own relu inputs per tag, in call order
    seen = {}
    O.RELU_HOOK = lambda t, tag: (seen.setdefault(tag, []).append(t.detach().clone()), torch.relu(t))[1]
    try:
        _run(O, sd, x, idxs)
    finally:
        O.RELU_HOOK = None
    assert any(tag.endswith(".head") for tag in seen) and any(tag.endswith(".mid") for tag in seen) and any(tag.endswith(".in") for tag in seen)
    assert len(seen["spatial_discriminator.d1.mid"]) == 8 and len(seen["temporal_discriminator.d1.mid"]) == 1
    ka = KinkAligner(None)
    ka.masks = {tag: torch.cat(ts, 0) > 0 for tag, ts in seen.items()}  # frame-major stacking = the HIP modules' batch layout
    with ka.oracle(O):
        out1, g1 = _run(O, sd, x, idxs)
    assert O.RELU_HOOK is None
    assert torch.equal(out0, out1) and all(torch.equal(g0[k], g1[k]) for k in g0)
    assert ka.report() == []
    # plant a foreign mask bit at the pre-activation closest to zero of a deep layer: the forward barely moves, the gradients do
    tag = "spatial_discriminator.d6.mid"
    pre = torch.cat(seen[tag], 0)
    flat = pre.abs().flatten()
    i = int(flat.argmin())
    ka.masks[tag].view(-1)[i] = not bool(ka.masks[tag].view(-1)[i])
    with ka.oracle(O):
        out2, g2 = _run(O, sd, x, idxs)
    rep = ka.report()
    assert len(rep) == 1 and rep[0][0] == tag and rep[0][2] == 1 and rep[0][3] <= float(flat[i] / flat.max()) * 2  # (per-call maximum)
    assert (out2 - out0).abs().max() <= 10 * flat[i].abs() * sd["spatial_discriminator.d6.last_conv_3x3.parametrizations.weight.original"].abs().max() * 1e3
    k = "spatial_discriminator.d6.first_conv_3x3.bias"
    assert not torch.equal(g2[k], g0[k]), "the planted mask bit did not reach the gradient"
