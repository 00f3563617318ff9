"""Deterministic mode (dgmr_set_deterministic, the default): every cross-workgroup sum of the step is formed in a fixed order - the
bias gradients (rows per slab instead of float atomics), <P_q, W> of the spectral-norm chain rule, per-channel statistics, the grid
cell loss, the frame-gather of the spatial discriminator's backward - so two identical runs must agree BIT FOR BIT in every parameter,
every buffer and every optimiser moment after several steps, in exact f32 and in the bench's `mixed` arithmetic.  With the mode off
the same comparison fails (float atomics meet in arrival order): asserted too, so that the test cannot pass vacuously.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

KW = dict(forecast_steps=2, output_shape=128, latent_channels=384, context_channels=192, generation_steps=2)


def _run(S, precision, steps, kw=KW, batch=2):
    S.set_precision(precision)
    try:
        torch.manual_seed(7)
        model = S.DGMR(**kw).to("cuda")
        torch.manual_seed(8)
        hw = kw["output_shape"]
        x = torch.rand(batch, 4, 1, hw, hw, device="cuda")
        y = torch.rand(batch, kw["forecast_steps"], 1, hw, hw, device="cuda")
        torch.manual_seed(9)
        losses = []
        for i in range(steps):
            out = model.training_step((x, y), i)
            losses.append(tuple(float(out[k]) for k in ("d_loss", "g_loss", "grid_loss")))
        torch.cuda.synchronize()
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        for oi, opt in enumerate(model.optimizers()):
            for pi, p in enumerate(opt.param_groups[0]["params"]):
                st = opt.state.get(p)
                if st:
                    state[f"opt{oi}.{pi}.exp_avg_sq"] = st["exp_avg_sq"].detach().clone()
        return state, losses
    finally:
        S.set_precision("f32")


def _differing(a, b):
    return [k for k in a if not torch.equal(a[k], b[k])]


@pytest.mark.parametrize("precision", ["f32", "mixed"])
def test_two_identical_runs_are_bit_identical(precision):
    import skillful_nowcasting_amd as S

    assert S.deterministic(), "deterministic mode is the default (DGMR_DETERMINISTIC=0 switches it off)"
    a, la = _run(S, precision, 3)
    b, lb = _run(S, precision, 3)
    assert la == lb, (la, lb)
    bad = _differing(a, b)
    assert not bad, f"{precision}: {len(bad)} of {len(a)} tensors differ between two identical runs, e.g. {bad[:5]}"


def test_bit_identity_at_the_paper_configuration():
    """One sample of the paper configuration (18 lead times, 256 x 256, six draws): the launch shapes the bench runs - wave-specialised
    weight gradients with 108 call groups, phase / pooled window launches, the 3-D blocks of the temporal discriminator."""
    import skillful_nowcasting_amd as S

    kw = dict(forecast_steps=18, output_shape=256, latent_channels=768, context_channels=384, generation_steps=6)
    a, la = _run(S, "mixed", 2, kw, batch=1)
    b, lb = _run(S, "mixed", 2, kw, batch=1)
    assert la == lb, (la, lb)
    bad = _differing(a, b)
    assert not bad, f"{len(bad)} of {len(a)} tensors differ, e.g. {bad[:5]}"


def test_the_mode_is_what_makes_the_runs_identical():
    import skillful_nowcasting_amd as S

    S.set_deterministic(False)
    try:
        differs = False
        ref, _ = _run(S, "mixed", 3)
        for _ in range(3):  # (arrival order usually differs in the first repeat already)
            other, _ = _run(S, "mixed", 3)
            if _differing(ref, other):
                differs = True
                break
    finally:
        S.set_deterministic(True)
    assert differs, "float-atomic sums gave identical results four times in a row: is the non-deterministic path still there?"


def test_nonfinite_count_and_detect_anomaly():
    """The stand-in for the reference's torch.autograd.set_detect_anomaly(True) (dgmr/dgmr.py:130): the counting kernel, and the opt-in
    check of DGMR.training_step that names the first parameter whose gradient went non-finite."""
    import skillful_nowcasting_amd as S
    from skillful_nowcasting_amd import ops

    x = torch.randn(1_000_003, device="cuda")
    x[17], x[999_999], x[500_000] = float("nan"), float("inf"), float("-inf")
    count = torch.zeros(1, device="cuda", dtype=torch.int32)
    ops.call("dgmr_nonfinite_count", x.data_ptr(), x.numel(), count.data_ptr(), ops._stream())
    ops.call("dgmr_nonfinite_count", x.data_ptr(), 400_000, count.data_ptr(), ops._stream())  # accumulates: + the NaN at 17
    assert int(count.item()) == 4

    torch.manual_seed(7)
    model = S.DGMR(**KW).to("cuda")
    model.detect_anomaly = True
    x = torch.rand(2, 4, 1, 128, 128, device="cuda")
    y = torch.rand(2, 2, 1, 128, 128, device="cuda")
    model.training_step((x, y), 0)  # a healthy step passes the check
    with torch.no_grad():
        model.discriminator.spatial_discriminator.d1.conv_1x1.bias[0] = float("nan")
    with pytest.raises(RuntimeError, match="detect_anomaly: non-finite"):
        model.training_step((x, y), 1)


def test_deferred_weight_gradients_change_nothing():
    """DGMR_WGRAD_DEFER (opt-in scheduling experiment: the generator pass's weight gradients are held back until the level's ConvGRU
    backward chain, _streams.defer_wgrads): the same kernels on the same operands, only issued later - bit-identical parameters,
    buffers and Adam moments after three steps."""
    import skillful_nowcasting_amd as S
    from skillful_nowcasting_amd import _streams

    ref, lref = _run(S, "mixed", 3)
    old = _streams._DEFER_ON
    _streams._DEFER_ON = True
    try:
        got, lgot = _run(S, "mixed", 3)
        assert not _streams._DEFERRED, "weight gradients left in the deferral queue after the step"
    finally:
        _streams._DEFER_ON = old
    assert lref == lgot, (lref, lgot)
    bad = _differing(ref, got)
    assert not bad, f"{len(bad)} tensors differ with deferred weight gradients, e.g. {bad[:5]}"
