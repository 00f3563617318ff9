"""Parity AT THE BENCHMARKED BATCH (paper config, per-GPU batch 16 - BASELINE.json configs[2] / [3], what bench.py times).

The oracle-anchored full-size tests (test_gpu_fullsize.py, test_gpu_fullstep.py) run B = 1 / 2: the float64 oracle of a whole step
takes minutes per sample.  The library's dispatch depends on the launch geometry (conv.hip: 256-pixel tiles from 2048 workgroups,
split-K below 192, the `small(<1024wg)` classes, implicit GEMM below the window kernels' threshold), so part of the kernel
variants behind the bench number never met the oracle at those sizes.  This module closes that gap three ways
(reference path: dgmr/dgmr.py:137-218 at the sizes of tests/test_model.py:227-259):

  1. train-mode generator forward of ONE draw at B = 16 and the discriminator forward on its 32 sequences against the CPU oracle in
     float32 AND float64 (conftest.band_check; the oracle's no-grad forward is ~2 s per sample);
  2. launch-geometry invariance: in eval mode (frozen sigma, running statistics: samples are independent) forward + backward at
     B = 16 with the step's own batching - six generator draws as one batch, six discriminator calls as one batch - against the
     SAME 16 samples run as 8 x B = 2, the geometry the oracle has checked (outputs, every parameter gradient, the gradient towards
     the generated frames);
  3. the kernel-variant set of one bench step (dgmr_profile_collect_detail rows: tile, mode, launch-size class, arithmetic) must be a
     subset of the variants parts 1 and 2 exercised.
"""
import contextlib
import ctypes

import pytest
import torch
import torch.nn.functional as F

from conftest import _log_band
from conftest import band_check as _band_check

pytestmark = pytest.mark.gpu

KW = dict(forecast_steps=18, output_shape=256, latent_channels=768, context_channels=384, generation_steps=6)
B = 16
COVERED = {}  # arithmetic mode -> kernel-variant names the parity tests of this module launched


@contextlib.contextmanager
def _record_variants(mode):
    """Collect the per-instantiated-kernel rows of everything launched inside the block into COVERED[mode]."""
    from skillful_nowcasting_amd import _lib

    lib = _lib.load()
    _drain(lib)
    lib.dgmr_profile_enable(1)
    try:
        yield
    finally:
        torch.cuda.synchronize()
        lib.dgmr_profile_enable(0)
        COVERED.setdefault(mode, set()).update(_drain(lib))


def _drain(lib):
    need = lib.dgmr_profile_collect_detail(None, 0)
    buf = ctypes.create_string_buffer(need + 16)
    lib.dgmr_profile_collect_detail(buf, need + 16)
    names = {line.split("\t")[0] for line in buf.value.decode().splitlines() if line}
    nv = lib.dgmr_profile_variants()
    a, b, c = (ctypes.c_double * nv)(), (ctypes.c_double * nv)(), (ctypes.c_int64 * nv)()
    lib.dgmr_profile_collect(a, b, c, nv)  # clears the records
    return names


def _distinct_sequences(n, seed):
    """n DISTINCT 22-frame sequences (per-sample amplitude, smooth structure + noise), as a batch of real and generated radar is -
    iid uniform noise makes the heads' BatchNorm1d degenerate (test_gpu_fullsize._d_inputs)."""
    g = torch.Generator().manual_seed(seed)
    amp = torch.linspace(0.15, 2.5, n).view(n, 1, 1, 1, 1)
    low = F.interpolate(torch.rand(n * 22, 1, 8, 8, generator=g), size=(256, 256), mode="bilinear", align_corners=False).view(n, 22, 1, 256, 256)
    return amp * (0.5 * torch.rand(n, 22, 1, 256, 256, generator=g) + low * torch.rand(n, 1, 1, 1, 1, generator=g) * 2)


@pytest.fixture(scope="module")
def setup():
    import skillful_nowcasting_amd as S

    torch.manual_seed(0)
    model = S.DGMR(**KW)
    sd_cpu = {k: v.detach().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(16)
    x = torch.rand(B, 4, 1, 256, 256, generator=g)
    y = torch.rand(B, 18, 1, 256, 256, generator=g)
    return model.to("cuda"), sd_cpu, x, y


def _reset(model, sd_cpu, train):
    import skillful_nowcasting_amd as S

    model.load_state_dict(sd_cpu)
    S.ops.bump_weights_epoch()
    model.train(train)
    for p in model.parameters():
        p.grad = None
        p.requires_grad_(True)


# ------------------------------------------------------------------------------------------------------------------------------
# 1. against the oracle, train mode, B = 16
# ------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def oracle_forward_b16(setup):
    """Oracle: train-mode generator forward of one draw on the 16 samples (BatchNorm statistics over the batch, one power iteration per
    call), float32 and float64; then the discriminator on [16 real, 16 distinct 'generated'] sequences."""
    from oracle import dgmr_oracle as O

    torch.set_num_threads(min(16, torch.get_num_threads()))
    _, sd_cpu, x, _ = setup
    seq = _distinct_sequences(2 * B, 32)
    res = {}
    for dt in (torch.float32, torch.float64):
        sd = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd_cpu.items()}
        torch.manual_seed(1)
        z = O.draw_latent((8, 8, 8)).to(dt)
        with torch.no_grad():
            out = O.generator(sd, "", x.to(dt), z, 18, True)
            torch.manual_seed(3)
            idxs = torch.randint(0, 22, (8,)).tolist()
            scores = O.discriminator(sd, "discriminator.", seq.to(dt), idxs, True)
        res[dt] = (out, scores, {k: sd[k] for k in sd if k.endswith(("._u", "._v", "running_mean", "running_var"))})
    return res, seq


@pytest.mark.parametrize("precision", ["f32", "mixed"])
def test_train_forward_b16_against_the_oracle(setup, oracle_forward_b16, precision):
    """The forward launches of the bench step's discriminator passes and logging forward (one generator draw at B = 16, D on 32
    sequences), train mode, against the float32 / float64 oracle: north-star bound 1e-3."""
    import skillful_nowcasting_amd as S

    model, sd_cpu, x, _ = setup
    ref, seq = oracle_forward_b16
    (o32, s32, b32), (o64, s64, b64) = ref[torch.float32], ref[torch.float64]
    _reset(model, sd_cpu, True)
    S.set_precision(precision)
    try:
        with _record_variants(precision), torch.no_grad():
            torch.manual_seed(1)
            out = model(x.cuda())
            torch.manual_seed(3)
            scores = model.discriminator(seq.cuda())
            torch.cuda.synchronize()
    finally:
        S.set_precision("f32")
    assert out.shape == (B, 18, 1, 256, 256) and scores.shape == (2 * B, 2, 1)
    rows = {"generator output [16, 18, 1, 256, 256]": (out.cpu(), o32, o64), "discriminator scores [32, 2, 1]": (scores.cpu(), s32, s64)}
    sd1 = model.state_dict()
    for k in ("sampler.up_g4.first_conv_3x3.parametrizations.weight.0._u", "sampler.convGRU1.cell.read_gate_conv.parametrizations.weight.0._v",
              "sampler.g2.bn1.running_var", "sampler.bn.running_mean", "conditioning_stack.d1.first_conv_3x3.parametrizations.weight.0._u",
              "discriminator.temporal_discriminator.d1.last_conv_3x3.parametrizations.weight.0._u",
              "discriminator.spatial_discriminator.bn.running_var"):
        rows["state " + k] = (sd1[k].detach().cpu().float(), b32[k], b64[k])
    _band_check(f"train-mode forward at the benchmarked batch B = {B}", precision, 1e-3, rows)


# ------------------------------------------------------------------------------------------------------------------------------
# 2. launch-geometry invariance: B = 16 in the step's own batching == 8 x B = 2 (eval mode: samples are independent)
# ------------------------------------------------------------------------------------------------------------------------------
def _grads(module):
    return {k: p.grad.detach().clone() for k, p in module.named_parameters() if p.grad is not None}


def _dist(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    e_max = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-300)
    e_l2 = ((a - b).norm() / b.norm().clamp_min(1e-300)).item()
    cos = 1.0 if b.numel() < 2 else torch.nn.functional.cosine_similarity(a, b, dim=0).item()
    return e_l2, e_max, 1.0 - cos


def _compare(what, precision, tol, big, small, other):
    """big / small / other: {name: tensor} of the SAME function of the SAME numbers, evaluated with all 16 samples in one batch, as 8
    batches of 2 (the geometry the float64 oracle has checked) and as 4 batches of 4.  What differs is the fp32 summation order (other
    tile variants, split-K, the order in which the samples' contributions meet in a weight gradient) - and, through it, a few relu
    masks: an activation within rounding of zero falls on the other side of the kink, its whole gradient toggles, and a fraction f of
    toggled elements moves a gradient by ~sqrt(f) in relative L2 (forward outputs agree to 3e-7 in f32, gradients to 1e-3: f ~ 1e-6).
    That band is MEASURED in the same run instead of assumed: the distance between the two small geometries.  The big batch must be
    within max(tol, 3 x that distance) of the 8 x 2 result, in relative L2 and in max-abs error over the largest magnitude, with
    1 - cosine within max(1e-9, 10 x): a misaddressed tile or a dropped sample is O(1) in all three and in no band.  (A gradient's own
    band is a single draw of that noise; it is floored by the median over the run's gradient tensors.)"""
    lines, bad = [], []
    noise = {k: _dist(other[k], small[k]) for k in small}
    # (a tensor's own band is one draw of that noise: it is floored by the median band of the gradient tensors of the run)
    gk = [k for k in small if k.startswith("grad ")]
    med = [sorted(noise[k][i] for k in gk)[len(gk) // 2] for i in range(3)]
    # One flipped relu mask in a head block (2 x 2 maps, 786 k activations: about one pair of geometries in six has one) moves that
    # block's weight gradient by ~1e-3 in L2 and 5e-3 at its worst element while the two small geometries happen to agree to 4e-7: the
    # measured band of a gradient is additionally floored at 200 x the forward tolerance (2e-3 in f32 - the largest distance between
    # the two SMALL geometries measured on any gradient tensor is 1.3e-3 - and 2e-2 where 16-bit products are involved)
    kink = 200.0 * tol
    for k in small:
        e = _dist(big[k], small[k])
        if k.startswith("grad "):
            n = tuple(max(noise[k][i], med[i]) for i in range(3))
            ok = e[0] <= max(kink, 3.0 * n[0]) and e[1] <= max(10.0 * kink, 3.0 * n[1]) and e[2] <= max(kink * kink, 10.0 * n[2])
        else:
            n = noise[k]
            ok = e[0] <= max(tol, 3.0 * n[0]) and e[1] <= max(10.0 * tol, 3.0 * n[1]) and e[2] <= max(1e-9, 10.0 * n[2])
        lines.append((e[0], e[1], e[2], n[0], n[1], k))
        if not ok:
            bad.append(f"{k}: l2 {e[0]:.2e} (band {n[0]:.2e}) max {e[1]:.2e} (band {n[1]:.2e}) 1-cos {e[2]:.1e} (band {n[2]:.1e})")
    lines.sort(reverse=True)
    fmt = lambda r: f"  {r[5]:100s} l2 {r[0]:.2e} (4x4 vs 8x2: {r[3]:.2e})  max {r[1]:.2e} ({r[4]:.2e})  1-cos {r[2]:.1e}"  # noqa: E731
    msg = (f"{what} [{precision}]: 16 samples in one batch vs the same samples as 8 x B = 2, {len(small)} tensors; bound max({tol:.0e}, 3 x the "
           f"distance between 4 x B = 4 and 8 x B = 2); largest:\n" + "\n".join(fmt(r) for r in lines[:8])
           + "\n  not gradients:\n" + "\n".join(fmt(r) for r in lines if not r[5].startswith("grad ")))
    print("\n" + msg)
    _log_band(msg)
    assert not bad, f"beyond the band: {bad[:10]}\n{msg}"


@pytest.mark.parametrize("precision,tol", [("f32", 1e-5), ("mixed", 1e-4)])
def test_generator_draws_b16_equal_eight_batches_of_two(setup, precision, tol):
    """Generator pass geometry: six draws of 16 samples as ONE batch (96 samples per forecast step, 1728 through the T-batched
    layers: 256-pixel tiles, no split-K, `big` classes) forward + backward, against 8 x (six draws of 2 samples) - the geometry of the
    oracle-anchored full-step test.  Eval mode: same sigma, same running statistics, same latents on both sides, so only the launch
    geometry differs.  Also the two-draw and one-draw forwards of the discriminator passes / the logging forward."""
    import skillful_nowcasting_amd as S

    model, sd_cpu, x, y = setup
    k = 6
    xd = x.cuda()
    cot = (torch.randn(k, B, 18, 1, 256, 256, generator=torch.Generator().manual_seed(5)) / (k * B)).cuda()

    def run(chunk):
        for p in model.parameters():
            p.grad = None
        outs, twos, ones = [], [], []
        for c in range(B // chunk):
            sl = slice(chunk * c, chunk * (c + 1))
            torch.manual_seed(21)
            out = model.generator.forward_draws(xd[sl], k)
            out.backward(cot[:, sl].reshape(k * chunk, 18, 1, 256, 256))  # (parameter gradients accumulate over the chunks)
            outs.append(out.detach().view(k, chunk, 18, 1, 256, 256))
            del out
            with torch.no_grad():
                torch.manual_seed(22)
                twos.append(model.generator.forward_draws(xd[sl], 2).view(2, chunk, 18, 1, 256, 256))
                ones.append(model.generator.forward_draws(xd[sl], 1))
        torch.cuda.synchronize()
        res = {"six draws: output": torch.cat(outs, dim=1), "two draws: output": torch.cat(twos, dim=1), "one draw: output": torch.cat(ones, dim=0)}
        res.update({"grad " + n: g for n, g in _grads(model.generator).items()})
        return res

    S.set_precision(precision)
    try:
        _reset(model, sd_cpu, False)
        with _record_variants(precision):
            big = run(B)
        small, other = run(2), run(4)
    finally:
        S.set_precision("f32")
    assert set(big) == set(small) == set(other) and len(big) >= 150
    _compare("generator, six draws forward + backward", precision, tol, big, small, other)


@pytest.mark.parametrize("precision,tol", [("f32", 1e-5), ("mixed", 1e-4)])
def test_discriminator_calls_b16_equal_eight_batches_of_two(setup, precision, tol):
    """Discriminator geometry of the step: ONE call on 32 sequences (discriminator passes: forward + backward with weight gradients) and
    SIX calls as one batch of 192 (generator pass: forward + data gradient only), against the same sequences as 8 chunks of B = 2.
    Eval mode (BatchNorm1d running statistics), the same random frame draw per call on both sides."""
    import skillful_nowcasting_amd as S

    model, sd_cpu, _, _ = setup
    k = 6
    real = _distinct_sequences(B, 41).cuda()
    gen = _distinct_sequences(k * B, 42).cuda().view(k, B, 22, 1, 256, 256)
    cot1 = torch.randn(2, B, 2, 1, generator=torch.Generator().manual_seed(6)).cuda()
    cot6 = torch.randn(k, 2, B, 2, 1, generator=torch.Generator().manual_seed(7)).cuda()

    def run(chunk):
        dparams = list(model.discriminator.parameters())
        for p in dparams:
            p.grad = None
            p.requires_grad_(True)
        s1, gx1, s6, gx6 = [], [], [], []
        for c in range(B // chunk):  # one call on cat(real, generated): scores, weight gradients (accumulating), d / d generated
            sl = slice(chunk * c, chunk * (c + 1))
            gn = gen[0, sl].detach().clone().requires_grad_(True)
            torch.manual_seed(3)
            s = model.discriminator(torch.cat([real[sl], gn], dim=0))
            (s.view(2, chunk, 2, 1) * cot1[:, sl]).sum().backward()
            s1.append(s.detach().view(2, chunk, 2, 1))
            gx1.append(gn.grad)
        grads = _grads(model.discriminator)
        for p in dparams:
            p.grad = None
            p.requires_grad_(False)
        for c in range(B // chunk):  # [K, (real, generated), b] as one batch of K calls (dgmr.py: _gen_losses), parameters frozen
            sl = slice(chunk * c, chunk * (c + 1))
            gn = gen[:, sl].detach().clone().requires_grad_(True)
            rl = real[sl]
            inp = torch.cat([rl.unsqueeze(0).expand(k, *rl.shape).unsqueeze(1), gn.unsqueeze(1)], dim=1)
            torch.manual_seed(4)
            s = model.discriminator(inp.reshape(2 * k * chunk, 22, 1, 256, 256), calls=k)
            (s.view(k, 2, chunk, 2, 1) * cot6[:, :, sl]).sum().backward()
            s6.append(s.detach().view(k, 2, chunk, 2, 1))
            gx6.append(gn.grad)
        torch.cuda.synchronize()
        for p in dparams:
            p.requires_grad_(True)
        res = {"one call: scores": torch.cat(s1, dim=1), "one call: d / d generated frames": torch.cat(gx1, dim=0),
               "six calls: scores": torch.cat(s6, dim=2), "six calls: d / d generated frames": torch.cat(gx6, dim=1)}
        res.update({"grad " + n: g for n, g in grads.items()})
        return res

    S.set_precision(precision)
    try:
        _reset(model, sd_cpu, False)
        with _record_variants(precision):
            big = run(B)
        small, other = run(2), run(4)
    finally:
        for p in model.discriminator.parameters():
            p.requires_grad_(True)
        S.set_precision("f32")
    assert set(big) == set(small) == set(other) and len(big) >= 60
    _compare("discriminator, one call of 32 and six calls of 32", precision, tol, big, small, other)


@pytest.mark.parametrize("precision,tol", [("f32", 1e-4), ("mixed", 1e-3)])
def test_train_mode_batched_draws_equal_sequential_forwards_b16(setup, precision, tol):
    """TRAIN mode at B = 16 (per-call sigma, BatchNorm batch statistics; the context stack then runs once per draw too):
    `forward_draws(x, 6)` - the generator pass's batch - and the two-draw replay of the discriminator passes against six / two
    consecutive `forward(x)` calls, whose geometry part 1 has checked against the oracle: outputs and every buffer afterwards
    (test_gpu_fullsize.py does this at B = 1 with three draws; same bounds)."""
    import skillful_nowcasting_amd as S

    model, sd_cpu, x, _ = setup
    xd = x.cuda()
    S.set_precision(precision)
    try:
        for k, shared_z in ((6, False), (2, True)):
            _reset(model, sd_cpu, True)
            torch.manual_seed(21)
            zs = torch.cat([model.latent_stack.draw(xd) for _ in range(k)], dim=0)
            if shared_z:
                zs = torch.cat([zs[:1]] * k, dim=0)  # (the replay: both draws share one latent, dgmr.py _training_step)
            with torch.no_grad():
                seq = torch.cat([model.generator.forward_draws(xd, 1, zs=zs[i:i + 1]) for i in range(k)], dim=0)
            sd_seq = {n: v.detach().clone() for n, v in model.state_dict().items()}
            _reset(model, sd_cpu, True)
            with _record_variants(precision), torch.no_grad():
                bat = model.generator.forward_draws(xd, k, zs=zs)
                torch.cuda.synchronize()
            e = (bat - seq).abs().max().item() / seq.abs().max().item()
            msg = f"train mode, B = {B}: forward_draws(x, {k}) vs {k} consecutive forwards [{precision}]: output {e:.2e} (bound {tol:.0e})"
            worst = (0.0, "")
            for n, v in model.state_dict().items():
                if n.endswith(("._u", "._v", "running_mean", "running_var")):
                    a, b = v.float(), sd_seq[n].float()
                    eb = (a - b).abs().max().item() / (b.abs().max().item() + 1e-7)
                    worst = max(worst, (eb, n))
            msg += f"; worst buffer {worst[0]:.2e} ({worst[1]})"
            print("\n" + msg)
            _log_band(msg)
            assert bat.shape == seq.shape and e <= tol, msg
            assert worst[0] <= tol, msg
    finally:
        S.set_precision("f32")


# ------------------------------------------------------------------------------------------------------------------------------
# 3. the bench step launches no conv kernel variant that parts 1 and 2 have not exercised
# ------------------------------------------------------------------------------------------------------------------------------
def test_bench_step_kernel_variants_are_covered_by_the_parity_tests(setup):
    """One training step exactly as bench.py times it (paper config, B = 16, `mixed`, strict reference semantics) with the library's
    per-launch records on: every (kernel, tile, mode, launch-size class, arithmetic) row it produces must also have been produced
    by the `mixed` runs of the tests above (weight-gradient rows included: the eval-mode backward of part 2 launches the same
    weight-gradient kernels on the same shapes)."""
    import skillful_nowcasting_amd as S
    from skillful_nowcasting_amd import _lib

    if "mixed" not in COVERED:
        pytest.skip("run the whole module: the `mixed` parity tests above collect the covered variant set")
    model, sd_cpu, x, y = setup
    _reset(model, sd_cpu, True)
    lib = _lib.load()
    S.set_precision("mixed")
    try:
        batch = (x.cuda(), y.cuda())
        model.training_step(batch, 0)  # (warm-up: the spectral-norm plans are traced in a first step)
        torch.cuda.synchronize()
        _drain(lib)
        lib.dgmr_profile_enable(1)
        model.training_step(batch, 1)
        torch.cuda.synchronize()
        lib.dgmr_profile_enable(0)
        step = _drain(lib)
    finally:
        lib.dgmr_profile_enable(0)
        S.set_precision("f32")
    missing = sorted(step - COVERED["mixed"])
    msg = (f"kernel variants of one bench step (B = {B}, mixed): {len(step)}; exercised by the B = {B} parity tests: {len(COVERED['mixed'])}; "
           f"in the step but not in the tests: {missing or 'none'}")
    print("\n" + msg)
    _log_band(msg)
    assert len(step) >= 40, step
    assert not missing, msg
