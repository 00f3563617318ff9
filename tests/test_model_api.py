"""The reference's API contract beyond the hot path, CPU only - no kernel runs:

  * Hugging Face `save_pretrained` / `from_pretrained` round trips of DGMR and its four publishable parts - the five tests of the
    reference's tests/test_model.py:341-399, through the `dgmr` import alias (skillful_nowcasting_amd.install_as);
  * interchange WITH the reference (build container only, where /root/reference exists): a state dict / safetensors directory /
    Lightning-style .ckpt written by the unmodified reference loads into this package and vice versa, key for key, bit for bit;
  * the data path (train/run.py:118-158) and the visualisation grids (dgmr/dgmr.py:302-327).
"""
import os
import sys

import numpy as np
import pytest
import torch
from torch.testing import assert_close

import skillful_nowcasting_amd as S

REF = "/root/reference"
needs_reference = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "dgmr")), reason="needs /root/reference (build container)")


@pytest.fixture(scope="module")
def dgmr():
    """`import dgmr` as the reference's tests do - resolved to this package."""
    mod = S.install_as("dgmr")
    import dgmr as alias
    from dgmr.common import DBlock, GBlock  # noqa: F401  (the import paths of tests/test_model.py:3-15)
    from dgmr.layers import ConvGRU  # noqa: F401
    from dgmr.layers.ConvGRU import ConvGRUCell  # noqa: F401

    assert alias is mod is S
    return alias


def assert_model_equal(actual, expected):  # tests/test_model.py:22-26
    assert actual.state_dict().keys() == expected.state_dict().keys()
    for x, y in zip(actual.state_dict().values(), expected.state_dict().values()):
        assert_close(x, y)


def test_model_serialization(tmp_path, dgmr):  # tests/test_model.py:341-362
    model = dgmr.DGMR(forecast_steps=1, input_channels=1, output_shape=128, gen_lr=1e-5, disc_lr=1e-4, visualize=True,
                      conv_type="standard", num_samples=1, grid_lambda=16.0, beta1=1.0, beta2=0.995, latent_channels=512,
                      context_channels=256, generation_steps=1, precip_weight_cap=12)
    model.save_pretrained(tmp_path / "dgmr")
    model_copy = dgmr.DGMR.from_pretrained(tmp_path / "dgmr")
    assert model.hparams == model_copy.hparams
    assert model_copy.grid_lambda == 16.0 and model_copy.beta2 == 0.995 and model_copy.visualize is True
    assert_model_equal(model, model_copy)
    # the kernels index conv weights as channels-last storage: a loaded model must come back in that layout
    w = model_copy.sampler.up_g4.first_conv_3x3.weight_orig
    assert w.is_contiguous(memory_format=torch.channels_last)


def test_discriminator_serialization(tmp_path, dgmr):  # :365-370
    discriminator = dgmr.Discriminator(input_channels=1, num_spatial_frames=1, conv_type="standard")
    discriminator.save_pretrained(tmp_path / "discriminator")
    assert_model_equal(discriminator, dgmr.Discriminator.from_pretrained(tmp_path / "discriminator"))


def test_sampler_serialization(tmp_path, dgmr):  # :373-380
    sampler = dgmr.Sampler(forecast_steps=1, latent_channels=256, context_channels=256, output_channels=1)
    sampler.save_pretrained(tmp_path / "sampler")
    assert_model_equal(sampler, dgmr.Sampler.from_pretrained(tmp_path / "sampler"))


def test_context_conditioning_stack_serialization(tmp_path, dgmr):  # :383-390
    ctz = dgmr.ContextConditioningStack(input_channels=2, output_channels=256, num_context_steps=1, conv_type="standard")
    ctz.save_pretrained(tmp_path / "context-conditioning-stack")
    assert_model_equal(ctz, dgmr.ContextConditioningStack.from_pretrained(tmp_path / "context-conditioning-stack"))


def test_latent_conditioning_stack_serialization(tmp_path, dgmr):  # :393-399
    lat = dgmr.LatentConditioningStack(shape=(4, 4, 4), output_channels=256, use_attention=True)
    lat.save_pretrained(tmp_path / "latent-conditioning-stack")
    assert_model_equal(lat, dgmr.LatentConditioningStack.from_pretrained(tmp_path / "latent-conditioning-stack"))


def test_lightning_checkpoint_round_trip(tmp_path):
    """A Lightning-format .ckpt ({"state_dict", "hyper_parameters"}) loads without Lightning installed."""
    kw = dict(forecast_steps=2, output_shape=128, latent_channels=384, context_channels=192, generation_steps=2, grid_lambda=7.0)
    torch.manual_seed(3)
    model = S.DGMR(**kw)
    path = tmp_path / "last.ckpt"
    torch.save({"state_dict": {k: v.clone().contiguous() for k, v in model.state_dict().items()}, "hyper_parameters": kw,
                "epoch": 3, "global_step": 17}, path)
    if not S.dgmr.HAVE_LIGHTNING:
        copy = S.DGMR.load_from_checkpoint(path)
        assert copy.grid_lambda == 7.0 and copy.generation_steps == 2
        assert_model_equal(copy, model)
        assert copy.sampler.g1.first_conv_3x3.weight_orig.is_contiguous(memory_format=torch.channels_last)
        with pytest.raises(KeyError):
            torch.save({"weights": {}}, tmp_path / "bad.ckpt")
            S.DGMR.load_from_checkpoint(tmp_path / "bad.ckpt")


def _reference():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import _stubs

    _stubs.install()
    for name in [k for k in sys.modules if k == "dgmr" or k.startswith("dgmr.")]:  # drop the alias: import the REAL reference
        del sys.modules[name]
    import dgmr as ref

    torch.autograd.set_detect_anomaly(False)
    return ref


@needs_reference
def test_interchange_with_the_reference(tmp_path):
    """Reference -> here and here -> reference: state_dict, save_pretrained directory, .ckpt."""
    kw = dict(forecast_steps=2, input_channels=1, output_shape=128, latent_channels=384, context_channels=192, generation_steps=2)
    try:
        ref = _reference()
        torch.manual_seed(11)
        theirs = ref.DGMR(**kw)
        torch.manual_seed(12)
        ours = S.DGMR(**kw)
        assert list(ours.state_dict().keys()) == list(theirs.state_dict().keys())
        # 1. reference state dict -> our model (strict), tensors equal afterwards, conv weights re-laid out channels-last
        ours.load_state_dict(theirs.state_dict(), strict=True)
        for (k, a), b in zip(ours.state_dict().items(), theirs.state_dict().values()):
            assert torch.equal(a, b), k
        assert ours.discriminator.temporal_discriminator.d1.first_conv_3x3.weight_orig.is_contiguous(memory_format=torch.channels_last_3d)
        # 2. a directory written by the reference's save_pretrained -> our from_pretrained
        theirs.save_pretrained(tmp_path / "theirs")
        loaded = S.DGMR.from_pretrained(tmp_path / "theirs")
        for (k, a), b in zip(loaded.state_dict().items(), theirs.state_dict().values()):
            assert torch.equal(a, b), k
        assert loaded.generation_steps == 2 and loaded.latent_channels == 384
        # 3. our save_pretrained -> the reference's from_pretrained
        torch.manual_seed(13)
        mine = S.DGMR(**kw)
        mine.save_pretrained(tmp_path / "mine")
        back = ref.DGMR.from_pretrained(tmp_path / "mine")
        for (k, a), b in zip(back.state_dict().items(), mine.state_dict().values()):
            assert torch.equal(a, b), k
        # 4. a reference-side Lightning checkpoint -> our loader
        torch.save({"state_dict": theirs.state_dict(), "hyper_parameters": kw}, tmp_path / "theirs.ckpt")
        if not S.dgmr.HAVE_LIGHTNING:
            from_ckpt = S.DGMR.load_from_checkpoint(tmp_path / "theirs.ckpt")
            for (k, a), b in zip(from_ckpt.state_dict().items(), theirs.state_dict().values()):
                assert torch.equal(a, b), k
    finally:
        for name in [k for k in sys.modules if k == "dgmr" or k.startswith("dgmr.")]:
            del sys.modules[name]


# ------------------------------------------------------------------------------------------------------------------------------
# data path (train/run.py:114-158)
# ------------------------------------------------------------------------------------------------------------------------------
def test_extract_input_and_target_frames():
    from skillful_nowcasting_amd import data

    frames = np.arange(24 * 4 * 6 * 1, dtype=np.float32).reshape(24, 4, 6, 1)
    x, y = data.extract_input_and_target_frames(frames)
    assert x.shape == (4, 4, 6, 1) and y.shape == (18, 4, 6, 1)
    assert np.array_equal(x, frames[2:6]) and np.array_equal(y, frames[6:])  # targets end the window, inputs right before
    xs, ys = data.row_to_sample({"radar_frames": frames})
    assert xs.shape == (4, 1, 4, 6) and ys.shape == (18, 1, 4, 6)  # [T, H, W, C] -> [T, C, H, W]
    assert np.array_equal(xs[:, 0], x[..., 0])


@needs_reference
def test_extract_matches_the_reference_function():
    """The reference's train/run.py cannot be imported here (wandb / datasets / Lightning), so its two data functions are executed
    from their source text."""
    import ast

    from skillful_nowcasting_amd import data

    src = open(os.path.join(REF, "train", "run.py")).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if (isinstance(n, ast.Assign) and any(getattr(t, "id", "").startswith("NUM_") for t in n.targets))
            or (isinstance(n, ast.FunctionDef) and n.name == "extract_input_and_target_frames")]
    ns = {}
    exec(compile(ast.Module(body=keep, type_ignores=[]), "run.py", "exec"), ns)
    frames = np.random.default_rng(0).random((30, 5, 7, 2)).astype(np.float32)
    rx, ry = ns["extract_input_and_target_frames"](frames)
    x, y = data.extract_input_and_target_frames(frames)
    assert np.array_equal(rx, x) and np.array_equal(ry, y)
    assert (ns["NUM_INPUT_FRAMES"], ns["NUM_TARGET_FRAMES"]) == (data.NUM_INPUT_FRAMES, data.NUM_TARGET_FRAMES)
    assert np.array_equal(np.moveaxis(rx, [0, 1, 2, 3], [0, 2, 3, 1]), data.to_model_layout(x))


@pytest.mark.parametrize("dtype", [np.uint8, np.float16, np.float32])
def test_radar_batch_loader_cpu(dtype):
    from skillful_nowcasting_amd import data

    rng = np.random.default_rng(1)
    rows = [{"radar_frames": (rng.random((24, 8, 8, 1)) * 200).astype(dtype)} for _ in range(5)]
    loader = data.RadarBatchLoader(rows, batch_size=2, scale=1 / 32.0, drop_last=False)
    batches = list(loader)
    assert [b[0].shape[0] for b in batches] == [2, 2, 1]
    for bi, (images, future) in enumerate(batches):
        assert images.dtype == torch.float32 and images.shape[1:] == (4, 1, 8, 8) and future.shape[1:] == (18, 1, 8, 8)
        for j in range(images.shape[0]):
            x, y = data.row_to_sample(rows[2 * bi + j])
            assert torch.allclose(images[j], torch.from_numpy(np.ascontiguousarray(x)).float() / 32.0)
            assert torch.allclose(future[j], torch.from_numpy(np.ascontiguousarray(y)).float() / 32.0)
    assert len(list(data.RadarBatchLoader(rows, batch_size=2))) == 2  # drop_last


@pytest.mark.gpu
def test_radar_batch_loader_gpu_matches_cpu_path():
    """The device path of the loader (pinned double-buffered slots, H2D copies on a side stream, fp32 conversion on the device;
    reference semantics train/run.py:118-158) against the CPU path, on uint8 rows and enough batches for every slot to be REUSED
    several times while earlier uploads may still be in flight (the host waits for a slot's previous copy before refilling it)."""
    from skillful_nowcasting_amd import data

    rng = np.random.default_rng(3)
    rows = [{"radar_frames": rng.integers(0, 255, (26, 64, 64, 1), dtype=np.uint8)} for _ in range(6 * 8 + 3)]
    cpu = list(data.RadarBatchLoader(rows, batch_size=8, scale=1 / 32.0, drop_last=False))
    dev = []
    for images, future in data.RadarBatchLoader(rows, batch_size=8, device="cuda", scale=1 / 32.0, drop_last=False):
        assert images.is_cuda and images.dtype == torch.float32
        # a consumer that lags behind the producer: the loader has already refilled the other slot when this batch is read
        torch.cuda._sleep(20_000_000)
        dev.append((images.clone(), future.clone()))
    torch.cuda.synchronize()
    assert len(dev) == len(cpu) == 7
    for (ci, cf), (di, df) in zip(cpu, dev):
        assert torch.equal(ci, di.cpu()) and torch.equal(cf, df.cpu())


def test_state_dict_is_current_after_an_optimiser_update_while_an_older_dict_is_alive():
    """conv weights are exported as contiguous copies; a copy kept alive by an earlier state_dict() must not be handed
    out again after the parameters changed (the optimiser kernels write parameters without bumping torch's version counter)."""
    torch.manual_seed(0)
    conv = S.common.DBlock(4, 8)
    sd1 = conv.state_dict()
    key = next(k for k in sd1 if k.endswith("first_conv_3x3.parametrizations.weight.original"))
    before = sd1[key].clone()
    with torch.no_grad():
        conv.first_conv_3x3.weight_orig.data.add_(1.0)  # out-of-band write: no version bump, exactly what the Adam kernel does
    sd2 = conv.state_dict()
    assert sd2[key] is not sd1[key]
    assert torch.equal(sd2[key], before + 1.0) and torch.equal(sd1[key], before)
    assert sd2[key].is_contiguous()
    # within ONE call the aliased names of DGMR still share a tensor (a checkpoint must not store the generator twice)
    model = S.DGMR(forecast_steps=2, output_shape=128, latent_channels=384, context_channels=192, generation_steps=2)
    sd = model.state_dict()
    k = "sampler.g1.first_conv_3x3.parametrizations.weight.original"
    assert sd[k] is sd["generator." + k]


def test_grid_cell_loss_without_weight_fn_constructs_like_the_reference():
    """dgmr/losses.py:161-190: GridCellLoss() constructs; its forward fails with a TypeError (`difference * None`)."""
    loss = S.losses.GridCellLoss()
    with pytest.raises(TypeError):
        loss(torch.zeros(1, 2, 1, 4, 4), torch.zeros(1, 2, 1, 4, 4))


# ------------------------------------------------------------------------------------------------------------------------------
# visualisation (dgmr/dgmr.py:302-327)
# ------------------------------------------------------------------------------------------------------------------------------
def test_make_grid_matches_torchvision_semantics():
    from skillful_nowcasting_amd.dgmr import make_grid

    one = make_grid([torch.arange(6.0).view(1, 2, 3)], nrow=1)
    assert one.shape == (3, 2, 3) and torch.equal(one[0], one[2])  # a single grey image: replicated to 3 channels, no padding
    imgs = [torch.full((1, 2, 3), float(i + 1)) for i in range(3)]
    grid = make_grid(imgs, nrow=2)  # 2 columns x 2 rows, 2-pixel padding of zeros
    assert grid.shape == (3, 2 * (2 + 2) + 2, 2 * (3 + 2) + 2)
    assert torch.all(grid[:, 2:4, 2:5] == 1) and torch.all(grid[:, 2:4, 7:10] == 2) and torch.all(grid[:, 6:8, 2:5] == 3)
    assert torch.all(grid[:, :2] == 0) and torch.all(grid[:, 6:8, 7:10] == 0)


def test_visualize_step_logs_the_reference_image_set():
    class Board:
        def __init__(self):
            self.calls = []

        def add_image(self, tag, img, global_step=None):
            self.calls.append((tag, tuple(img.shape), global_step))

    class Logger:
        def __init__(self):
            self.experiment = [Board()]

    model = S.DGMR(forecast_steps=4, output_shape=128, latent_channels=384, context_channels=192)
    model.logger = Logger()  # (the Lightning-free stand-in base has no logger property)
    x, y, y_hat = torch.rand(2, 4, 1, 8, 8), torch.rand(2, 4, 1, 8, 8), torch.rand(2, 4, 1, 8, 8)
    model.visualize_step(x, y, y_hat, 5, step="train")
    calls = model.logger.experiment[0].calls
    assert len(calls) == 12  # 4 input frames x (input, target, generated)
    assert calls[0] == ("train/Input_Image_Stack_Frame_0", (3, 8, 8), 5)
    assert {c[0] for c in calls} == {f"train/{n}_Frame_{i}" for n in ("Input_Image_Stack", "Target_Image", "Generated_Image") for i in range(4)}


def test_default_logger_is_an_in_memory_recorder():
    """Without Lightning `model.logger` exists (dgmr/dgmr.py:307 reads `self.logger.experiment[0]`) and records the images."""
    model = S.DGMR(forecast_steps=4, output_shape=128, latent_channels=384, context_channels=192)
    x, y, y_hat = torch.rand(1, 4, 1, 8, 8), torch.rand(1, 4, 1, 8, 8), torch.rand(1, 4, 1, 8, 8)
    model.visualize_step(x, y, y_hat, 7, step="val")
    imgs = model.logger.experiment[0].images
    assert len(imgs) == 12 and imgs[0][0] == "val/Input_Image_Stack_Frame_0" and imgs[0][2] == 7
    assert tuple(imgs[0][1].shape) == (3, 8, 8) and torch.equal(imgs[0][1][0], x[0, 0, 0])


@pytest.mark.gpu
def test_training_and_validation_step_with_visualize_on_gpu():
    """visualize=True through a real step on the HIP path (dgmr/dgmr.py:213-218,285-290): the logging forward's output is what gets
    drawn - 4 input-frame indices x (input, target, generated) per call, finite, the generated frame equal to an eval of the
    same state is not expected (train-mode forward), so only shapes, tags, steps and the input / target pixels are pinned."""
    torch.manual_seed(0)
    model = S.DGMR(forecast_steps=4, output_shape=128, latent_channels=384, context_channels=192, generation_steps=2, visualize=True).to("cuda")
    x, y = torch.rand(2, 4, 1, 128, 128, device="cuda"), torch.rand(2, 4, 1, 128, 128, device="cuda")
    model.training_step((x, y), 0)
    model.validation_step((x, y), 0)
    torch.cuda.synchronize()
    imgs = model.logger.experiment[0].images
    assert len(imgs) == 24
    tags = [t for t, _, _ in imgs]
    assert tags[:3] == ["train/Input_Image_Stack_Frame_0", "train/Target_Image_Frame_0", "train/Generated_Image_Frame_0"]
    assert tags[12] == "val/Input_Image_Stack_Frame_0"
    assert all(step == 1 for _, _, step in imgs)  # global_iteration after the first training_step
    for tag, img, _ in imgs:
        assert tuple(img.shape) == (3, 128, 128) and torch.isfinite(img).all(), tag
    assert torch.equal(imgs[0][1][0], x[0, 0, 0].cpu()) and torch.equal(imgs[1][1][0], y[0, 0, 0].cpu())
    gen = imgs[2][1][0]
    assert gen.abs().max().item() > 0 and not torch.equal(gen, imgs[14][1][0])  # a real forward; the val forward is another draw
