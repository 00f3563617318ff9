"""Host logic of round 6 (no GPU): which cached weight images an optimiser step invalidates (_core.note_optimizer_step / weight_tag) and the
upload helper's CPU path.  The GPU side - bit-identical training state with the old and the new keying - is tools/state_digest.py
(profiles/r06_per_parameter_epochs_ab.log)."""
import torch

from skillful_nowcasting_amd import _core


def test_an_optimizer_step_moves_only_its_own_parameters_tags():
    g = [torch.nn.Parameter(torch.zeros(4, 3, 3, 3)) for _ in range(3)]
    d = [torch.nn.Parameter(torch.zeros(2, 3, 3, 3)) for _ in range(2)]
    _core.note_optimizer_step(g)  # both networks have been stepped once: every parameter is known by address
    _core.note_optimizer_step(d)
    tg, td = [_core.weight_tag(p) for p in g], [_core.weight_tag(p) for p in d]
    _core.note_optimizer_step(d)  # a discriminator step
    assert [_core.weight_tag(p) for p in g] == tg, "the generator's images must survive a discriminator step"
    assert all(a != b for a, b in zip([_core.weight_tag(p) for p in d], td))
    _core.note_optimizer_step(g[:2])  # a step that skipped a parameter without a gradient
    assert _core.weight_tag(g[2]) == tg[2]
    assert _core.weight_tag(g[0]) != tg[0] and _core.weight_tag(g[1]) != tg[1]


def test_tensors_that_are_not_stepped_parameters_follow_every_step():
    p = torch.nn.Parameter(torch.zeros(8, 4))
    other = torch.zeros(8, 4)          # e.g. a temporary holding tap sums
    view = p.detach()[2:]              # a view at an offset: another address
    _core.note_optimizer_step([p])
    t_other, t_view = _core.weight_tag(other), _core.weight_tag(view)
    q = torch.nn.Parameter(torch.zeros(2))
    _core.note_optimizer_step([q])     # some other optimiser's step
    assert _core.weight_tag(other) != t_other and _core.weight_tag(view) != t_view  # conservative: as with the global counter
    same_address = p.detach()          # a view at offset 0 IS the parameter's memory
    assert _core.weight_tag(same_address)[2] == _core.weight_tag(p)[2]


def test_out_of_band_writes_and_torch_version_counters_invalidate():
    p = torch.nn.Parameter(torch.zeros(4))
    _core.note_optimizer_step([p])
    t0 = _core.weight_tag(p)
    _core.bump_weights_epoch()         # load_state_dict, the parameter broadcast, a test writing p.data
    t1 = _core.weight_tag(p)
    assert t1 != t0
    with torch.no_grad():
        p.add_(1.0)                    # a torch optimiser: the version counter moves
    assert _core.weight_tag(p) != t1


def test_global_keying_switch(monkeypatch):
    g, d = torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(3))
    _core.note_optimizer_step([g])
    _core.note_optimizer_step([d])
    monkeypatch.setattr(_core, "_PER_PARAM", False)  # DGMR_PARAM_EPOCH=0: rounds 1 - 5
    tg = _core.weight_tag(g)
    _core.note_optimizer_step([d])
    assert _core.weight_tag(g) != tg


def test_upload_on_cpu_is_a_plain_conversion():
    idx = torch.arange(12, dtype=torch.int64).reshape(3, 4)
    out = _core.upload(idx, "cpu", torch.int32)
    assert out.dtype == torch.int32 and torch.equal(out.to(torch.int64), idx)
    z = torch.randn(8, 8, 8, 1).permute(3, 0, 1, 2)
    assert torch.equal(_core.upload(z, "cpu", torch.float32), z)
    assert _core._process_group_active() is False


def test_uploads_stay_blocking_under_a_process_group():
    """_core.upload keeps the synchronous copy whenever torch.distributed is initialised (profiles/r06_async_upload_and_host_lead.log, part 3)."""
    import os

    import torch.distributed as dist

    if dist.is_initialized():  # (another test of this process owns a group: the predicate is all there is to check)
        assert _core._process_group_active()
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29631")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        assert _core._process_group_active()
        assert torch.equal(_core.upload(torch.arange(4), "cpu", torch.int32), torch.arange(4, dtype=torch.int32))
    finally:
        dist.destroy_process_group()
    assert not _core._process_group_active()
