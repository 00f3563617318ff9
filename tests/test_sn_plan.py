"""Host logic of the spectral-norm plan (nn.SNPlan): descriptor table and arena layout.  CPU only, no kernel runs."""
import ctypes

import torch


def _modules():
    from skillful_nowcasting_amd.nn import SNConv, SNLinear1

    torch.manual_seed(0)
    from skillful_nowcasting_amd.ops import CallLayout

    # (module, calls, call layout): the first stands for a sampler conv under 2 batched generator draws x 2 forecast steps
    return [(SNConv(8, 16, 3), 4, CallLayout(2, 2, time_major=True)), (SNConv(16, 8, 1), 1, None), (SNLinear1(32), 3, None),
            (SNConv(4, 12, 3, ndim=3), 2, None)]


def test_plan_layout_is_disjoint_and_complete():
    from skillful_nowcasting_amd._lib import SNDesc
    from skillful_nowcasting_amd.nn import SNPlan

    entries = _modules()
    plan = SNPlan(entries)
    raw = bytes(plan.descs_dev.cpu().numpy().tobytes())
    descs = (SNDesc * len(entries)).from_buffer_copy(raw)
    used = []
    rows = cols = its = 0
    assert descs[0].perm == plan.perms[0].data_ptr() and plan.perms[0].tolist() == [0, 2, 1, 3]
    assert all(d.perm is None for d in descs[1:])
    for d, (m, calls, _) in zip(descs, entries):
        w = m.weight_orig
        cout, cin = w.shape[0], w.shape[1]
        taps = w.numel() // (cout * cin)
        k = cin * taps
        assert (d.Cout, d.Cin, d.taps, d.T) == (cout, cin, taps, calls)
        assert d.w == w.data_ptr() and d.gram == m._gram_buffer().data_ptr()
        assert abs(d.eps - m.eps) <= 1e-12 * max(1.0, m.eps) or d.eps == ctypes.c_float(m.eps).value
        assert d.row_block0 == rows and d.col_block0 == cols and d.iter_block0 == its
        rows += cout
        cols += (k + 63) // 64
        its += (cout + 31) // 32
        # arena regions: inv_sigma[T] | u_hist[T*Cout] | v_hist[T*K] | tmp[3*Cout + T]
        regions = [(d.inv_sigma_off, calls), (d.u_hist_off, calls * cout), (d.v_hist_off, calls * k), (d.tmp_off, 3 * cout + calls)]
        for off, n in regions:
            assert off >= 0 and off + n <= plan.total
            used.append((off, off + n))
    assert (rows, cols, its) == (plan.rows, plan.cols, plan.iters) and plan.max_calls == 4
    used.sort()
    for (a0, a1), (b0, b1) in zip(used, used[1:]):
        assert a1 <= b0, "arena regions overlap"
    assert plan.max_cout == 16


def test_plan_valid_tracks_pointers():
    from skillful_nowcasting_amd.nn import SNPlan

    entries = _modules()
    plan = SNPlan(entries)
    assert plan.valid()
    m = entries[0][0]
    vec = getattr(m.parametrizations.weight, "0")
    vec._u = vec._u.clone()  # a buffer moved (e.g. module.to(device)): the plan must notice
    assert not plan.valid()


def test_scope_is_noop_in_eval_and_traces_in_train():
    from skillful_nowcasting_amd.nn import SNScope

    owner = torch.nn.Module()
    owner.eval()
    with SNScope(owner, "k") as sc:
        assert sc.noop and SNScope._active is None
    owner.train()
    with SNScope(owner, "k") as sc:
        assert not sc.noop and SNScope._active is sc and sc.trace == []
    assert SNScope._active is None


def test_call_layout_slots():
    """ops.CallLayout: which group of a batched launch each call of the reference's sequence is."""
    from skillful_nowcasting_amd.ops import CallLayout

    # 3 draws x 2 steps, groups [step][draw]; reference order: draw-major
    assert CallLayout(3, 2, time_major=True).slots() == [0, 3, 1, 4, 2, 5]
    # checkpoint recompute: draws visited last-to-first
    assert CallLayout(3, 2, time_major=True, reverse=True).slots() == [2, 5, 1, 4, 0, 3]
    # context stack: groups [draw][step] are already in call order
    assert CallLayout(3, 2, time_major=False).is_identity()
    assert CallLayout(3, 2, time_major=False, reverse=True).slots() == [4, 5, 2, 3, 0, 1]
    assert CallLayout(1, 5).is_identity() and CallLayout(4, 1).is_identity() and CallLayout(1, 5, reverse=True).is_identity()
    for lay in (CallLayout(6, 18), CallLayout(6, 18, reverse=True), CallLayout(6, 4, time_major=False, reverse=True)):
        assert sorted(lay.slots()) == list(range(lay.calls))


def test_prefetch_schedule_state_machine(monkeypatch):
    """SNScope.step: a step's sequence of scope keys is prefetched one forward ahead only after two identical steps, never across an
    optimiser step, and a forward that leaves the announced sequence with a prefetched one pending is an error (host logic only: plans
    and streams are replaced by recorders)."""
    import pytest
    import torch

    from skillful_nowcasting_amd.nn import SNScope

    class Owner(torch.nn.Module):
        pass

    owner = Owner().train()
    issued, ran = [], []

    def fake_prefetch(self, key):
        if key[1] == "weights changed":
            return False
        issued.append(key[1])
        SNScope._pending[key] = ({}, None, None)
        return True

    monkeypatch.setattr(SNScope, "_prefetch", fake_prefetch)
    orig_enter = SNScope.__enter__

    def enter(self):  # the pending records of the fake carry no event / arena: consume them here, then run the real bookkeeping
        if self.key in SNScope._pending:
            SNScope._pending.pop(self.key)
            ran.append(("prefetched", self.key[1]))
        else:
            ran.append(("inline", self.key[1]))
        return orig_enter(self)

    monkeypatch.setattr(SNScope, "__enter__", enter)

    def one_step(keys, change_after=None):
        with SNScope.step(owner):
            for i, k in enumerate(keys):
                with SNScope(owner, k):
                    pass
                if change_after == i:
                    SNScope.weights_changed(owner)

    seq = ["a", "a", "b", "c", "d"]
    try:
        one_step(seq, change_after=3)
        one_step(seq, change_after=3)
        assert issued == []  # two identical steps first
        one_step(seq, change_after=3)
        # a -> a -> b -> c prefetched one ahead; nothing from in front of the optimiser step for d; the step's first scope never
        assert issued == ["a", "b", "c"]
        assert ran[-5:] == [("inline", "a"), ("prefetched", "a"), ("prefetched", "b"), ("prefetched", "c"), ("inline", "d")]
        assert not SNScope._pending
        # a step that leaves the sequence: the prefetch stops where it diverges ...
        del issued[:]
        with SNScope.step(owner):
            with SNScope(owner, "x"):
                pass
        assert issued == [] and not SNScope._pending
        # ... and needs two identical steps again
        one_step(seq, change_after=3)
        assert issued == []
        one_step(seq, change_after=3)
        assert issued == []
        one_step(seq, change_after=3)
        assert issued == ["a", "b", "c"]
        # leaving the sequence while a prefetched forward is pending cannot be repaired (u / v have advanced): refused loudly
        with pytest.raises(RuntimeError, match="prefetched"):
            with SNScope.step(owner):
                with SNScope(owner, "a"):
                    pass
                with SNScope(owner, "zzz"):
                    pass
        assert not SNScope._pending  # the step's exit dropped it
        # outside a step nothing is recorded or prefetched
        del issued[:]
        with SNScope(owner, "a"):
            pass
        assert issued == []
    finally:
        SNScope._steps.pop(id(owner), None)
        SNScope._pending.clear()
