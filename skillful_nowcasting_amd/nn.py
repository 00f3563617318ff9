"""Parameter containers with the reference's ``state_dict`` layout, computing through the HIP operators.

``SNConv`` / ``SNLinear1`` reproduce the key set that ``torch.nn.utils.parametrizations.spectral_norm``
gives a Conv/Linear (``bias``, ``parametrizations.weight.original``, ``parametrizations.weight.0._u``,
``parametrizations.weight.0._v``) and the same construction-time RNG consumption (kaiming-uniform weight,
uniform bias, normal u/v, 15 power iterations — torch/nn/modules/conv.py reset_parameters and
torch/nn/utils/parametrizations.py:432-439), so a seeded construction yields the reference's tensors.
Construction runs on the host with torch's init functions; every forward/backward op is a HIP kernel.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .ops import BNState, ConvSpec


class SNPlan:
    """Every spectral-norm call sequence of one forward, drawn in three launches (dgmr_spectral_sigma_seq_multi).

    `entries`: [(module, calls, layout)] in first-use order, each module at most once; `layout` (ops.CallLayout or None) says
    which group of the consuming batch each call of the sequence belongs to.  The descriptor table (static pointers to W, W W^T,
    u, v, the call -> group map and offsets into a per-run output arena) is uploaded once; `run()` refreshes stale Gram matrices,
    allocates the arena and returns {id(module): SNCall} whose tensors are views into it, in GROUP order.
    """

    def __init__(self, entries):
        import ctypes

        from ._lib import SNDesc

        self.entries = list(entries)
        n = len(self.entries)
        descs = (SNDesc * n)()
        off = row0 = col0 = it0 = 0
        self.max_calls = 1
        self.layout = []
        self.max_cout = 1
        self.ptrs = []
        self.perms = []  # keeps the device call -> group maps alive
        dev = None
        for i, (m, calls, layout) in enumerate(self.entries):
            w = m.weight_orig
            dev = w.device
            perm = ops.call_slots(layout, dev)
            self.perms.append(perm)
            vec = getattr(m.parametrizations.weight, "0")
            cout, cin = w.shape[0], w.shape[1]
            taps = w.numel() // (cout * cin)
            k = cin * taps
            gram = m._gram_buffer()
            d = descs[i]
            d.w, d.gram, d.u, d.v = w.data_ptr(), gram.data_ptr(), vec._u.data_ptr(), vec._v.data_ptr()
            d.perm = perm.data_ptr() if perm is not None else None
            d.inv_sigma_off = off
            d.u_hist_off = off + calls
            d.v_hist_off = d.u_hist_off + calls * cout
            d.tmp_off = d.v_hist_off + calls * k
            self.layout.append((off, calls, cout, k))
            off = d.tmp_off + 3 * cout + calls  # t0[Cout] | dnorm[calls] | y[2][Cout]
            off = (off + 3) // 4 * 4
            d.Cout, d.Cin, d.taps, d.T, d.eps = cout, cin, taps, calls, float(m.eps)
            d.row_block0, d.col_block0, d.iter_block0 = row0, col0, it0
            row0 += cout
            col0 += (k + 63) // 64
            it0 += (cout + 31) // 32
            self.max_cout = max(self.max_cout, cout)
            self.max_calls = max(self.max_calls, calls)
            self.ptrs.append((w.data_ptr(), vec._u.data_ptr(), vec._v.data_ptr(), gram.data_ptr()))
        self.total, self.rows, self.cols, self.iters = off, row0, col0, it0
        raw = bytes(ctypes.string_at(ctypes.addressof(descs), ctypes.sizeof(descs)))
        self.descs_dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self.device = dev

    def valid(self) -> bool:
        for (m, _, _), ptrs in zip(self.entries, self.ptrs):
            vec = getattr(m.parametrizations.weight, "0")
            if (m.weight_orig.data_ptr(), vec._u.data_ptr(), vec._v.data_ptr(), m._gram_buffer().data_ptr()) != ptrs:
                return False
        return True

    def run(self):
        for m, _, _ in self.entries:
            m._gram()  # refresh W W^T in place if the optimiser moved W
        arena = torch.empty(self.total, device=self.device, dtype=torch.float32)
        self.last_arena = arena  # (SNScope._prefetch: the allocation every record of this run is a view of)
        ops.call("dgmr_spectral_sigma_seq_multi", self.descs_dev.data_ptr(), len(self.entries), self.rows, self.cols, self.iters,
                 self.max_cout, self.max_calls, arena.data_ptr(), ops._stream())
        out = {}
        for (m, calls, lay), (off, _, cout, k) in zip(self.entries, self.layout):
            inv_sigma = arena[off:off + calls]
            u = arena[off + calls:off + calls + calls * cout].view(calls, cout)
            v = arena[off + calls + calls * cout:off + calls + calls * cout + calls * k].view(calls, k)
            out[id(m)] = (ops.SNCall(inv_sigma, u, v, calls), lay)
        return out


class SNScope:
    """`with SNScope(owner, key):` around a forward whose spectral-norm calls are data-independent (every DGMR forward is).

    First time for (owner, key): the forward runs with per-module launches while the (module, calls) requests are traced.
    Afterwards the traced plan is executed up front and each `_sigma()` request is served from it.  Eval mode: no-op.
    """

    _active = None
    _plans = {}
    # Prefetch (DGMR.training_step): the generator's weights do not change between the optimiser steps, so the power iterations of its
    # NEXT forward depend on nothing but u / v after the current one.  Inside `with SNScope.step(owner):` the sequence of scope keys
    # of one step is recorded; once two consecutive steps have shown the same sequence, each scope entry issues the plan of the next
    # key on a stream of its own, beside the forward that is about to run, and the next entry only waits for its event
    # (17 ms of latency-bound launches per paper-size step sat on the main stream; two thirds of them can run ahead).
    _steps = {}     # id(owner) -> {"seen": [...keys of this step], "last": [... of the step before], "plan": [...] or None, "pos": int}
    _pending = {}   # (id(owner), key) -> (records, event, arena)
    _streams = {}
    _PREFETCH = __import__("os").environ.get("DGMR_SN_PREFETCH", "1") != "0"

    def __init__(self, owner: nn.Module, key=()):
        self.owner, self.key = owner, (id(owner), key)
        self.records = None
        self.trace = None

    class step:
        """`with SNScope.step(module):` brackets one training step of `module` (see above).  Leaving it with a prefetched sequence
        unconsumed (an exception mid-step) drops the records; u / v have then advanced by that forward's iterations."""

        def __init__(self, owner: nn.Module):
            self.oid = id(owner)

        def __enter__(self):
            st = SNScope._steps.setdefault(self.oid, {"seen": [], "last": None, "plan": None, "pos": 0})
            st["seen"], st["pos"] = [], 0
            return self

        def __exit__(self, *exc):
            st = SNScope._steps.get(self.oid)
            if st is None:
                return False
            for k in [k for k in SNScope._pending if k[0] == self.oid]:
                del SNScope._pending[k]
            complete = exc[0] is None and (st["plan"] is None or st["pos"] == len(st["plan"]))
            # the next step prefetches only if this one repeated the one before (and followed the plan it was given, if any)
            st["plan"] = list(st["seen"]) if (complete and st["last"] == st["seen"] and st["seen"]) else None
            st["last"] = list(st["seen"]) if exc[0] is None else None
            st["seen"] = None  # outside a step: nothing is recorded, nothing prefetched
            return False

    @classmethod
    def weights_changed(cls, owner: nn.Module):
        """An optimiser step has just changed `owner`'s weights: nothing behind this point of the step may be prefetched from in front
        of it (recorded in the step's sequence like a scope)."""
        st = cls._steps.get(id(owner))
        if st is None or st["seen"] is None:
            return
        key = (id(owner), "weights changed")
        st["seen"].append(key)
        seq = st["plan"]
        if seq is not None:
            if st["pos"] < len(seq) and seq[st["pos"]] == key:
                st["pos"] += 1
            else:
                st["plan"] = None

    def _prefetch(self, key):
        if key[1] == "weights changed":
            return False
        hit = SNScope._plans.get(key)
        if hit is None or hit[0]() is not self.owner or not hit[1].valid():
            return False
        plan = hit[1]
        dev = plan.device
        side = SNScope._streams.get(dev)
        if side is None:
            side = SNScope._streams[dev] = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)
        side.wait_stream(main)  # behind this forward's own sequence (u / v) and every optimiser update issued so far
        with torch.cuda.stream(side):
            records = plan.run()
            ev = side.record_event()
        SNScope._pending[key] = (records, ev, plan.last_arena)
        return True

    def __enter__(self):
        self.prev = SNScope._active
        if not self.owner.training or self.prev is not None:
            self.noop = True
            return self
        self.noop = False
        st = SNScope._steps.get(self.key[0])
        if st is not None and st["seen"] is None:
            st = None
        pend = SNScope._pending.pop(self.key, None)
        if any(k[0] == self.key[0] for k in SNScope._pending):
            raise RuntimeError("spectral norm: a prefetched power-iteration sequence is pending for another forward of this module "
                               "(the step did not repeat the previous one; DGMR_SN_PREFETCH=0 switches the prefetch off)")
        hit = SNScope._plans.get(self.key)
        plan = None
        if hit is not None and hit[0]() is self.owner and hit[1].valid():  # same live owner, pointers unchanged
            plan = hit[1]
        if pend is not None:
            records, ev, arena = pend
            cur = torch.cuda.current_stream(arena.device)
            cur.wait_event(ev)
            arena.record_stream(cur)
            self.records = records
        elif plan is not None:
            self.records = plan.run()
        else:
            self.trace = []
        if st is not None:
            st["seen"].append(self.key)
            seq = st["plan"]
            if seq is not None:
                if st["pos"] < len(seq) and seq[st["pos"]] == self.key:
                    st["pos"] += 1
                    if SNScope._PREFETCH and st["pos"] < len(seq):
                        self._prefetch(seq[st["pos"]])
                else:
                    st["plan"] = None  # not the announced sequence: no more prefetching in this step
        SNScope._active = self
        return self

    def __exit__(self, *exc):
        if self.noop:
            return False
        SNScope._active = self.prev
        if self.trace is not None and exc[0] is None:
            seen = {id(m) for m, _, _ in self.trace}
            if self.trace and len(seen) == len(self.trace):  # a module requested twice cannot be planned (dependent sequences)
                import weakref

                SNScope._plans[self.key] = (weakref.ref(self.owner), SNPlan(self.trace))
        return False

    def take(self, module, calls, layout=None):
        if self.records is None:
            return None
        rec = self.records.get(id(module))
        if rec is not None and rec[0].groups == calls and rec[1] == layout:
            del self.records[id(module)]
            return rec[0]
        return None


def _mf(ndim: int):
    return torch.channels_last if ndim == 4 else (torch.channels_last_3d if ndim == 5 else torch.contiguous_format)


def _conv_init(out_channels: int, in_channels: int, ks):
    w = torch.empty(out_channels, in_channels, *ks)
    nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    fan_in = in_channels * math.prod(ks)
    bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
    b = torch.empty(out_channels)
    nn.init.uniform_(b, -bound, bound)
    return w, b


def _relayout_hook(module, incompatible_keys=None):
    """load_state_dict post-hook: conv weights go back to channels-last storage (load_state_dict(assign=True) swaps in the
    checkpoint's NCHW-contiguous tensors) and every cache keyed on the weights is invalidated."""
    for name in ("original", "weight"):
        p = module._parameters.get(name)
        if p is not None and p.dim() in (4, 5) and not p.is_contiguous(memory_format=_mf(p.dim())):
            p.data = p.data.contiguous(memory_format=_mf(p.dim()))
    ops.bump_weights_epoch()


def _export_hook(module, state_dict, prefix, local_metadata):
    """state_dict post-hook: conv weights leave in the standard contiguous (OIHW) layout, like the reference's, so that
    safetensors / torch.save files interchange with it (safetensors refuses non-contiguous tensors outright).  The parameter itself
    stays channels-last (what the kernels index).  One copy per parameter and state_dict() CALL: DGMR lists its generator parts
    twice (`sampler.*` and `generator.sampler.*`), both names must keep pointing at ONE tensor or a checkpoint would store it twice.
    Nothing is reused across calls: the optimiser kernels write parameters without touching torch's version counter, so a copy
    kept from an earlier call would be stale (a dict from an earlier call keeps its own, older, tensors - like a checkpoint).
    A call is recognised by its destination dict, which torch hands to every module's hook of that call."""
    import weakref

    for name in ("original", "weight"):
        key = prefix + name
        t = state_dict.get(key)
        if t is None or t.dim() not in (4, 5) or t.is_contiguous():
            continue
        cache = module.__dict__.setdefault("_export_refs", {})
        hit = cache.get(name)
        copy = hit[1]() if hit is not None and hit[0]() is state_dict else None
        if copy is None:
            copy = t.detach().contiguous()
            try:
                cache[name] = (weakref.ref(state_dict), weakref.ref(copy))
            except TypeError:
                # a destination that cannot be weakly referenced (state_dict(destination={}): a plain dict): no per-call cache, the
                # second name of a shared tensor gets its own copy - correct, merely not deduplicated
                cache.pop(name, None)
        state_dict[key] = copy


class _SNVectors(nn.Module):
    def __init__(self, u, v):
        super().__init__()
        self.register_buffer("_u", u)
        self.register_buffer("_v", v)


class _SNWeight(nn.Module):
    def __init__(self, weight, u, v):
        super().__init__()
        self.original = nn.Parameter(weight)
        self.add_module("0", _SNVectors(u, v))
        self.register_load_state_dict_post_hook(_relayout_hook)
        self.register_state_dict_post_hook(_export_hook)


class _Parametrizations(nn.Module):
    def __init__(self, snw):
        super().__init__()
        self.weight = snw


class _SpectralNormBase(nn.Module):
    """Holds W (``parametrizations.weight.original``), bias, u, v; one power iteration per train-mode call."""

    def _init_sn(self, w: torch.Tensor, b: torch.Tensor, eps: float):
        self.eps = eps
        self.bias = nn.Parameter(b)
        wm = w.flatten(1)
        h, wd = wm.shape
        u = F.normalize(wm.new_empty(h).normal_(0, 1), dim=0, eps=eps)
        v = F.normalize(wm.new_empty(wd).normal_(0, 1), dim=0, eps=eps)
        with torch.no_grad():
            # 15 warm-up iterations (parametrizations.py:432-439) + the train-mode forward that
            # register_parametrization runs once as its consistency check = 16 before the first user call
            for _ in range(16):
                u = F.normalize(torch.mv(wm, v), dim=0, eps=eps)
                v = F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps)
        self.parametrizations = _Parametrizations(_SNWeight(w.contiguous(memory_format=_mf(w.dim())), u, v))
        self.register_buffer("_scratch", torch.zeros(4), persistent=False)

    @property
    def weight_orig(self) -> torch.Tensor:
        return self.parametrizations.weight.original

    def _gram_buffer(self) -> torch.Tensor:
        w = self.weight_orig
        buf = getattr(self, "_gram_buf", None)
        if buf is None or buf.device != w.device:
            buf = torch.empty(w.shape[0], w.shape[0], device=w.device, dtype=torch.float32)
            self._gram_buf = buf
            self._gram_tag = None
        return buf

    def _gram(self) -> torch.Tensor:
        """W W^T (in a persistent buffer), recomputed only when the optimiser (or a load_state_dict) has changed W."""
        w = self.weight_orig
        buf = self._gram_buffer()
        tag = ops._core.weight_tag(w)
        if self._gram_tag != tag:
            ops.weight_gram(w, out=buf)
            self._gram_tag = tag
        return buf

    def _sigma(self, calls: int = 1, layout: Optional[ops.CallLayout] = None) -> ops.SNCall:
        """Spectral-norm record of `calls` consecutive calls of this module (train: one power iteration per call), indexed by the
        GROUP of the consuming batch that each call belongs to (`layout`; default: group q == call q)."""
        if layout is not None:
            if layout.is_identity():
                layout = None
            elif layout.calls != calls:
                raise RuntimeError(f"spectral norm: {calls} calls requested but the call layout describes {layout.calls}")
        vec = getattr(self.parametrizations.weight, "0")
        scope = SNScope._active
        if scope is not None and self.training:
            rec = scope.take(self, calls, layout)
            if rec is not None:
                return rec
            if scope.trace is not None:
                scope.trace.append((self, calls, layout))
        if calls > 1 and self.training:
            # outside a planned forward (first, traced call; stand-alone blocks): a one-module plan, cached on the module
            plans = self.__dict__.setdefault("_solo_plans", {})
            plan = plans.get((calls, layout))
            if plan is None or not plan.valid():
                plan = plans[(calls, layout)] = SNPlan([(self, calls, layout)])
            return plan.run()[id(self)][0]
        # one call, or eval mode (no iteration: every call sees the same sigma)
        return ops.spectral_sigma(self.weight_orig, vec._u, vec._v, self._scratch, self.eps, self.training)


class SNConv(_SpectralNormBase):
    """spectral_norm(Conv2d/Conv3d(k in {1,3}, stride 1, 'same' padding))."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, ndim: int = 2, eps: float = 1e-12):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size, self.ndim = in_channels, out_channels, kernel_size, ndim
        w, b = _conv_init(out_channels, in_channels, (kernel_size,) * ndim)
        self._init_sn(w, b, eps)

    def forward(self, x, *, pre_relu: bool = False, bn: Optional[BNState] = None, upsample: bool = False, residual=None,
                act_relu: bool = False, calls: int = 1, sn: Optional[ops.SNCall] = None, residual_up: bool = False,
                layout: Optional[ops.CallLayout] = None, want_stats: bool = False, pool_out: bool = False):
        """`calls` > 1: x is a batch of `calls` groups (forecast steps / frames [x generator draws]), each group being one call of
        this module in the reference (own power iteration, own sigma); `layout`: which call each group is (ops.CallLayout).
        `sn`: a record drawn earlier with `_sigma` (ConvGRU steps)."""
        if layout is not None:
            calls = layout.calls
        if sn is None:
            sn = self._sigma(calls, layout)
        # want_stats: -> (y, partials): per-tile sums of y and y^2 from the conv's epilogue for the BatchNorm that follows (None when
        # the dispatched kernel has none; BatchNorm.prepare then reads y)
        # pool_out: -> AvgPool2d(2) / AvgPool3d(2) of the conv (+ `residual`, which is then at the pooled resolution): DBlock's tail
        spec = ConvSpec(upsample=upsample, pre_relu=pre_relu, bn=bn, sn=sn, act_relu=act_relu, residual_up=residual_up,
                        want_stats=want_stats and self.training, pool_out=pool_out)
        out = ops.conv(x, self.weight_orig, self.bias, sn.inv_sigma, residual, spec)
        if want_stats and not spec.want_stats:
            return out, None
        return out


class SNLinear1(_SpectralNormBase):
    """spectral_norm(Linear(C, 1)) — the discriminator heads (discriminators.py:100,192)."""

    def __init__(self, in_features: int, eps: float = 1e-12):
        super().__init__()
        self.in_features = in_features
        w = torch.empty(1, in_features)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_features)
        b = torch.empty(1)
        nn.init.uniform_(b, -bound, bound)
        self._init_sn(w, b, eps)

    def forward(self, x, calls: int = 1, layout: Optional[ops.CallLayout] = None):
        if layout is not None:
            calls = layout.calls
        sn = self._sigma(calls, layout)
        return ops.SNLinear1Fn.apply(x, self.weight_orig, self.bias, sn)


class Conv(nn.Module):
    """Plain Conv2d(k in {1,3}, 'same' padding) with keys ``weight`` / ``bias`` (LBlock, Attention)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, bias: bool = True):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        w, b = _conv_init(out_channels, in_channels, (kernel_size, kernel_size)) if bias else (None, None)
        if not bias:
            w = torch.empty(out_channels, in_channels, kernel_size, kernel_size)
            nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        self.weight = nn.Parameter(w.contiguous(memory_format=torch.channels_last))
        if bias:
            self.bias = nn.Parameter(b)
        else:
            self.register_parameter("bias", None)
        self.register_load_state_dict_post_hook(_relayout_hook)
        self.register_state_dict_post_hook(_export_hook)

    def forward(self, x, *, pre_relu: bool = False, residual=None, scale=None, gamma_scale: bool = False):
        spec = ConvSpec(pre_relu=pre_relu, gamma_scale=gamma_scale)
        return ops.conv(x, self.weight, self.bias, scale, residual, spec)


class BatchNorm(nn.BatchNorm2d):
    """Parameter/buffer container with BatchNorm2d's keys and init; never run through torch's batch_norm.

    ``prepare(x)`` launches the statistics kernels and returns the per-channel affine that the NEXT conv
    applies (with the ReLU) while loading its operand: the normalised tensor is never written to HBM.
    """

    def forward(self, x):
        """Stand-alone torch.nn.BatchNorm2d semantics (module-level drop-in: the reference's tests call `sampler.bn(h)` directly,
        tests/test_model.py:211).  The model's own forward never comes here: it folds the affine into the next conv (`prepare`)."""
        ops.require_hip(x)
        xc = ops.to_cl(x)
        n, c, h, w = xc.shape
        y = ops.BatchNorm1dFn.apply(xc.permute(0, 2, 3, 1).reshape(n * h * w, c), self.weight, self.bias, self.running_mean,
                                    self.running_var, self.num_batches_tracked, self.eps, self.momentum, self.training, 1, None)
        return y.view(n, h, w, c).permute(0, 3, 1, 2)

    def prepare(self, x, groups: int = 1, layout: Optional[ops.CallLayout] = None, partials=None) -> BNState:
        """`groups` > 1: x holds that many calls of the reference's module, each with its own batch statistics; `layout` gives the
        order in which the reference made them (= the order of the running-statistics updates).  `partials`: per-tile sums of x and
        x^2 already taken by the conv that produced x (SNConv(..., want_stats=True))."""
        if layout is not None:
            groups = layout.calls
        return ops.bn_prepare(x, self.weight, self.bias, self.running_mean, self.running_var, self.num_batches_tracked, self.eps,
                              self.momentum, self.training, groups, layout, partials)


class BatchNorm1d(nn.BatchNorm1d):
    """BatchNorm1d over [N, C] through the HIP kernels (discriminator heads)."""

    def forward(self, x, groups: int = 1, layout: Optional[ops.CallLayout] = None):
        """`groups` > 1: x is [groups*N, C]; each group is one call of the reference's module (own batch statistics)."""
        if layout is not None:
            groups = layout.calls
        return ops.BatchNorm1dFn.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.num_batches_tracked,
                                       self.eps, self.momentum, self.training, groups, layout)
