"""Second-stream scheduling of the backward pass: weight gradients beside the data-gradient chain, an optional branch stream for
the temporal discriminator, and the joins that order the main stream behind them (no host synchronisation anywhere)."""
from __future__ import annotations

import ctypes
import weakref
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import torch
from torch.autograd import Function

from ._lib import ConvArgs, WgradArgs, call

# Weight gradients off the critical path: the backward chain only needs each conv's DATA gradient; its weight gradient (window /
# im2col kernel, slab reduce, spectral-norm finalize - latency-bound kernels at 20-30 % matrix-pipe occupancy) runs on a second
# stream beside the data-gradient convs of the layers below.  The main stream joins it when the backward pass ends
# (autograd engine callback), i.e. before anything can read a .grad.
_SIDE_STREAMS = {}
_SIDE_PENDING = {}
_SIDE_KEEP = []  # (event after the side work, tensors it reads)


_DEFER_JOIN = [0]


class defer_side_join:
    """Inside: the end of a backward pass does NOT make the main stream wait for the weight-gradient stream; the caller does, with
    join_side_streams(), before it reads a .grad - after putting work that does not need the gradients in between
    (DGMR.training_step: the generator forward of the next discriminator iteration runs beside the tail of the weight gradients)."""

    def __enter__(self):
        _DEFER_JOIN[0] += 1

    def __exit__(self, *exc):
        _DEFER_JOIN[0] -= 1


_BRANCH_STREAMS = {}
# opt-in: measured 1044.7 vs 1049.6 ms/step (-0.5 %) with all parity tests green; off by default - autograd warns about the
# AccumulateGrad stream of inputs shared by the two branches, and half a percent does not pay for a second compute stream's risk
_BRANCH_ON = __import__("os").environ.get("DGMR_BRANCH_STREAM", "0") != "0"


def branch_stream(dev):
    """A second compute stream for an independent branch of the forward (the temporal discriminator beside the spatial one); autograd
    runs the branch's backward on it as well.  None: disabled."""
    if not _BRANCH_ON:
        return None
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _BRANCH_STREAMS.get(idx)
    if st is None:
        st = _BRANCH_STREAMS[idx] = torch.cuda.Stream(device=dev)
    return st


def join_side_streams():
    """The current stream waits for everything issued on the weight-gradient streams and on the branch stream (no host
    synchronisation): parameter gradients are written by the kernels, not handed to autograd, so its own end-of-backward stream
    synchronisation does not cover them."""
    flush_deferred()
    cur = torch.cuda.current_stream()
    _tail_end_of_pass(cur, [side for (idx, _lane), side in _SIDE_STREAMS.items() if cur.device.index == idx])
    for idx, st in _BRANCH_STREAMS.items():
        if cur.device.index == idx:
            cur.wait_stream(st)
    for (idx, _lane), side in _SIDE_STREAMS.items():
        if cur.device.index == idx:
            cur.wait_stream(side)
    _SIDE_PENDING.clear()
    _SIDE_KEEP.clear()


def _join_side_streams():
    flush_deferred()  # (end of a backward pass: nothing may stay behind)
    if _DEFER_JOIN[0]:
        return
    if _SIDE_PENDING:
        key0, main0 = next(iter(_SIDE_PENDING.items()))
        _tail_end_of_pass(main0, [_SIDE_STREAMS[k] for k in _SIDE_PENDING])
    for key, main in list(_SIDE_PENDING.items()):
        main.wait_stream(_SIDE_STREAMS[key])
    for idx, st in _BRANCH_STREAMS.items():  # (its own parameter gradients; the default stream is the one readers use)
        torch.cuda.default_stream(torch.device("cuda", idx)).wait_stream(st)
        for (i2, _lane), side in _SIDE_STREAMS.items():
            if i2 == idx:
                torch.cuda.default_stream(torch.device("cuda", idx)).wait_stream(side)
    _SIDE_PENDING.clear()
    _SIDE_KEEP.clear()  # everything the main stream does from here on is ordered behind the side work


# Deferred weight gradients (DGMR_WGRAD_DEFER=1, inside `with defer_wgrads():` = the generator's backward pass): instead of starting
# beside the fat data-gradient convs of the G-blocks (both compute-bound: they take CUs from each other, the main chain measured
# ~14 % slower), the weight gradients of a sampler level are held back until the level's ConvGRU starts its backward-through-time
# chain - small, latency-bound launches that leave most of the chip idle - and are flushed onto the side stream there
# (ops.ConvGRUFn.backward calls flush_deferred()).  Whatever is still held when a join happens is flushed first.
_DEFER_ON = __import__("os").environ.get("DGMR_WGRAD_DEFER", "0") != "0"
_DEFER_OPEN = [0]
_DEFERRED = []


class defer_wgrads:
    def __enter__(self):
        _DEFER_OPEN[0] += 1

    def __exit__(self, exc_type, exc, tb):
        _DEFER_OPEN[0] -= 1
        if not _DEFER_OPEN[0]:
            if exc_type is not None:
                _DEFERRED.clear()  # the backward pass died: its held-back weight gradients are not wanted (and their operands may be stale)
            else:
                flush_deferred()


def flush_deferred():
    if not _DEFERRED:
        return
    todo = list(_DEFERRED)
    _DEFERRED.clear()
    for dev, fn, tensors, lane in todo:
        _on_side_stream(dev, fn, tensors, lane, _now=True)


# Tail balancing (round 6).  The weight-gradient stream runs behind the data-gradient chain all through a backward pass; when the chain
# ends, the main stream sits at the join with nothing to do while the side stream works off its backlog alone - rocprofv3 of the bench
# step: ONE gap of 58 ms at the end of the generator pass, 6.6 % of the step (profiles/r06_tails_kernel_streams.txt).  So the LAST weight
# gradients of a pass are issued on the main stream instead, behind the chain, and the two streams finish together.  Which ones: every
# call carries a cost (its multiply-adds); a pass (the calls between two joins) is recognised by its first calls' costs; at each join
# the wait is MEASURED (two events: the main stream reaching the join, the side stream running dry) and the pass's inline budget moves by
# half of it - a few steps to settle, no model of kernel speeds.  Same kernels, same operands, same order of gradient-buffer touches:
# results are bit-identical wherever a call runs.  DGMR_WGRAD_TAIL=0 switches it off.
_TAIL_ON = __import__("os").environ.get("DGMR_WGRAD_TAIL", "1") != "0"
_TAIL_KEY = 4  # a pass is keyed by the costs of its first calls


class _TailPass:
    """One backward pass being issued: costs of its side-stream calls so far, how much went inline, the events of its first side launch."""

    def __init__(self):
        self.costs: List[float] = []
        self.inline_cost = 0.0
        self.first_ev = None


class _TailProfile:
    """What is known about a recurring pass: its call costs, and the cost budget of the trailing calls that run on the main stream."""

    def __init__(self):
        self.costs: List[float] = []
        self.budget = 0.0
        self.pending = None  # (event main-at-join, event side-dry, event first side launch, side cost of that pass)
        self.history: List[float] = []  # measured waits (ms), newest last


_tail_pass = _TailPass()
_tail_profiles = {}


def tail_stats():
    """{pass key: (inline budget as a share of the pass's weight-gradient cost, the last measured waits in ms)} - for bench.py / tests."""
    out = {}
    for k, pr in _tail_profiles.items():
        tot = sum(pr.costs) or 1.0
        out[k] = (pr.budget / tot, list(pr.history[-6:]))
    return out


def _tail_inline(cost: float, lane) -> bool:
    """Called for every weight-gradient job: True = run it on the current (main) stream."""
    ps = _tail_pass
    i = len(ps.costs)
    ps.costs.append(float(cost))
    if not _TAIL_ON or lane is not None or i < _TAIL_KEY:  # (fixed-lane work shares scratch with its lane: it stays there)
        return False
    pr = _tail_profiles.get(tuple(ps.costs[:_TAIL_KEY]))
    if pr is None or i >= len(pr.costs) or pr.costs[i] != ps.costs[i]:
        return False  # unknown pass, or it no longer follows the recorded one
    if sum(pr.costs[i:]) <= pr.budget:
        ps.inline_cost += float(cost)
        return True
    return False


def _tail_end_of_pass(main, sides):
    """At a join, BEFORE the main stream is made to wait: close the pass, read the previous measurement of its profile, start a new one."""
    global _tail_pass
    ps, _tail_pass = _tail_pass, _TailPass()
    if not _TAIL_ON or len(ps.costs) <= _TAIL_KEY or not sides:
        return
    key = tuple(ps.costs[:_TAIL_KEY])
    pr = _tail_profiles.get(key)
    if pr is None:
        pr = _tail_profiles[key] = _TailProfile()
    if pr.pending is not None:
        ev_main, ev_side, ev_first, side_cost = pr.pending
        pr.pending = None
        if ev_side.query() and ev_main.query():
            wait = ev_main.elapsed_time(ev_side)  # > 0: the main stream waited that long for the side stream; < 0: the side stream was done first
            span = max(ev_first.elapsed_time(ev_side), 1e-3) if ev_first is not None else 0.0
            pr.history.append(round(wait, 2))
            if span > 0 and abs(wait) > 1.0:
                rate = side_cost / span  # cost the side stream works off per ms over the pass
                pr.budget = min(max(pr.budget + 0.5 * wait * rate, 0.0), 0.5 * sum(ps.costs))
    pr.costs = list(ps.costs)
    ev_main, ev_side = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev_main.record(main)
    ev_side.record(sides[0])
    pr.pending = (ev_main, ev_side, ps.first_ev, sum(ps.costs) - ps.inline_cost)


_SIDE_LANES = int(__import__("os").environ.get("DGMR_WGRAD_LANES", "1"))  # more lanes measured no gain (1051-1060 ms for 1, 2, 3)
_side_rr = [0]


def _on_side_stream(dev, fn, tensors, lane=None, _now=False, cost=0.0):
    """Run fn() (kernel launches through _stream()) on one of the device's side streams, ordered after everything issued so far on
    the current stream; `tensors`: what fn reads that the caller may free right after (kept alive for the side stream's work).
    lane: a fixed stream for work that shares a scratch buffer (the pair-sum planes: lane 0); None: round robin.
    cost: the job's multiply-adds - the tail balancer (above) runs the last jobs of a pass on the current stream instead."""
    if _DEFER_ON and _DEFER_OPEN[0] and not _now:
        _DEFERRED.append((dev, fn, tensors, lane))  # (the closure and `tensors` keep every operand alive until the flush)
        torch.autograd.Variable._execution_engine.queue_callback(_join_side_streams)
        return
    if not _now and _tail_inline(cost, lane):
        fn()  # on the main stream, behind the chain: nothing to order, nothing to keep alive
        torch.autograd.Variable._execution_engine.queue_callback(_join_side_streams)
        return
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if lane is None:
        _side_rr[0] = (_side_rr[0] + 1) % _SIDE_LANES
        lane = _side_rr[0]
    idx = (idx, lane)
    main = torch.cuda.current_stream(dev)
    side = _SIDE_STREAMS.get(idx)
    if side is None:
        side = _SIDE_STREAMS[idx] = torch.cuda.Stream(device=dev)
    while _SIDE_KEEP and _SIDE_KEEP[0][0].query():
        _SIDE_KEEP.pop(0)
    side.wait_stream(main)
    if _TAIL_ON and _tail_pass.first_ev is None:
        _tail_pass.first_ev = torch.cuda.Event(enable_timing=True)
        _tail_pass.first_ev.record(side)
    with torch.cuda.stream(side):
        fn()
    tensors = tuple(t for t in tensors if isinstance(t, torch.Tensor))
    for t in tensors:
        t.record_stream(side)
    # A reference is held until the side work is done: autograd accumulates a second gradient INTO a buffered one in place when
    # nobody else holds it (InputBuffer) - e.g. the gradient this conv hands to its residual - and would overwrite dy on the main
    # stream under the weight-gradient kernel still reading it here.
    _SIDE_KEEP.append((side.record_event(), tensors))
    # a callback per call (the first to run joins, the rest find nothing pending): a backward pass that died on an exception must
    # not leave a stale "callback already queued" state behind
    _SIDE_PENDING[idx] = main
    torch.autograd.Variable._execution_engine.queue_callback(_join_side_streams)


def side_streams(device) -> List["torch.cuda.Stream"]:
    """The weight-gradient / branch streams of `device` that exist so far (ddp: a collective must be ordered behind them)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return [st for (i, _lane), st in _SIDE_STREAMS.items() if i == idx] + [st for i, st in _BRANCH_STREAMS.items() if i == idx]

