"""Shared plumbing of the host-side operators (ops.py and its siblings): device pointers and the current stream, layout checks,
scratch buffers, the gradient-destination hook, call-group layouts, spectral-norm call records and BatchNorm statistics.

No arithmetic lives here that is not a call into libdgmr_hip.so; there is no CPU fallback (require_hip)."""
from __future__ import annotations

import ctypes
import os
import weakref
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import torch
from torch.autograd import Function

from ._lib import DETERMINISTIC_DEFAULT, ConvArgs, WgradArgs, call

_WEIGHTS_EPOCH = 0  # bumped by the optimiser after every in-place parameter update (flip cache key)


def bump_weights_epoch():
    global _WEIGHTS_EPOCH
    _WEIGHTS_EPOCH += 1


def weights_epoch() -> int:
    return _WEIGHTS_EPOCH


# Round 6: an optimiser step marks ITS parameters, not every weight of the process.  The caches of derived weight images (bf16 planes,
# flipped / phase / pooled tap sums, W W^T) were keyed on one global counter: a discriminator step also invalidated every image of the
# generator's unchanged weights and vice versa - of the three rebuilds per training step two (generator) and one (discriminator) were
# redundant, ~400 small launches on the forward passes' critical path.  Keyed by the address of the parameter's storage; a tensor that
# is not a stepped parameter (a view at an offset, a temporary) falls back to the count of ALL optimiser steps - the old behaviour.
# Out-of-band writes (load_state_dict, the parameter broadcast, tests) keep using bump_weights_epoch(), which invalidates everything.
_PARAM_STEP = {}   # parameter data_ptr -> number of optimiser steps that wrote it
_ANY_STEP = [0]    # optimiser steps of any optimiser
_PER_PARAM = os.environ.get("DGMR_PARAM_EPOCH", "1") != "0"  # A/B switch: 0 = every optimiser step invalidates every image


def note_optimizer_step(params):
    """The optimiser has just written `params` in place (raw pointers: torch's version counters do not move)."""
    _ANY_STEP[0] += 1
    for p in params:
        k = p.data_ptr()
        _PARAM_STEP[k] = _PARAM_STEP.get(k, 0) + 1


def weight_tag(w: torch.Tensor):
    """What a cached image of weight `w` is valid for: torch's version counter, the global out-of-band epoch, the optimiser steps that
    wrote this parameter (or, for a tensor that is not a stepped parameter, all optimiser steps), address and shape."""
    e = _PARAM_STEP.get(w.data_ptr()) if _PER_PARAM else None
    return (w._version, _WEIGHTS_EPOCH, e if e is not None else -1 - _ANY_STEP[0], w.data_ptr(), tuple(w.shape))


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    """Device pointer of a tensor (None -> NULL); raw integer addresses pass through (slices of step buffers)."""
    if t is None or isinstance(t, int):
        return t
    return t.data_ptr()


# ---- deterministic mode (dgmr_set_deterministic, include/dgmr_hip.h ABI 11) -----------------------------------------------------------
# Every cross-workgroup sum of the step in a fixed order: two identical runs give bit-identical parameters and buffers.  The library
# needs rows of scratch for it, which the caller owns like every other buffer: the helpers below size them.
_DETERMINISTIC = DETERMINISTIC_DEFAULT  # (the library is put into the same mode when it is loaded, _lib.load)


def set_deterministic(on: bool = True):
    global _DETERMINISTIC
    call("dgmr_set_deterministic", int(bool(on)))
    _DETERMINISTIC = bool(on)


def deterministic() -> bool:
    return _DETERMINISTIC


def sums_buffer(groups: int, rows: int, c: int, device, row_blocks: bool = True, zero: bool = True) -> torch.Tensor:
    """`sums` / `tmp` of a per-channel reduction over [groups][rows][c]: groups * 2 * c doubles, zeroed; in deterministic mode followed
    by the library's per-workgroup rows (dgmr_reduce_doubles; row_blocks=False: dgmr_bn_partial_reduce needs none).
    zero=False: scratch of dgmr_colsum, which clears what it uses itself."""
    n = groups * 2 * c
    if not (_DETERMINISTIC and row_blocks):
        return torch.zeros(n, device=device, dtype=torch.float64) if zero else torch.empty(n, device=device, dtype=torch.float64)
    from ._lib import load

    buf = torch.empty(int(load().dgmr_reduce_doubles(groups, rows, c)), device=device, dtype=torch.float64)
    if zero:
        buf[:n].zero_()
    return buf


def colsum_tmp(rows: int, c: int, device) -> torch.Tensor:
    """`tmp` of dgmr_colsum over [rows][c].  Few rows x many columns go through a column-parallel kernel that never touches tmp (ops.hip:
    R <= 4096, C >= 4096, C % 4 == 0 - the backward of a batch repeat / a group sum): two doubles stand in there instead of the
    deterministic mode's per-workgroup rows (tens of MB per call)."""
    if rows <= 4096 and c >= 4096 and c % 4 == 0:
        return torch.empty(2, device=device, dtype=torch.float64)
    return sums_buffer(1, rows, c, device, zero=False)


def dot_buffer(groups: int, device) -> torch.Tensor:
    """<P_q, W> accumulators of dgmr_wgrad_reduce: `groups` floats, zeroed (+ the workgroups' rows in deterministic mode)."""
    if not _DETERMINISTIC:
        return torch.zeros(groups, device=device, dtype=torch.float32)
    from ._lib import load

    buf = torch.empty(int(load().dgmr_wgrad_dot_floats(groups)), device=device, dtype=torch.float32)
    buf[:groups].zero_()
    return buf


def bias_rows(wa, device) -> Optional[torch.Tensor]:
    """After dgmr_conv_wgrad_plan: the row workspace of the deterministic bias gradient (dgmr_wgrad_args.bias_partial); the caller keeps
    the returned tensor alive until the launch has been issued (stream-ordered allocator: that is enough)."""
    if not (_DETERMINISTIC and wa.bias_grad):
        return None
    rows = torch.empty(int(wa.bias_rows) * int(wa.Cout), device=device, dtype=torch.float32)
    wa.bias_partial = rows.data_ptr()
    return rows


_SPLITK_WS = {}
SPLITK_WS_BYTES = 64 << 20


def _splitk_ws(device) -> torch.Tensor:
    """Per-(device, stream) scratch for split-K partial sums (launches are stream-ordered, so one buffer serves every conv of a stream)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _SPLITK_WS.get(key)
    if ws is None:
        ws = torch.empty(SPLITK_WS_BYTES // 4, device=device, dtype=torch.float32)
        _SPLITK_WS[key] = ws
    return ws


_SCRATCH = {}


def _scratch(numel: int, device, key: str) -> torch.Tensor:
    """A persistent fp32 scratch buffer per (device, key), grown to the largest request: for multi-GB temporaries that live inside one
    backward function (launches are stream-ordered: the next user overwrites it only after the previous kernels).  Kept out of
    torch's caching allocator on purpose - a 22.8 GB temporary that comes and goes made the allocator split its cached block for
    other requests and hipMalloc a second one a few steps later (0.4 s stall in the third step, tools/mem_segments.py)."""
    k = (device, key)
    buf = _SCRATCH.get(k)
    if buf is None or buf.numel() < numel:
        _SCRATCH.pop(k, None)
        buf = torch.empty(numel, device=device, dtype=torch.float32)
        _SCRATCH[k] = buf
    return buf[:numel]


_ASYNC_UPLOAD = os.environ.get("DGMR_ASYNC_UPLOAD", "1") != "0"  # A/B switch: 0 = the blocking `.to(device)` of rounds 1 - 6


def _process_group_active() -> bool:
    d = torch.distributed
    return d.is_available() and d.is_initialized()


def upload(t: torch.Tensor, device, dtype=None) -> torch.Tensor:
    """A small host tensor (a latent draw, drawn frame indices - both come from the CPU generator, as in the reference) on `device`
    WITHOUT blocking the host.  `t.to(device)` from pageable memory is a synchronous copy, ordered behind everything queued on the
    stream: the host waits for the GPU to drain - eleven times per training step - and the GPU then idles until the host has issued
    new work (the 0.4 - 6 ms gaps after `__amd_rocclr_copyBuffer` in `profiles/r06_final2_B16_kernel_gaps.txt`, host time per step =
    device time per step).  Staged through pinned memory the copy is asynchronous; torch's caching host allocator keeps the pinned
    block alive until the copy has executed (it records an event for `non_blocking` copies out of pinned memory)."""
    src = t if dtype is None or t.dtype == dtype else t.to(dtype)
    if not _ASYNC_UPLOAD or torch.device(device).type != "cuda" or _process_group_active():
        # Under a process group the blocking copies STAY (measured, `profiles/r06_async_upload_and_host_lead.log`): with the host
        # running ahead, the one-rank RCCL configuration went from 820 - 825 to 896 - 899 ms per step; draining the device once per
        # step gave half of that back (861 - 863).  There the host's lead lets the weight-gradient stream's backlog compete with the
        # data-gradient chain much earlier (it finishes 134 ms before BOTH discriminator joins instead of 114 / 22), and HIP has no
        # stream priority below normal (priority_range() = (0, -1)) to hold it back with: the eleven copies are the throttle.
        return src.to(device)
    pinned = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
    pinned.copy_(src)
    return pinned.to(device, non_blocking=True)


def require_hip(t: torch.Tensor, what: str = "input"):
    if not t.is_cuda:
        raise RuntimeError(
            f"skillful_nowcasting_amd: {what} is on '{t.device}'. The DGMR kernels are HIP-only (gfx950); "
            "there is no CPU fallback — move the module and its inputs to a HIP device."
        )
    if t.dtype != torch.float32:
        raise RuntimeError(f"skillful_nowcasting_amd: {what} must be float32, got {t.dtype}")


def require_weight_layout(w: torch.Tensor, what: str = "conv weight"):
    """The kernels index conv weights as O[D]HWI storage (an OIHW parameter in channels_last memory format).  A parameter that
    was replaced by an NCHW-contiguous tensor (load_state_dict(assign=True), `p.data = ...`) would be read with the wrong
    strides: refuse it instead of computing garbage (nn.SNConv / nn.Conv re-layout such tensors when a state_dict is loaded)."""
    if w.dim() in (4, 5):
        mf = torch.channels_last if w.dim() == 4 else torch.channels_last_3d
        if not w.is_contiguous(memory_format=mf):
            raise RuntimeError(
                f"skillful_nowcasting_amd: {what} of shape {tuple(w.shape)} has strides {tuple(w.stride())}; the HIP kernels need "
                f"channels-last (O[D]HWI) storage. Use `p.data = p.data.contiguous(memory_format=torch.channels_last[_3d])` "
                "and ops.bump_weights_epoch() after writing a parameter out of band.")


def to_cl(x: torch.Tensor) -> torch.Tensor:
    """Channels-last contiguous view/copy of a 4-D or 5-D activation (no-op on the hot path)."""
    if x.dim() == 4:
        return x.contiguous(memory_format=torch.channels_last)
    if x.dim() == 5:
        return x.contiguous(memory_format=torch.channels_last_3d)
    return x.contiguous()


def empty_cl(shape: Sequence[int], like: torch.Tensor) -> torch.Tensor:
    mf = torch.channels_last if len(shape) == 4 else torch.channels_last_3d
    return torch.empty(tuple(shape), device=like.device, dtype=torch.float32, memory_format=mf)


def _dims(x: torch.Tensor):
    """(N, C, D, H, W) of a 4-D / 5-D logical NC[D]HW tensor."""
    if x.dim() == 4:
        n, c, h, w = x.shape
        return n, c, 1, h, w
    n, c, d, h, w = x.shape
    return n, c, d, h, w


_GRAD_TOUCH_HOOK = [None]


def set_grad_touch_hook(fn):
    """fn(p) is called whenever a kernel is about to accumulate into p.grad (None: off).  ddp.GradSync uses the sequence of these
    "touches" to launch each gradient bucket's all-reduce as soon as the backward pass is done with it."""
    _GRAD_TOUCH_HOOK[0] = fn



def grad_buffer(p: torch.Tensor) -> torch.Tensor:
    """``p.grad`` with p's physical layout, zero-initialised on first touch.  THE way a kernel launch obtains the destination of a
    parameter gradient (see set_grad_touch_hook)."""
    if p.grad is None:
        p.grad = torch.zeros_like(p)  # preserve_format: same strides as the parameter
    if _GRAD_TOUCH_HOOK[0] is not None:
        _GRAD_TOUCH_HOOK[0](p)
    return p.grad


# ---------------------------------------------------------------------------------------------------
# call groups: which reference CALL of a module each group of a batched launch stands for
# ---------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class CallLayout:
    """`outer * inner` consecutive calls of one module, run as the groups of ONE batched launch.

    The reference calls a sampler / discriminator module once per forecast step (or frame) inside every generator draw (or
    discriminator call): call number  c = d' * inner + i  for draw position d' and step i.  State that moves per call - the
    spectral-norm power iteration (u, v, sigma) and BatchNorm's running statistics - must follow THAT order, while the batch that
    carries all calls at once is laid out for the kernels:

      time_major   groups [inner][outer]  (the ConvGRU needs every draw's samples of one step contiguous), or
      draw-major   groups [outer][inner]  (context stack: the four context frames of one draw together);
      reverse      the reference visits the draws last-to-first: activation checkpointing recomputes the generator forwards of a
                   step in reverse order during the backward pass (dgmr/dgmr.py:176, torch.utils.checkpoint), and that recompute
                   is what the gradients (and the second advance of u / v / running statistics) come from.
    """

    outer: int = 1
    inner: int = 1
    time_major: bool = True
    reverse: bool = False

    @property
    def calls(self) -> int:
        return self.outer * self.inner

    def slots(self) -> List[int]:
        """slots()[c] = group of the batch that call c of the reference's sequence belongs to."""
        out = []
        for c in range(self.calls):
            dpos, i = divmod(c, self.inner)
            d = self.outer - 1 - dpos if self.reverse else dpos
            out.append(i * self.outer + d if self.time_major else d * self.inner + i)
        return out

    def is_identity(self) -> bool:
        return self.slots() == list(range(self.calls))


_SLOT_CACHE = {}


def call_slots(layout: Optional[CallLayout], device) -> Optional[torch.Tensor]:
    """Device int32 array of `layout.slots()` (None for the identity order), cached per (layout, device)."""
    if layout is None or layout.is_identity():
        return None
    key = (layout, str(device))
    t = _SLOT_CACHE.get(key)
    if t is None:
        t = torch.tensor(layout.slots(), dtype=torch.int32).to(device)
        _SLOT_CACHE[key] = t
    return t


# ---------------------------------------------------------------------------------------------------
# spectral norm (torch/nn/utils/parametrizations.py:454-521)
# ---------------------------------------------------------------------------------------------------
@dataclass
class SNCall:
    """Record of `groups` consecutive calls of one spectral-norm module (1 for an ordinary call)."""

    inv_sigma: torch.Tensor  # [groups]
    u: torch.Tensor  # [groups, Cout]  copies of the u, v that sigma was computed with (the module buffers move on)
    v: torch.Tensor  # [groups, K]
    groups: int = 1

    def at(self, t: int) -> "SNCall":
        """The t-th call of a sequence as a single-call record (views, no copies)."""
        return SNCall(self.inv_sigma[t:t + 1], self.u[t:t + 1], self.v[t:t + 1], 1)


# ---------------------------------------------------------------------------------------------------
# batch norm statistics -> per-channel affine consumed by the next conv's operand load
# ---------------------------------------------------------------------------------------------------
@dataclass
class BNState:
    a: torch.Tensor  # [G, C]   y = relu(a*x + b)
    b: torch.Tensor
    mean: torch.Tensor
    rstd: torch.Tensor
    gamma: Optional[torch.Tensor]
    beta: Optional[torch.Tensor]
    train: bool
    groups: int
    group_size: int  # samples per group


def bn_prepare(x: torch.Tensor, gamma, beta, running_mean, running_var, num_batches_tracked, eps: float, momentum: float,
               train: bool, groups: int = 1, layout: Optional[CallLayout] = None, partials: Optional[torch.Tensor] = None) -> BNState:
    """BatchNorm statistics of x (train) or running statistics (eval) folded to y = a*x + b.

    torch.nn.BatchNorm2d semantics (dgmr/common.py:38-39,108-109; generators.py:113): biased batch variance
    for normalisation, unbiased for the running estimate, momentum 0.1, one running update per group, applied in the
    reference's call order (`layout`, see CallLayout; default: group order).
    """
    require_hip(x)
    x = to_cl(x)
    n, c, d, h, w = _dims(x)
    if not train:
        groups = 1
    assert n % groups == 0
    r = (n // groups) * d * h * w
    dev = x.device
    a = torch.empty(groups, c, device=dev)
    b = torch.empty(groups, c, device=dev)
    mean = torch.empty(groups, c, device=dev)
    rstd = torch.empty(groups, c, device=dev)
    if train:
        from_partials = partials is not None and partials.shape[0] % groups == 0 and partials.shape[2] == c
        sums = sums_buffer(groups, r, c, dev, row_blocks=not from_partials)
        if from_partials:
            # the conv that produced x already summed y and y^2 per pixel tile in its epilogue (`want_stats`): x is not read again
            call("dgmr_bn_partial_reduce", _p(partials), _p(sums), groups, partials.shape[0] // groups, c, _stream())
        else:
            call("dgmr_bn_stats", _p(x), _p(sums), groups, r, c, _stream())
        if layout is not None and layout.calls != groups:
            raise RuntimeError(f"batch norm: {groups} call groups but the call layout describes {layout.calls}")
        call("dgmr_bn_finalize", _p(sums), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(num_batches_tracked),
             _p(a), _p(b), _p(mean), _p(rstd), groups, r, c, float(eps), float(momentum), _p(call_slots(layout, dev)), _stream())
    else:
        call("dgmr_bn_finalize", None, _p(gamma), _p(beta), _p(running_mean), _p(running_var), None, _p(a), _p(b), _p(mean),
             _p(rstd), 1, r, c, float(eps), float(momentum), None, _stream())
    return BNState(a, b, mean, rstd, gamma, beta, train, groups, n // groups)



def _copy(src_ptr, dst_ptr, n):
    call("dgmr_axpby", src_ptr, None, dst_ptr, 1.0, 0.0, n, _stream())

