"""Data-parallel gradient exchange for the DGMR step: one process per GPU, RCCL all-reduce over xGMI.

The reference has no distributed code; Lightning would wrap it in stock DDP.  The step has
two optimisers and three backward passes, so the exchange is explicit here:

* gradients live in ONE flat fp32 buffer per network (generator 204 MiB, discriminator 170 MiB at the paper
  config); each ``param.grad`` is a strided view into it, so the kernels' direct ``param.grad`` accumulation
  fills the buffer and no pack/unpack copies exist;
* after a D backward only D's buffer is reduced, after the G backward only G's — the cross-network gradients
  the reference computes and discards (Q6) never touch the wire;
* the buffer is reduced in 64 MB buckets (xGMI links are point-to-point, ~153 GB/s each: fewer, larger
  collectives), SUM then a 1/world scale (HIP kernel on GPU tensors).  Buckets are launched ASYNCHRONOUSLY
  while the backward pass is still running, in the order in which they complete — the reverse of the forward
  order, since the buffer is laid out in registration (= forward) order and the backward walks the layers
  back to front: see `Overlap` below.  Only ``wait()`` (before ``optimizer.step()``) orders the main stream
  behind the collectives;
* the 12 parameters that never receive a gradient simply stay zero in the buffer (a stock DDP reducer would
  stall on them);
* buffers (u/v, BN running statistics) live in ONE persistent flat tensor (each module buffer is a view) that
  is broadcast from rank 0 once per step, like DDP's broadcast_buffers — one collective, no pack / unpack;
* parameters are broadcast once at attach time as one coalesced collective per network.

Overlap.  Parameter gradients are written by kernels, not handed to autograd, so there are no AccumulateGrad hooks
to hang a reducer on.  Every kernel obtains its destination through ``ops.grad_buffer(p)``: that call is the
"touch" of p.  The first backward of each kind ('d', 'g') runs un-overlapped and RECORDS its touch sequence (the
step is a static graph: the sequence is the same every step); from it, the position of the LAST touch of every
bucket is known.  In later backwards a bucket is launched once the sequence has moved `MARGIN` touches past the
bucket's last one (a kernel is issued at most two touches after its ``grad_buffer`` call — `MARGIN` = 4): the
collective is enqueued behind the CURRENT state of the main and the weight-gradient streams, i.e. behind every
kernel that writes the bucket, and overlaps whatever the backward pass still has to do for the layers in front.
A sequence that deviates from the recording stops launching early and reduces the remaining buckets at ``sync()`` - IN THE RECORDED
ORDER: every rank issues its collectives in one canonical order (the recorded launch order, cross-checked between the ranks when
it is recorded), whether it launched a bucket during the backward pass or late, so equal-sized buckets can never be paired across
ranks by accident.  A touch of a bucket that has already been launched raises (never a silent wrong gradient).

The CPU branch of ``_scale`` exists only so the collective logic can be exercised under gloo in tests.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional

import torch
import torch.distributed as dist

MARGIN = 4  # touches between a parameter's grad_buffer() call and the latest point at which its kernels are issued (see above)


class FlatGrads:
    """One flat gradient buffer for a list of parameters; ``p.grad`` become views with p's own strides."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        seen, self.params = set(), []
        for p in params:
            if id(p) not in seen and p.requires_grad:
                seen.add(id(p))
                self.params.append(p)
        total = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        self.flat = torch.zeros(total, device=p0.device, dtype=p0.dtype)
        self.offset: Dict[int, int] = {}
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].as_strided(p.shape, p.stride())
            self.offset[id(p)] = off
            off += n

    def zero_(self):
        self.flat.zero_()
        off = 0
        for p in self.params:  # re-attach in case something dropped the views
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * off:
                p.grad = self.flat[off:off + n].as_strided(p.shape, p.stride())
            off += n


class _Pass:
    """State of one backward pass kind ('d' or 'g'): the recorded touch sequence and, per bucket, its last touch."""

    def __init__(self):
        self.recorded: Optional[List[int]] = None  # parameter ids in touch order (first backward of this kind)
        self.last_touch: List[int] = []            # per bucket: index of its last touch in `recorded`
        self.order: List[int] = []                 # buckets sorted by last touch (= launch order)


class GradSync:
    def __init__(self, model, process_group=None, chunk_mb: int = 64, overlap: bool = True, force_exchange: bool = False):
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # force_exchange: run every collective even in a one-rank group (all-reduce and broadcast over one rank are the identity, the
        # scale is 1/1): the whole RCCL path - library load, communicator on device_id, comm-stream fan-in, work.wait() - executes on a
        # single GPU and must leave the parameters bit-identical to the run without it (tests/test_gpu_rccl.py)
        self.exchange = self.world > 1 or (bool(force_exchange) and dist.is_initialized())
        if dist.is_initialized() and process_group is None and "WORLD_SIZE" in os.environ:
            # the launcher's idea of the job and the process group's must agree (a rank that fell back to a private group would train alone)
            if int(os.environ["WORLD_SIZE"]) != self.world:
                raise RuntimeError(f"process group reports {self.world} ranks, the launcher set WORLD_SIZE={os.environ['WORLD_SIZE']}")
        self.gen = FlatGrads(model.generator.parameters())
        self.disc = FlatGrads(model.discriminator.parameters())
        self.chunk = chunk_mb * (1 << 20) // 4
        self.model = model
        self.overlap = overlap
        self.check_exchange = False  # tests: keep every bucket's local values and verify reduced == mean over ranks, bit for bit
        self._passes = {"d": _Pass(), "g": _Pass()}
        self._active: Optional[str] = None
        self._touches: List[int] = []
        self._launched: Dict[int, object] = {}   # bucket -> work handle (or None for a synchronous reduce)
        self._next = 0                            # index into the pass's launch order
        self._deviated = False
        self._locals: Dict[int, torch.Tensor] = {}
        self._comm_stream = None
        # late_buckets: launched at sync() instead of during the backward pass; of those, late_buckets_recording_step belong to the first
        # pass of a network (nothing recorded yet: every bucket is late) - what remains afterwards is one bucket per pass, the one the pass
        # writes last, plus whatever a deviation from the recorded order left over
        self.stats = {"overlapped_buckets": 0, "late_buckets": 0, "late_buckets_recording_step": 0, "deviations": 0}
        self._flatten_buffers(model)

    # ------------------------------------------------------------------------------------------------------------------
    # buffers and parameters
    # ------------------------------------------------------------------------------------------------------------------
    def _flatten_buffers(self, model):
        """Every floating-point module buffer (u, v, BatchNorm running statistics) becomes a view into one persistent flat tensor:
        broadcast_buffers is then ONE collective on that tensor, no torch.cat and no copies back.  Kernels update the buffers in
        place through their data pointers, load_state_dict copies in place: the views stay attached."""
        seen, bufs = set(), []
        for name, b in model.named_buffers():
            if b.is_floating_point() and b.numel() > 0 and b.dim() > 0 and not name.endswith("_scratch") and id(b) not in seen:
                seen.add(id(b))
                bufs.append(b)
        self._buffers = bufs
        self._buf_flat = None
        if not bufs:
            return
        flat = torch.empty(sum(b.numel() for b in bufs), device=bufs[0].device, dtype=torch.float32)
        off = 0
        with torch.no_grad():
            for b in bufs:
                n = b.numel()
                view = flat[off:off + n].view(b.shape)
                view.copy_(b)
                b.data = view
                off += n
        self._buf_flat = flat

    def broadcast_buffers(self, src: int = 0):
        if not self.exchange or self._buf_flat is None:
            return
        dist.broadcast(self._buf_flat, src=src, group=self.pg)

    def broadcast_parameters(self, src: int = 0):
        """One coalesced broadcast per network (two collectives instead of one per tensor)."""
        if not self.exchange:
            return
        with torch.no_grad():
            for fg in (self.gen, self.disc):
                flat = torch.cat([p.data.reshape(-1) for p in fg.params])  # logical (OIHW) order; layouts are restored by copy_
                dist.broadcast(flat, src=src, group=self.pg)
                off = 0
                for p in fg.params:
                    n = p.numel()
                    p.data.copy_(flat[off:off + n].view(p.shape))
                    off += n
        # p.data writes bump neither p._version nor the optimiser's epoch, which key the W W^T / flipped / split-plane caches:
        # any forward that ran before this broadcast (warm-up, smoke, eval) must not leave stale copies behind on ranks != src
        from . import ops

        ops.bump_weights_epoch()

    # ------------------------------------------------------------------------------------------------------------------
    # gradient exchange
    # ------------------------------------------------------------------------------------------------------------------
    def flat_for(self, which: str) -> FlatGrads:
        return self.gen if which == "g" else self.disc

    def _scale(self, t: torch.Tensor, s: float):
        if t.is_cuda:
            from . import ops

            ops.call("dgmr_axpby", t.data_ptr(), None, t.data_ptr(), float(s), 0.0, t.numel(), ops._stream())
        else:
            t.mul_(s)

    def _nbuckets(self, fg: FlatGrads) -> int:
        return (fg.flat.numel() + self.chunk - 1) // self.chunk

    def _buckets_of(self, fg: FlatGrads, p) -> range:
        off = fg.offset[id(p)]
        return range(off // self.chunk, (off + p.numel() - 1) // self.chunk + 1)

    def begin(self, which: str):
        """Call right before the backward pass that fills network `which`'s gradients ('d' or 'g')."""
        if not self.exchange:
            return
        from . import ops

        self._active = which
        self._touches = []
        self._launched = {}
        self._locals = {}
        self._next = 0
        self._deviated = False
        self._main_stream = torch.cuda.current_stream() if torch.cuda.is_available() and self.gen.flat.is_cuda else None
        ops.set_grad_touch_hook(self._touch)

    def abort(self):
        """Drop the touch hook and the pass state without exchanging anything (a backward pass that raised): whatever runs backward
        next must not find this pass's hook.  Collectives already launched are waited for, so that no rank is left behind."""
        from . import ops

        ops.set_grad_touch_hook(None)
        self._active = None
        for work in self._launched.values():
            try:
                work.wait()
            except Exception:  # noqa: BLE001 - the backward pass already failed; its error is the one to surface
                pass
        self._launched = {}

    def _touch(self, p):
        which = self._active
        if which is None:
            return
        fg = self.flat_for(which)
        if id(p) not in fg.offset:
            return  # a parameter of the other network (its gradient is never exchanged in this pass)
        ps = self._passes[which]
        i = len(self._touches)
        self._touches.append(id(p))
        for b in self._buckets_of(fg, p):
            if b in self._launched:
                raise RuntimeError(f"gradient bucket {b} of network '{which}' was written (touch {i}) after its all-reduce had been "
                                   "launched: the backward pass no longer follows the recorded order")
        if ps.recorded is None or not self.overlap or self._deviated:
            return
        if i >= len(ps.recorded) or ps.recorded[i] != id(p):
            self._deviated = True  # the rest of this pass is reduced at sync()
            self.stats["deviations"] += 1
            return
        while self._next < len(ps.order) and i >= ps.last_touch[ps.order[self._next]] + MARGIN:
            self._launch(fg, ps.order[self._next], overlapped=True)
            self._next += 1

    def _launch(self, fg: FlatGrads, b: int, overlapped: bool):
        piece = fg.flat[b * self.chunk:(b + 1) * self.chunk]
        if piece.is_cuda:
            from . import ops

            # a touch may fire on the weight-gradient stream (grad_buffer() inside weight_grad()) or, with DGMR_BRANCH_STREAM=1, on the
            # branch stream: order the collective behind the stream begin() was called on, the CURRENT one, the device's default
            # stream and every side stream - whichever of them carries kernels that write this bucket
            main = torch.cuda.current_stream(piece.device)
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=piece.device)
            comm = self._comm_stream
            comm.wait_stream(main)
            comm.wait_stream(torch.cuda.default_stream(piece.device))
            if getattr(self, "_main_stream", None) is not None:
                comm.wait_stream(self._main_stream)
            for st in ops.side_streams(piece.device):  # the weight-gradient kernels run there
                comm.wait_stream(st)
            with torch.cuda.stream(comm):
                if self.check_exchange:
                    self._locals[b] = piece.clone()
                work = dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        else:
            if self.check_exchange:
                self._locals[b] = piece.clone()
            work = dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        self._launched[b] = work
        self.stats["overlapped_buckets" if overlapped else "late_buckets"] += 1

    def sync(self, which: str):
        """All-reduce (mean) the gradient buffer of network `which` ('g' or 'd'): launch whatever has not been launched during the
        backward pass, wait for every bucket, scale by 1/world.  Call after the backward pass (and after the weight-gradient
        streams have been joined), before ``optimizer.step()``."""
        if not self.exchange:
            return
        from . import ops

        fg = self.flat_for(which)
        ps = self._passes[which]
        if self._active != which:  # begin() was not called: plain post-backward reduction
            self._launched, self._locals, self._touches = {}, {}, []
        ops.set_grad_touch_hook(None)
        self._active = None
        nb = self._nbuckets(fg)
        # ONE canonical order on every rank: the recorded launch order once there is one (a rank that stopped launching early - a
        # deviation - continues exactly where the others went on), back to front before that (the recording pass itself)
        canonical = ps.order if (ps.recorded is not None and self.overlap) else list(range(nb - 1, -1, -1))
        for b in canonical:
            if b not in self._launched:
                self._launch(fg, b, overlapped=False)
                if ps.recorded is None:
                    self.stats["late_buckets_recording_step"] += 1
        for b, work in self._launched.items():
            work.wait()  # (RCCL: the current stream waits for the collective; gloo: the host does)
        if fg.flat.is_cuda and self._comm_stream is not None:
            torch.cuda.current_stream(fg.flat.device).wait_stream(self._comm_stream)
        if self.check_exchange:
            self._verify_exchange(fg)
        self._scale(fg.flat, 1.0 / self.world)
        if ps.recorded is None and self.overlap and self._touches:  # (a mismatch switches overlap off: checked once, not every step)
            self._record(fg, ps)

    def _record(self, fg: FlatGrads, ps: _Pass):
        recorded = list(self._touches)
        nb = self._nbuckets(fg)
        last = [-1] * nb
        by_id = {id(p): p for p in fg.params}
        for i, pid in enumerate(recorded):
            for b in self._buckets_of(fg, by_id[pid]):
                last[b] = i
        order = sorted(range(nb), key=lambda b: last[b])
        # the launch order IS the pairing of the collectives across ranks: it must be the same list everywhere.  Checked on LOCAL
        # values; the pass state is committed only on success - a caller that catches the error and goes on keeps exchanging
        # un-overlapped, back to front, which pairs by bucket index on every rank
        if self.exchange:
            # (seeded with the one process-level switch that moves the touch order - DGMR_WGRAD_DEFER fires the gradient-buffer touches at
            #  flush time: ranks that disagree on it must not overlap)
            from . import _streams

            h = 7 if _streams._DEFER_ON else 0
            for b in order:
                h = (h * 1000003 + b + 1) % 2147483629
            t = torch.tensor([h, -h], dtype=torch.int64, device=fg.flat.device if fg.flat.is_cuda else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.pg)
            if int(t[0]) != h or int(t[1]) != -h:
                self.overlap = False
                raise RuntimeError("the ranks recorded different gradient-bucket launch orders: the backward passes are not the same "
                                   "static graph on every rank; overlap has been switched off for this GradSync")
        ps.recorded = recorded
        ps.last_touch = last  # -1: a bucket nothing writes (dead parameters only): launchable from the start
        ps.order = order

    def _verify_exchange(self, fg: FlatGrads):
        """check_exchange: the reduced bucket must equal the sum over ranks of what each rank held when it launched the bucket (fp32
        addition of `world` values in rank order is what a tree / ring of two produces exactly; for more ranks: to 1e-6)."""
        for b, local in sorted(self._locals.items()):
            gathered = [torch.empty_like(local) for _ in range(self.world)]
            dist.all_gather(gathered, local, group=self.pg)
            total = gathered[0].clone()
            for g in gathered[1:]:
                total += g
            piece = fg.flat[b * self.chunk:(b + 1) * self.chunk]
            if self.world == 2:
                ok = torch.equal(piece, total)
            else:
                ok = torch.allclose(piece, total, rtol=1e-6, atol=1e-12)
            if not ok:
                raise RuntimeError(f"bucket {b}: all-reduced gradient differs from the sum of the ranks' local buckets "
                                   f"(max abs diff {(piece - total).abs().max().item():.3e}): a kernel wrote the bucket after its launch")
