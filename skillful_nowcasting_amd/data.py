"""The data path on either side of the training step (mirror of train/run.py:114-158 of the reference).

The reference's `TFDataset.__getitem__` takes a dataset row's `radar_frames` ([T_all, H, W, C], channels last), keeps the LAST
4 + 18 frames (targets aligned to the end of the window, inputs right before them) and moves the channel axis:
[T, H, W, C] -> [T, C, H, W].  That is all the arithmetic there is; what matters on an MI355X is that the host side keeps up with
~300 frames/s/GPU without stalling the step:

  * rows are copied into PINNED staging buffers in their storage dtype (uint8 / int16 / float16 / float32 - a uint8 frame is a
    quarter of the PCIe bytes of fp32),
  * two staging slots alternate: while the step runs on batch k, batch k+1 is uploaded on a side stream
    (`non_blocking` copies from pinned memory), converted to fp32 and laid out [B, T, C, H, W] on the device,
  * the consumer only waits on an event, never on the host; the PRODUCER waits (on the host) for a slot's previous upload before
    it overwrites that slot - a `non_blocking` copy reads the pinned buffer asynchronously, and nothing else stops the host from
    running two batches ahead of the copy engine.

No HIP kernel of ours is involved: layout moves and dtype conversion are torch copy kernels (plumbing, not the hot path).
"""
from __future__ import annotations

from typing import Iterable, Iterator, Optional, Tuple

import numpy as np
import torch

NUM_INPUT_FRAMES = 4
NUM_TARGET_FRAMES = 18


def extract_input_and_target_frames(radar_frames, num_input_frames: int = NUM_INPUT_FRAMES, num_target_frames: int = NUM_TARGET_FRAMES):
    """train/run.py:118-123: targets are the last `num_target_frames` frames of the window, inputs the frames right before them."""
    input_frames = radar_frames[-num_target_frames - num_input_frames: -num_target_frames]
    target_frames = radar_frames[-num_target_frames:]
    return input_frames, target_frames


def to_model_layout(frames):
    """[T, H, W, C] -> [T, C, H, W] (train/run.py:156-158, `np.moveaxis(x, [0, 1, 2, 3], [0, 2, 3, 1])`); numpy or torch."""
    if isinstance(frames, np.ndarray):
        return np.moveaxis(frames, [0, 1, 2, 3], [0, 2, 3, 1])
    return frames.permute(0, 3, 1, 2)


def row_to_sample(row, num_input_frames: int = NUM_INPUT_FRAMES, num_target_frames: int = NUM_TARGET_FRAMES):
    """What `TFDataset.__getitem__` returns for a dataset row (a mapping with a `radar_frames` entry, or the array itself)."""
    frames = row["radar_frames"] if isinstance(row, dict) else row
    x, y = extract_input_and_target_frames(frames, num_input_frames, num_target_frames)
    return to_model_layout(x), to_model_layout(y)


class RadarBatchLoader:
    """Iterate `(images [B,4,C,H,W], future [B,T,C,H,W])` fp32 device batches from an iterable of rows (`radar_frames` arrays
    [T_all, H, W, C] of any real dtype), double-buffered through pinned memory and a copy stream.

    `scale` / `offset`: optional affine applied after the conversion to fp32 (e.g. 1/32 for the 1/32-mm/h integer encoding of the
    NIMROD composites).  `device=None` or a CPU device: same semantics without streams (used by the CPU tests).
    `drop_last`: discard a final partial batch (the reference's loader yields whatever the dataset yields).
    """

    def __init__(self, rows: Iterable, batch_size: int, device=None, num_input_frames: int = NUM_INPUT_FRAMES,
                 num_target_frames: int = NUM_TARGET_FRAMES, scale: float = 1.0, offset: float = 0.0, drop_last: bool = True):
        self.rows, self.batch_size = rows, int(batch_size)
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.n_in, self.n_out = num_input_frames, num_target_frames
        self.scale, self.offset, self.drop_last = float(scale), float(offset), drop_last
        self._slots = [None, None]
        self._slot_uploaded = [None, None]  # event recorded right after the H2D copy out of each slot
        self._stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None

    # -- host side: rows -> one staging slot ------------------------------------------------------------------------------
    def _stage(self, batch_rows, slot: int):
        t = self.n_in + self.n_out
        first = np.asarray(batch_rows[0]["radar_frames"] if isinstance(batch_rows[0], dict) else batch_rows[0])
        shape = (len(batch_rows), t) + tuple(first.shape[1:])  # [B, T, H, W, C] in the rows' dtype
        dtype = torch.from_numpy(first[:1]).dtype
        if self._slot_uploaded[slot] is not None:
            self._slot_uploaded[slot].synchronize()  # the previous upload from this slot has left the pinned buffer
            self._slot_uploaded[slot] = None
        buf = self._slots[slot]
        if buf is None or tuple(buf.shape) != shape or buf.dtype != dtype:
            buf = torch.empty(shape, dtype=dtype, pin_memory=self.device.type == "cuda")
            self._slots[slot] = buf
        for i, row in enumerate(batch_rows):
            frames = np.asarray(row["radar_frames"] if isinstance(row, dict) else row)
            if frames.shape[0] < t:
                raise ValueError(f"row {i}: {frames.shape[0]} frames, need at least {t}")
            buf[i].copy_(torch.from_numpy(np.ascontiguousarray(frames[-t:])))  # the last 4 + T frames of the window
        return buf

    # -- device side: staging slot -> fp32 [B, T, C, H, W] -----------------------------------------------------------------
    def _upload(self, buf, slot: int):
        if self._stream is None:
            dev = buf.to(self.device)
            return self._finish(dev), None
        with torch.cuda.stream(self._stream):
            dev = buf.to(self.device, non_blocking=True)
            copied = torch.cuda.Event()
            copied.record(self._stream)
            self._slot_uploaded[slot] = copied
            out = self._finish(dev)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        return out, ev

    def _finish(self, dev):
        x = dev.permute(0, 1, 4, 2, 3).float()  # [B, T, H, W, C] -> [B, T, C, H, W], fp32
        if self.scale != 1.0 or self.offset != 0.0:
            x = x * self.scale + self.offset
        x = x.contiguous()
        return x[:, :self.n_in], x[:, self.n_in:]

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        pending: Optional[tuple] = None
        chunk, slot = [], 0
        for row in self.rows:
            chunk.append(row)
            if len(chunk) < self.batch_size:
                continue
            nxt = self._upload(self._stage(chunk, slot), slot)
            chunk, slot = [], slot ^ 1
            if pending is not None:
                yield self._ready(pending)
            pending = nxt
        if chunk and not self.drop_last:
            nxt = self._upload(self._stage(chunk, slot), slot)
            if pending is not None:
                yield self._ready(pending)
            pending = nxt
        if pending is not None:
            yield self._ready(pending)

    def _ready(self, item):
        (images, future), ev = item
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
            for t_ in (images, future):  # the tensors were produced on the copy stream: tell the allocator about their consumer
                t_.record_stream(torch.cuda.current_stream(self.device))
        return images, future
