"""Does any kernel WRITE outside the tensor it was given?  Every torch.empty / empty_like / zeros of the step is carved out of a larger
allocation with a 64 KB guard band in front and behind, filled with a bit pattern; every tensor is kept alive until the step is over (so
that no memory is recycled and a stray write cannot hide in a freed block), then all guard bands are checked.
(Found the reason for a run-to-run difference that showed only when two processes shared the GPU: the caching allocator's layout then
differs from run to run, so a stray write lands in another neighbour every time.)   python tools/guard_probe.py [precision] [paper]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

GUARD = 64 * 1024  # bytes, each side
PATTERN = 0x5A
_empty = torch.empty
ON = [False]
LIVE = []  # (raw uint8 allocation, nbytes of the payload, description)


def guarded(shape, dtype, device, memory_format=None, what=""):
    numel = 1
    for s in shape:
        numel *= int(s)
    nbytes = numel * torch.empty((), dtype=dtype).element_size()
    pad = (-nbytes) % 256
    raw = _empty(GUARD + nbytes + pad + GUARD, dtype=torch.uint8, device=device)
    raw[:GUARD].fill_(PATTERN)
    raw[GUARD + nbytes:].fill_(PATTERN)
    t = raw[GUARD:GUARD + nbytes].view(dtype).view(tuple(int(s) for s in shape)) if numel else _empty(shape, dtype=dtype, device=device)
    if memory_format in (torch.channels_last, torch.channels_last_3d) and numel:
        perm = (0, 2, 3, 1) if len(shape) == 4 else (0, 2, 3, 4, 1)
        inv = (0, 3, 1, 2) if len(shape) == 4 else (0, 4, 1, 2, 3)
        t = raw[GUARD:GUARD + nbytes].view(dtype).view(tuple(int(shape[i]) for i in perm)).permute(inv)
    LIVE.append((raw, nbytes, what))
    return t


def _shape_of(args):
    if len(args) == 1 and isinstance(args[0], (tuple, list, torch.Size)):
        return tuple(args[0])
    return tuple(args)


def p_empty(*args, **kw):
    dev = kw.get("device")
    if not ON[0] or dev is None or not str(dev).startswith("cuda") or kw.get("pin_memory"):
        return _empty(*args, **kw)
    return guarded(_shape_of(args), kw.get("dtype") or torch.get_default_dtype(), dev, kw.get("memory_format"), "empty")


_empty_like, _zeros = torch.empty_like, torch.zeros


def p_empty_like(t, **kw):
    if not ON[0] or not t.is_cuda or kw.get("dtype") not in (None, t.dtype) or not t.numel():
        return _empty_like(t, **kw)
    mf = kw.get("memory_format")
    if mf in (None, torch.preserve_format):
        if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous():
            mf = torch.channels_last
        elif t.dim() == 5 and t.is_contiguous(memory_format=torch.channels_last_3d) and not t.is_contiguous():
            mf = torch.channels_last_3d
        elif not t.is_contiguous():
            return _empty_like(t, **kw)
    return guarded(tuple(t.shape), t.dtype, t.device, mf, "empty_like")


def p_zeros(*args, **kw):
    dev = kw.get("device")
    if not ON[0] or dev is None or not str(dev).startswith("cuda"):
        return _zeros(*args, **kw)
    t = guarded(_shape_of(args), kw.get("dtype") or torch.get_default_dtype(), dev, None, "zeros")
    return t.zero_()


torch.empty, torch.empty_like, torch.zeros = p_empty, p_empty_like, p_zeros

import skillful_nowcasting_amd as S  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "mixed"
paper = len(sys.argv) > 2 and sys.argv[2] == "paper"
KW = dict(forecast_steps=18, output_shape=256, latent_channels=768, context_channels=384, generation_steps=6) if paper else \
    dict(forecast_steps=2, output_shape=128, latent_channels=384, context_channels=192, generation_steps=2)
B = 1 if paper else 2
S.set_precision(prec)
torch.manual_seed(7)
model = S.DGMR(**KW).to("cuda")
hw = KW["output_shape"]
x = torch.rand(B, 4, 1, hw, hw, device="cuda")
y = torch.rand(B, KW["forecast_steps"], 1, hw, hw, device="cuda")
for step in range(2):
    LIVE.clear()
    ON[0] = True
    model.training_step((x, y), step)
    torch.cuda.synchronize()
    ON[0] = False
    bad = 0
    for raw, nbytes, what in LIVE:
        front, back = raw[:GUARD], raw[GUARD + nbytes:]
        fb, bb = (front != PATTERN), (back != PATTERN)
        if bool(fb.any()) or bool(bb.any()):
            bad += 1
            if bad <= 12:
                fi, bi = fb.nonzero().flatten(), bb.nonzero().flatten()
                print(f"  STRAY WRITE next to a {what} tensor of {nbytes} bytes: {fi.numel()} bytes in the front guard "
                      f"(offsets {(-GUARD + int(fi.min())) if fi.numel() else None} .. {(-GUARD + int(fi.max())) if fi.numel() else None}), "
                      f"{bi.numel()} behind it (offsets +{int(bi.min()) if bi.numel() else None} .. +{int(bi.max()) if bi.numel() else None})")
    print(f"step {step}: {len(LIVE)} guarded allocations, {sum(r.numel() for r, _, _ in LIVE) / 2**30:.2f} GiB, {bad} with a damaged guard band", flush=True)
