"""Tile / split-K sweep for the sequential (h-part) ConvGRU convs through the C ABI (GPU only).

    python tools/gru_sweep.py [--prec=bf16x3] [--batch=16]

For each sampler level (8x8x384, 16x16x192, 32x32x96, 64x64x48 hidden maps at the paper configuration) the candidate conv of one
step (3x3, Cin = Cout = hidden channels, blend epilogue) is timed with every generic tile variant x split-K factor and with
the LDS-window kernel, via dgmr_conv_tune().  The library's own choice is the line marked `auto`.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from skillful_nowcasting_amd import ops
from skillful_nowcasting_amd._lib import call, load

LEVELS = [("gru1 8x8x384", 8, 384), ("gru2 16x16x192", 16, 192), ("gru3 32x32x96", 32, 96), ("gru4 64x64x48", 64, 48)]
VARIANTS = {0: "128x128", 1: "64x64", 2: "128x96", 3: "128x64", 4: "128x32"}


def bench(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


def main():
    load()
    prec, batch = "bf16x3", 16
    for a in sys.argv[1:]:
        if a.startswith("--prec="):
            prec = a.split("=")[1]
        if a.startswith("--batch="):
            batch = int(a.split("=")[1])
    ops.set_precision(prec)
    dev = "cuda"
    print(f"precision {prec}  batch {batch}", flush=True)
    for name, hw, ch in LEVELS:
        cx = 2 * ch  # the layer's weight has cx + ch input channels; the h part is the slice [cx, cx + ch)
        ct = cx + ch
        m = batch * hw * hw
        x = torch.randn(m * ch, device=dev)
        wt = torch.randn(ch * 9 * ct, device=dev) * 0.05
        y, pre, h, pu, add = (torch.randn(m * ch, device=dev) for _ in range(5))
        bias = torch.randn(ch, device=dev)
        scale = torch.full((1,), 0.5, device=dev)
        wsp = torch.empty(2 * ch * 9 * ch, device=dev, dtype=torch.int16)
        call("dgmr_split_weights", wt.data_ptr(), wsp.data_ptr(), ch * 9, ch, ct, cx, 2, 0, ops._stream())
        flops = 2.0 * m * 9 * ch * ch

        def fwd(use_split=True):
            ops._launch_conv(x, wt.data_ptr(), bias, scale, y, batch, 1, hw, hw, ch, ch, 1, 3, 3, w_cin=ct, w_coff=cx, addend=add,
                             epi_mode=ops.EPI_GRU_BLEND, gru_h=h, gru_pu=pu, pre_out=pre, w_split=wsp if use_split else None)

        print(f"--- {name}: M={m} K={9 * ch} N={ch}  ({flops / 1e9:.2f} GF)", flush=True)
        call("dgmr_conv_tune", -1, -1, -1, -1)
        us = bench(fwd)
        print(f"  auto                      {us:8.1f} us  {flops / us / 1e6:7.1f} TF", flush=True)
        for mode, label in ((1, "window kernel (registers)"), (3, "window kernel (LDS-DMA)  ")):
            call("dgmr_conv_tune", -1, -1, mode, -1)
            us = bench(fwd)
            print(f"  {label} {us:8.1f} us  {flops / us / 1e6:7.1f} TF", flush=True)
        nk = (9 * ch + 31) // 32
        for v, vname in VARIANTS.items():
            bn = int(vname.split("x")[1])
            if bn > 64 and bn > ch + 31:
                continue
            row = []
            for ks in (1, 2, 3, 4, 6, 8, 12, 16, 24):
                if ks > nk // 2:
                    break
                call("dgmr_conv_tune", v, ks, 0, -1)
                us = bench(fwd)
                row.append(f"S{ks}:{us:6.1f}")
            print(f"  {vname:8s} " + "  ".join(row), flush=True)
        call("dgmr_conv_tune", -1, -1, -1, -1)


if __name__ == "__main__":
    main()
