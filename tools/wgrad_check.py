"""Two weight-gradient kernels against each other through the C ABI (GPU only): same partial sums, timing of both.

    python tools/wgrad_check.py [--prec=bf16x3] [--modes=1,2] [--dbg=16] [shape substrings]

--modes: two dgmr_conv_tune wgrad_window values - 0 im2col, 1 one-role LDS-window kernel (wgrad_win.h), 2 / 3 wave-specialised one
(wgrad_ws.h) with three / four matrix waves (3 = the library's choice); default 1,3.
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from skillful_nowcasting_amd import ops
from skillful_nowcasting_amd._lib import WgradArgs, call, load

SHAPES = [
    # name, N, H, W (output), Cin, Cout, upsample, bn, groups
    ("small 2x32x32 40->96 g2", 2, 32, 32, 40, 96, False, True, 2),
    ("small up 2x64x64 32->48", 2, 64, 64, 32, 48, True, True, 1),
    ("small relu 3x32x64 96->192", 3, 32, 64, 96, 192, False, False, 3),
    ("small 16x16 2x16x16 64->96 g2", 2, 16, 16, 64, 96, False, True, 2),
    ("small up 16x16 3x16x16 32->64", 3, 16, 16, 32, 64, True, False, 1),
    ("up_g1.first T18", 288, 16, 16, 768, 768, True, True, 18),
    ("g2.first T18", 288, 16, 16, 384, 384, False, True, 18),
    ("up_g4.first T18", 288, 128, 128, 96, 96, True, True, 18),
    ("up_g4.last T18", 288, 128, 128, 96, 48, False, True, 18),
    ("up_g3.first T18", 288, 64, 64, 192, 192, True, True, 18),
    ("up_g3.last T18", 288, 64, 64, 192, 96, False, True, 18),
    ("g3.first T18", 288, 32, 32, 192, 192, False, True, 18),
    ("up_g2.last T18", 288, 32, 32, 384, 192, False, True, 18),
    ("spatD.d2.first f8", 256, 64, 64, 48, 96, False, False, 1),
    ("spatD.d1.last f8", 256, 128, 128, 48, 48, False, False, 1),
    ("gru4.h T18", 288, 64, 64, 48, 48, False, False, 18),
    ("gru3.x T18", 288, 32, 32, 192, 96, False, False, 18),
]


def bench(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


MODES = (1, 3)


def main():
    global MODES
    load()
    prec = "bf16x3"
    for a in sys.argv[1:]:
        if a.startswith("--prec="):
            prec = a.split("=")[1]
        if a.startswith("--dbg="):  # dgmr_debug_flags timing probes (16: wgrad_ws.h without its matrix work)
            call("dgmr_debug_flags", int(a.split("=")[1]))
        if a.startswith("--modes="):  # two dgmr_conv_tune wgrad_window values to compare
            MODES = tuple(int(v) for v in a.split("=")[1].split(","))
    ops.set_precision(prec)
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    dev = "cuda"
    torch.manual_seed(0)
    for name, n, h, w, cin, cout, up, bn, groups in SHAPES:
        if only and not any(o in name for o in only):
            continue
        hin, win = (h // 2, w // 2) if up else (h, w)
        x = torch.randn(n * hin * win * cin, device=dev)
        dy = torch.randn(n * h * w * cout, device=dev)
        a = torch.rand(groups * cin, device=dev) + 0.5
        b = torch.randn(groups * cin, device=dev) * 0.3
        m, k = n * h * w, 9 * cin
        res = {}
        nss = []
        for mode in MODES:
            call("dgmr_conv_tune", -1, -1, -1, mode)
            bias = torch.zeros(cout, device=dev)
            wa = WgradArgs()
            wa.x, wa.dy = x.data_ptr(), dy.data_ptr()
            wa.pre_a, wa.pre_b = (a.data_ptr(), b.data_ptr()) if bn else (None, None)
            wa.N, wa.D, wa.H, wa.W, wa.Cin, wa.Cout = n, 1, h, w, cin, cout
            wa.KD, wa.KH, wa.KW = 1, 3, 3
            wa.upsample, wa.pre_relu, wa.pre_group, wa.groups = int(up), int(not bn), n // groups, groups
            wa.bias_grad = bias.data_ptr()
            call("dgmr_conv_wgrad_plan", ctypes.byref(wa))
            ns = wa.nsplit
            nss.append(ns)
            partial = torch.full((ns, cout, k), float("nan"), device=dev)
            wa.partial = partial.data_ptr()
            call("dgmr_conv_wgrad", ctypes.byref(wa), ops._stream())
            torch.cuda.synchronize()
            per_group = partial.view(groups, ns // groups, cout, k).double().sum(1)
            wa.bias_grad = None
            ms = bench(lambda: call("dgmr_conv_wgrad", ctypes.byref(wa), ops._stream()))
            res[mode] = (per_group, bias.double(), ms)
        call("dgmr_conv_tune", -1, -1, -1, -1)
        g0, b0, t0 = res[MODES[0]]
        g1, b1, t1 = res[MODES[1]]
        flops = 2.0 * m * k * cout
        err = (g0 - g1).abs().max().item() / g0.abs().max().item()
        berr = (b0 - b1).abs().max().item() / b0.abs().max().item()
        nan = int(torch.isnan(g1).sum().item())
        print(f"{name:28s} ns={nss[0]:4d}/{nss[1]:4d}  A {t0*1e3:8.1f} us {flops/t0/1e9:6.1f} TF | B {t1*1e3:8.1f} us {flops/t1/1e9:6.1f} TF"
              f" | rel diff {err:.2e} bias {berr:.2e} nan {nan}", flush=True)


if __name__ == "__main__":
    main()
