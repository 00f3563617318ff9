"""Micro-benchmark of the conv kernels through the C ABI (GPU only).  python tools/conv_bench.py [--bwd]"""
import ctypes
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from skillful_nowcasting_amd import ops
from skillful_nowcasting_amd._lib import WgradArgs, call, load

SHAPES = [
    # name, N, D, H, W (output), Cin, Cout, k(d,h,w), upsample, bn
    ("up_g4.first t1", 16, 1, 128, 128, 96, 96, (1, 3, 3), True, True),
    ("up_g4.first T18", 288, 1, 128, 128, 96, 96, (1, 3, 3), True, True),
    ("up_g4.last T18", 288, 1, 128, 128, 96, 48, (1, 3, 3), False, True),
    ("up_g3.first t1", 16, 1, 64, 64, 192, 192, (1, 3, 3), True, True),
    ("up_g3.first T18", 288, 1, 64, 64, 192, 192, (1, 3, 3), True, True),
    ("up_g3.last T18", 288, 1, 64, 64, 192, 96, (1, 3, 3), False, True),
    ("g3.first T18", 288, 1, 32, 32, 192, 192, (1, 3, 3), False, True),
    ("up_g2.first t1", 16, 1, 32, 32, 384, 384, (1, 3, 3), True, True),
    ("up_g2.first T18", 288, 1, 32, 32, 384, 384, (1, 3, 3), True, True),
    ("g1.first t1", 16, 1, 8, 8, 768, 768, (1, 3, 3), False, True),
    ("g1.first T18", 288, 1, 8, 8, 768, 768, (1, 3, 3), False, True),
    ("c64 T18", 288, 1, 64, 64, 192, 64, (1, 3, 3), False, True),
    ("lat Cin32", 16, 1, 64, 64, 32, 64, (1, 3, 3), False, False),
    ("lat Cin64", 16, 1, 64, 64, 64, 64, (1, 3, 3), False, False),
    ("lat Cin128", 16, 1, 64, 64, 128, 64, (1, 3, 3), False, False),
    ("lat Cin256", 16, 1, 64, 64, 256, 64, (1, 3, 3), False, False),
    ("gru1.gate", 16, 1, 8, 8, 1152, 384, (1, 3, 3), False, False),
    ("gru1.h-only", 16, 1, 8, 8, 384, 384, (1, 3, 3), False, False),
    ("gru4.gate", 16, 1, 64, 64, 144, 48, (1, 3, 3), False, False),
    ("gru_1x1_4 T18", 288, 1, 64, 64, 48, 96, (1, 1, 1), False, False),
    ("tempD.d1.last 3d", 32, 22, 64, 64, 48, 48, (3, 3, 3), False, False),
    ("tempD.d2.first 3d", 32, 11, 32, 32, 48, 96, (3, 3, 3), False, False),
    ("tempD.d2.last 3d", 32, 11, 32, 32, 96, 96, (3, 3, 3), False, False),
    ("tempD.d1.first 3d", 32, 22, 64, 64, 4, 48, (3, 3, 3), False, False),
    ("spatD.d2.first f8", 256, 1, 32, 32, 48, 96, (1, 3, 3), False, False),
    ("spatD.d5 f8", 256, 1, 4, 4, 384, 768, (1, 3, 3), False, False),
    ("gru1.h-step B4", 4, 1, 8, 8, 384, 384, (1, 3, 3), False, False),
    ("gru2.h-step B4", 4, 1, 16, 16, 192, 192, (1, 3, 3), False, False),
    ("gru3.h-step B4", 4, 1, 32, 32, 96, 96, (1, 3, 3), False, False),
    ("gru4.h-step B4", 4, 1, 64, 64, 48, 48, (1, 3, 3), False, False),
    ("gru3.h-step B16", 16, 1, 32, 32, 96, 96, (1, 3, 3), False, False),
    ("gru2.x-part T18B4", 72, 1, 16, 16, 384, 192, (1, 3, 3), False, False),
    # pointwise convs of the sampler at the full draw batch (HBM-bound)
    ("pw gru_1x1_4 full", 1728, 1, 64, 64, 48, 96, (1, 1, 1), False, False),
    ("pw gru_1x1_3 full", 1728, 1, 32, 32, 96, 192, (1, 1, 1), False, False),
    ("pw up_g4.conv_1x1 full", 1728, 1, 64, 64, 96, 48, (1, 1, 1), False, False),
    ("pw up_g3.conv_1x1 full", 1728, 1, 32, 32, 192, 96, (1, 1, 1), False, False),
    # recurrent step convs with the six draws batched (96 samples per step)
    ("gru4.h-step B96", 96, 1, 64, 64, 48, 48, (1, 3, 3), False, False),
    ("gru3.h-step B96", 96, 1, 32, 32, 96, 96, (1, 3, 3), False, False),
    ("gru2.h-step B96", 96, 1, 16, 16, 192, 192, (1, 3, 3), False, False),
    ("gru1.h-step B96", 96, 1, 8, 8, 384, 384, (1, 3, 3), False, False),
    # weight gradient of the upsampling convs as a 1x1 problem on the 9 pair-summed planes of dy (Cout' = 9 Cout)
    ("z9 up_g1", 288, 1, 8, 8, 768, 3456, (1, 1, 1), False, True),
    ("z9 up_g2", 288, 1, 16, 16, 384, 1728, (1, 1, 1), False, True),
    ("z9 up_g3", 288, 1, 32, 32, 192, 864, (1, 1, 1), False, True),
    ("z9 up_g4", 288, 1, 64, 64, 96, 432, (1, 1, 1), False, True),
    # the same layers as they run now (real channel counts)
    ("real up_g1.first", 288, 1, 16, 16, 768, 384, (1, 3, 3), True, True),
    ("real up_g2.first", 288, 1, 32, 32, 384, 192, (1, 3, 3), True, True),
    ("real up_g3.first", 288, 1, 64, 64, 192, 96, (1, 3, 3), True, True),
    ("real up_g4.first", 288, 1, 128, 128, 96, 48, (1, 3, 3), True, True),
    ("real up_g4.last", 288, 1, 128, 128, 48, 48, (1, 3, 3), False, True),
    # the upsampling convs as the step runs them: all six draws in one batch, 108 spectral-norm call groups (--groups=108)
    ("full up_g1.first", 1728, 1, 16, 16, 768, 768, (1, 3, 3), True, True),
    ("full up_g2.first", 1728, 1, 32, 32, 384, 384, (1, 3, 3), True, True),
    ("full up_g3.first", 1728, 1, 64, 64, 192, 192, (1, 3, 3), True, True),
    ("full up_g4.first", 1728, 1, 128, 128, 96, 96, (1, 3, 3), True, True),
    # launch-size sweep of the dominant layer (308 TF at N = 1728 against 484 at N = 288 / 576 inside the step)
    ("nsweep up_g4.first N576", 576, 1, 128, 128, 96, 96, (1, 3, 3), True, True),
    ("nsweep up_g4.first N864", 864, 1, 128, 128, 96, 96, (1, 3, 3), True, True),
    # plain 3x3 layers of the sampler at the full draw batch
    ("full g4.first", 1728, 1, 64, 64, 96, 96, (1, 3, 3), False, True),
    ("full up_g3.last", 1728, 1, 64, 64, 192, 96, (1, 3, 3), False, True),
    ("full g3.first", 1728, 1, 32, 32, 192, 192, (1, 3, 3), False, True),
    ("full g2.first", 1728, 1, 16, 16, 384, 384, (1, 3, 3), False, True),
    ("full up_g4.last", 1728, 1, 128, 128, 96, 48, (1, 3, 3), False, True),
    ("half up_g4.last", 576, 1, 128, 128, 96, 48, (1, 3, 3), False, True),
]


def bench(fn, iters=int(os.environ.get("CONV_BENCH_ITERS", "10"))):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


GROUPS = 1


def main():
    global GROUPS
    for a_ in sys.argv[1:]:
        if a_.startswith("--groups="):
            GROUPS = int(a_.split("=")[1])
    load()
    dev = "cuda"
    for a in sys.argv[1:]:
        if a.startswith("--prec="):
            ops.set_precision(a.split("=")[1])
        if a.startswith("--tune="):  # variant,ksplit,window for dgmr_conv_tune
            call("dgmr_conv_tune", *[int(v) for v in a.split("=")[1].split(",")])
        if a.startswith("--dbg="):  # kernel-phase timing switches: 1 no epilogue, 2 one halo only, 3 both (dgmr_debug_flags)
            call("dgmr_debug_flags", int(a.split("=")[1]))
    print("precision:", ops.get_precision(), flush=True)
    only = [a for a in sys.argv[1:] if not a.startswith("--")] + [a.split("=", 1)[1].replace("+", " ") for a in sys.argv[1:] if a.startswith("--shape=")]
    
    for name, n, d, h, w, cin, cout, ks, up, bn in SHAPES:
        if only and not any(o in name for o in only):
            continue
        kd, kh, kw = ks
        hin, win = (h // 2, w // 2) if up else (h, w)
        x = torch.randn(n * d * hin * win * cin, device=dev)
        wt = torch.randn(cout * kd * kh * kw * cin, device=dev) * 0.05
        y = torch.empty(n * d * h * w * cout, device=dev)
        bias = torch.randn(cout, device=dev)
        scale = torch.full((1,), 0.5, device=dev)
        a = torch.rand(cin, device=dev) + 0.5
        b = torch.randn(cin, device=dev) * 0.1
        flops = 2.0 * n * d * h * w * cout * cin * kd * kh * kw

        wsp = None
        if ops.get_precision() != "f32" and ks in ((1, 3, 3), (3, 3, 3)) and cin % 8 == 0 and "--nowin" not in sys.argv:
            wsp = torch.empty(2 * wt.numel(), device=dev, dtype=torch.int16)
            call("dgmr_split_weights", wt.data_ptr(), wsp.data_ptr(), cout * kd * 9, cin, 0, 0, 2, 0, ops._stream())

        wph0 = None
        if "--phases-only" in sys.argv and up and wsp is not None and ks == (1, 3, 3):  # PMC runs: only the path the product takes
            sums0 = torch.empty(16 * cout * cin, device=dev)
            call("dgmr_upsample_phase_weights", wt.data_ptr(), sums0.data_ptr(), cout, cin, ops._stream())
            wph0 = torch.empty(2 * sums0.numel(), device=dev, dtype=torch.int16)
            call("dgmr_split_weights", sums0.data_ptr(), wph0.data_ptr(), 16 * cout, cin, 0, 0, 2, 0, ops._stream())

        def fwd():
            ops._launch_conv(x, wt.data_ptr(), bias, scale, y, n, d, h, w, cin, cout, kd, kh, kw, upsample=up,
                             pre_a=a if bn else None, pre_b=b if bn else None, pre_group=n, w_split=wsp, w_phase=wph0)

        ms = bench(fwd)
        line = f"{name:22s} M={n*d*h*w:8d} K={cin*kd*kh*kw:6d} N={cout:4d}  fwd {ms*1e3:9.1f} us {flops/ms/1e9:7.1f} TF"
        if up and wsp is not None and ks == (1, 3, 3) and wph0 is None:  # the same conv as four 2x2 phase convs on the low-resolution input
            sums = torch.empty(16 * cout * cin, device=dev)
            call("dgmr_upsample_phase_weights", wt.data_ptr(), sums.data_ptr(), cout, cin, ops._stream())
            wph = torch.empty(2 * sums.numel(), device=dev, dtype=torch.int16)
            call("dgmr_split_weights", sums.data_ptr(), wph.data_ptr(), 16 * cout, cin, 0, 0, 2, 0, ops._stream())
            y_direct = y.clone()
            y.zero_()

            def fwd_ph():
                ops._launch_conv(x, wt.data_ptr(), bias, scale, y, n, d, h, w, cin, cout, kd, kh, kw, upsample=up,
                                 pre_a=a if bn else None, pre_b=b if bn else None, pre_group=n, w_split=wsp, w_phase=wph)

            msp = bench(fwd_ph)
            diff = (y - y_direct).abs().max().item() / y_direct.abs().max().item()
            line += f" | phases {msp*1e3:9.1f} us {flops/msp/1e9:7.1f} TF (algorithmic) diff {diff:.1e}"
        if "--bwd" in sys.argv:
            m = n * d * h * w
            k = cin * kd * kh * kw
            ns = ops.call_nsplit(m, cout, k)
            partial = torch.empty(ns * cout * k, device=dev)
            wa = WgradArgs()
            wa.x, wa.dy, wa.partial = x.data_ptr(), y.data_ptr(), partial.data_ptr()
            wa.pre_a, wa.pre_b = (a.data_ptr(), b.data_ptr()) if bn else (None, None)
            wa.N, wa.D, wa.H, wa.W, wa.Cin, wa.Cout = n, d, h, w, cin, cout
            wa.KD, wa.KH, wa.KW = kd, kh, kw
            wa.upsample, wa.pre_relu, wa.pre_group, wa.nsplit = int(up), 0, n, ns
            if GROUPS > 1:
                wa.groups = GROUPS
                call("dgmr_conv_wgrad_plan", ctypes.byref(wa))
                ns = wa.nsplit
                partial = torch.empty(ns * cout * k, device=dev)
                wa.partial = partial.data_ptr()

            def wg():
                call("dgmr_conv_wgrad", ctypes.byref(wa), ops._stream())

            ms2 = bench(wg)
            line += f" | wgrad ns={ns:4d} {ms2*1e3:9.1f} us {flops/ms2/1e9:7.1f} TF"
            skip_old = os.environ.get("CONV_BENCH_SKIP_OLD") == "1"  # PMC runs: only the paths the product takes
            if up and wsp is not None and ks == (1, 3, 3) and not skip_old:  # weight gradient through the 9 pair-summed planes (1x1 problem)
                z9 = torch.empty(n * (h // 2) * (w // 2) * 9 * cout, device=dev)
                wz = WgradArgs()
                wz.x, wz.dy = x.data_ptr(), z9.data_ptr()
                wz.pre_a, wz.pre_b = (a.data_ptr(), b.data_ptr()) if bn else (None, None)
                wz.N, wz.D, wz.H, wz.W, wz.Cin, wz.Cout = n, 1, h // 2, w // 2, cin, 9 * cout
                wz.KD, wz.KH, wz.KW, wz.upsample, wz.pre_relu, wz.pre_group, wz.groups = 1, 1, 1, 0, 0, n, GROUPS
                call("dgmr_conv_wgrad_plan", ctypes.byref(wz))
                pz = torch.empty(wz.nsplit * 9 * cout * cin, device=dev)
                wz.partial = pz.data_ptr()

                def zb():
                    call("dgmr_upsample_wgrad_sums", y.data_ptr(), z9.data_ptr(), n, h // 2, w // 2, cout, ops._stream())

                def zg():
                    call("dgmr_conv_wgrad", ctypes.byref(wz), ops._stream())

                t1, t2 = bench(zb), bench(zg)
                gold = partial.view(ns, -1).sum(0)
                gnew = pz.view(wz.nsplit, -1).sum(0)
                diff = (gnew - gold).abs().max().item() / gold.abs().max().item()
                line += f" | z9: sums {t1*1e3:7.1f} us + 1x1 wgrad ns={wz.nsplit} {t2*1e3:7.1f} us diff {diff:.1e}"
            if up and wsp is not None:  # data gradient of the upsampling conv: conv at full resolution + 2x2 sum, vs one pooled pass
                dx = torch.empty_like(x)
                hi = torch.empty(n * h * w * cin, device=dev)
                wflip = torch.randn(cin * 9 * cout, device=dev) * 0.05
                wfs = torch.empty(2 * wflip.numel(), device=dev, dtype=torch.int16)
                call("dgmr_split_weights", wflip.data_ptr(), wfs.data_ptr(), cin * 9, cout, 0, 0, 2, 0, ops._stream())
                sums = torch.empty(16 * cout * cin, device=dev)
                call("dgmr_pool2_phase_weights", wflip.data_ptr(), sums.data_ptr(), cin, cout, ops._stream())
                wpl = torch.empty(2 * sums.numel(), device=dev, dtype=torch.int16)
                call("dgmr_split_weights", sums.data_ptr(), wpl.data_ptr(), 16 * cin, cout, 0, 0, 2, 0, ops._stream())

                def dg_old():
                    ops._launch_conv(y, wflip.data_ptr(), None, scale, hi, n, d, h, w, cout, cin, kd, kh, kw, w_split=wfs)
                    call("dgmr_pool_fwd", hi.data_ptr(), None, dx.data_ptr(), n, d, h, w, cin, 1, 1.0, x.data_ptr(),
                         a.data_ptr() if bn else None, b.data_ptr() if bn else None, n, ops._stream())

                ms3 = 0.0 if skip_old else bench(dg_old)
                dx_old = dx.clone()
                dx.zero_()

                def dg_new():
                    r = ops._launch_conv(y, wflip.data_ptr(), None, scale, dx, n, d, h, w, cout, cin, kd, kh, kw, mask_src=x,
                                         mask_a=a if bn else None, mask_b=b if bn else None, mask_group=n, w_split=wfs, w_phase=wpl,
                                         pool2=True)
                    assert r is not NotImplemented

                ms4 = bench(dg_new)
                diff = (dx - dx_old).abs().max().item() / max(dx_old.abs().max().item(), 1e-30)
                line += f" | dgrad conv+pool {ms3*1e3:9.1f} us, pooled pass {ms4*1e3:9.1f} us {flops/ms4/1e9:7.1f} TF diff {diff:.1e}"
            if not up:
                dx = torch.empty_like(x)
                wflip = torch.empty_like(wt)

                def dg():
                    ops._launch_conv(y, wflip.data_ptr(), None, scale, dx, n, d, h, w, cout, cin, kd, kh, kw, mask_src=x,
                                     mask_a=a if bn else None, mask_b=b if bn else None, mask_group=n)

                ms3 = bench(dg)
                line += f" | dgrad {ms3*1e3:9.1f} us {flops/ms3/1e9:7.1f} TF"
        print(line, flush=True)


if __name__ == "__main__":
    main()
