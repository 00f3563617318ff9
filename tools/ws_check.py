"""Wave-specialised window conv (conv_win_ws.h, dgmr_conv_tune window = 7) against the one-role LDS-DMA kernel with the same block
shapes (window = 6) and against the library's own choice (-1 with DGMR_WS_AUTO=0 semantics: window = 3 family incl. 256-pixel tiles),
through the C ABI on the GPU:

  * y must be BIT-IDENTICAL to window = 6 (same MFMA sequence and epilogue expressions per element);
  * the fused BatchNorm partial sums (stats_out) are summed in another order: compared after folding all rows in float64;
  * timing of all three (us per launch, algorithmic TF).

    python tools/ws_check.py [--prec=bf16x3|bf16] [--big] [--time-only] [name ...]

Small cases cover every mode (plain / phase / pooled), both column blocks (96 / 128), fused operands (residual, half-resolution
residual, relu mask with and without BatchNorm affine), ragged item counts, 16- and 8-pixel-wide maps, Cin tails; --big adds the
paper-configuration layers at the generator pass's batch."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.pop("DGMR_WS_AUTO", None)  # "lib" below = the library's choice WITHOUT the wave-specialised kernel (read once, at load)
import torch

from skillful_nowcasting_amd import ops
from skillful_nowcasting_amd._lib import call, load

# name, N, H, W (output map), Cin, Cout, mode (plain | phase | pooled), bn prologue, fused operand (None | res | res_up | mask | maskbn), stats, groups
SMALL = [
    ("plain96 res", 6, 64, 64, 96, 96, "plain", True, "res", True, 3),
    ("plain96 nostat", 5, 32, 32, 96, 96, "plain", False, None, False, 1),
    ("plain96 mask", 6, 64, 64, 128, 96, "plain", False, "maskbn", True, 2),
    ("plain192 resup", 4, 64, 64, 192, 192, "plain", True, "res_up", True, 2),
    ("plain128 res", 7, 32, 32, 128, 128, "plain", True, "res", True, 1),
    ("plain384 mask", 6, 16, 16, 384, 384, "plain", False, "mask", True, 3),
    ("plain768 8x8", 16, 8, 8, 768, 768, "plain", True, None, True, 4),
    ("plain96 tail", 6, 32, 32, 80, 96, "plain", True, "res", True, 1),
    ("plain Cout200", 6, 32, 32, 96, 200, "plain", True, None, True, 1),
    ("phase96", 6, 64, 64, 96, 96, "phase", True, None, True, 3),
    ("phase192", 5, 32, 32, 192, 192, "phase", True, None, True, 1),
    ("phase384", 6, 16, 16, 384, 384, "phase", True, None, True, 2),
    ("pooled96", 6, 64, 64, 96, 96, "pooled", False, "maskbn", True, 3),
    ("pooled192", 5, 32, 32, 192, 192, "pooled", False, "maskbn", True, 1),
    ("pooled96 nomask", 4, 32, 32, 96, 96, "pooled", False, None, False, 1),
]
BIG = [
    ("full g4.first", 1728, 64, 64, 96, 96, "plain", True, None, True, 108),
    ("full g4.last", 1728, 64, 64, 96, 96, "plain", True, "res", True, 108),
    ("full g4 dgrad", 1728, 64, 64, 96, 96, "plain", False, "maskbn", True, 108),
    ("full up_g3.last", 1728, 64, 64, 192, 96, "plain", True, "res_up", True, 108),
    ("full g3.first", 1728, 32, 32, 192, 192, "plain", True, None, True, 108),
    ("full g2.first", 1728, 16, 16, 384, 384, "plain", True, None, True, 108),
    ("full g1.first", 1728, 8, 8, 768, 768, "plain", True, None, True, 108),
    ("full up_g4.first", 1728, 64, 64, 96, 96, "phase", True, None, True, 108),
    ("full up_g3.first", 1728, 32, 32, 192, 192, "phase", True, None, True, 108),
    ("full up_g2.first", 1728, 16, 16, 384, 384, "phase", True, None, True, 108),
    ("full up_g4 dgrad", 1728, 64, 64, 96, 96, "pooled", False, "maskbn", True, 108),
    ("full up_g3 dgrad", 1728, 32, 32, 192, 192, "pooled", False, "maskbn", True, 108),
    ("B16 g4.first", 288, 64, 64, 96, 96, "plain", True, None, True, 18),
    ("B16 up_g4.first", 288, 64, 64, 96, 96, "phase", True, None, True, 18),
]


def bench(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def run_case(case, prec="bf16x3", time_it=False, dev="cuda"):
    """One case through dgmr_conv_tune window = 6 (one-role reference), 7 (wave-specialised) and -1 (the library's choice):
    -> dict(exact, nan, stats_rel, vs_lib, times)."""
    name, n, h, w, cin, cout, mode, bn, eop, stats, groups = case
    planes = 2
    torch.manual_seed(sum(map(ord, name)) % 1000)
    # `h, w`: the map the window kernel walks = the LOW-resolution map of a phase (input) / pooled (output) conv
    if mode == "phase":
        ih, iw, oh, ow = h, w, 2 * h, 2 * w
    elif mode == "pooled":
        ih, iw, oh, ow = 2 * h, 2 * w, h, w
    else:
        ih, iw, oh, ow = h, w, h, w
    x = torch.randn(n * ih * iw * cin, device=dev)
    wt = torch.randn(cout * 9 * cin, device=dev) * 0.05
    bias = torch.randn(cout, device=dev)
    ng = groups
    scale = torch.rand(ng, device=dev) + 0.5
    a = (torch.rand(ng * cin, device=dev) + 0.5) if bn else None
    b = (torch.randn(ng * cin, device=dev) * 0.1) if bn else None
    wsp = torch.empty(planes * wt.numel(), device=dev, dtype=torch.int16)
    call("dgmr_split_weights", wt.data_ptr(), wsp.data_ptr(), cout * 9, cin, 0, 0, planes, 0, ops._stream())
    wph = None
    if mode in ("phase", "pooled"):
        sums = torch.empty(16 * cout * cin, device=dev)
        call("dgmr_upsample_phase_weights" if mode == "phase" else "dgmr_pool2_phase_weights", wt.data_ptr(), sums.data_ptr(), cout, cin, ops._stream())
        wph = torch.empty(planes * sums.numel(), device=dev, dtype=torch.int16)
        call("dgmr_split_weights", sums.data_ptr(), wph.data_ptr(), 16 * cout, cin, 0, 0, planes, 0, ops._stream())
    kw = dict(pre_a=a, pre_b=b, pre_group=n // ng, scale_group=n // ng, w_split=wsp, w_phase=wph, want_stats=stats)
    if mode == "phase":
        kw["upsample"] = True
    if mode == "pooled":
        kw["pool2"] = True
    if eop == "res":
        kw["residual"] = torch.randn(n * oh * ow * cout, device=dev)
    elif eop == "res_up":
        kw["residual"] = torch.randn(n * (oh // 2) * (ow // 2) * cout, device=dev)
        kw["residual_up"] = True
    elif eop in ("mask", "maskbn"):
        kw["mask_src"] = torch.randn(n * oh * ow * cout, device=dev)
        if eop == "maskbn":
            kw["mask_a"] = torch.rand(ng * cout, device=dev) + 0.5
            kw["mask_b"] = torch.randn(ng * cout, device=dev) * 0.3
            kw["mask_group"] = n // ng
    # conv extent as dgmr_conv_fwd wants it: the conv's own map (phase: the upsampled one; pooled: the full-resolution one)
    ch, cw = (oh, ow) if mode != "pooled" else (ih, iw)
    outs, times = {}, {}
    flops = 2.0 * n * (oh * ow if mode != "pooled" else ih * iw) * cout * cin * 9
    try:
        for tag, win in (("ref6", 6), ("ws7", 7), ("lib", -1)):
            call("dgmr_conv_tune", -1, -1, win, -1)
            y = torch.full((n * oh * ow * cout,), float("nan"), device=dev)

            def run():
                return ops._launch_conv(x, wt.data_ptr(), bias, scale, y, n, 1, ch, cw, cin, cout, 1, 3, 3, **kw)

            part = run()
            if part is NotImplemented and tag == "lib":  # (below the library's own size threshold for this mode: nothing to compare with)
                outs[tag] = (outs["ws7"][0], None)
                continue
            assert part is not NotImplemented, (name, tag)
            torch.cuda.synchronize()
            outs[tag] = (y.clone(), None if part is None else part.double().sum(0))
            if time_it:
                times[tag] = bench(run, 3 if n >= 288 else 10)
    finally:
        call("dgmr_conv_tune", -1, -1, -1, -1)
    y6, s6 = outs["ref6"]
    y7, s7 = outs["ws7"]
    yl, _ = outs["lib"]
    nan7 = int(torch.isnan(y7).sum().item())
    exact = bool(torch.equal(y6, y7))
    dmax = (y6 - y7).abs().max().item() if not exact and nan7 == 0 else 0.0
    dlib = (yl - y7).abs().max().item() / max(yl.abs().max().item(), 1e-30) if nan7 == 0 else float("nan")
    sdiff = 0.0
    if s6 is not None and s7 is not None:
        sdiff = ((s6 - s7).abs().max() / s6.abs().max().clamp_min(1e-30)).item()
    return dict(exact=exact, nan=nan7, max_diff=dmax, stats_rel=sdiff, vs_lib=dlib, times=times, flops=flops)


def main():
    load()
    prec = "bf16x3"
    for a in sys.argv[1:]:
        if a.startswith("--prec="):
            prec = a.split("=")[1]
    ops.set_precision(prec)
    dbg = 0
    for a in sys.argv[1:]:
        if a.startswith("--dbg="):  # dgmr_debug_flags: 1 no epilogue, 2 no halo staging (timing only: results are garbage)
            dbg = int(a.split("=")[1])
    call("dgmr_debug_flags", dbg)
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    cases = SMALL + (BIG if "--big" in sys.argv else [])
    if "--big-only" in sys.argv:
        cases = BIG
    bad = 0
    for case in cases:
        name, n, h, w, cin, cout, mode, bn, eop, stats, groups = case
        if only and not any(o in name for o in only):
            continue
        r = run_case(case, prec, time_it="--no-time" not in sys.argv)
        ok = r["exact"] and r["nan"] == 0 and r["stats_rel"] < 1e-5
        bad += 0 if ok else 1
        line = (f"{name:18s} {mode:6s} N={n:4d} {h:3d}x{w:<3d} {cin:3d}->{cout:<3d} eop={str(eop):7s} | y==ref6: {r['exact']} (max diff {r['max_diff']:.2e}, "
                f"NaN {r['nan']}) stats rel {r['stats_rel']:.1e} vs lib {r['vs_lib']:.1e} {'OK' if ok else 'FAIL'}")
        if r["times"]:
            line += " | " + "  ".join(f"{t} {r['times'][t]*1e3:8.1f} us {r['flops']/r['times'][t]/1e9:6.1f} TF" for t in ("ref6", "ws7", "lib") if t in r["times"])
        print(line, flush=True)
    print("ws_check:", "ALL OK" if bad == 0 else f"{bad} FAILED", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
