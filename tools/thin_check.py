"""16-column window tile for convs with <= 16 output channels (the data gradients towards the discriminators' 4-channel inputs)
against the 64-column tile (dgmr_conv_tune window = 3) and the generic kernel (window = 0): results and time.  python tools/thin_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from skillful_nowcasting_amd import ops
from skillful_nowcasting_amd._lib import call, load

load()
ops.set_precision("bf16x3")
dev = "cuda"
for name, n, d, h, w, cin, cout, kd in [("spatial D d1 dgrad", 6 * 32 * 8, 1, 64, 64, 48, 4, 1), ("temporal D d1 dgrad (3-D), G pass", 192, 22, 64, 64, 48, 4, 3),
                                          ("temporal D d1 dgrad (3-D), D pass", 32, 22, 64, 64, 48, 4, 3), ("Cout 8, 2-D", 64, 1, 32, 32, 96, 8, 1),
                                          ("Cout 16, 2-D", 64, 1, 32, 32, 64, 16, 1)]:
    torch.manual_seed(1)
    x = torch.randn(n * d * h * w * cin, device=dev)
    wt = torch.randn(cout * kd * 9 * cin, device=dev) * 0.05
    bias = torch.randn(cout, device=dev)
    scale = torch.full((1,), 0.7, device=dev)
    msk = torch.randn(n * d * h * w * cout, device=dev)
    wsp = torch.empty(2 * wt.numel(), device=dev, dtype=torch.int16)
    call("dgmr_split_weights", wt.data_ptr(), wsp.data_ptr(), cout * kd * 9, cin, 0, 0, 2, 0, ops._stream())
    res, tms = {}, {}
    for tag, win in (("thin16", -1), ("tile64", 3), ("generic", 0)):
        call("dgmr_conv_tune", -1, -1, win, -1)
        y = torch.empty(n * d * h * w * cout, device=dev)

        def run():
            ops._launch_conv(x, wt.data_ptr(), bias, scale, y, n, d, h, w, cin, cout, kd, 3, 3, mask_src=msk, w_split=wsp)

        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record()
        torch.cuda.synchronize()
        res[tag], tms[tag] = y.clone(), e0.elapsed_time(e1) / 5
    call("dgmr_conv_tune", -1, -1, -1, -1)
    sc = res["generic"].abs().max().item()
    print(f"{name:36s} N={n} D={d} {h}x{w} {cin}->{cout}: thin16 {tms['thin16']*1e3:8.1f} us  tile64 {tms['tile64']*1e3:8.1f} us  generic {tms['generic']*1e3:8.1f} us | "
          f"thin vs tile64 {(res['thin16']-res['tile64']).abs().max().item()/sc:.1e}  thin vs generic {(res['thin16']-res['generic']).abs().max().item()/sc:.1e}", flush=True)
