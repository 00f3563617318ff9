// Wave-specialised LDS-window 3x3 convolution kernels (conv_win_ws.h) instantiated for ONE arithmetic mode and ONE kernel mode:
// compile with -DDGMR_NS=1|3 -DDGMR_WS_MODE=0|1|2 (plain / phase / pooled).  Called from dgmr_conv_fwd (conv.hip) through
// dgmr_tu::launch_window_ws_ns<NS>.
#include "conv_launch.h"
#include "conv_win_ws.h"

#if !defined(DGMR_NS) || !defined(DGMR_WS_MODE)
#error "compile with -DDGMR_NS=1|3 -DDGMR_WS_MODE=0|1|2"
#endif

namespace dgmr_tu {

#define DGMR_WS_FN_(ns, mode) launch_window_ws_ns##ns##_m##mode
#define DGMR_WS_FN(ns, mode) DGMR_WS_FN_(ns, mode)

int DGMR_WS_FN(DGMR_NS, DGMR_WS_MODE)(const dgmr_conv_args& p, const WinPlan& wp, int grid, hipStream_t s) {
    constexpr int NS = DGMR_NS, MODE = DGMR_WS_MODE;
    const int n_nb = (p.Cout + wp.bnw - 1) / wp.bnw;
    const bool eop = p.residual != nullptr || p.mask_src != nullptr;
#define DGMR_WS(BN_, M16_, EOP_)                                                                                                 \
    hipLaunchKernelGGL((conv3x3_ws_kernel<BN_, 2, NS, M16_, EOP_, MODE>), dim3(grid), dim3(768), 0, s, p, wp.tw_shift, wp.tiles_w, \
                       wp.tiles_hw, wp.g_shift, wp.grid_x, n_nb, wp.ws_ups)
    if (wp.bnw == 96) {
#if DGMR_WS_MODE != 1
        if (eop) DGMR_WS(96, true, true);
        else
#endif
            DGMR_WS(96, true, false);
    } else if (wp.bnw == 128) {
#if DGMR_WS_MODE != 1
        if (eop) DGMR_WS(128, false, true);
        else
#endif
            DGMR_WS(128, false, false);
    } else {
        dgmr_set_error("wave-specialised window conv: no kernel for %d-column tiles", wp.bnw);
        return -1;
    }
#undef DGMR_WS
    return 0;
}

}  // namespace dgmr_tu
