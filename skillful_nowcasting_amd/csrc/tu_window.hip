// LDS-window 3x3 / 3x3x3 convolution kernels (conv_win_glds.h, conv_bf16.h) instantiated for ONE arithmetic mode: compile with
// -DDGMR_NS=1 | 3 | 6.  Called from dgmr_conv_fwd (conv.hip) through dgmr_tu::launch_window_ns<NS>.
#include "conv_launch.h"
#include "conv_win_glds.h"

#ifndef DGMR_NS
#error "compile with -DDGMR_NS=1|3|6"
#endif

namespace dgmr_tu {

int DGMR_TU_CAT(launch_window_ns, DGMR_NS)(const dgmr_conv_args& p, const WinPlan& wp, bool phases, int tune_window, hipStream_t s) {
    constexpr int NS = DGMR_NS;
    const int tw_shift = wp.tw_shift, g_shift = wp.g_shift, tiles_w = wp.tiles_w, tiles_hw = wp.tiles_hw, bnw = wp.bnw;
    const dim3 grid((unsigned)wp.grid_x * (phases ? (wp.pair ? 2u : 4u) : 1u), (unsigned)((p.Cout + bnw - 1) / bnw));
#define DGMR_GLDS(BN_, WM_, WN_, ...) \
    hipLaunchKernelGGL((conv3x3_glds_kernel<BN_, WM_, WN_, NS, __VA_ARGS__>), grid, dim3(256), 0, s, p, tw_shift, tiles_w, tiles_hw, g_shift)
#if DGMR_NS != 6
    if (phases && wp.pair) {  // both column parities of a row parity per workgroup (128-pixel tiles, 96 columns: window_plan)
        DGMR_GLDS(96, 4, 1, 128, false, false, true);
        return 0;
    }
#endif
    if (wp.big) {  // 256-pixel tiles (never with 128 columns: window_plan)
        if (bnw == 16) DGMR_GLDS(16, 4, 1, 256, false, true);
        else if (bnw == 96) DGMR_GLDS(96, 4, 1, 256);
        else if (bnw == 48) DGMR_GLDS(48, 4, 1, 256, false, true);
        else DGMR_GLDS(64, 4, 1, 256);
    } else if (wp.glds) {
        // weight stages by LDS-DMA (conv_win_glds.h; measured +4..17 % over the register-staged kernel below, bit-identical results)
        if (bnw == 48) DGMR_GLDS(48, 4, 1, 128, false, true);
        // <= 16 output channels (the data gradients towards the discriminators' 4-channel inputs, discriminators.py:113,189 backwards):
        // ONE 16-column block per wave row instead of a 64-column tile that is 94 % padding
        else if (bnw == 16) DGMR_GLDS(16, 4, 1, 128, false, true);
#if DGMR_NS != 6
        // dgmr_conv_tune window = 6: the one-role kernels with the wave-specialised kernels' own block shapes (conv_win_ws.h: 16 x 16
        // blocks at 96 columns, four row waves at 128) - the bit-for-bit reference of tests/test_gpu_kernels.py
        else if (bnw == 96 && tune_window == 6) DGMR_GLDS(96, 4, 1, 128, false, true);
        else if (bnw == 128 && tune_window == 6) DGMR_GLDS(128, 4, 1, 128);
        else if (bnw == 128 && tune_window == 4) DGMR_GLDS(128, 1, 4, 128, true);
        else if (bnw == 64 && tune_window == 4) DGMR_GLDS(64, 2, 2, 128, true);
#endif
        else if (bnw == 128) DGMR_GLDS(128, 2, 2, 128);
        else if (bnw == 96) DGMR_GLDS(96, 4, 1, 128);
        else DGMR_GLDS(64, 4, 1, 128);
    }
#if DGMR_NS != 6
    // the register-staged predecessor (dgmr_conv_tune window = 1): the A/B reference of tests/test_gpu_kernels.py
    else if (bnw == 128) {
        // bf16x3: one weight stage + halo fetched at the chunk boundary = 53 KB of LDS and <= 168 VGPRs -> three workgroups per CU
        // (measured 320 -> 350 TF); plain bf16 keeps the two-stage pipeline
        if constexpr (NS == 3) hipLaunchKernelGGL((conv3x3_win_kernel<128, 2, 2, NS, 1, true>), grid, dim3(256), 0, s, p, tw_shift, tiles_w, tiles_hw, g_shift);
        else hipLaunchKernelGGL((conv3x3_win_kernel<128, 2, 2, NS>), grid, dim3(256), 0, s, p, tw_shift, tiles_w, tiles_hw, g_shift);
    } else if (bnw == 96) {
        // bf16x3 at 96 channels: ONE weight stage and the halo fetched at the chunk boundary (48 KB of LDS instead of 63, no spill at
        // 168 VGPRs) let three workgroups share a CU (measured 261 -> 286 TF); plain bf16 already fits three with two stages
        if constexpr (NS == 3) hipLaunchKernelGGL((conv3x3_win_kernel<96, 4, 1, NS, 1, true>), grid, dim3(256), 0, s, p, tw_shift, tiles_w, tiles_hw, g_shift);
        else hipLaunchKernelGGL((conv3x3_win_kernel<96, 4, 1, NS, 2>), grid, dim3(256), 0, s, p, tw_shift, tiles_w, tiles_hw, g_shift);
    } else {
        hipLaunchKernelGGL((conv3x3_win_kernel<64, 4, 1, NS>), grid, dim3(256), 0, s, p, tw_shift, tiles_w, tiles_hw, g_shift);
    }
#else
    else {
        dgmr_set_error("window conv: bf16x6 has no register-staged kernel (dgmr_conv_tune window = 1 / 2)");
        return -1;
    }
#endif
#undef DGMR_GLDS
    return 0;
}

}  // namespace dgmr_tu
