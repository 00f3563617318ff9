// Weight-gradient kernels on the bf16 matrix cores (wgrad_win.h: LDS-window 3x3; conv_bf16.h: im2col) for ONE arithmetic mode:
// compile with -DDGMR_NS=1 | 3 | 6.  Called from dgmr_conv_wgrad (conv.hip).
#include "conv_launch.h"
#include "wgrad_win.h"
#include "wgrad_ws.h"

#ifndef DGMR_NS
#error "compile with -DDGMR_NS=1|3|6"
#endif

namespace dgmr_tu {

int DGMR_TU_CAT(launch_wgrad_window_ns, DGMR_NS)(const dgmr_wgrad_args& p, dim3 grid, int tw_shift, int tiles_w, int tiles_hw,
                                                 int tiles_per_split, int splits_per_group, int tiles_per_group, int ws, hipStream_t s) {
    constexpr int NS = DGMR_NS;
    const bool b96 = p.Cout % 96 == 0;
    if (ws) {  // wave-specialised kernel (wgrad_ws.h): 3 matrix + 4 loader waves
        // ws bit 2: four matrix waves (16 x 16 x 32 MFMAs; bf16 / bf16x3 only)
#define DGMR_WGS(BI_, TWS_)                                                                                                      \
    do {                                                                                                                         \
        if constexpr (NS != 6) {                                                                                                 \
            if (ws & 8) { /* one output-pixel parity of an upsampling conv (ws bits 16, 17), four matrix waves */                 \
                hipLaunchKernelGGL((conv_wgrad_ws_kernel<BI_, NS, TWS_, 4, true>), dim3(grid.x * grid.y * grid.z), dim3(512), 0, s, p,     \
                                   tiles_w, tiles_hw, tiles_per_split, splits_per_group, tiles_per_group, ws & 0xff, 0, (ws >> 16) & 3, 4); \
                break;                                                                                                           \
            }                                                                                                                    \
            if (ws & 4) {                                                                                                        \
                hipLaunchKernelGGL((conv_wgrad_ws_kernel<BI_, NS, TWS_, 4>), dim3(grid.x * grid.y * grid.z), dim3(512), 0, s, p, tiles_w, \
                                   tiles_hw, tiles_per_split, splits_per_group, tiles_per_group, ws & 0xff, (ws >> 8) & 0xff);   \
                break;                                                                                                           \
            }                                                                                                                    \
        }                                                                                                                        \
        hipLaunchKernelGGL((conv_wgrad_ws_kernel<BI_, NS, TWS_, 3>), dim3(grid.x * grid.y * grid.z), dim3(448), 0, s, p, tiles_w,    \
                           tiles_hw, tiles_per_split, splits_per_group, tiles_per_group, ws & 0xff, (ws >> 8) & 0xff);           \
    } while (0)
        // <= 48 output channels, four matrix waves, bf16 / bf16x3, no phases: the pixel-split 48-column tile (ws bit 32 keeps the
        // 64-column one: dgmr_conv_tune wgrad_window = 5, the A/B reference)
        bool done = false;
        if constexpr (NS != 6) {
            if (p.Cout <= 48 && (ws & 4) && !(ws & 8) && !(ws & 32)) {
                if (tw_shift == 5)
                    hipLaunchKernelGGL((conv_wgrad_ws_kernel<48, NS, 5, 4, false, true>), dim3(grid.x * grid.y * grid.z), dim3(512), 0, s, p, tiles_w,
                                       tiles_hw, tiles_per_split, splits_per_group, tiles_per_group, ws & 0x1f, (ws >> 8) & 0xff);
                else
                    hipLaunchKernelGGL((conv_wgrad_ws_kernel<48, NS, 4, 4, false, true>), dim3(grid.x * grid.y * grid.z), dim3(512), 0, s, p, tiles_w,
                                       tiles_hw, tiles_per_split, splits_per_group, tiles_per_group, ws & 0x1f, (ws >> 8) & 0xff);
                done = true;
            }
        }
        if (done) return 0;
        if (b96) {
            if (tw_shift == 5) DGMR_WGS(96, 5);
            else DGMR_WGS(96, 4);
        } else {
            if (tw_shift == 5) DGMR_WGS(64, 5);
            else DGMR_WGS(64, 4);
        }
#undef DGMR_WGS
        return 0;
    }
#define DGMR_WGW(BI_, TWS_)                                                                                                       \
    hipLaunchKernelGGL((conv_wgrad_win_kernel<BI_, NS, TWS_>), grid, dim3(192), 0, s, p, tiles_w, tiles_hw, tiles_per_split, \
                       splits_per_group, tiles_per_group)
    if (b96) {
        if (tw_shift == 5) DGMR_WGW(96, 5);
        else DGMR_WGW(96, 4);
    } else {
        if (tw_shift == 5) DGMR_WGW(64, 5);
        else DGMR_WGW(64, 4);
    }
#undef DGMR_WGW
    return 0;
}

int DGMR_TU_CAT(launch_wgrad_gemm_ns, DGMR_NS)(const dgmr_wgrad_args& p, int bi, dim3 grid, int M, int Ktot, int rows_per_split,
                                               int splits_per_group, int rows_per_group, hipStream_t s) {
    constexpr int NS = DGMR_NS;
    const dim3 blk(256);
    switch (bi) {
        case 32: hipLaunchKernelGGL((conv_wgrad_bf16_kernel<32, 1, 4, NS>), grid, blk, 0, s, p, M, Ktot, rows_per_split, splits_per_group, rows_per_group); break;
        case 64: hipLaunchKernelGGL((conv_wgrad_bf16_kernel<64, 2, 2, NS>), grid, blk, 0, s, p, M, Ktot, rows_per_split, splits_per_group, rows_per_group); break;
        case 96: hipLaunchKernelGGL((conv_wgrad_bf16_kernel<96, 1, 4, NS>), grid, blk, 0, s, p, M, Ktot, rows_per_split, splits_per_group, rows_per_group); break;
        default: hipLaunchKernelGGL((conv_wgrad_bf16_kernel<128, 2, 2, NS>), grid, blk, 0, s, p, M, Ktot, rows_per_split, splits_per_group, rows_per_group); break;
    }
    return 0;
}

}  // namespace dgmr_tu
