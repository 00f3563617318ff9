// Host-side seams between the translation units of libdgmr_hip.so.  conv.hip owns the C-ABI entry points, the dispatch policy and the
// exact-f32 kernels; the bf16 matrix-core kernels are compiled once per arithmetic mode NS (1: bf16, 3: bf16x3, 6: bf16x6; see
// conv_bf16.h) in their own translation units (tu_window.hip, tu_gemm.hip, tu_wgrad.hip with -DDGMR_NS=...), so that the eleven
// objects build side by side.  Nothing here is exported from the shared library.
#pragma once
#include "common.h"

#define DGMR_HIDDEN __attribute__((visibility("hidden")))

// tile variants of the implicit-GEMM forward / data-gradient kernels, weight-gradient classes, window classes (profiling rows)
enum { V_F128x128 = 0, V_F64x64, V_F128x96, V_F128x64, V_F128x32, V_W128, V_W64, V_W32, V_WIN128, V_WIN96, V_WIN64, V_COUNT };

// Which LDS-window 3x3 kernel takes a conv, and with which tiling (conv.hip: window_plan)
struct WinPlan {
    int tw_shift, g_shift, tiles_w, tiles_hw, bnw, grid_x;
    bool big, glds;  // 256-pixel tiles; LDS-DMA kernel (has the fused output statistics)
    bool ws;         // the wave-specialised persistent kernel (conv_win_ws.h): 128-pixel tiles, one workgroup per CU
    int ws_ups;      // its epilogue units per tap (1 | 2)
    bool pair;       // phase mode: one workgroup per ROW parity computes both column parities from one halo (conv_win_glds.h PAIR)
};

namespace dgmr_tu {

// MFMAs per product of precision code `prec` (dgmr_set_precision): 1 -> 3, 2 -> 1, 3 -> 6
inline int ns_of_precision(int prec) { return prec == 1 ? 3 : (prec == 2 ? 1 : 6); }

#define DGMR_TU_DECLARE(NS)                                                                                                          \
    DGMR_HIDDEN int launch_window_ns##NS(const dgmr_conv_args& p, const WinPlan& wp, bool phases, int tune_window, hipStream_t s);   \
    DGMR_HIDDEN int launch_gemm_ns##NS(int variant, const dgmr_conv_args& p, int M, int Ktot, int kt_per_split, dim3 grid,           \
                                       hipStream_t s);                                                                               \
    DGMR_HIDDEN int launch_wgrad_window_ns##NS(const dgmr_wgrad_args& p, dim3 grid, int tw_shift, int tiles_w, int tiles_hw,         \
                                               int tiles_per_split, int splits_per_group, int tiles_per_group, int ws,         \
                                               hipStream_t s);                                                                        \
    DGMR_HIDDEN int launch_wgrad_gemm_ns##NS(const dgmr_wgrad_args& p, int bi, dim3 grid, int M, int Ktot, int rows_per_split,       \
                                             int splits_per_group, int rows_per_group, hipStream_t s);
DGMR_TU_DECLARE(1)
DGMR_TU_DECLARE(3)
DGMR_TU_DECLARE(6)
#undef DGMR_TU_DECLARE
// streaming 1x1 conv (tu_1x1.hip, conv1x1.h): bf16 and bf16x3
DGMR_HIDDEN int launch_conv1x1_ns1(const dgmr_conv_args& p, int M, int rows_per_sample, hipStream_t s);
DGMR_HIDDEN int launch_conv1x1_ns3(const dgmr_conv_args& p, int M, int rows_per_sample, hipStream_t s);

// wave-specialised window kernels (tu_ws.hip): one translation unit per (arithmetic, kernel mode: 0 plain, 1 phase, 2 pooled)
#define DGMR_TU_DECLARE_WS(NS, MODE) \
    DGMR_HIDDEN int launch_window_ws_ns##NS##_m##MODE(const dgmr_conv_args& p, const WinPlan& wp, int grid, hipStream_t s);
DGMR_TU_DECLARE_WS(1, 0)
DGMR_TU_DECLARE_WS(1, 1)
DGMR_TU_DECLARE_WS(1, 2)
DGMR_TU_DECLARE_WS(3, 0)
DGMR_TU_DECLARE_WS(3, 1)
DGMR_TU_DECLARE_WS(3, 2)
#undef DGMR_TU_DECLARE_WS

}  // namespace dgmr_tu

#define DGMR_TU_CAT_(a, b) a##b
#define DGMR_TU_CAT(a, b) DGMR_TU_CAT_(a, b)
