// Generic implicit-GEMM forward / data-gradient kernel on the bf16 matrix cores (conv_bf16_kernel) for ONE arithmetic mode:
// compile with -DDGMR_NS=1 | 3 | 6.  Called from dgmr_conv_fwd (conv.hip) through dgmr_tu::launch_gemm_ns<NS>.
#include "conv_launch.h"
#include "conv_bf16.h"

#ifndef DGMR_NS
#error "compile with -DDGMR_NS=1|3|6"
#endif

namespace dgmr_tu {

int DGMR_TU_CAT(launch_gemm_ns, DGMR_NS)(int variant, const dgmr_conv_args& p, int M, int Ktot, int kt_per_split, dim3 grid, hipStream_t s) {
    constexpr int NS = DGMR_NS;
    // two register stages: the 128 x 128 tile needs 8 waves to fit
    switch (variant) {
        case V_F128x128: hipLaunchKernelGGL((conv_bf16_kernel<128, 128, 32, 2, 4, NS>), grid, dim3(512), 0, s, p, M, Ktot, kt_per_split); break;
        case V_F64x64: hipLaunchKernelGGL((conv_bf16_kernel<64, 64, 32, 2, 2, NS>), grid, dim3(256), 0, s, p, M, Ktot, kt_per_split); break;
        case V_F128x96: hipLaunchKernelGGL((conv_bf16_kernel<128, 96, 32, 4, 1, NS>), grid, dim3(256), 0, s, p, M, Ktot, kt_per_split); break;
        case V_F128x64: hipLaunchKernelGGL((conv_bf16_kernel<128, 64, 32, 4, 1, NS>), grid, dim3(256), 0, s, p, M, Ktot, kt_per_split); break;
        default: hipLaunchKernelGGL((conv_bf16_kernel<128, 32, 32, 4, 1, NS>), grid, dim3(256), 0, s, p, M, Ktot, kt_per_split); break;
    }
    return 0;
}

}  // namespace dgmr_tu
