// The streaming 1x1-conv kernel (conv1x1.h) for ONE arithmetic mode: compile with -DDGMR_NS=1 | 3.  Called from dgmr_conv_fwd
// (conv.hip) through dgmr_tu::launch_conv1x1_ns<NS>.
#include "conv_launch.h"
#include "conv1x1.h"

#ifndef DGMR_NS
#error "compile with -DDGMR_NS=1|3"
#endif

namespace dgmr_tu {

int DGMR_TU_CAT(launch_conv1x1_ns, DGMR_NS)(const dgmr_conv_args& p, int M, int rows_per_sample, hipStream_t s) {
    constexpr int NS = DGMR_NS;
    const int C = p.Cout;
    // output-channel block: the whole row of Y where it fits (X is then read once), 96-column blocks for multiples of 96
    const int bn = C <= 64 ? 64 : ((C <= 96 || (C % 96 == 0 && C % 128 != 0)) ? 96 : 128);
    const dim3 grid((unsigned)((M + 255) / 256), (unsigned)((C + bn - 1) / bn));
    if (bn == 64) hipLaunchKernelGGL((conv1x1_kernel<64, NS>), grid, dim3(256), 0, s, p, M, rows_per_sample);
    else if (bn == 96) hipLaunchKernelGGL((conv1x1_kernel<96, NS>), grid, dim3(256), 0, s, p, M, rows_per_sample);
    else hipLaunchKernelGGL((conv1x1_kernel<128, NS>), grid, dim3(256), 0, s, p, M, rows_per_sample);
    return 0;
}

}  // namespace dgmr_tu
