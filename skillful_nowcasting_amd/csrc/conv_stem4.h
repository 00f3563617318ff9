// 3x3 / 3x3x3 convolution of a FOUR-channel map - the discriminators' and the conditioning stack's first convs behind the
// space-to-depth of single-channel radar frames (dgmr/discriminators.py:71-76,168-173; dgmr/common.py:333) - for gfx950 (MI355X).
//
// K = 36 (108 in 3-D) is too short for the bf16 kernels: the LDS-window kernel stages 32-channel chunks (1/8 filled), the implicit-GEMM
// kernel gathers its A tile element by element (21 TF algorithmic in bf16x6: 11 ms of the step for the temporal discriminator's first
// conv).  Here the whole problem sits in registers and one small LDS image:
//   * v_mfma_f32_16x16x4_f32: K = 4 per instruction = exactly the four input channels of ONE filter tap; exact fp32 products and sums
//     (the 157 TF pipe - the layer needs 1.5 ms of it per step), so the result does not depend on the library's arithmetic mode;
//   * the B operand of a tap and a 16-channel block is ONE register per lane (B[k = lane >> 4][j = lane & 15] = w[co0 + j][tap][k]): the
//     27 x 3 registers of a wave's 48 output channels are loaded once and stay - no weight traffic, no weight LDS, no barrier in the loop;
//   * the A operand is one ds_read_b32 per tap and 16-pixel block from the fp32 halo image of the tile (A[i = lane & 15][k = lane >> 4] =
//     halo[pixel i + tap shift][k]: 16 pixels x 16 bytes, contiguous, conflict-free), feeding three MFMAs;
//   * a workgroup = 8 x 32 output pixels x 48 channels: 4 waves x (4 pixel blocks x 3 channel blocks) accumulators; the halo (10 x 34
//     pixels x 16 bytes per depth plane) is staged once, ONE barrier per workgroup;
//   * epilogue as the window kernels' 16-byte one: 4 x 4 quad transposes (DPP), then 1/sigma, bias, (relu), 16-byte stores.
// HBM: 16 bytes in, 4 * Cout bytes out per pixel - the layer is bound by its output stream once the matrix work is this small.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace {

typedef float stem_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float stem_qx1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
}
__device__ __forceinline__ float stem_qx2(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
}
// 4 x 4 transpose across a quad of lanes (conv_win_glds.h quad_transpose): lane c holds rows 0..3 of column c -> lane j holds columns
// 0..3 of row j
__device__ __forceinline__ stem_f32x4 stem_quad_transpose(float a0, float a1, float a2, float a3, int lane) {
    const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
    const float r01 = stem_qx1(b0 ? a0 : a1), r23 = stem_qx1(b0 ? a2 : a3);
    a0 = b0 ? r01 : a0;
    a1 = b0 ? a1 : r01;
    a2 = b0 ? r23 : a2;
    a3 = b0 ? a3 : r23;
    const float r02 = stem_qx2(b1 ? a0 : a2), r13 = stem_qx2(b1 ? a1 : a3);
    a0 = b1 ? r02 : a0;
    a2 = b1 ? a2 : r02;
    a1 = b1 ? r13 : a1;
    a3 = b1 ? a3 : r13;
    return (stem_f32x4){a0, a1, a2, a3};
}

constexpr int STEM_TH = 8, STEM_TW = 32, STEM_HW = STEM_TW + 2, STEM_NPIX = (STEM_TH + 2) * STEM_HW;  // 10 x 34 halo pixels
constexpr int STEM_BN = 48;                                                                          // output channels per workgroup

// grid: (N * D * tiles_hw, Cout / 48); p.Cin == 4, p.H % 8 == 0, p.W % 32 == 0, p.Cout % 48 == 0, plain epilogue without fused operands
template <int KD>
__global__ __launch_bounds__(256, 2) void conv_stem4_kernel(const dgmr_conv_args p, const int tiles_w, const int tiles_hw) {
    constexpr int TAPS = 9 * KD;
    __shared__ __attribute__((aligned(16))) float halo[KD * STEM_NPIX * 4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tile = blockIdx.x;
    const int n = tile / tiles_hw;  // image of the tile; 3-D: depth plane (sample * D + d)
    const int trem = tile - n * tiles_hw;
    const int th = trem / tiles_w;
    const int h0 = th * STEM_TH, w0 = (trem - th * tiles_w) * STEM_TW;
    const int smp = KD == 3 ? n / p.D : n;
    const int dpl = KD == 3 ? n - smp * p.D : 0;
    const int co0 = (int)blockIdx.y * STEM_BN;

    // ---- B: every tap of this workgroup's 48 output channels, one register per tap and 16-channel block ----
    float wv[TAPS][3];
    {
        const float* wl = p.w + ((size_t)(co0 + (lane & 15)) * TAPS) * 4 + (lane >> 4);
#pragma unroll
        for (int jb = 0; jb < 3; ++jb)
#pragma unroll
            for (int t = 0; t < TAPS; ++t) wv[t][jb] = wl[((size_t)jb * 16 * TAPS + t) * 4];
    }
    // ---- A: the halo of the tile, KD depth planes, fp32 (zero outside the map / the depth range) ----
    const stem_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kd = 0; kd < KD; ++kd) {
        const int dz = KD == 3 ? kd - 1 : 0;
        const bool pok = (unsigned)(dpl + dz) < (unsigned)p.D;
        for (int i = tid; i < STEM_NPIX; i += 256) {
            const int lr = i / STEM_HW, lc = i - lr * STEM_HW;
            const int ih = h0 - 1 + lr, iw = w0 - 1 + lc;
            const bool ok = pok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            stem_f32x4 v = *reinterpret_cast<const stem_f32x4*>(p.x + (ok ? (((size_t)(n + dz) * p.H + ih) * p.W + iw) * 4 : (size_t)0));
            if (p.pre_relu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            *reinterpret_cast<stem_f32x4*>(halo + (kd * STEM_NPIX + i) * 4) = ok ? v : zero4;
        }
    }
    __syncthreads();

    // ---- 4 pixel blocks (16 consecutive pixels of a tile row) x 3 channel blocks per wave ----
    stem_f32x4 acc[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int jb = 0; jb < 3; ++jb) acc[q][jb] = zero4;
    const float* ab[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int pb = wid * 4 + q, r = pb >> 1, c0 = (pb & 1) * 16;
        ab[q] = halo + ((r * STEM_HW + c0 + (lane & 15)) * 4 + (lane >> 4));
    }
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        const int kd = t / 9, ky = (t % 9) / 3, kx = t % 3;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float a = ab[q][(kd * STEM_NPIX + ky * STEM_HW + kx) * 4];
#pragma unroll
            for (int jb = 0; jb < 3; ++jb) {
#if defined(__HIP_DEVICE_COMPILE__)
                acc[q][jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wv[t][jb], acc[q][jb], 0, 0, 0);
#endif
            }
        }
    }

    // ---- epilogue: lane holds rows 4 (lane >> 4) + r, column lane & 15 of a block; after the quad transpose 4 channels of one pixel ----
    const float sc = p.scale ? p.scale[smp / p.scale_group] : 1.f;
    const int j4 = lane & 3, q4 = (lane & 15) >> 2, rsel = lane >> 4;
#pragma unroll
    for (int jb = 0; jb < 3; ++jb) {
        const int col = co0 + jb * 16 + 4 * q4;
        const stem_f32x4 b4 = p.bias ? *reinterpret_cast<const stem_f32x4*>(p.bias + col) : zero4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pb = wid * 4 + q, r = pb >> 1, c0 = (pb & 1) * 16;
            const int hh = h0 + r, ww = w0 + c0 + 4 * rsel + j4;
            stem_f32x4 v = stem_quad_transpose(acc[q][jb][0], acc[q][jb][1], acc[q][jb][2], acc[q][jb][3], lane);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                v[c] = fmaf(v[c], sc, b4[c]);
                if (p.act_relu) v[c] = fmaxf(v[c], 0.f);
            }
            *reinterpret_cast<stem_f32x4*>(p.y + (((size_t)n * p.H + hh) * p.W + ww) * p.Cout + col) = v;
        }
    }
}

}  // namespace
