// Weight gradient of a 3x3 convolution with the INPUT WINDOW staged in LDS (bf16 matrix cores, fp32 tensors and accumulation).
//
//   G[co][tap][ci] = sum over pixels  dY[n,h,w,co] * pre(x)[n, h+dy, w+dx, ci]          (reference: autograd of F.conv2d at every
//                                                                                        `spectral_norm(conv)` call)
//
// conv_wgrad_bf16_kernel treats this as a GEMM over an im2col matrix: every workgroup re-gathers (and re-applies the fused
// BatchNorm/ReLU prologue to, and re-splits into bf16 pairs) its 128 im2col columns for every pixel, i.e. each input element is
// fetched and converted 9 x (Cout / tile) times; the kernel is bound by that VALU work and the L2 traffic, not by the matrix pipe.
// Here a workgroup owns 32 INPUT channels x BI output channels and walks a slab of 64-pixel tiles (2 rows x 32 columns, or 4 x 16
// on 16-pixel-wide maps):
//   * the 4 x 34 (6 x 18) input halo of the tile is fetched, prologue'd and split ONCE and serves all nine taps;
//   * dY of the tile is fetched and split once per 32 input channels;
//   * both land TRANSPOSED in LDS ([channel][pixel], the reduction index contiguous) through a 4-pixel x 4-channel register
//     transpose, so MFMA fragments are plain 16-byte reads; the +-1 column shift of a tap is applied in registers (one extra dword
//     and four v_alignbit per fragment) because a 2-byte-shifted 16-byte LDS read would be replayed at 64 cycles;
//   * wave w owns filter row w (dy = w - 1): 3 taps x BI/32 output-channel blocks = up to 9 accumulators [32 co x 32 ci] that
//     stay in registers for the whole slab; per 16-pixel step it reads BI/32 dY fragments and ONE input fragment (+2 dwords) per
//     plane and issues up to 27 MFMAs - the LDS is nowhere near its bandwidth.
// Output: partial[slab][co][tap*Cin + ci], the layout dgmr_wgrad_reduce* already consume.
#pragma once
#include "conv_bf16.h"

namespace {

// Two workgroups per CU (the nine accumulators take 144 registers).  Holding the next tile's loads in registers under the MFMAs
// needs > 256 registers, i.e. one workgroup per CU: measured 1.7x slower - co-residency hides the fetch better than prefetching.
// TWS: log2 of the tile width (5: tiles of 2 x 32 pixels, 4: 4 x 16) - compile time, the inner loop's addresses fold to constants
template <int BI, int NS, int TWS>
__global__ __launch_bounds__(192, 2) void conv_wgrad_win_kernel(const dgmr_wgrad_args p, const int tiles_w, const int tiles_hw,
                                                                const int tiles_per_split, const int splits_per_group,
                                                                const int tiles_per_group) {
    constexpr int NT = 192, CK = 32;
    constexpr int NP = planes_of<NS>::value;
    // dwords per input channel: 4 halo rows of 48 bf16 slots (TW = 32) or 6 rows of 32 slots (TW = 16), column c at slot 8 + c of
    // its row, + 16 bytes of padding
    constexpr int XLD = 4 * 24 + 4;
    constexpr int YLD = 32 + 4;      // dwords per output channel: 64 pixels + 16 bytes
    constexpr int CB = BI / 32;      // output-channel blocks per wave
    constexpr int XBLK = 4 * 10 * 8, YBLK = 16 * (BI / 4);  // 4 px x 4 ch blocks: halo rows x 4-px groups x channel quads (TW = 16: 6 x 6 x 8)
    constexpr int XPASS = (XBLK + NT - 1) / NT, YPASS = (YBLK + NT - 1) / NT;
    static_assert(BI == 64 || BI == 96, "BI");

    __shared__ __attribute__((aligned(16))) uint32_t smem[NP * CK * XLD + NP * BI * YLD];
    uint32_t* Xs = smem;                  // [plane][ci][XLD]
    uint32_t* Ys = smem + NP * CK * XLD;  // [plane][co][YLD]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int chunk = blockIdx.x, co0 = blockIdx.y * BI;
    const int grp = blockIdx.z / splits_per_group;
    const int t_begin = grp * tiles_per_group + (blockIdx.z - grp * splits_per_group) * tiles_per_split;
    const int t_end = min((grp + 1) * tiles_per_group, t_begin + tiles_per_split);
    const int us = p.upsample ? 1 : 0;
    const int Hs = p.H >> us, Ws = p.W >> us;
    const int Ktot = 9 * p.Cin;
    constexpr int tw_shift = TWS;
    constexpr int TW = 1 << TWS, TH = 64 >> TWS;  // 32 x 2 or 16 x 4
    constexpr int XG = (TW >> 2) + 2;              // 4-pixel groups per halo row (columns -4 .. TW+3)
    constexpr int XPG = (TH + 2) * XG;             // pixel groups per channel quad: 40 or 36
    constexpr int ROWDW = (TW + 16) >> 1;          // dwords per halo row: 24 or 16
    static_assert(TWS == 5 || TWS == 4, "tile width");

    // ---- staging geometry: quads of lanes = 4 consecutive channel quads of one pixel group (64 contiguous bytes per pixel in HBM;
    // in LDS the 4-way channel stride falls on 2 banks x 2 and the 8-byte pixel groups of the next lanes fill the rest) ----
    int x_r[XPASS], x_g[XPASS], x_ci[XPASS];
    bool x_on[XPASS];
#pragma unroll
    for (int i = 0; i < XPASS; ++i) {
        const int idx = tid + i * NT;
        const int t2 = idx >> 2;
        const int pgid = t2 % XPG, cq = (t2 / XPG) * 4 + (idx & 3);
        x_r[i] = pgid / XG;
        x_g[i] = pgid - x_r[i] * XG;
        x_ci[i] = cq * 4;  // channel inside the chunk
        x_on[i] = idx < XPG * 8;
    }
    int y_pg[YPASS], y_co[YPASS];
    bool y_on[YPASS];
#pragma unroll
    for (int i = 0; i < YPASS; ++i) {
        const int idx = tid + i * NT;
        const int t2 = idx >> 2;
        y_pg[i] = t2 & 15;
        y_co[i] = ((t2 >> 4) * 4 + (idx & 3)) * 4;  // channel inside the tile
        y_on[i] = idx < YBLK;
    }
    f32x4 bsum[YPASS];
#pragma unroll
    for (int i = 0; i < YPASS; ++i) bsum[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x16 acc[CB][3];
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][d][r] = 0.f;

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 rx[XPASS][4], ry[YPASS][4], bn_a[XPASS], bn_b[XPASS];
    unsigned xmask[XPASS], ymask[YPASS];
    // ---- fetch of one tile: every load unconditional (clamped address), masks zero the invalid elements afterwards ----
    auto fetch = [&](int t) {
        const int n = t / tiles_hw;
        const int trem = t - n * tiles_hw;
        const int th = trem / tiles_w;
        const int h0 = th * TH, w0 = (trem - th * tiles_w) << tw_shift;
#pragma unroll
        for (int i = 0; i < XPASS; ++i) {
            const int ci = chunk * CK + x_ci[i];
            const bool c_ok = x_on[i] && ci < p.Cin;
            const int ih = h0 - 1 + x_r[i];
            const bool r_ok = c_ok && (unsigned)ih < (unsigned)p.H;
            const uint32_t rbase = ((uint32_t)n * Hs + (r_ok ? (ih >> us) : 0)) * Ws;
            unsigned m = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int iw = w0 - 4 + 4 * x_g[i] + j;
                const bool ok = r_ok && (unsigned)iw < (unsigned)p.W;
                rx[i][j] = *reinterpret_cast<const f32x4*>(p.x + (ok ? (rbase + (iw >> us)) * p.Cin + ci : 0u));
                m |= (ok ? 1u : 0u) << j;
            }
            xmask[i] = m;
            if (p.pre_a) {
                const uint32_t g = c_ok ? (uint32_t)(n / p.pre_group) * p.Cin + ci : 0u;
                bn_a[i] = *reinterpret_cast<const f32x4*>(p.pre_a + g);
                bn_b[i] = *reinterpret_cast<const f32x4*>(p.pre_b + g);
            }
        }
#pragma unroll
        for (int i = 0; i < YPASS; ++i) {
            const int co = co0 + y_co[i];
            const bool ok = y_on[i] && co < p.Cout;
            const int hh = h0 + ((y_pg[i] * 4) >> tw_shift), ww = w0 + ((y_pg[i] * 4) & (TW - 1));
            const size_t m0 = ((size_t)n * p.H + hh) * p.W + ww;
#pragma unroll
            for (int j = 0; j < 4; ++j) ry[i][j] = *reinterpret_cast<const f32x4*>(p.dy + (ok ? (m0 + j) * p.Cout + co : 0));
            ymask[i] = ok ? 0xfu : 0u;
        }
    };
    for (int t = t_begin; t < t_end; ++t) {
        fetch(t);

        // ---- prologue, split, 4x4 transpose -> LDS ----
#pragma unroll
        for (int i = 0; i < XPASS; ++i) {
            if (!x_on[i]) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = rx[i][j];
                if (p.pre_a) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = fmaxf(fmaf(v[c], bn_a[i][c], bn_b[i][c]), 0.f);
                } else if (p.pre_relu) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
                }
                rx[i][j] = ((xmask[i] >> j) & 1u) ? v : zero4;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                u32x2 pl[NP];
                split_planes4<NP>((f32x4){rx[i][0][c], rx[i][1][c], rx[i][2][c], rx[i][3][c]}, pl);
                uint32_t* dst = Xs + (x_ci[i] + c) * XLD + x_r[i] * ROWDW + 2 + 2 * x_g[i];
#pragma unroll
                for (int q = 0; q < NP; ++q) *reinterpret_cast<u32x2*>(dst + q * CK * XLD) = pl[q];
            }
        }
#pragma unroll
        for (int i = 0; i < YPASS; ++i) {
            if (!y_on[i]) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ry[i][j] = ymask[i] ? ry[i][j] : zero4;
                bsum[i] += ry[i][j];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                u32x2 pl[NP];
                split_planes4<NP>((f32x4){ry[i][0][c], ry[i][1][c], ry[i][2][c], ry[i][3][c]}, pl);
                uint32_t* dst = Ys + (y_co[i] + c) * YLD + y_pg[i] * 2;
#pragma unroll
                for (int q = 0; q < NP; ++q) *reinterpret_cast<u32x2*>(dst + q * BI * YLD) = pl[q];
            }
        }
        __syncthreads();

        // ---- multiply: wave `wid` = filter row dy = wid - 1; k index = pixel ----
        const int kg = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            // 16-pixel step kk = pixels [16 kk, 16 kk + 16) of the tile: tile row (16 kk) / TW, first column (16 kk) % TW
            const int hrow = ((kk * 16) >> tw_shift) + wid;  // its halo row under the wave's dy
            const uint32_t* xb = Xs + (lane & 31) * XLD + hrow * ROWDW + 4 + (((kk * 16) & (TW - 1)) >> 1) + kg * 4;
            const uint32_t* yb = Ys + (lane & 31) * YLD + kk * 8 + kg * 4;
            bf16x8_t xf[NP][3], yf[NP][CB];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                const u32x4 q = *reinterpret_cast<const u32x4*>(xb + pl * CK * XLD);
                const uint32_t dl = xb[pl * CK * XLD - 1], dr = xb[pl * CK * XLD + 4];
                const u32x4 m1 = {__builtin_amdgcn_alignbit(q[0], dl, 16), __builtin_amdgcn_alignbit(q[1], q[0], 16),
                                  __builtin_amdgcn_alignbit(q[2], q[1], 16), __builtin_amdgcn_alignbit(q[3], q[2], 16)};
                const u32x4 p1 = {__builtin_amdgcn_alignbit(q[1], q[0], 16), __builtin_amdgcn_alignbit(q[2], q[1], 16),
                                  __builtin_amdgcn_alignbit(q[3], q[2], 16), __builtin_amdgcn_alignbit(dr, q[3], 16)};
                xf[pl][0] = __builtin_bit_cast(bf16x8_t, m1);
                xf[pl][1] = __builtin_bit_cast(bf16x8_t, q);
                xf[pl][2] = __builtin_bit_cast(bf16x8_t, p1);
#pragma unroll
                for (int c = 0; c < CB; ++c)
                    yf[pl][c] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(yb + (pl * BI + c * 32) * YLD));
            }
            __builtin_amdgcn_s_setprio(1);
            for_each_product<NP>([&](auto qa, auto qb) {
#pragma unroll
                for (int c = 0; c < CB; ++c)
#pragma unroll
                    for (int d = 0; d < 3; ++d) acc[c][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[qa][c], xf[qb][d], acc[c][d], 0, 0, 0);
            });
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
    }

    // ---- partial[slab][co][tap*Cin + ci]: lane = input channel, 16 output channels per MFMA block ----
    float* out = p.partial + (size_t)blockIdx.z * p.Cout * Ktot;
    const int ci = chunk * CK + (lane & 31);
    if (ci < p.Cin) {
#pragma unroll
        for (int c = 0; c < CB; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + c * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co >= p.Cout) continue;
#pragma unroll
                for (int d = 0; d < 3; ++d) out[(size_t)co * Ktot + (wid * 3 + d) * p.Cin + ci] = acc[c][d][r];
            }
    }
    // bias gradient (first input-channel chunk only): lanes with equal channel quad differ in bits 2..5 (pixel group)
    if (p.bias_grad && chunk == 0 && t_end > t_begin) {
#pragma unroll
        for (int i = 0; i < YPASS; ++i) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float v = bsum[i][c];
                v += __shfl_xor(v, 4, 64);
                v += __shfl_xor(v, 8, 64);
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if ((lane >> 2) == 0 && y_on[i] && co0 + y_co[i] + c < p.Cout) atomicAdd(p.bias_grad + co0 + y_co[i] + c, v);
            }
        }
    }
}

}  // namespace
