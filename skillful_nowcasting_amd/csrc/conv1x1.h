// 1x1 convolution (forward and data gradient) of the big maps as a streaming GEMM on the bf16 matrix cores, for gfx950 (MI355X).
//
//   Y[m][co] = scale * sum_ci X[m][ci] * W[co][ci] + bias[co]        m = pixel (n, d, h, w) flattened, channels-last
//
// UpsampleGBlock's shortcut and the sampler's convs behind each ConvGRU (dgmr/common.py:139, dgmr/generators.py:62-95) and their data
// gradients: K = 96 ... 768, 48 ... 384 output channels, millions of pixels - bound by the HBM stream of X and Y (16 ... 40 flops per
// byte in bf16x3).  The implicit-GEMM kernel gathers its A tile like a 3x3 conv (row decode, tap decode and bounds per 16-byte item)
// and reached a third of that stream (30 - 53 TF algorithmic).  Here
//   * a workgroup owns 256 consecutive pixels x BN output channels; per 32-channel chunk every thread fetches eight 16-byte items of
//     eight rows (128 contiguous bytes per row and chunk), splits them into bf16 planes and stores them with the window
//     kernels' swizzle (conv_win_glds.h); the items of chunk c + 1 are in flight in registers while chunk c is multiplied;
//   * the weights of chunk c + 1 arrive by LDS-DMA under the MFMAs of chunk c (two stages), as in the window kernels;
//   * epilogue: the window kernels' 16-byte one (quad transposes, 1/sigma, bias, 16-byte stores).
// A tile never straddles two samples (D * H * W % 256 == 0 is the library's condition), so 1/sigma is one scalar per workgroup.
#pragma once
#include "conv_win_glds.h"  // lds_dma16, dma_drain, mfma_blk, lds_swz, quad_transpose

namespace {

template <int BN, int NS>
__global__ __launch_bounds__(256, 2) void conv1x1_kernel(const dgmr_conv_args p, const int M, const int rows_per_sample) {
    constexpr int CK = 32, ROW = CK / 2, BM = 256;
    constexpr int NP = planes_of<NS>::value;
    constexpr int TM = 2, TN = BN / 32;   // a wave: 64 rows x BN columns in 32 x 32 blocks
    constexpr int APASS = BM * 8 / 256;   // 16-byte items per thread and chunk
    constexpr int BUNITS = BN * 4 * NP, BPASS = (BUNITS + 255) / 256;
    constexpr int BSTAGE = NP * BN * ROW;
    static_assert(BN % 32 == 0 && BUNITS % 64 == 0, "tile");
    typedef float accv_t __attribute__((ext_vector_type(16)));

    __shared__ __attribute__((aligned(16))) uint32_t smem[NP * BM * ROW + 2 * BSTAGE];
    uint32_t* As = smem;                    // [plane][row][ROW]
    uint32_t* Bs = smem + NP * BM * ROW;    // [stage][plane][co][ROW]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int m0 = (int)blockIdx.x * BM, n0 = (int)blockIdx.y * BN;
    const int nchunks = (p.Cin + CK - 1) / CK;
    const int cq = tid & 7;

    // ---- A: this thread's rows (tid >> 3) + 32 i; rows beyond M are fetched from the last row and stored as zeros ----
    uint32_t a_goff[APASS];
    unsigned a_valid = 0;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
        const int r = (tid >> 3) + i * 32;
        const bool ok = m0 + r < M;
        a_goff[i] = (uint32_t)(ok ? m0 + r : M - 1) * (uint32_t)p.Cin + cq * 4;
        a_valid |= (ok ? 1u : 0u) << i;
    }
    f32x4 ra[APASS];
    auto issue_a = [&](int chunk) {
        const int cb = chunk * CK + cq * 4;
        const uint32_t shift = cb < p.Cin ? (uint32_t)(chunk * CK) : 0u;  // (channels beyond Cin: any valid address, zeroed below)
#pragma unroll
        for (int i = 0; i < APASS; ++i) ra[i] = *reinterpret_cast<const f32x4*>(p.x + a_goff[i] + shift - (cb < p.Cin ? 0u : (uint32_t)(cq * 4)));
    };
    auto store_a = [&](int chunk) {
        const bool kok = chunk * CK + cq * 4 < p.Cin;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            const int r = (tid >> 3) + i * 32;
            f32x4 v = ra[i];
            if (p.pre_relu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            v = (kok && ((a_valid >> i) & 1u)) ? v : zero4;
            u32x2 pl[NP];
            split_planes4<NP>(v, pl);
            uint32_t* dst = As + r * ROW + (((cq >> 1) ^ lds_swz<false>(r)) << 2) + (cq & 1) * 2;
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<u32x2*>(dst + q * BM * ROW) = pl[q];
        }
    };

    // ---- B: stage = one 32-channel chunk of the BN rows, by LDS-DMA (unit u of a stage = 16 bytes at LDS offset 16 u) ----
    const size_t plane_stride = (size_t)p.Cout * p.Cin;  // bf16 elements per plane
    uint32_t b_off[BPASS], b_tail[BPASS];
    const int c_last = (nchunks - 1) * CK;
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
        const int u = min(tid + i * 256, BUNITS - 1);
        const int plane = u / (BN * 4);
        const int r = (u >> 2) % BN;
        const int ch = ((u & 3) ^ lds_swz<false>(r)) * 8;
        const uint32_t row = (uint32_t)(plane * plane_stride) + (uint32_t)min(n0 + r, p.Cout - 1) * (uint32_t)p.Cin;
        b_off[i] = row + ch;
        b_tail[i] = row + (c_last + ch < p.Cin ? c_last + ch : 0);
    }
    const bool has_tail = (p.Cin & (CK - 1)) != 0;
    auto dma_b = [&](int chunk, int stage) {
        const bool tail = has_tail && chunk == nchunks - 1;
        const uint16_t* base = p.w_split + (tail ? 0 : chunk * CK);
#pragma unroll
        for (int i = 0; i < BPASS; ++i)
            if (i * 256 + wid * 64 < BUNITS) lds_dma16(base + (tail ? b_tail[i] : b_off[i]), Bs + stage * BSTAGE + (i * 256 + wid * 64) * 4);
    };

    accv_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int kg = lane >> 5;
    const int bsw = lds_swz<false>(lane);
    const uint32_t* Bb0 = Bs + (lane & 31) * ROW;
    const bool tail16 = (p.Cin & (CK - 1)) != 0 && (p.Cin & (CK - 1)) <= 16;
    auto mma = [&](int stage, bool half) {
        const uint32_t* Ab[TM];
        int asw[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int r = wid * 64 + i * 32 + (lane & 31);
            Ab[i] = As + r * ROW;
            asw[i] = lds_swz<false>(r);
        }
        const uint32_t* Bb = Bb0 + stage * BSTAGE;
#pragma unroll
        for (int kk = 0; kk < CK / 16; ++kk) {
            if (kk == 1 && half) break;  // (wave-uniform)
            const int ks = kk * 2 + kg;
            bf16x8_t af[NP][TM], bf[NP][TN];
            const int ob = (ks ^ bsw) << 2;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int t = 0; t < NP; ++t) {  // (in the order the products consume them: schedule_split_products, conv_bf16.h)
                const int qa = NP - 1 - t, qb = t;
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[qa][i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(Ab[i] + qa * BM * ROW + ((ks ^ asw[i]) << 2)));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bf[qb][j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(Bb + (qb * BN + j * 32) * ROW + ob));
            }
            for_each_product<NP>([&](auto qa, auto qb) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma_blk<false>(af[qa][i], bf[qb][j], acc[i][j]);
            });
            schedule_split_products<NP, TM, TN>();
            __builtin_amdgcn_s_setprio(0);
        }
    };

    issue_a(0);
    dma_b(0, 0);
    store_a(0);
    dma_drain();
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < nchunks; ++c) {
        const int st = c & 1;
        const bool more = c + 1 < nchunks;
        if (more) {  // (loads issued here are consumed in this same iteration: nothing in flight across the loop edge)
            issue_a(c + 1);
            dma_b(c + 1, st ^ 1);
        }
        mma(st, tail16 && c == nchunks - 1);
        __syncthreads();  // every wave is done with this chunk's rows
        if (more) store_a(c + 1);
        dma_drain();
        __syncthreads();
    }

    // ---- 16-byte epilogue (Cout % 4 == 0, aligned tensors: the library's condition) ----
    const float sc = p.scale ? p.scale[(m0 / rows_per_sample) / p.scale_group] : 1.f;
    const int j4 = lane & 3, q4 = (lane & 31) >> 2, rsel = lane >> 5;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col4 = n0 + j * 32 + 4 * q4;
        const bool cok = col4 < p.Cout;
        const f32x4 b4 = (p.bias && cok) ? *reinterpret_cast<const f32x4*>(p.bias + col4) : zero4;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int m = m0 + wid * 64 + i * 32 + j4 + 8 * g + 4 * rsel;
                f32x4 v = quad_transpose(acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3], lane);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    v[c] = fmaf(v[c], sc, b4[c]);
                    if (p.act_relu) v[c] = fmaxf(v[c], 0.f);
                }
                if (cok && m < M) *reinterpret_cast<f32x4*>(p.y + (size_t)m * p.Cout + col4) = v;
            }
        }
    }
}

}  // namespace
