// Device helpers shared by the convolution kernels (conv.hip: exact-f32 MFMA; conv_bf16.hip: split-bf16 MFMA).
#pragma once
#include "common.h"

namespace {

struct RowCoord {
    int n, d, h, w;
};

__device__ __forceinline__ RowCoord decode_row(int m, int D, int H, int W) {
    RowCoord r;
    r.w = m % W;
    int t = m / W;
    r.h = t % H;
    t /= H;
    r.d = t % D;
    r.n = t / D;
    return r;
}

// Decoded position of a thread's 4-channel group inside the flattened K axis.
struct KPos {
    int ci, dz, dy, dx;  // channel, tap offsets relative to the output pixel (already minus padding)
    bool ok;
};

__device__ __forceinline__ KPos decode_k(int k, int Ktot, int Cin, int KW, int KHW, int pd, int ph, int pw) {
    KPos p;
    p.ok = k < Ktot;
    const int tap = k / Cin;
    p.ci = k - tap * Cin;
    const int kz = tap / KHW;
    const int r2 = tap - kz * KHW;
    const int ky = r2 / KW;
    p.dz = kz - pd;
    p.dy = ky - ph;
    p.dx = r2 - ky * KW - pw;
    return p;
}

// Issue the 16-byte load of 4 consecutive input channels of im2col element (pixel rc, position kp).  The value is
// returned RAW (no relu / affine): the fused prologue is applied later, at LDS-store time (finish_a), so that the
// load stays in flight under the MFMAs of the current tile instead of being waited for right here.
__device__ __forceinline__ f32x4 issue_a(const float* __restrict__ x, const RowCoord& rc, bool row_ok, const KPos& kp, int D,
                                         int H, int W, int Cin, int upsample, bool& valid) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    int id = rc.d + kp.dz, ih = rc.h + kp.dy, iw = rc.w + kp.dx;
    valid = row_ok && kp.ok && (unsigned)id < (unsigned)D && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    if (valid) {
        int Hin = H, Win = W;
        if (upsample) {
            ih >>= 1;
            iw >>= 1;
            Hin >>= 1;
            Win >>= 1;
        }
        const size_t off = ((((size_t)rc.n * D + id) * Hin + ih) * Win + iw) * (size_t)Cin + kp.ci;
        v = *reinterpret_cast<const f32x4*>(x + off);
    }
    return v;
}

__device__ __forceinline__ f32x4 finish_a(f32x4 v, bool valid, const float* __restrict__ pre_a, const float* __restrict__ pre_b,
                                          int n, int ci, int Cin, int pre_relu, int pre_group) {
    if (pre_a) {
        if (valid) {
            const size_t g = (size_t)(n / pre_group) * Cin + ci;
            const f32x4 a = *reinterpret_cast<const f32x4*>(pre_a + g);
            const f32x4 b = *reinterpret_cast<const f32x4*>(pre_b + g);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaf(v[j], a[j], b[j]), 0.f);
        }
    } else if (pre_relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    return v;
}

__device__ __forceinline__ float sigmoid_(float x) { return 1.f / (1.f + __expf(-x)); }

// One output element: (acc + addend) * scale + bias, then the fused tail selected by epi_mode.  Shared by the in-kernel
// epilogue and the split-K reduce kernel.
// Per-row quantities of the epilogue, computed once per output row instead of once per element (the integer divisions are
// ~35 VALU instructions each): sample index, 1/sigma of its call group, index of its mask (BatchNorm) group.
struct RowEpi {
    int n;
    float sc;
    int mg;
    size_t rres;  // element offset of this row in the residual tensor
};
// residual_up: the residual is at half resolution [N][H/2][W/2][Cout] and is added with nearest-2x upsampling — the shortcut of an
// upsampling G-block computed BEFORE the upsample: conv1x1(up(x)) == up(conv1x1(x)) exactly (common.py:142-143,154).
__device__ __forceinline__ size_t residual_row_base(const dgmr_conv_args& p, int n, int pix /* index inside the sample */) {
    const int h = pix / p.W, w = pix - h * p.W;
    return (((size_t)n * (p.H >> 1) + (h >> 1)) * (p.W >> 1) + (w >> 1)) * p.Cout;
}
__device__ __forceinline__ RowEpi row_epi(const dgmr_conv_args& p, int row, int DHW) {
    RowEpi e;
    e.n = row / DHW;
    e.sc = p.scale ? p.scale[e.n / p.scale_group] : 1.f;
    e.mg = p.mask_a ? e.n / p.mask_group : 0;
    e.rres = (p.residual && p.residual_up) ? residual_row_base(p, e.n, row - e.n * DHW) : (size_t)row * p.Cout;
    return e;
}

// One output element with the row quantities and the bias value hoisted by the caller.
__device__ __forceinline__ void epilogue_store_row(const dgmr_conv_args& p, float v, const RowEpi& e, float bias_v, int col, size_t idx) {
    if (p.addend) v += p.addend[idx];
    v = fmaf(v, e.sc, bias_v);
    if (p.epi_mode == DGMR_EPI_GRU_GATE) {
        if (p.pre_out) p.pre_out[idx] = v;
        v = sigmoid_(v) * p.gru_h[idx];
    } else if (p.epi_mode == DGMR_EPI_GRU_BLEND) {
        if (p.pre_out) p.pre_out[idx] = v;
        const float s = sigmoid_(p.gru_pu[idx]);
        v = s * p.gru_h[idx] + (1.f - s) * fmaxf(v, 0.f);
    } else {
        if (p.act_relu) v = fmaxf(v, 0.f);
        if (p.residual) v += p.residual[e.rres + col];
        if (p.mask_src) {
            float ms = p.mask_src[idx];
            if (p.mask_a) {
                const size_t g = (size_t)e.mg * p.Cout + col;
                ms = fmaf(ms, p.mask_a[g], p.mask_b[g]);
            }
            v = ms > 0.f ? v : 0.f;
        }
    }
    p.y[idx] = v;
}

__device__ __forceinline__ void epilogue_store(const dgmr_conv_args& p, float v, int n, int col, size_t idx, size_t ridx) {
    if (p.addend) v += p.addend[idx];
    if (p.scale) v *= p.scale[n / p.scale_group];
    if (p.bias) v += p.bias[col];
    if (p.epi_mode == DGMR_EPI_GRU_GATE) {  // r * h with r = sigmoid(pre)   (ConvGRU.py:69-71,78)
        if (p.pre_out) p.pre_out[idx] = v;
        v = sigmoid_(v) * p.gru_h[idx];
    } else if (p.epi_mode == DGMR_EPI_GRU_BLEND) {  // u*h + (1-u)*relu(pre_c)   (ConvGRU.py:80-84)
        if (p.pre_out) p.pre_out[idx] = v;
        const float s = sigmoid_(p.gru_pu[idx]);
        v = s * p.gru_h[idx] + (1.f - s) * fmaxf(v, 0.f);
    } else {
        if (p.act_relu) v = fmaxf(v, 0.f);
        if (p.residual) v += p.residual[ridx];
        if (p.mask_src) {
            float ms = p.mask_src[idx];
            if (p.mask_a) {
                const size_t g = (size_t)(n / p.mask_group) * p.Cout + col;
                ms = fmaf(ms, p.mask_a[g], p.mask_b[g]);
            }
            v = ms > 0.f ? v : 0.f;
        }
    }
    p.y[idx] = v;
}


}  // namespace
