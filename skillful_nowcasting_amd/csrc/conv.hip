// Implicit-GEMM convolution kernels for gfx950 (MI355X), fp32 on v_mfma_f32_32x32x2_f32.
//
// Forward / data-gradient:   Y[m][co] = sum_k A[m][k] * W[co][k]
//     m  = output pixel (n,d,h,w) flattened (channels-last activations => A rows are contiguous in ci)
//     k  = (kd,kh,kw,ci) flattened; Cin % 4 == 0 so a 16-byte load never straddles a tap
//     A is gathered on the fly (zero padding, optional nearest-2x upsample, optional relu / affine+relu)
// Weight-gradient:           G[co][k] = sum_m dY[m][co] * A[m][k]     (split over m, partial slabs)
//
// Tiling is for 64-wide wavefronts: a workgroup of WM x WN waves owns a BM x BN output tile, each wave
// a (BM/WM) x (BN/WN) sub-tile made of 32x32 MFMA blocks (16 accumulator VGPRs each).  Operands are
// staged global -> registers -> LDS (double-buffered, one barrier per K step) so the next tile's HBM
// latency hides under the current tile's MFMAs; LDS rows are padded to BK+4 floats, which makes the
// ds_read_b128 fragment reads (lane = row, 4 consecutive k) conflict-free.
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "conv_bf16.h"  // split_weights_kernel (the bf16 MFMA kernels themselves are instantiated in tu_*.hip, see conv_launch.h)
#include "conv_device.h"
#include "conv_launch.h"
#include "conv_stem4.h"

namespace {

template <int BM, int BN, int BK, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_kernel(const dgmr_conv_args p, const int M, const int Ktot,
                                                                   const int kt_per_split) {
    constexpr int NT = 64 * WM * WN;
    constexpr int LD = BK + 4;
    constexpr int KQ = BK / 4;
    constexpr int RPP = NT / KQ;  // tile rows filled per pass
    constexpr int AP = BM / RPP, BP = BN / RPP;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(AP >= 1 && BP >= 1 && TM >= 1 && TN >= 1, "bad tile");
    static_assert(BM % RPP == 0 && BN % RPP == 0, "bad tile");

    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LD];
    float* As = smem;
    float* Bs = smem + 2 * BM * LD;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int kq = tid % KQ, lrow = tid / KQ;

    const int KHW = p.KH * p.KW;
    const int pd = p.KD >> 1, ph = p.KH >> 1, pw = p.KW >> 1;
    const size_t wrow = (size_t)p.KD * KHW * p.w_cin;  // weight row stride (floats)

    RowCoord rc[AP];
    bool rok[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int m = m0 + i * RPP + lrow;
        rok[i] = m < M;
        rc[i] = decode_row(rok[i] ? m : 0, p.D, p.H, p.W);
    }

    f32x4 ra[AP], rb[BP];
    bool va[AP];
    int cur_ci = 0;
    auto load_tiles = [&](int kt) {
        const int k = kt * BK + kq * 4;
        const KPos kp = decode_k(k, Ktot, p.Cin, p.KW, KHW, pd, ph, pw);
        cur_ci = kp.ci;
#pragma unroll
        for (int i = 0; i < AP; ++i) ra[i] = issue_a(p.x, rc[i], rok[i], kp, p.D, p.H, p.W, p.Cin, p.upsample, va[i]);
        // weights may be an input-channel slice [w_coff, w_coff + Cin) of a [Cout][taps][w_cin] tensor (ConvGRU x / h parts)
        const int tap = k / p.Cin;
        const size_t wk = (size_t)tap * p.w_cin + p.w_coff + kp.ci;
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            const int co = n0 + i * RPP + lrow;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (co < p.Cout && k < Ktot) v = *reinterpret_cast<const f32x4*>(p.w + (size_t)co * wrow + wk);
            rb[i] = v;
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AP; ++i)
            *reinterpret_cast<f32x4*>(&As[buf * BM * LD + (i * RPP + lrow) * LD + kq * 4]) =
                finish_a(ra[i], va[i], p.pre_a, p.pre_b, rc[i].n, cur_ci, p.Cin, p.pre_relu, p.pre_group);
#pragma unroll
        for (int i = 0; i < BP; ++i) *reinterpret_cast<f32x4*>(&Bs[buf * BN * LD + (i * RPP + lrow) * LD + kq * 4]) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // split-K: blockIdx.z owns k tiles [kt0, kt1); partial sums go to the workspace, the reduce kernel finishes
    const int nk_all = (Ktot + BK - 1) / BK;
    const int kt0 = blockIdx.z * kt_per_split;
    const int nk = min(nk_all, kt0 + kt_per_split);
    load_tiles(kt0);
    store_tiles(0);
    __syncthreads();
    for (int kt = kt0; kt < nk; ++kt) {
        const int cur = (kt - kt0) & 1;
        if (kt + 1 < nk) load_tiles(kt + 1);
        const float* Ab = As + cur * BM * LD + (wm * TM * 32 + (lane & 31)) * LD + (lane >> 5) * 4;
        const float* Bb = Bs + cur * BN * LD + (wn * TN * 32 + (lane & 31)) * LD + (lane >> 5) * 4;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            f32x4 a4[TM], b4[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a4[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LD + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) b4[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LD + kk * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i][s], b4[j][s], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tiles(cur ^ 1);
        __syncthreads();
    }

    const int DHW = p.D * p.H * p.W;
    // Epilogue.  Lane = output channel, 16 rows per MFMA block; row quantities (sample, 1/sigma, mask group) once per row, bias
    // once per column.
    float* ws = gridDim.z > 1 ? p.splitk_ws + (size_t)blockIdx.z * M * p.Cout : nullptr;
    float bj[TN];
    int colj[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        colj[j] = n0 + wn * TN * 32 + j * 32 + (lane & 31);
        bj[j] = (p.bias && colj[j] < p.Cout) ? p.bias[colj[j]] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row >= M) continue;
            const size_t rbase = (size_t)row * p.Cout;
            if (ws) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (colj[j] < p.Cout) ws[rbase + colj[j]] = acc[i][j][r];
            } else {
                const RowEpi e = row_epi(p, row, DHW);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (colj[j] < p.Cout) epilogue_store_row(p, acc[i][j][r], e, bj[j], colj[j], rbase + colj[j]);
            }
        }
    }
}

// y = epilogue(sum_z ws[z]) for a split-K launch; one thread per output element (these problems are small by construction).
__global__ void splitk_reduce_kernel(const dgmr_conv_args p, const int M, const int S) {
    const size_t total = (size_t)M * p.Cout;
    const int DHW = p.D * p.H * p.W;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int z = 0; z < S; ++z) v += p.splitk_ws[(size_t)z * total + idx];
        const int row = idx / p.Cout, col = idx - (size_t)row * p.Cout;
        const int n = row / DHW;
        const size_t ridx = (p.residual && p.residual_up) ? residual_row_base(p, n, row - n * DHW) + col : idx;
        epilogue_store(p, v, n, col, idx, ridx);
    }
}

// The same for Cout % 4 == 0 and M * Cout < 2^31 (every ConvGRU step): four columns per thread, the S partial sums and all the
// epilogue's operands (addend, ConvGRU state, residual, mask source) fetched as independent 16-byte loads before any arithmetic -
// the scalar kernel above chains ~S + 5 dependent memory round trips and three integer divisions per element.
__global__ void splitk_reduce4_kernel(const dgmr_conv_args p, const int M, const int S) {
    const uint32_t total4 = (uint32_t)((size_t)M * p.Cout / 4);
    const uint32_t DHW = (uint32_t)(p.D * p.H * p.W);
    const uint32_t c4 = (uint32_t)p.Cout / 4;
    const f32x4* ws = reinterpret_cast<const f32x4*>(p.splitk_ws);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += gridDim.x * blockDim.x) {
        f32x4 a0 = zero4, a1 = zero4, a2 = zero4, a3 = zero4;
        int z = 0;
        for (; z + 4 <= S; z += 4) {
            a0 += ws[(size_t)z * total4 + i];
            a1 += ws[(size_t)(z + 1) * total4 + i];
            a2 += ws[(size_t)(z + 2) * total4 + i];
            a3 += ws[(size_t)(z + 3) * total4 + i];
        }
        for (; z < S; ++z) a0 += ws[(size_t)z * total4 + i];
        const uint32_t row = i / c4, col = (i - row * c4) * 4;
        const uint32_t n = row / DHW;
        const size_t idx = (size_t)i * 4;
        f32x4 ad = zero4, hv = zero4, pv = zero4, rs = zero4, ms = zero4, bs = zero4, ma = {1.f, 1.f, 1.f, 1.f}, mb = zero4;
        const float sc = p.scale ? p.scale[n / p.scale_group] : 1.f;
        if (p.addend) ad = *reinterpret_cast<const f32x4*>(p.addend + idx);
        if (p.bias) bs = *reinterpret_cast<const f32x4*>(p.bias + col);
        if (p.epi_mode != DGMR_EPI_PLAIN) hv = *reinterpret_cast<const f32x4*>(p.gru_h + idx);
        if (p.epi_mode == DGMR_EPI_GRU_BLEND) pv = *reinterpret_cast<const f32x4*>(p.gru_pu + idx);
        if (p.epi_mode == DGMR_EPI_PLAIN && p.residual)
            rs = *reinterpret_cast<const f32x4*>(p.residual + (p.residual_up ? residual_row_base(p, n, row - n * DHW) + col : idx));
        if (p.epi_mode == DGMR_EPI_PLAIN && p.mask_src) {
            ms = *reinterpret_cast<const f32x4*>(p.mask_src + idx);
            if (p.mask_a) {
                const size_t g = (size_t)(n / p.mask_group) * p.Cout + col;
                ma = *reinterpret_cast<const f32x4*>(p.mask_a + g);
                mb = *reinterpret_cast<const f32x4*>(p.mask_b + g);
            }
        }
        f32x4 v = ((a0 + a1) + (a2 + a3)) + ad;
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = p.scale ? v[j] * sc : v[j];
            v[j] += bs[j];
            if (p.epi_mode == DGMR_EPI_GRU_GATE) {
                o[j] = sigmoid_(v[j]) * hv[j];
            } else if (p.epi_mode == DGMR_EPI_GRU_BLEND) {
                const float sg = sigmoid_(pv[j]);
                o[j] = sg * hv[j] + (1.f - sg) * fmaxf(v[j], 0.f);
            } else {
                float t = p.act_relu ? fmaxf(v[j], 0.f) : v[j];
                if (p.residual) t += rs[j];
                if (p.mask_src) t = fmaf(ms[j], ma[j], mb[j]) > 0.f ? t : 0.f;
                o[j] = t;
            }
        }
        if (p.epi_mode != DGMR_EPI_PLAIN && p.pre_out) *reinterpret_cast<f32x4*>(p.pre_out + idx) = v;
        *reinterpret_cast<f32x4*>(p.y + idx) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// Optional per-launch timing (bench.py's roofline leg): HIP events recorded on the launch stream around every
// conv / wgrad kernel, grouped by tile variant.  Off by default; never used inside the timed region.
// ------------------------------------------------------------------------------------------------
int g_precision = 0;  // (documented below, with the dispatch switches)
struct ProfRec {
    int variant;
    uint32_t detail;  // the instantiated kernel behind the class row: dgmr_profile_detail (prof_detail_* below)
    double flops;     // algorithmic: 2 * MACs of the dense convolution as the reference states it
    double executed;  // flops the matrix pipe really issues: x 16/36 for the phase / pooled decompositions, x MFMAs per product (3 in
                      // bf16x3, 6 in bf16x6) - the mode is the launch's own, a step may mix them
    hipEvent_t e0, e1;
};
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_event_pool;
std::mutex g_prof_mu;

hipEvent_t prof_event() {
    if (!g_event_pool.empty()) {
        hipEvent_t e = g_event_pool.back();
        g_event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

struct ProfScope {
    bool on;
    ProfRec r;
    hipStream_t s;
    ProfScope(int variant, double flops, hipStream_t st, double exec_frac = 1.0, uint32_t detail = 0) : on(g_prof_on), s(st) {
        if (!on) return;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        r.variant = variant;
        r.detail = detail | ((uint32_t)g_precision << 28);
        r.flops = flops;
        r.executed = flops * exec_frac * (g_precision == 1 ? 3.0 : (g_precision == 3 ? 6.0 : 1.0));
        r.e0 = prof_event();
        r.e1 = prof_event();
        (void)hipEventRecord(r.e0, s);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(r.e1, s);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof.push_back(r);
    }
};

const char* const kVariantNames[V_COUNT] = {
    "conv_fwd_dgrad<128,128>", "conv_fwd_dgrad<64,64>", "conv_fwd_dgrad<128,96>", "conv_fwd_dgrad<128,64>",
    "conv_fwd_dgrad<128,32>",  "conv_wgrad<128|96,128>", "conv_wgrad<64,128>", "conv_wgrad<32,128>",
    "conv_fwd_dgrad_win3x3<128px,128>", "conv_fwd_dgrad_win3x3<128px,96>", "conv_fwd_dgrad_win3x3<128px,64>"};

// 0: exact fp32 (v_mfma_f32_32x32x2_f32)   1: bf16x3 (two bf16 planes per operand, three MFMAs per product: 16-bit products)
// 2: plain bf16   3: bf16x6 (three planes, six MFMAs: products as accurate as an fp32 product's own rounding)
// (int g_precision: defined above ProfRec)

#define DGMR_BY_NS(fn, ...) \
    (g_precision == 1 ? dgmr_tu::fn##_ns3(__VA_ARGS__) : (g_precision == 2 ? dgmr_tu::fn##_ns1(__VA_ARGS__) : dgmr_tu::fn##_ns6(__VA_ARGS__)))
// dgmr_conv_tune(): -1 = automatic
int g_tune_variant = -1, g_tune_ksplit = -1, g_tune_window = -1, g_tune_wgrad_window = -1;
int g_debug_flags = 0;  // dgmr_debug_flags(): kernel-phase timing switches of tools/conv_bench.py, 0 in every product launch
static const bool g_thin_auto = []() {  // DGMR_THIN_TILE=0: A/B switch for the 16-column tile of <= 16-channel outputs
    const char* e = getenv("DGMR_THIN_TILE");
    return !(e && e[0] == '0');
}();
bool g_m16_auto = true;  // 16-column blocks for <= 48 output channels: measured +14 ... +28 % on the 48-channel layers (tune window 3 = the 64-column tile)

// WM x WN: wave grid of the f32 kernel (the bf16 kernels of the same tile: tu_gemm.hip)
template <int VARIANT, int BM, int BN, int WM, int WN>
int launch_conv(const dgmr_conv_args& a, int M, int Ktot, hipStream_t s) {
    const int BK = 32;
    const int nk = (Ktot + BK - 1) / BK;
    const int S = a.ksplit > 1 ? a.ksplit : 1;
    const int per = (nk + S - 1) / S;
    const int Seff = (nk + per - 1) / per;  // no empty splits
    dim3 grid((M + BM - 1) / BM, (a.Cout + BN - 1) / BN, Seff);
    if (g_precision != 0)
        DGMR_BY_NS(launch_gemm, VARIANT, a, M, Ktot, per, grid, s);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, 32, WM, WN>), grid, dim3(64 * WM * WN), 0, s, a, M, Ktot, per);
    if (Seff > 1) {
        const size_t total = (size_t)M * a.Cout;
        if (a.Cout % 4 == 0 && total < (1ull << 31)) {
            const int blocks = (int)std::min<size_t>((total / 4 + 255) / 256, 2048);
            hipLaunchKernelGGL(splitk_reduce4_kernel, dim3(blocks), dim3(256), 0, s, a, M, Seff);
        } else {
            const int blocks = (int)std::min<size_t>((total + 255) / 256, 2048);
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, a, M, Seff);
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------
template <int BI, int BJ, int WI, int WJ>
__global__ __launch_bounds__(64 * WI * WJ) void conv_wgrad_kernel(const dgmr_wgrad_args p, const int M, const int Ktot,
                                                                   const int rows_per_split, const int splits_per_group,
                                                                   const int rows_per_group) {
    constexpr int NT = 64 * WI * WJ;
    constexpr int BR = 32;               // pixels per reduction step
    constexpr int TPR = NT / BR;         // threads per tile row (8 for NT = 256)
    constexpr int YP = BI / 4 / TPR;     // float4 per thread per row, dY tile
    constexpr int XP = BJ / 4 / TPR;     // float4 per thread per row, X tile
    constexpr int LDY = BI + 4, LDX = BJ + 4;
    constexpr int TM = BI / WI / 32, TN = BJ / WJ / 32;
    static_assert(YP >= 1 && XP >= 1 && TM >= 1 && TN >= 1, "bad tile");

    __shared__ __attribute__((aligned(16))) float smem[2 * BR * (LDY + LDX)];
    float* Ys = smem;
    float* Xs = smem + 2 * BR * LDY;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wi = wid / WJ, wj = wid % WJ;
    const int k0 = blockIdx.x * BJ, co0 = blockIdx.y * BI;
    // slab z covers rows of ONE group (samples that share a spectral-norm sigma): group = z / splits_per_group
    const int grp = blockIdx.z / splits_per_group;
    const int r_begin = grp * rows_per_group + (blockIdx.z - grp * splits_per_group) * rows_per_split;
    const int r_end = min(min(M, (grp + 1) * rows_per_group), r_begin + rows_per_split);
    const int lr = tid / TPR, lq = tid % TPR;

    const int KHW = p.KH * p.KW;
    const int pd = p.KD >> 1, ph = p.KH >> 1, pw = p.KW >> 1;

    // this thread's XP positions on the K axis never change: decode them once
    KPos kps[XP];
#pragma unroll
    for (int i = 0; i < XP; ++i) kps[i] = decode_k(k0 + (lq + i * TPR) * 4, Ktot, p.Cin, p.KW, KHW, pd, ph, pw);

    f32x4 ry[YP], rx[XP];
    f32x4 bsum[YP];
#pragma unroll
    for (int i = 0; i < YP; ++i) bsum[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bool vx[XP];
    int cur_n = 0;
    auto load_tiles = [&](int r0) {
        const int m = r0 + lr;
        const bool ok = m < r_end;
        const RowCoord rc = decode_row(ok ? m : 0, p.D, p.H, p.W);
        cur_n = rc.n;
#pragma unroll
        for (int i = 0; i < YP; ++i) {
            const int co = co0 + (lq + i * TPR) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok && co < p.Cout) v = *reinterpret_cast<const f32x4*>(p.dy + (size_t)m * p.Cout + co);
            ry[i] = v;
            bsum[i] += v;
        }
#pragma unroll
        for (int i = 0; i < XP; ++i) rx[i] = issue_a(p.x, rc, ok, kps[i], p.D, p.H, p.W, p.Cin, p.upsample, vx[i]);
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < YP; ++i) *reinterpret_cast<f32x4*>(&Ys[buf * BR * LDY + lr * LDY + (lq + i * TPR) * 4]) = ry[i];
#pragma unroll
        for (int i = 0; i < XP; ++i)
            *reinterpret_cast<f32x4*>(&Xs[buf * BR * LDX + lr * LDX + (lq + i * TPR) * 4]) =
                finish_a(rx[i], vx[i], p.pre_a, p.pre_b, cur_n, kps[i].ci, p.Cin, p.pre_relu, p.pre_group);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nr = (r_end - r_begin + BR - 1) / BR;
    if (nr > 0) {
        load_tiles(r_begin);
        store_tiles(0);
    }
    __syncthreads();
    for (int it = 0; it < nr; ++it) {
        const int cur = it & 1;
        if (it + 1 < nr) load_tiles(r_begin + (it + 1) * BR);
        const float* Yb = Ys + cur * BR * LDY + (lane >> 5) * LDY + wi * TM * 32 + (lane & 31);
        const float* Xb = Xs + cur * BR * LDX + (lane >> 5) * LDX + wj * TN * 32 + (lane & 31);
#pragma unroll
        for (int rr = 0; rr < BR; rr += 2) {
            float a1[TM], b1[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a1[i] = Yb[rr * LDY + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b1[j] = Xb[rr * LDX + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[j], acc[i][j], 0, 0, 0);
        }
        if (it + 1 < nr) store_tiles(cur ^ 1);
        __syncthreads();
    }

    float* out = p.partial + (size_t)blockIdx.z * p.Cout * Ktot;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wi * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (co >= p.Cout) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int k = k0 + wj * TN * 32 + j * 32 + (lane & 31);
                if (k < Ktot) out[(size_t)co * Ktot + k] = acc[i][j][r];
            }
        }
    // bias gradient (column sums of dY over this slab): thread (lr, lq) summed rows lr, lr+.. of its channels
    if (p.bias_grad && blockIdx.x == 0 && nr > 0) {
#pragma unroll
        for (int i = 0; i < YP; ++i) {
            const int co = co0 + (lq + i * TPR) * 4;
            if (co < p.Cout)
#pragma unroll
                for (int c = 0; c < 4; ++c) atomicAdd(p.bias_grad + co + c, bsum[i][c]);
        }
    }
}

__global__ void flip_weights_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int Cin, int taps, int w_cin,
                                    int w_coff) {
    // wt[ci][taps-1-t][co] = w[co][t][w_coff + ci]; one thread per (ci, t, co) destination element, co fastest.
    const size_t total = (size_t)Cout * Cin * taps;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int co = i % Cout;
        const size_t r = i / Cout;
        const int t = r % taps;
        const int ci = r / taps;
        wt[i] = w[((size_t)co * taps + (taps - 1 - t)) * w_cin + w_coff + ci];
    }
}

// g[i] = sum_grp scale[grp] * P_grp[i],  dot[grp] += <P_grp, w>   with P_grp = sum of the group's slabs.
// groups = calls of the module that one batched launch covers: forecast steps / frames (<= 32) or, with the generator draws of
// a step batched as well, draws x steps (6 x 18 = 108 at the paper configuration)
constexpr int WG_MAX_GROUPS = 128;
// Partial element i = (co, tap, ci) of a [Cout][taps][cs] slab lands at j = (co*taps + tap)*ct + coff + ci of g / w
// (cs == ct, coff == 0: j == i).
// (two workgroups per CU = 256 registers per lane: the MAXG running dot products plus the 16-byte lanes spilled 636 bytes per lane
//  under the compiler's own occupancy target of four)
template <int MAXG, bool VEC4 = true>
__global__ __launch_bounds__(256, 2) void wgrad_reduce_kernel(const float* __restrict__ partial, int nsplit, int groups, size_t numel,
                                    const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ g,
                                    float* __restrict__ dot, int taps, int cs, int ct, int coff, int dot_rows) {
    __shared__ float red[MAXG][4];
    const int spg = nsplit / groups;
    const size_t rowlen = (size_t)taps * cs;
    float d[MAXG];  // one running <P_q, W> per group, in registers (MAXG = 128: ~3 waves per SIMD, still an HBM-bound stream)
#pragma unroll
    for (int q = 0; q < MAXG; ++q) d[q] = 0.f;
    // four consecutive elements per thread (16-byte loads: the slab walk of one element is a chain of nsplit dependent-address
    // loads, 4-byte lanes left the stream at 0.5 TB/s); a quad never leaves its (co, tap) row: cs, ct, coff are multiples of 4
    const bool vec4 = VEC4 && (numel & 3) == 0 && (cs & 3) == 0 && (ct & 3) == 0 && (coff & 3) == 0 &&
                      (reinterpret_cast<uintptr_t>(partial) & 15) == 0 && (reinterpret_cast<uintptr_t>(g) & 15) == 0 &&
                      (!w || (reinterpret_cast<uintptr_t>(w) & 15) == 0);
    if (vec4) {
        const size_t n4 = numel >> 2;
        for (size_t i4 = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i4 < n4; i4 += (size_t)gridDim.x * blockDim.x) {
            const size_t i = i4 << 2;
            size_t j = i;
            if (cs != ct) {
                const size_t co = i / rowlen, r = i - co * rowlen;
                const size_t tap = r / cs, ci = r - tap * cs;
                j = (co * taps + tap) * ct + coff + ci;
            }
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            const f32x4 wi = w ? *reinterpret_cast<const f32x4*>(w + j) : zero;
            f32x4 tot = zero;
#pragma unroll
            for (int q = 0; q < MAXG; ++q) {
                if (q < groups) {
                    f32x4 s = zero;
                    for (int k = 0; k < spg; ++k) s += *reinterpret_cast<const f32x4*>(partial + (size_t)(q * spg + k) * numel + i);
                    const float sq = scale ? scale[q] : 1.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {  // the element order of the scalar path: bit-identical sums per element
                        tot[e] = fmaf(s[e], sq, tot[e]);
                        d[q] = fmaf(s[e], wi[e], d[q]);
                    }
                }
            }
            *reinterpret_cast<f32x4*>(g + j) = tot;
        }
    }
    for (size_t i = vec4 ? numel : blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < numel; i += (size_t)gridDim.x * blockDim.x) {
        size_t j = i;
        if (cs != ct) {
            const size_t co = i / rowlen, r = i - co * rowlen;
            const size_t tap = r / cs, ci = r - tap * cs;
            j = (co * taps + tap) * ct + coff + ci;
        }
        const float wi = w ? w[j] : 0.f;
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < MAXG; ++q) {
            if (q < groups) {
                float s = 0.f;
                for (int k = 0; k < spg; ++k) s += partial[(size_t)(q * spg + k) * numel + i];
                tot = fmaf(s, scale ? scale[q] : 1.f, tot);
                d[q] = fmaf(s, wi, d[q]);
            }
        }
        g[j] = tot;
    }
    if (dot) {
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
        for (int q = 0; q < MAXG; ++q) {
            if (q < groups) {
                const float v = wave_sum(d[q]);
                if (lane == 0) red[q][wid] = v;
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < groups) {
            const float v = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
            // deterministic mode: this workgroup's row behind the first `groups` floats of dot (wgrad_dot_finish_kernel adds the rows in order)
            if (dot_rows) dot[(size_t)(1 + blockIdx.x) * groups + threadIdx.x] = v;
            else atomicAdd(dot + threadIdx.x, v);
        }
    }
}

// dot[q] += sum over workgroup rows b of dot[(1 + b) * groups + q] in a FIXED order: eight lanes per group take rows b = lane (mod 8)
// in increasing order, then the eight partial sums are added in lane order (1024 rows as 128 dependent adds instead of 1024)
__global__ __launch_bounds__(1024) void wgrad_dot_finish_kernel(float* __restrict__ dot, int groups, int nb) {
    __shared__ float part[8][WG_MAX_GROUPS];
    const int q = threadIdx.x & (WG_MAX_GROUPS - 1), l = threadIdx.x / WG_MAX_GROUPS;
    float a = 0.f;
    if (q < groups) {
        int b = l;
        for (; b + 56 < nb; b += 64) {  // eight independent loads in flight, added in row order
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = dot[(size_t)(1 + b + 8 * k) * groups + q];
#pragma unroll
            for (int k = 0; k < 8; ++k) a += v[k];
        }
        for (; b < nb; b += 8) a += dot[(size_t)(1 + b) * groups + q];
    }
    part[l][q] = a;
    __syncthreads();
    if (l == 0 && q < groups) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += part[k][q];
        dot[q] += t;
    }
}

// gw (+)= g - sum_grp dot[grp] * inv_sigma[grp]^2 * u[grp][co] * v[grp][perm(k)]   (g already carries the 1/sigma factors)
__global__ void sn_wgrad_finalize_kernel(const float* __restrict__ g, float* __restrict__ gw, const float* __restrict__ dot,
                                         const float* __restrict__ inv_sigma, const float* __restrict__ u,
                                         const float* __restrict__ v, int Cout, int Cin, int taps, int groups, int accumulate) {
    const size_t K = (size_t)Cin * taps;
    const size_t total = (size_t)Cout * K;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int co = i / K;
        const int k = i - (size_t)co * K;
        const int t = k / Cin, ci = k - t * Cin;
        float val = g[i];
        if (u) {
            const size_t vj = (size_t)ci * taps + t;
            for (int q = 0; q < groups; ++q) {
                const float is = inv_sigma[q];
                val = fmaf(-dot[q] * is * is * u[(size_t)q * Cout + co], v[(size_t)q * K + vj], val);
            }
        }
        gw[i] = accumulate ? gw[i] + val : val;
    }
}

__global__ void zero_n_kernel(float* p, int n) {
    if ((int)threadIdx.x < n) p[threadIdx.x] = 0.f;
}

}  // namespace

static int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return n;
}
// DGMR_WS_AUTO=1 in the environment: the library picks the wave-specialised window kernel by itself wherever it applies and every
// CU gets a run of items (A/B runs of the whole step).  Off by default: measured on the paper configuration's layers it is
// bit-identical to the one-role kernels and faster on launches of at most one workgroup per CU (+10 ... +40 %), but 10 - 25 % slower
// than the 256-pixel-tile one-role kernel on the big launches (profiles/r04_ws_*.log)
static const bool g_ws_auto = []() {
    const char* e = getenv("DGMR_WS_AUTO");
    return e && e[0] == '1';
}();

// DGMR_PHASE_PAIR=1 (or dgmr_debug_flags 256): phase launches with one workgroup per ROW parity (conv_win_glds.h PAIR).  Off by default:
// measured in isolation (tools/conv_bench.py, profiles/r05_phase_pair_*.log) 4 % / 3 % SLOWER on the two biggest upsampling convs
// (up_g4.first 8.79 vs 8.43 ms, up_g3.first 7.00 vs 6.78 ms) and 2.5 % faster on up_g2.first - half the halo staging per MFMA, but a
// 128-pixel tile with one block row per wave reads 1.33 LDS fragments per MFMA where the 256-pixel tile reads 0.83
static const bool g_phase_pair = []() {
    const char* e = getenv("DGMR_PHASE_PAIR");
    return e && e[0] == '1';
}();

// Which LDS-window 3x3 kernel (if any) takes a conv, and with which tiling: shared by the launch and by dgmr_conv_stats_rows.
static bool window_plan(const dgmr_conv_args& p, WinPlan* w) {
    const int64_t M64 = (int64_t)p.N * p.D * p.H * p.W;
    const int C = p.Cout;
    // 8x8 maps: a tile is two whole images; every per-sample group (1/sigma, BatchNorm statistics, relu mask) must then hold
    // an even number of samples so that a tile never straddles two groups
    const bool small8 = p.H == 8 && p.W == 8 && p.D == 1 && !p.upsample && p.N % 2 == 0 && (!p.scale || p.scale_group % 2 == 0) &&
                        (!p.pre_a || p.pre_group % 2 == 0) && (!p.mask_a || p.mask_group % 2 == 0);
    // the LDS-DMA kernel (both bf16 modes) also takes 3x3x3 convs, plane by plane
    const bool glds_ok = g_precision != 0 && (g_tune_window < 0 || g_tune_window >= 3);
    const bool is3d = p.KD == 3 && p.D > 1 && glds_ok && !p.upsample && !p.residual_up;
    if (!(g_precision != 0 && p.w_split && ((p.KD == 1 && p.D == 1) || is3d) && p.KH == 3 && p.KW == 3 && p.Cin % 8 == 0 &&
          (p.W == 16 || p.W % 32 == 0 || small8) &&
          (g_tune_window < 0 ? (M64 / 128) * ((C + 63) / 64) * (p.reserved0 ? 4 : 1) >= 192 : g_tune_window >= 1)))
        return false;
    w->tw_shift = small8 ? 3 : (p.W == 16 ? 4 : 5);
    w->g_shift = small8 ? 1 : 0;
    w->bnw = C % 128 == 0 ? 128 : (C % 96 == 0 ? 96 : (C <= 64 ? 64 : 128));
    // <= 48 output channels in 16-column blocks (v_mfma_f32_16x16x32): three blocks instead of two 32-column ones, a quarter less
    // matrix work (LDS-DMA kernel, 128-pixel tiles)
    if (glds_ok && C <= 48 && C % 16 == 0 && (g_tune_window == 5 || (g_tune_window < 0 && g_m16_auto))) w->bnw = 48;
    // <= 8 output channels (16-column blocks; the discriminators' first convs backwards: Cout = 4): 16-column tile
    if (glds_ok && C <= 8 && C % 4 == 0 && (g_tune_window == 5 || g_tune_window < 0) && g_thin_auto) w->bnw = 16;  // (measured: 1.5 - 1.8 x at 4 and 8 channels, slower at 16)
    // few pixels, many channels (the ConvGRU steps on 8x8 / 16x16 maps: 6144 pixels x 384 channels): 128-column tiles would leave
    // half the CUs without a workgroup - 64-column tiles double the grid (measured 146 -> see profiles/README.md, us per step conv)
    if (g_tune_window < 0 && C % 64 == 0 && (M64 / 128) * ((C + w->bnw - 1) / w->bnw) * (p.reserved0 ? 4 : 1) < 256) w->bnw = 64;
    if (g_tune_window < 0 && (M64 / 128) * ((C + w->bnw - 1) / w->bnw) * (p.reserved0 ? 4 : 1) < 192) return false;
    // 256-pixel tiles (8 x 32 or 16 x 16 pixels of one image), LDS-DMA kernel
    // (not at 128 output channels: 8 accumulator blocks per wave spill at two workgroups per CU).  Measured +2 ... +8 % on the
    // 96- / 64-channel layers of the sampler at T x B maps (gpurun r2o), -10 ... -20 % on launches of a few hundred workgroups:
    // automatic only when the 256-pixel tiles still fill the chip four times over
    const int64_t big_wgs = (M64 / 256) * ((C + w->bnw - 1) / w->bnw) * (p.reserved0 ? 4 : 1);
    // Wave-specialised persistent kernel (conv_win_ws.h; dgmr_conv_tune window = 7 forces it wherever it applies, -1 takes it when
    // every CU gets a run of items): 2-D, plain epilogue in its 16-byte form, at most ONE fused epilogue operand, 96 / 128-column
    // blocks, 32-bit element offsets, and enough taps per item to hide the previous item's epilogue behind (ws_ups)
    w->ws = false;
    w->ws_ups = 1;
    w->pair = false;
    if (glds_ok && g_precision != 3 && p.KD == 1 && p.D == 1 && !p.upsample && p.epi_mode == DGMR_EPI_PLAIN && (p.reserved1 & 4) &&
        !p.addend && !(p.residual && p.mask_src) && (w->bnw == 96 || w->bnw == 128) && M64 * (int64_t)C < (1ll << 32) &&
        !(p.reserved1 & (64 | 128)) && (g_tune_window == 7 || g_tune_window < 0)) {
        const int mode = p.reserved0;                       // 0 plain, 1 phase, 2 pooled
        const int T = mode == 0 ? 9 : 4, D = mode == 0 ? 8 : 3, upsmax = mode == 1 ? 2 : 1;  // (ws_mode<> in conv_win_ws.h)
        const int nchunks = (p.Cin + 31) / 32, nsl = (mode == 2 ? 4 * nchunks : nchunks) * T;
        const int NU = w->bnw == 96 ? 12 : 16;              // epilogue units per loader wave and item
        int ups = 0;
        for (int u = 1; u <= upsmax && !ups; ++u)
            if ((NU + u - 1) / u + D <= nsl - 1) ups = u;
        const int64_t items = (M64 / 128) * ((C + w->bnw - 1) / w->bnw) * (mode == 1 ? 4 : 1);
        const bool eop_ok = (mode != 1 || (!p.residual && !p.mask_src)) && !(mode == 2 && p.residual);
        if (ups && eop_ok && (g_tune_window == 7 || (g_ws_auto && items >= 4 * (int64_t)num_cus()))) {
            w->ws = true;
            w->ws_ups = ups;
        }
    }
    // phase mode, 96 / 128 output columns, bf16 / bf16x3: both column parities of a row parity from ONE staged halo (conv_win_glds.h
    // PAIR; 128-pixel tiles).  DGMR_PHASE_PAIR=0: the four-workgroups-per-tile scheme (A/B)
    w->pair = !w->ws && p.reserved0 == 1 && glds_ok && g_precision != 3 && (g_phase_pair || (g_debug_flags & 256)) && g_tune_window < 0 && C % 96 == 0 && !small8 && p.KD == 1;
    if (w->pair) w->bnw = 96;  // (384 / 768 columns too: the 128-column tile with two accumulator sets spills, 96 sits at 217 registers)
    w->big = !w->ws && !w->pair && !small8 && w->bnw != 128 && p.KD == 1 && p.H % (256 >> w->tw_shift) == 0 && g_precision != 3 &&  // (bf16x6: 102 KB of LDS)
             (g_tune_window == 2 || (g_tune_window < 0 && big_wgs >= 2048));
    w->glds = glds_ok || w->big;
    const int TWv = 1 << w->tw_shift, THv = ((w->big ? 256 : 128) >> w->tw_shift) >> w->g_shift;
    if (p.H % THv != 0) return false;
    w->tiles_w = p.W / TWv;
    w->tiles_hw = w->tiles_w * (p.H / THv);
    w->grid_x = w->g_shift ? p.N >> w->g_shift : p.N * p.D * w->tiles_hw;
    return true;
}

// nearest-2x upsample + 3x3 conv == four 2x2 convs ("phases", one per output-pixel parity) on the low-resolution input with tap sums
// as weights (w_phase): 16 instead of 36 multiply steps per input pixel.  The library's private copy of the arguments then describes
// the LOW-resolution map, reserved0 = 1 tells the LDS-DMA window kernel to walk the phases (conv_win_glds.h).
static bool phase_plan(dgmr_conv_args& p, WinPlan* w) {
    if (!(p.upsample && p.w_phase && g_precision != 0 && p.KD == 1 && p.D == 1 && p.KH == 3 && p.KW == 3 && p.epi_mode == DGMR_EPI_PLAIN &&
          !p.addend && !p.mask_src && p.H % 2 == 0 && p.W % 2 == 0))
        return false;
    dgmr_conv_args q = p;
    q.H = p.H / 2;
    q.W = p.W / 2;
    q.upsample = 0;
    q.reserved0 = 1;
    q.w_split = p.w_phase;
    if (!(window_plan(q, w) && w->glds)) return false;
    p = q;
    return true;
}

// 3x3 conv followed by a 2x2 sum pool (pool2: the data gradient of an upsampling conv) as ONE pass of the window kernel over the four
// pixel-parity planes of the input with the 2x2 tap sums of w_phase; p then describes the pooled (output) map, reserved0 = 2.
// (round 4: also the FORWARD of a DBlock's last conv + AvgPool - a residual at the pooled resolution rides in the epilogue - and 3x3x3
//  convs, plane by plane: w_phase then holds 16 tap sums per depth tap and the pool is the 2 x 2 spatial part of AvgPool3d)
static bool pooled_plan(dgmr_conv_args& p, WinPlan* w) {
    if (!(p.pool2 && !p.upsample && p.w_phase && g_precision != 0 && ((p.KD == 1 && p.D == 1) || (p.KD == 3 && p.D > 1)) && p.KH == 3 &&
          p.KW == 3 && p.epi_mode == DGMR_EPI_PLAIN && !p.addend && !p.residual_up && p.H % 2 == 0 && p.W % 2 == 0))
        return false;
    dgmr_conv_args q = p;
    q.H = p.H / 2;
    q.W = p.W / 2;
    q.reserved0 = 2;
    q.w_split = p.w_phase;
    if (!(window_plan(q, w) && w->glds)) return false;
    p = q;
    return true;
}

extern "C" int dgmr_conv_pool2_supported(const dgmr_conv_args* a) {
    if (!a || a->N <= 0 || a->H <= 0 || a->W <= 0 || a->Cout <= 0 || a->Cin <= 0) return 0;
    dgmr_conv_args p = *a;
    p.reserved0 = 0;
    if (p.scale_group < 1) p.scale_group = 1;
    if (p.pre_group < 1) p.pre_group = 1;
    if (p.mask_group < 1) p.mask_group = 1;
    WinPlan w;
    return pooled_plan(p, &w) ? 1 : 0;
}

static inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
// DGMR_STEM4=0: A/B switch for the four-channel first-conv kernel
static const bool g_stem4 = []() {
    const char* e = getenv("DGMR_STEM4");
    return !(e && e[0] == '0');
}();
// DGMR_CONV1X1=0: A/B switch for the streaming 1x1-conv kernel
static const bool g_conv1x1 = []() {
    const char* e = getenv("DGMR_CONV1X1");
    return !(e && e[0] == '0');
}();
static bool conv1x1_ok(const dgmr_conv_args& p, int64_t M64) {
    return g_conv1x1 && g_tune_variant < 0 && (g_precision == 1 || g_precision == 2) && p.w_split && p.KD == 1 && p.KH == 1 && p.KW == 1 &&
           p.Cin % 8 == 0 && p.Cin >= 32 && p.Cout % 4 == 0 && p.w_cin == p.Cin && p.w_coff == 0 && p.epi_mode == DGMR_EPI_PLAIN && !p.upsample &&
           !p.pool2 && !p.pre_a && !p.addend && !p.residual && !p.mask_src && !p.stats_out && (p.reserved1 & 4) && al16(p.x) &&
           ((int64_t)p.D * p.H * p.W) % 256 == 0 && M64 / 256 >= 512;  // (enough workgroups to fill the chip twice; tiles inside one sample)
}
static bool stem4_ok(const dgmr_conv_args& p) {
    return g_stem4 && g_tune_variant < 0 && p.Cin == 4 && p.w_cin == 4 && p.w_coff == 0 && p.KH == 3 && p.KW == 3 &&
           ((p.KD == 1 && p.D == 1) || (p.KD == 3 && p.D >= 1)) && p.Cout % STEM_BN == 0 && p.H % STEM_TH == 0 && p.W % STEM_TW == 0 &&
           p.epi_mode == DGMR_EPI_PLAIN && !p.upsample && !p.pool2 && !p.pre_a && !p.addend && !p.residual && !p.mask_src && !p.stats_out &&
           al16(p.x) && al16(p.y) && al16(p.bias) && (int64_t)p.N * p.D * (p.H / STEM_TH) * (p.W / STEM_TW) < (1ll << 31);
}
static void conv_args_defaults(dgmr_conv_args& p) {
    p.reserved0 = 0;
    p.reserved1 = g_debug_flags & (3 | 64 | 128);
    // bit 2: the window kernels may use their 16-byte epilogue (conv_win_glds.h) - four consecutive output channels per lane
    if (!(g_debug_flags & 8) && p.Cout % 4 == 0 && al16(p.y) && al16(p.bias) && al16(p.addend) && al16(p.residual) && al16(p.mask_src) &&
        al16(p.mask_a) && al16(p.mask_b) && al16(p.gru_h) && al16(p.gru_pu) && al16(p.pre_out) && al16(p.bias2) && al16(p.addend2) && al16(p.y2))
        p.reserved1 |= 4;
    if (p.scale_group < 1) p.scale_group = 1;
    if (p.pre_group < 1) p.pre_group = 1;
    if (p.mask_group < 1) p.mask_group = 1;
    if (p.w_cin == 0) {
        p.w_cin = p.Cin;
        p.w_coff = 0;
    }
}

// DGMR_EPI_GRU_GATES2 lives in the LDS-DMA window kernel's 16-byte epilogue only
static bool gates2_ok(const dgmr_conv_args& p) {
    WinPlan w;
    return p.epi_mode == DGMR_EPI_GRU_GATES2 && p.gru_split > 0 && p.gru_split % 4 == 0 && p.Cout == 2 * p.gru_split && p.gru_h && p.y2 &&
           !p.upsample && !p.pool2 && !p.residual && !p.mask_src && !p.stats_out && (p.reserved1 & 4) && window_plan(p, &w) && w.glds;
}

extern "C" int dgmr_conv_gates2_supported(const dgmr_conv_args* a) {
    if (!a || a->N <= 0 || a->H <= 0 || a->W <= 0 || a->Cout <= 0 || a->Cin <= 0 || !a->y) return 0;
    dgmr_conv_args p = *a;
    conv_args_defaults(p);
    return gates2_ok(p) ? 1 : 0;
}

extern "C" int dgmr_conv_stats_rows(const dgmr_conv_args* a) {
    if (!a || a->N <= 0 || a->H <= 0 || a->W <= 0 || a->Cout <= 0 || a->epi_mode != DGMR_EPI_PLAIN) return 0;
    dgmr_conv_args p = *a;
    conv_args_defaults(p);
    WinPlan w;
    if (phase_plan(p, &w)) return 4 * w.grid_x;
    if (p.pool2) return pooled_plan(p, &w) ? w.grid_x : 0;
    return (window_plan(p, &w) && w.glds) ? w.grid_x : 0;
}

extern "C" int dgmr_conv_fwd(const dgmr_conv_args* a, void* stream) {
    DGMR_CHECK_ARG(a && a->x && a->w && a->y, "dgmr_conv_fwd: null pointer");
    DGMR_CHECK_ARG(a->Cin % 4 == 0 && a->Cin > 0, "dgmr_conv_fwd: Cin=%d must be a positive multiple of 4", a->Cin);
    DGMR_CHECK_ARG(a->Cout > 0, "dgmr_conv_fwd: Cout=%d", a->Cout);
    DGMR_CHECK_ARG((a->KD == 1 || a->KD == 3) && (a->KH == 1 || a->KH == 3) && (a->KW == 1 || a->KW == 3),
                   "dgmr_conv_fwd: kernel %dx%dx%d unsupported", a->KD, a->KH, a->KW);
    DGMR_CHECK_ARG(!a->upsample || (a->H % 2 == 0 && a->W % 2 == 0), "dgmr_conv_fwd: upsample needs even H,W");
    DGMR_CHECK_ARG(!a->residual_up || (a->D == 1 && a->H % 2 == 0 && a->W % 2 == 0), "dgmr_conv_fwd: residual_up needs a 2-D map with even H,W");
    DGMR_CHECK_ARG((a->pre_a == nullptr) == (a->pre_b == nullptr), "dgmr_conv_fwd: pre_a/pre_b must come together");
    const int64_t M64 = (int64_t)a->N * a->D * a->H * a->W;
    DGMR_CHECK_ARG(M64 > 0 && M64 < (1ll << 31), "dgmr_conv_fwd: M=%lld out of range", (long long)M64);
    // the kernels address the input with 32-bit element offsets
    DGMR_CHECK_ARG((M64 >> (a->upsample ? 2 : 0)) * a->Cin < (1ll << 32), "dgmr_conv_fwd: input of %lld x %d elements exceeds 2^32",
                   (long long)M64, a->Cin);
    DGMR_CHECK_ARG(a->epi_mode == DGMR_EPI_PLAIN || (a->gru_h && (a->epi_mode != DGMR_EPI_GRU_BLEND || a->gru_pu)),
                   "dgmr_conv_fwd: epi_mode %d needs gru_h / gru_pu", a->epi_mode);
    DGMR_CHECK_ARG(a->w_cin == 0 || (a->w_cin >= a->w_coff + a->Cin && a->w_coff >= 0 && a->w_coff % 4 == 0 && a->w_cin % 4 == 0),
                   "dgmr_conv_fwd: weight slice [%d, %d) of %d channels is invalid", a->w_coff, a->w_coff + a->Cin, a->w_cin);
    dgmr_conv_args p = *a;
    conv_args_defaults(p);
    DGMR_CHECK_ARG(p.epi_mode != DGMR_EPI_GRU_GATES2 || gates2_ok(p),
                   "dgmr_conv_fwd: DGMR_EPI_GRU_GATES2 needs a conv the LDS-DMA window kernel takes, Cout == 2 * gru_split, gru_split %% 4 == 0, "
                   "gru_h / y2 and 16-byte aligned tensors (ask dgmr_conv_gates2_supported)");
    const int M = (int)M64, Ktot = a->KD * a->KH * a->KW * a->Cin;
    hipStream_t s = (hipStream_t)stream;
    const int C = a->Cout;
    // Tile choice: N tile from Cout (32x32 MFMA granularity), M tile shrinks when the grid would not fill 256 CUs.
    int bn;
    if (C <= 32) bn = 32;
    else if (C <= 64) bn = 64;
    else if (C % 128 == 0) bn = 128;
    else if (C % 96 == 0) bn = 96;
    else bn = 128;
    const int64_t wgs128 = ((M64 + 127) / 128) * ((C + bn - 1) / bn);
    const double flops = 2.0 * (double)M64 * (double)Ktot * (double)C;  // algorithmic: 2*MACs of the dense conv
    int variant;
    if (bn == 128) variant = wgs128 >= 256 ? V_F128x128 : V_F64x64;
    else if (bn == 96) variant = V_F128x96;
    else if (bn == 64) variant = wgs128 >= 256 ? V_F128x64 : V_F64x64;
    else variant = V_F128x32;
    // bf16x6: three planes per operand - the 128-row tiles with 64 / 96 columns hold 92 / 107 KB of LDS (one workgroup per CU); the
    // 64 x 64 tile keeps two resident, which is what the short-K layers that land here need (the discriminators' Cin = 4 first convs)
    if (g_precision == 3 && (variant == V_F128x64 || variant == V_F128x96)) variant = V_F64x64;
    if (g_tune_variant >= V_F128x128 && g_tune_variant <= V_F128x32) variant = g_tune_variant;
    // 3x3 / 3x3x3 convs of four-channel maps (the first convs behind a space-to-depth): exact fp32 on v_mfma_f32_16x16x4_f32, whatever
    // the arithmetic mode (conv_stem4.h)
    if (stem4_ok(p)) {
        const uint32_t detail = 4u | ((p.KD == 3 ? 1u : 0u) << 11);
        ProfScope ps(V_F64x64, flops, s, g_precision == 1 ? 1.0 / 3.0 : (g_precision == 3 ? 1.0 / 6.0 : 1.0), detail);
        const int tiles_w = p.W / STEM_TW, tiles_hw = tiles_w * (p.H / STEM_TH);
        const dim3 grid((unsigned)(p.N * p.D * tiles_hw), (unsigned)(p.Cout / STEM_BN));
        if (p.KD == 3) hipLaunchKernelGGL(conv_stem4_kernel<3>, grid, dim3(256), 0, s, p, tiles_w, tiles_hw);
        else hipLaunchKernelGGL(conv_stem4_kernel<1>, grid, dim3(256), 0, s, p, tiles_w, tiles_hw);
        DGMR_CHECK_LAUNCH();
        return 0;
    }
    // 1x1 convs of the big maps, bf16 / bf16x3: streaming GEMM (conv1x1.h; needs pre-split weights like the window kernels)
    if (conv1x1_ok(p, M64)) {
        const uint32_t detail = 5u | ((p.KD == 1 && p.D > 1 ? 1u : 0u) << 11);
        ProfScope ps(variant, flops, s, 1.0, detail);
        const int rows = p.D * p.H * p.W;
        if ((g_precision == 1 ? dgmr_tu::launch_conv1x1_ns3(p, M, rows, s) : dgmr_tu::launch_conv1x1_ns1(p, M, rows, s)) != 0) return -1;
        DGMR_CHECK_LAUNCH();
        return 0;
    }
    // 3x3 convs of the big feature maps in the bf16 modes: LDS-window kernel (needs pre-split weights)
    WinPlan wp;
    const bool phases = phase_plan(p, &wp);  // (rewrites p to the low-resolution map when it applies)
    DGMR_CHECK_ARG(!p.pool2 || pooled_plan(p, &wp), "dgmr_conv_fwd: pool2 needs a conv the window kernel takes (dgmr_conv_pool2_supported)");
    DGMR_CHECK_ARG(!p.stats_out || (window_plan(p, &wp) && wp.glds && p.epi_mode == DGMR_EPI_PLAIN),
                   "dgmr_conv_fwd: stats_out given but the dispatched kernel has no fused statistics (ask dgmr_conv_stats_rows first)");
    if (window_plan(p, &wp)) {
        const int v = wp.bnw == 128 ? V_WIN128 : (wp.bnw == 96 ? V_WIN96 : V_WIN64);
        const int64_t wgs = (int64_t)wp.grid_x * (phases ? 4 : 1) * ((p.Cout + wp.bnw - 1) / wp.bnw);
        const uint32_t detail = 1u | ((wp.bnw == 48 ? 0u : (wp.bnw == 64 ? 1u : (wp.bnw == 96 ? 2u : (wp.bnw == 128 ? 3u : 4u)))) << 4) | ((wp.big ? 1u : 0u) << 8) |
                                ((uint32_t)p.reserved0 << 9) | ((p.KD == 3 ? 1u : 0u) << 11) | ((wgs < 1024 ? 1u : 0u) << 12) |
                                ((wp.glds ? 0u : 1u) << 13) | ((p.epi_mode != DGMR_EPI_PLAIN ? 1u : 0u) << 14);
        ProfScope ps(v, flops, s, (phases || p.reserved0 == 2) ? 16.0 / 36.0 : 1.0, detail | ((wp.ws ? 1u : 0u) << 15));
        if (wp.ws) {
            const int64_t items = (int64_t)wp.grid_x * (phases ? 4 : 1) * ((p.Cout + wp.bnw - 1) / wp.bnw);
            const int grid = (int)std::min<int64_t>(items, num_cus());
            int rc;
            if (g_precision == 1) rc = p.reserved0 == 0 ? dgmr_tu::launch_window_ws_ns3_m0(p, wp, grid, s) : (p.reserved0 == 1 ? dgmr_tu::launch_window_ws_ns3_m1(p, wp, grid, s) : dgmr_tu::launch_window_ws_ns3_m2(p, wp, grid, s));
            else rc = p.reserved0 == 0 ? dgmr_tu::launch_window_ws_ns1_m0(p, wp, grid, s) : (p.reserved0 == 1 ? dgmr_tu::launch_window_ws_ns1_m1(p, wp, grid, s) : dgmr_tu::launch_window_ws_ns1_m2(p, wp, grid, s));
            if (rc != 0) return -1;
            DGMR_CHECK_LAUNCH();
            return 0;
        }
        if (DGMR_BY_NS(launch_window, p, wp, phases, g_tune_window, s) != 0) return -1;
        DGMR_CHECK_LAUNCH();
        return 0;
    }
    // Split-K when the output grid cannot fill the 256 CUs (ConvGRU steps, latent stack, deep discriminator layers): the k
    // loop is the only parallelism left.  Needs a workspace; chosen so that grid * ksplit ~ 512 workgroups, >= 4 k-tiles each.
    p.ksplit = 1;
    if (p.splitk_ws && p.splitk_ws_bytes > 0) {
        const int bm = variant == V_F64x64 ? 64 : 128;
        const int bnv = variant == V_F128x128 ? 128 : (variant == V_F128x96 ? 96 : (variant == V_F128x32 ? 32 : 64));
        const int64_t wgs = ((M64 + bm - 1) / bm) * ((C + bnv - 1) / bnv);
        const int bk = 32;
        const int nk = (Ktot + bk - 1) / bk;
        if (wgs < 192 && nk >= 8) {
            int64_t S = 512 / wgs;
            if (S > nk / 4) S = nk / 4;
            const int64_t cap = p.splitk_ws_bytes / ((int64_t)M64 * C * 4);
            if (S > cap) S = cap;
            if (S > 64) S = 64;
            if (S > 1) p.ksplit = (int)S;
        }
        if (g_tune_ksplit >= 1) {
            const int64_t cap = p.splitk_ws_bytes / ((int64_t)M64 * C * 4);
            p.ksplit = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(g_tune_ksplit, cap), nk));
        }
    }
    {
        const int bm_v = variant == V_F64x64 ? 64 : 128;
        const int bn_v = variant == V_F128x128 ? 128 : (variant == V_F128x96 ? 96 : (variant == V_F128x32 ? 32 : 64));
        const int64_t wgs_v = ((M64 + bm_v - 1) / bm_v) * ((C + bn_v - 1) / bn_v) * (p.ksplit > 1 ? p.ksplit : 1);
        const uint32_t detail = 2u | ((uint32_t)variant << 4) | ((p.ksplit > 1 ? 1u : 0u) << 8) | ((Ktot == p.Cin ? 1u : 0u) << 9) |
                                ((p.KD == 3 ? 1u : 0u) << 11) | ((wgs_v < 1024 ? 1u : 0u) << 12) | ((p.epi_mode != DGMR_EPI_PLAIN ? 1u : 0u) << 14);
        ProfScope ps(variant, flops, s, 1.0, detail);
        switch (variant) {
            case V_F128x128: launch_conv<V_F128x128, 128, 128, 2, 2>(p, M, Ktot, s); break;
            case V_F64x64: launch_conv<V_F64x64, 64, 64, 2, 2>(p, M, Ktot, s); break;
            case V_F128x96: launch_conv<V_F128x96, 128, 96, 4, 1>(p, M, Ktot, s); break;
            case V_F128x64: launch_conv<V_F128x64, 128, 64, 4, 1>(p, M, Ktot, s); break;
            default: launch_conv<V_F128x32, 128, 32, 4, 1>(p, M, Ktot, s); break;
        }
    }
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_conv_flip_weights(const float* w, float* w_t, int Cout, int Cin, int KD, int KH, int KW, int w_cin, int w_coff,
                                      void* stream) {
    DGMR_CHECK_ARG(w && w_t, "dgmr_conv_flip_weights: null pointer");
    if (w_cin == 0) {
        w_cin = Cin;
        w_coff = 0;
    }
    DGMR_CHECK_ARG(w_coff >= 0 && w_coff + Cin <= w_cin, "dgmr_conv_flip_weights: bad slice");
    const int taps = KD * KH * KW;
    const size_t total = (size_t)Cout * Cin * taps;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(flip_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, w_t, Cout, Cin, taps, w_cin, w_coff);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_conv_wgrad_nsplit(int M, int Cout, int K, int groups) {
    if (groups < 1) groups = 1;
    const int bi = Cout <= 32 ? 32 : (Cout <= 64 ? 64 : 128);
    const int64_t tiles = (int64_t)((K + 127) / 128) * ((Cout + bi - 1) / bi);
    int64_t ns = 1024 / tiles;
    const int64_t cap = M / 256;  // at least 256 pixels per slab
    if (ns > cap) ns = cap;
    if (ns > 1024) ns = 1024;
    // a slab never straddles two groups: round to a multiple of `groups` (>= 1 slab per group)
    int64_t per = ns / groups;
    if (per < 1) per = 1;
    return (int)(per * groups);
}

// ---- deterministic bias gradient (dgmr_wgrad_args.bias_partial) ----
// rows[b][c] = sum of dy[r][c] over the b-th of `nrows` contiguous row ranges: fixed order inside a thread (stride RL), then over the
// RL row lanes in lane order.  For the weight-gradient kernels whose workgroups meet in a channel (wgrad_win.h, the im2col kernels).
__global__ __launch_bounds__(256) void colsum_rows_kernel(const float* __restrict__ dy, float* __restrict__ rows, int64_t M, int Cout, int nrows) {
    __shared__ f32x4 part[256];
    const int Cq = Cout / 4;
    const int64_t per = (M + nrows - 1) / nrows;
    const int64_t r0 = blockIdx.x * per, r1 = min(M, r0 + per);
    for (int cb = 0; cb < Cq; cb += 256) {
        const int Wd = min(256, Cq - cb), RL = 256 / Wd;
        const int q = cb + (int)threadIdx.x % Wd, rl = threadIdx.x / Wd;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (rl < RL)
            for (int64_t r = r0 + rl; r < r1; r += RL) a += *reinterpret_cast<const f32x4*>(dy + (size_t)r * Cout + q * 4);
        part[threadIdx.x] = a;
        __syncthreads();
        if ((int)threadIdx.x < Wd) {
            f32x4 t = part[threadIdx.x];
            for (int k = 1; k < RL; ++k) t += part[threadIdx.x + k * Wd];
            *reinterpret_cast<f32x4*>(rows + (size_t)blockIdx.x * Cout + q * 4) = t;
        }
        __syncthreads();
    }
}
// bias_grad[c] += sum of rows[r][c] in a fixed order: four lanes per channel take rows r = lane (mod 4) in increasing order, then the
// four partial sums are added in lane order
__global__ __launch_bounds__(256) void bias_rows_finish_kernel(const float* __restrict__ rows, int nrows, int Cout, float* __restrict__ bias_grad) {
    __shared__ float part[4][64];
    const int col = threadIdx.x & 63, l = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + col;
    float a = 0.f;
    if (c < Cout) {
        int r = l;
        for (; r + 28 < nrows; r += 32) {  // eight independent loads in flight, added in row order
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = rows[(size_t)(r + 4 * k) * Cout + c];
#pragma unroll
            for (int k = 0; k < 8; ++k) a += v[k];
        }
        for (; r < nrows; r += 4) a += rows[(size_t)r * Cout + c];
    }
    part[l][col] = a;
    __syncthreads();
    if (l == 0 && c < Cout) bias_grad[c] += (part[0][col] + part[1][col]) + (part[2][col] + part[3][col]);
}
static inline int colsum_rows_for(int64_t M) { return (int)std::max<int64_t>(1, std::min<int64_t>(1024, M / 512)); }

// the LDS-window weight gradient (wgrad_win.h) applies to 3x3 convs on 2-D maps made of whole rows of 32 (or 16) pixels, bf16 modes
// which of the two window kernels: the wave-specialised one (wgrad_ws.h; dgmr_conv_tune wgrad_window 2: three matrix waves, 3 = automatic:
// four matrix waves in bf16 / bf16x3, three in bf16x6) or the one-role kernel of round 2 (wgrad_win.h; 1)
static bool wgrad_ws() { return g_tune_wgrad_window != 1; }
// DGMR_WGRAD_PSPLIT=0: the 48-channel layers' weight gradients on the 64-column tile as in rounds 3 - 5 (A/B switch for the pixel-split
// 48-column tile of wgrad_ws.h; dgmr_conv_tune wgrad_window = 5 does the same inside a process, without the phase launches)
static const bool g_wgrad_psplit = []() {
    const char* e = getenv("DGMR_WGRAD_PSPLIT");
    return !(e && e[0] == '0');
}();
// (3 x 3 x 3 convs, without upsampling: the wave-specialised kernel only, one launch per depth tap)
static bool wgrad_uses_window(const dgmr_wgrad_args* a) {
    const bool k2d = a->KD == 1 && a->D == 1, k3d = a->KD == 3 && a->D >= 1 && !a->upsample && wgrad_ws();
    return g_precision != 0 && g_tune_wgrad_window != 0 && (k2d || k3d) && a->KH == 3 && a->KW == 3 &&
           ((a->W % 32 == 0 && a->H % 2 == 0) || (a->W == 16 && a->H % 4 == 0));
}
// tiles of 64 pixels: 2 x 32, or 4 x 16 on 16-pixel-wide maps
static int wgrad_window_tw_shift(const dgmr_wgrad_args* a) { return a->W % 32 == 0 ? 5 : 4; }
// Upsampling convs: the weight gradient by output-pixel parity on the LOW-resolution map (wgrad_ws.h PHASE: 16 instead of 36 multiply
// steps per input pixel; measured 57.9 -> 44.5 ms for the step's four launches, 877.7 -> 864.7 ms per step, profiles/r04_step6_*).
// DGMR_WGRAD_PHASES=0 switches it off (A/B).  Needs the four-matrix-wave kernel (bf16 / bf16x3) and a low-resolution map the window
// kernel tiles.
static const bool g_wgrad_phases = []() {
    const char* e = getenv("DGMR_WGRAD_PHASES");
    return !(e && e[0] == '0');
}();
static bool wgrad_by_phases(const dgmr_wgrad_args* a) {
    // (dgmr_conv_tune wgrad_window: -1 = the library's choice = phases unless switched off, 4 = phases, 1 / 2 / 3 = the named 3 x 3 kernel)
    if (!((g_tune_wgrad_window == 4 || (g_tune_wgrad_window < 0 && g_wgrad_phases)) && a->upsample && a->KD == 1 && a->D == 1 && a->KH == 3 && a->KW == 3 &&
          (g_precision == 1 || g_precision == 2) && a->H % 2 == 0 && a->W % 2 == 0))
        return false;
    const int h = a->H / 2, w = a->W / 2;
    return (w % 32 == 0 && h % 2 == 0) || (w == 16 && h % 4 == 0);
}

extern "C" int dgmr_conv_wgrad_plan(dgmr_wgrad_args* a) {
    DGMR_CHECK_ARG(a && a->N > 0 && a->Cin > 0 && a->Cout > 0, "dgmr_conv_wgrad_plan: bad args");
    const int groups = a->groups < 1 ? 1 : a->groups;
    DGMR_CHECK_ARG(a->N % groups == 0, "dgmr_conv_wgrad_plan: N=%d not divisible by groups=%d", a->N, groups);
    const int64_t M = (int64_t)a->N * a->D * a->H * a->W;
    if (!wgrad_uses_window(a)) {
        a->nsplit = dgmr_conv_wgrad_nsplit((int)M, a->Cout, a->KD * a->KH * a->KW * a->Cin, groups);
        a->bias_rows = std::max(a->nsplit, colsum_rows_for(M));
        return 0;
    }
    // workgroups = 32-channel input chunks x output tiles x slabs.  Two are resident per CU: one full round (<= 512 workgroups)
    // measures better than 1.5 rounds (tail) and than many small slabs (partial-sum traffic); >= 2 tiles of 64 pixels per slab
    const int per_slab = ((a->Cin + 31) / 32) * (a->Cout % 96 == 0 ? a->Cout / 96 : (a->Cout + 63) / 64);
    const bool phases = wgrad_by_phases(a);  // tiles of the low-resolution map; every slab is four rows of `partial` (one per parity)
    const int64_t tiles_per_group = (int64_t)(a->N / groups) * a->D * ((int64_t)a->H * a->W / (phases ? 256 : 64));
    int64_t per = 512 / ((int64_t)per_slab * groups);  // slabs per group
    per = std::min<int64_t>(per, tiles_per_group / 2);
    per = std::max<int64_t>(per, 1);
    // the wave-specialised kernel (wgrad_ws.h) has ONE workgroup per CU: one round of <= 256.  (Several rounds of smaller slabs fill
    // the chip better when the launch runs alone - 3 chunks x 18 groups: 216 of 256 CUs with 4 slabs per group, 756 of 768 with 14,
    // +8 % in isolation - but in the training step the weight gradients share the chip with the other streams' kernels and the
    // larger partial sums cost what the fill gains: 999 vs 1000 - 1004 ms per step)
    // (a weight-gradient workgroup - 84 KB of LDS, 8 waves x 222 registers - leaves no room on its CU for a 256-pixel-tile conv workgroup
    //  of the main stream: beside each other the two streams share the chip by whole CUs.  With 108 call groups x 3 input chunks a launch
    //  is 324 workgroups whatever this cap says - round 5 tried 128 ... 512 here: no difference, 880.5 ... 885.2 ms on one box)
    if (wgrad_ws()) per = std::max<int64_t>(1, std::min<int64_t>(256 / ((int64_t)per_slab * groups), tiles_per_group / 2));
    a->nsplit = (int)std::min<int64_t>(per * groups, 4096) * (phases ? 4 : 1);
    a->bias_rows = std::max(a->nsplit, colsum_rows_for(M));  // rows of dgmr_wgrad_args.bias_partial (deterministic bias gradient)
    return 0;
}

extern "C" int dgmr_conv_wgrad(const dgmr_wgrad_args* a, void* stream) {
    DGMR_CHECK_ARG(a && a->x && a->dy && a->partial, "dgmr_conv_wgrad: null pointer");
    DGMR_CHECK_ARG(a->Cin % 4 == 0 && a->Cout % 4 == 0, "dgmr_conv_wgrad: Cin=%d Cout=%d must be multiples of 4", a->Cin,
                   a->Cout);
    DGMR_CHECK_ARG(a->nsplit >= 1, "dgmr_conv_wgrad: nsplit=%d", a->nsplit);
    const int groups = a->groups < 1 ? 1 : a->groups;
    DGMR_CHECK_ARG(a->nsplit % groups == 0 && a->N % groups == 0, "dgmr_conv_wgrad: nsplit=%d / N=%d not divisible by groups=%d",
                   a->nsplit, a->N, groups);
    const int64_t M64 = (int64_t)a->N * a->D * a->H * a->W;
    DGMR_CHECK_ARG(M64 > 0 && M64 < (1ll << 31), "dgmr_conv_wgrad: M out of range");
    // 32-bit element offsets into x (as in dgmr_conv_fwd)
    DGMR_CHECK_ARG((M64 >> (a->upsample ? 2 : 0)) * a->Cin < (1ll << 32), "dgmr_conv_wgrad: input of %lld x %d elements exceeds 2^32",
                   (long long)M64, a->Cin);
    dgmr_wgrad_args p = *a;
    if (p.pre_group < 1) p.pre_group = 1;
    p.bias_stride = 0;
    // deterministic bias gradient: rows of bias_partial instead of float atomics on bias_grad (dgmr_hip.h, ABI 11)
    const bool bias_rows_mode = a->bias_grad && a->bias_partial;
    // (deterministic mode without bias_partial: the float-atomic path - the caller's choice; the package always passes the rows)
    DGMR_CHECK_ARG(!bias_rows_mode || a->bias_rows >= std::max(a->nsplit, colsum_rows_for(M64)),
                   "dgmr_conv_wgrad: bias_rows=%d (dgmr_conv_wgrad_plan fills it in)", a->bias_rows);
    hipStream_t s_ = (hipStream_t)stream;
    auto bias_rows_begin = [&](int rows) { (void)hipMemsetAsync(a->bias_partial, 0, sizeof(float) * (size_t)rows * a->Cout, s_); };
    auto bias_rows_finish = [&](int rows) {
        hipLaunchKernelGGL(bias_rows_finish_kernel, dim3((a->Cout + 63) / 64), dim3(256), 0, s_, a->bias_partial, rows, a->Cout, a->bias_grad);
    };
    auto bias_by_colsum = [&]() {  // kernels whose workgroups meet in a channel: a fixed-order column-sum pass over dy instead
        const int rows = colsum_rows_for(M64);
        hipLaunchKernelGGL(colsum_rows_kernel, dim3(rows), dim3(256), 0, s_, a->dy, a->bias_partial, M64, a->Cout, rows);
        bias_rows_finish(rows);
    };
    const int M = (int)M64, Ktot = a->KD * a->KH * a->KW * a->Cin;
    const int spg = a->nsplit / groups, rows_per_group = M / groups;
    int rows = (rows_per_group + spg - 1) / spg;
    rows = (rows + 31) / 32 * 32;
    hipStream_t s = (hipStream_t)stream;
    const int kt = (Ktot + 127) / 128;
    const int wclass = a->Cout <= 32 ? V_W32 : (a->Cout <= 64 ? V_W64 : V_W128);
    const uint32_t wdetail = 3u | ((wgrad_uses_window(a) ? (wgrad_ws() ? 2u : 1u) : 0u) << 4) | ((uint32_t)(wclass - V_W128) << 6) |
                             ((a->KD == 3 ? 1u : 0u) << 8) | ((a->upsample ? 1u : 0u) << 9) | ((Ktot == a->Cin ? 1u : 0u) << 10);
    ProfScope ps(wclass, 2.0 * (double)M64 * (double)Ktot * (double)a->Cout, s, 1.0, wdetail);
    // 3x3 convs on maps with whole rows of 32 pixels, bf16 modes: LDS-window weight gradient (wgrad_win.h)
    if (wgrad_uses_window(a) && wgrad_by_phases(a) && a->nsplit % (4 * groups) == 0) {
        dgmr_wgrad_args q = p;
        q.H = a->H / 2, q.W = a->W / 2, q.upsample = 0;
        const int spg4 = a->nsplit / 4 / groups;
        const int tw_shift = wgrad_window_tw_shift(&q);
        const int tiles_w = q.W >> tw_shift, tiles_hw = (q.H / (64 >> tw_shift)) * tiles_w;
        const int tiles_per_group = (a->N / groups) * tiles_hw;
        const int tiles_per_split = (tiles_per_group + spg4 - 1) / spg4;
        const bool b96 = a->Cout % 96 == 0;
        const dim3 grid((a->Cin + 31) / 32, b96 ? a->Cout / 96 : (a->Cout + 63) / 64, a->nsplit / 4);
        if (bias_rows_mode) bias_rows_begin(a->nsplit);
        for (int ph = 0; ph < 4; ++ph) {  // (every parity holds a quarter of dY's pixels: the bias gradient adds up over the four launches)
            if (bias_rows_mode) q.bias_grad = a->bias_partial + (size_t)ph * (a->nsplit / 4) * a->Cout, q.bias_stride = a->Cout;
            DGMR_BY_NS(launch_wgrad_window, q, grid, tw_shift, tiles_w, tiles_hw, tiles_per_split, spg4, tiles_per_group,
                       1 | ((g_debug_flags & 16) >> 3) | 4 | 8 | (ph << 16), s);
            DGMR_CHECK_LAUNCH();
        }
        if (bias_rows_mode) bias_rows_finish(a->nsplit);
        return 0;
    }
    if (wgrad_uses_window(a)) {
        const int tw_shift = wgrad_window_tw_shift(a);
        const int tiles_w = a->W >> tw_shift, tiles_hw = (a->H / (64 >> tw_shift)) * tiles_w;
        const int tiles_per_group = (a->N / groups) * a->D * tiles_hw;
        const int tiles_per_split = (tiles_per_group + spg - 1) / spg;
        const bool b96 = a->Cout % 96 == 0;
        const dim3 grid((a->Cin + 31) / 32, b96 ? a->Cout / 96 : (a->Cout + 63) / 64, a->nsplit);
        const bool rows_in_kernel = bias_rows_mode && wgrad_ws();  // (the wave-specialised kernel: one writer per slab and channel)
        if (rows_in_kernel) bias_rows_begin(a->nsplit);
        for (int kd = 0; kd < a->KD; ++kd) {  // (3-D: one launch per depth tap; the bias gradient rides in the centre one, which skips no plane)
            dgmr_wgrad_args q = p;
            if (a->KD == 3 && kd != 1) q.bias_grad = nullptr;
            else if (rows_in_kernel) q.bias_grad = a->bias_partial, q.bias_stride = a->Cout;
            else if (bias_rows_mode) q.bias_grad = nullptr;
            DGMR_BY_NS(launch_wgrad_window, q, grid, tw_shift, tiles_w, tiles_hw, tiles_per_split, spg, tiles_per_group,
                       wgrad_ws() ? (1 | ((g_debug_flags & 16) >> 3) | (g_tune_wgrad_window != 2 && g_precision != 3 ? 4 : 0) | ((g_tune_wgrad_window == 5 || !g_wgrad_psplit) ? 32 : 0) | (kd << 8)) : 0, s);
            DGMR_CHECK_LAUNCH();
        }
        if (rows_in_kernel) bias_rows_finish(a->nsplit);
        else if (bias_rows_mode) bias_by_colsum();
        return 0;
    }
    if (bias_rows_mode && g_precision != 0) {
        // conv_wgrad_bf16_kernel: ONE thread per (slab, channel) adds the slab's column sum - into a row of its own per slab (blockIdx.z)
        bias_rows_begin(a->nsplit);
        p.bias_grad = a->bias_partial, p.bias_stride = a->Cout;
    } else if (bias_rows_mode) {
        p.bias_grad = nullptr;  // conv_wgrad_kernel (exact f32): several threads of a workgroup meet in a channel - column-sum pass instead
    }
    if (g_precision != 0) {
        // output-channel tile: 32 / 64 / 96 (96, 192, 288 channels: no idle MFMA rows) / 128
        const int bi = a->Cout <= 32 ? 32 : (a->Cout <= 64 ? 64 : ((a->Cout % 96 == 0 && a->Cout % 128 != 0) ? 96 : 128));
        const dim3 grid(kt, (a->Cout + bi - 1) / bi, a->nsplit);
        DGMR_BY_NS(launch_wgrad_gemm, p, bi, grid, M, Ktot, rows, spg, rows_per_group, s);
        if (bias_rows_mode) bias_rows_finish(a->nsplit);
    } else if (a->Cout <= 32) {
        hipLaunchKernelGGL((conv_wgrad_kernel<32, 128, 1, 4>), dim3(kt, (a->Cout + 31) / 32, a->nsplit), dim3(256), 0, s, p, M,
                           Ktot, rows, spg, rows_per_group);
    } else if (a->Cout <= 64) {
        hipLaunchKernelGGL((conv_wgrad_kernel<64, 128, 2, 2>), dim3(kt, (a->Cout + 63) / 64, a->nsplit), dim3(256), 0, s, p, M,
                           Ktot, rows, spg, rows_per_group);
    } else {
        hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, 2, 2>), dim3(kt, (a->Cout + 127) / 128, a->nsplit), dim3(256), 0, s, p,
                           M, Ktot, rows, spg, rows_per_group);
    }
    if (bias_rows_mode && g_precision == 0) bias_by_colsum();
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_wgrad_dot_floats(int groups) { return g_deterministic ? std::max(groups, 1) * (1 + 1024) : std::max(groups, 1); }

extern "C" int dgmr_wgrad_reduce(const float* partial, int nsplit, int groups, int64_t numel, const float* w, const float* scale,
                                 float* g, float* dot, void* stream) {
    DGMR_CHECK_ARG(partial && g && numel > 0 && nsplit >= 1, "dgmr_wgrad_reduce: bad args");
    if (groups < 1) groups = 1;
    DGMR_CHECK_ARG(groups <= WG_MAX_GROUPS && nsplit % groups == 0, "dgmr_wgrad_reduce: groups=%d (max %d) must divide nsplit=%d",
                   groups, WG_MAX_GROUPS, nsplit);
    DGMR_CHECK_ARG(!dot || w, "dgmr_wgrad_reduce: dot needs w");
    const int blocks = (int)std::min<int64_t>((numel + 255) / 256, 1024);
    if (groups <= 32)
        hipLaunchKernelGGL(wgrad_reduce_kernel<32>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, partial, nsplit, groups,
                           (size_t)numel, w, scale, g, dot, 1, 1, 1, 0, (dot && g_deterministic) ? 1 : 0);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel<WG_MAX_GROUPS>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, partial, nsplit, groups,
                           (size_t)numel, w, scale, g, dot, 1, 1, 1, 0, (dot && g_deterministic) ? 1 : 0);
    if (dot && g_deterministic)
        hipLaunchKernelGGL(wgrad_dot_finish_kernel, dim3(1), dim3(8 * WG_MAX_GROUPS), 0, (hipStream_t)stream, dot, groups, blocks);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_wgrad_reduce_slice(const float* partial, int nsplit, int groups, int Cout, int taps, int cin_slice, int cin_total,
                                       int coff, const float* w, const float* scale, float* g, float* dot, void* stream) {
    DGMR_CHECK_ARG(partial && g && Cout > 0 && taps > 0 && cin_slice > 0 && nsplit >= 1, "dgmr_wgrad_reduce_slice: bad args");
    if (groups < 1) groups = 1;
    DGMR_CHECK_ARG(groups <= WG_MAX_GROUPS && nsplit % groups == 0, "dgmr_wgrad_reduce_slice: groups=%d must divide nsplit=%d", groups,
                   nsplit);
    DGMR_CHECK_ARG(coff >= 0 && coff + cin_slice <= cin_total, "dgmr_wgrad_reduce_slice: bad slice");
    DGMR_CHECK_ARG(!dot || w, "dgmr_wgrad_reduce_slice: dot needs w");
    const int64_t numel = (int64_t)Cout * taps * cin_slice;
    const int blocks = (int)std::min<int64_t>((numel + 255) / 256, 1024);
    // cs == ct would short-circuit the index map: force the mapped path whenever this is a true slice
    if (groups <= 32)
        hipLaunchKernelGGL(wgrad_reduce_kernel<32>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, partial, nsplit, groups,
                           (size_t)numel, w, scale, g, dot, taps, cin_slice, cin_total, coff, (dot && g_deterministic) ? 1 : 0);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel<WG_MAX_GROUPS>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, partial, nsplit, groups,
                           (size_t)numel, w, scale, g, dot, taps, cin_slice, cin_total, coff, (dot && g_deterministic) ? 1 : 0);
    if (dot && g_deterministic)
        hipLaunchKernelGGL(wgrad_dot_finish_kernel, dim3(1), dim3(8 * WG_MAX_GROUPS), 0, (hipStream_t)stream, dot, groups, blocks);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_sn_wgrad_finalize(const float* g, float* gw, float* dot, const float* inv_sigma, const float* u,
                                      const float* v, int Cout, int Cin, int taps, int groups, int accumulate, void* stream) {
    DGMR_CHECK_ARG(g && gw && ((u == nullptr) == (v == nullptr)) && (!u || (dot && inv_sigma)), "dgmr_sn_wgrad_finalize: null pointer");
    if (groups < 1) groups = 1;
    DGMR_CHECK_ARG(groups <= WG_MAX_GROUPS, "dgmr_sn_wgrad_finalize: groups=%d > %d", groups, WG_MAX_GROUPS);
    const size_t total = (size_t)Cout * Cin * taps;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 2048);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sn_wgrad_finalize_kernel, dim3(blocks), dim3(256), 0, s, g, gw, dot, inv_sigma, u, v, Cout, Cin, taps, groups,
                       accumulate);
    if (dot) hipLaunchKernelGGL(zero_n_kernel, dim3(1), dim3(WG_MAX_GROUPS), 0, s, dot, groups);
    DGMR_CHECK_LAUNCH();
    return 0;
}

// out[(ph*Cout + co)][a][b][ci]: see dgmr_hip.h; one thread per output element
__global__ void phase_weights_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin) {
    const int64_t total = (int64_t)16 * Cout * Cin;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        int64_t r = i / Cin;
        const int b = (int)(r & 1), a = (int)((r >> 1) & 1);
        r >>= 2;
        const int co = (int)(r % Cout), ph = (int)(r / Cout), py = ph >> 1, px = ph & 1;
        // taps of filter row ky that land on input row offset a for output parity py: py = 0: {0} | {1, 2};  py = 1: {0, 1} | {2}
        const int ky0 = py == 0 ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2), ky1 = py == 0 ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2);
        const int kx0 = px == 0 ? (b == 0 ? 0 : 1) : (b == 0 ? 0 : 2), kx1 = px == 0 ? (b == 0 ? 0 : 2) : (b == 0 ? 1 : 2);
        float acc = 0.f;
        for (int ky = ky0; ky <= ky1; ++ky)
            for (int kx = kx0; kx <= kx1; ++kx) acc += w[(((size_t)co * 3 + ky) * 3 + kx) * Cin + ci];
        out[i] = acc;
    }
}

// out[co][(p*2+q)*4 + a*2+b][ci] = sum of w[co][ky][kx][ci] over ky + i = 2a + 1 - p (i in {0, 1}: the pooled rows), kx likewise: the
// 4 x 4 stride-2 kernel of "3x3 conv, then 2x2 sum pool", split by the parity (p, q) of the input pixel it multiplies
__global__ void pool2_weights_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin) {
    const int64_t total = (int64_t)16 * Cout * Cin;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        int64_t r = i / Cin;
        const int b = (int)(r & 1), a = (int)((r >> 1) & 1), q = (int)((r >> 2) & 1), pp = (int)((r >> 3) & 1);
        const int co = (int)(r >> 4);
        const int u = 2 * a + 1 - pp, v = 2 * b + 1 - q;  // 0..3: row / column of the 4x4 kernel
        float acc = 0.f;
        for (int ky = max(0, u - 1); ky <= min(2, u); ++ky)
            for (int kx = max(0, v - 1); kx <= min(2, v); ++kx) acc += w[(((size_t)co * 3 + ky) * 3 + kx) * Cin + ci];
        out[i] = acc;
    }
}

extern "C" int dgmr_pool2_phase_weights(const float* w, float* out, int Cout, int Cin, void* stream) {
    DGMR_CHECK_ARG(w && out && Cout > 0 && Cin > 0, "dgmr_pool2_phase_weights: bad args");
    const int64_t total = (int64_t)16 * Cout * Cin;
    hipLaunchKernelGGL(pool2_weights_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, w,
                       out, Cout, Cin);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_upsample_phase_weights(const float* w, float* out, int Cout, int Cin, void* stream) {
    DGMR_CHECK_ARG(w && out && Cout > 0 && Cin > 0, "dgmr_upsample_phase_weights: bad args");
    const int64_t total = (int64_t)16 * Cout * Cin;
    hipLaunchKernelGGL(phase_weights_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, w,
                       out, Cout, Cin);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_split_weights(const float* w, uint16_t* out, int64_t rows, int Cin, int w_cin, int w_coff, int planes,
                                  int64_t plane_stride, void* stream) {
    DGMR_CHECK_ARG(w && out && rows > 0 && Cin > 0 && Cin % 2 == 0, "dgmr_split_weights: bad args (Cin=%d must be even)", Cin);
    DGMR_CHECK_ARG(planes >= 1 && planes <= 3, "dgmr_split_weights: planes=%d (1 .. 3)", planes);
    if (w_cin == 0) {
        w_cin = Cin;
        w_coff = 0;
    }
    DGMR_CHECK_ARG(w_coff >= 0 && w_coff + Cin <= w_cin, "dgmr_split_weights: bad slice");
    const int64_t total = rows * Cin;
    if (plane_stride == 0) plane_stride = total;
    DGMR_CHECK_ARG(plane_stride >= total && plane_stride % 2 == 0, "dgmr_split_weights: plane_stride=%lld < %lld elements", (long long)plane_stride,
                   (long long)total);
    const int blocks = (int)std::min<int64_t>((total / 2 + 255) / 256, 2048);
    hipLaunchKernelGGL(split_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, out, total, Cin, w_cin, w_coff, planes,
                       plane_stride);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_set_precision(int mode) {
    DGMR_CHECK_ARG(mode >= 0 && mode <= 3, "dgmr_set_precision: mode %d (0 f32, 1 bf16x3, 2 bf16, 3 bf16x6)", mode);
    g_precision = mode;
    return 0;
}
extern "C" int dgmr_get_precision(void) { return g_precision; }

extern "C" int dgmr_conv_tune(int variant, int ksplit, int window, int wgrad_window) {
    DGMR_CHECK_ARG(variant >= -1 && variant <= V_F128x32 && ksplit >= -1 && window >= -1 && window <= 7 && wgrad_window >= -1 &&
                       wgrad_window <= 5,
                   "dgmr_conv_tune: variant %d ksplit %d window %d wgrad_window %d", variant, ksplit, window, wgrad_window);
    g_tune_variant = variant;
    g_tune_ksplit = ksplit;
    g_tune_window = window;
    g_tune_wgrad_window = wgrad_window;
    return 0;
}

extern "C" int dgmr_debug_flags(int flags) {
    DGMR_CHECK_ARG(flags >= 0 && flags <= 511 && !(flags & 4), "dgmr_debug_flags: %d", flags);  // (256: phase launches one workgroup per ROW parity - tests, A/B)
    g_debug_flags = flags;
    return 0;
}

extern "C" int dgmr_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return 0;
}

extern "C" int dgmr_profile_variants(void) { return V_COUNT; }
extern "C" const char* dgmr_profile_variant_name(int v) { return (v >= 0 && v < V_COUNT) ? kVariantNames[v] : ""; }

// Synchronises the events recorded so far and returns per-variant totals; clears the records.
extern "C" int dgmr_profile_collect(double* total_ms, double* total_flops, int64_t* launches, int n) {
    return dgmr_profile_collect2(total_ms, total_flops, nullptr, launches, n);
}

// Per instantiated kernel (class row x tile / mode / launch size): "name\tlaunches\ttotal_ms\talgorithmic_flops\texecuted_flops\n" lines,
// NUL-terminated; returns the number of bytes needed (call before dgmr_profile_collect*, which clears the records).
static std::string prof_detail_name(int variant, uint32_t d) {
    static const char* const prec[] = {"f32", "bf16x3", "bf16", "bf16x6"};
    char b[160];
    const int kind = d & 15;
    const char* pr = prec[(d >> 28) & 3];
    if (kind == 1) {
        static const int bn[] = {48, 64, 96, 128, 16, 0, 0, 0};
        static const char* const mode[] = {"plain", "phase", "pooled", "phase4"};
        snprintf(b, sizeof b, "win3x3<bn%d,%dpx>%s%s%s%s %s %s", bn[(d >> 4) & 7], (d >> 8) & 1 ? 256 : 128, (d >> 13) & 1 ? " reg-staged" : "",
                 (d >> 11) & 1 ? " 3d" : "", (d >> 14) & 1 ? " gru" : "", (d >> 15) & 1 ? " ws" : "", mode[(d >> 9) & 3], (d >> 12) & 1 ? "small(<1024wg)" : "big");
    } else if (kind == 2) {
        snprintf(b, sizeof b, "%s%s%s%s%s %s", kVariantNames[variant], (d >> 8) & 1 ? " splitk" : "", (d >> 9) & 1 ? " 1x1" : "",
                 (d >> 11) & 1 ? " 3d" : "", (d >> 14) & 1 ? " gru" : "", (d >> 12) & 1 ? "small(<1024wg)" : "big");
    } else if (kind == 5) {
        snprintf(b, sizeof b, "conv1x1 stream (%s class)", kVariantNames[variant]);
    } else if (kind == 4) {
        snprintf(b, sizeof b, "stem3x3 Cin=4 <f32 mfma 16x16x4>%s", (d >> 11) & 1 ? " 3d" : "");
    } else if (kind == 3) {
        static const char* const k[] = {"im2col", "window(one-role)", "window(wave-specialised)", "?"};
        snprintf(b, sizeof b, "%s %s%s%s%s", kVariantNames[variant], k[(d >> 4) & 3], (d >> 8) & 1 ? " 3d" : "", (d >> 9) & 1 ? " upsample" : "",
                 (d >> 10) & 1 ? " 1x1" : "");
    } else {
        snprintf(b, sizeof b, "%s", variant >= 0 && variant < V_COUNT ? kVariantNames[variant] : "?");
    }
    return std::string(b) + " [" + pr + "]";
}

extern "C" int dgmr_profile_collect_detail(char* buf, int cap) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    struct Row { int64_t n = 0; double ms = 0, fl = 0, ex = 0; };
    std::map<std::string, Row> rows;
    for (auto& r : g_prof) {
        if (hipEventSynchronize(r.e1) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
        Row& w = rows[prof_detail_name(r.variant, r.detail)];
        w.n += 1, w.ms += ms, w.fl += r.flops, w.ex += r.executed;
    }
    std::string out;
    char line[320];
    for (auto& kv : rows) {
        snprintf(line, sizeof line, "%s\t%lld\t%.6f\t%.6e\t%.6e\n", kv.first.c_str(), (long long)kv.second.n, kv.second.ms, kv.second.fl, kv.second.ex);
        out += line;
    }
    if (buf && cap > 0) {
        const size_t nb = std::min<size_t>(out.size(), (size_t)cap - 1);
        memcpy(buf, out.data(), nb);
        buf[nb] = 0;
    }
    return (int)out.size() + 1;
}

extern "C" int dgmr_profile_collect2(double* total_ms, double* total_flops, double* executed_flops, int64_t* launches, int n) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < n; ++i) {
        total_ms[i] = 0.0;
        total_flops[i] = 0.0;
        if (executed_flops) executed_flops[i] = 0.0;
        launches[i] = 0;
    }
    for (auto& r : g_prof) {
        if (hipEventSynchronize(r.e1) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess && r.variant < n) {
            total_ms[r.variant] += ms;
            total_flops[r.variant] += r.flops;
            if (executed_flops) executed_flops[r.variant] += r.executed;
            launches[r.variant] += 1;
        }
        g_event_pool.push_back(r.e0);
        g_event_pool.push_back(r.e1);
    }
    g_prof.clear();
    return 0;
}
