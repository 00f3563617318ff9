// Implicit-GEMM convolution on the bf16 matrix cores of gfx950 (v_mfma_f32_32x32x16_bf16, 16x the f32 MFMA rate) with
// fp32 tensors in HBM and fp32 accumulation.
//
//   NS == 3  "bf16x3": every fp32 operand is split on the fly into a + a' (a = bf16(x), a' = bf16(x - a)) and the product is
//            formed as a*b + a*b' + a'*b on three MFMAs: products carry ~16 mantissa bits (error ~2^-16 |ab|, far inside the
//            1e-3 parity bound; per-block tests hold 1e-4), at an effective 2.5 PF / 3 = 833 TF ceiling instead of 157 TF.
//   NS == 1  plain bf16 operands (BASELINE.json configs[1]): 2.5 PF ceiling, ~3 significant digits.
//   NS == 6  "bf16x6": three planes a = a0 + a1 + a2 (exact: 3 x 8 significant bits hold an fp32 mantissa) and the six products
//            a_i * b_j with i + j <= 2; the dropped ones (a1 b2, a2 b1, a2 b2) are <= 2^-25 |ab|, i.e. below the rounding of an
//            fp32 product - fp32-faithful contractions at 2.5 PF / 6 = 417 TF instead of the 157 TF of the exact-f32 MFMA.
//
// Same tiling, staging and fused prologue / epilogue as conv_igemm_kernel (conv.hip); what differs is the LDS image: two bf16
// planes (hi, lo) per operand, rows of BK bf16 padded by 16 bytes so that the 16-byte fragment reads (lane = row, 8 consecutive
// k) of a 16-lane group land on 16 distinct 16-byte slots of the 256-byte bank row.
#pragma once
#include <type_traits>

#include "conv_device.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {  // round-to-nearest-even: v_cvt_pk_bf16_f32
    const bf16x2_t r = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);
    return __builtin_bit_cast(uint32_t, r);
}

// bf16 planes per operand of arithmetic mode NS (= MFMAs per product): 1 -> 1, 3 -> 2 (hi, lo), 6 -> 3 (hi, mid, lo)
template <int NS>
struct planes_of {
    static_assert(NS == 1 || NS == 3 || NS == 6, "NS");
    static constexpr int value = NS == 1 ? 1 : (NS == 3 ? 2 : 3);
};

// 4 floats -> NP planes of 4 bf16 (2 dwords each): plane k = bf16(v - plane 0 - ... - plane k-1); every residual is exact in fp32
template <int NP>
__device__ __forceinline__ void split_planes4(f32x4 v, u32x2 (&pl)[NP]) {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        pl[k][0] = pack_bf16(v[0], v[1]);
        pl[k][1] = pack_bf16(v[2], v[3]);
        if (k + 1 < NP) {
            v[0] -= __uint_as_float(pl[k][0] << 16);
            v[1] -= __uint_as_float(pl[k][0] & 0xffff0000u);
            v[2] -= __uint_as_float(pl[k][1] << 16);
            v[3] -= __uint_as_float(pl[k][1] & 0xffff0000u);
        }
    }
}

// The products of a split contraction, small terms first: planes (i, j) with i + j = NP - 1, then NP - 2, ... , (0, 0).
// NP 2: (1,0) (0,1) (0,0) = the three terms of bf16x3;  NP 3: (2,0) (1,1) (0,2) (1,0) (0,1) (0,0) = bf16x6.
// (compile-time plane indices: the fragments live in register arrays)
template <int T, int I, typename F>
__device__ __forceinline__ void product_step(F& f) {
    f(std::integral_constant<int, I>{}, std::integral_constant<int, T - I>{});
    if constexpr (I > 0) product_step<T, I - 1>(f);
    else if constexpr (T > 0) product_step<T - 1, T - 1>(f);
}
template <int NP, typename F>
__device__ __forceinline__ void for_each_product(F&& f) {
    product_step<NP - 1, NP - 1>(f);
}

// Instruction order asked of the scheduler for one 16-channel step of a split contraction (round 6).  The fragments are READ in the order
// the products consume them - for_each_product starts with A's last plane x B's first, so read (A[NP-1], B[0]), (A[NP-2], B[1]), ... -
// and the reads of pair t + 1 are spread between the MFMAs of product t: LDS reads return in order, so the first MFMAs issue as soon as
// the first TM + TN fragments have landed.  Left alone the compiler clusters all reads of a step and waits for every one of them
// (`s_waitcnt lgkmcnt(0)`) before the first MFMA - the LDS latency of a step fully exposed, twice per tap of the window kernel
// (803.6 vs 816.4 ms per training step from that kernel alone).  The s_setprio pair must enclose reads AND MFMAs: it bounds a
// scheduling region.  The builtin wants literal counts, hence the templates.
template <int READS, int M, int I>
__device__ __forceinline__ void mfma_read_interleave() {
    if constexpr (I < M) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        constexpr int N = (READS * (I + 1)) / M - (READS * I) / M;
        if constexpr (N > 0) __builtin_amdgcn_sched_group_barrier(0x100, N, 0);
        mfma_read_interleave<READS, M, I + 1>();
    }
}
template <int NP, int TM, int TN, int T = 1>
__device__ __forceinline__ void schedule_split_products() {
    if constexpr (NP >= 2) {
        if constexpr (T == 1) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
        if constexpr (T < NP) {
            mfma_read_interleave<TM + TN, TM * TN, 0>();
            schedule_split_products<NP, TM, TN, T + 1>();
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, (NP * (NP + 1) / 2 - (NP - 1)) * TM * TN, 0);
        }
    }
}

template <int BM, int BN, int BK, int WM, int WN, int NS>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_bf16_kernel(const dgmr_conv_args p, const int M, const int Ktot,
                                                                  const int kt_per_split) {
    constexpr int NT = 64 * WM * WN;
    constexpr int KQ = BK / 4;           // threads per tile row (4 k each)
    constexpr int RPP = NT / KQ;         // tile rows filled per pass
    constexpr int AP = BM / RPP, BP = BN / RPP;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int LDW = BK / 2 + 4;      // row stride in dwords: BK bf16 + 16 bytes of padding
    constexpr int NP = planes_of<NS>::value;  // bf16 planes per operand
    static_assert(AP >= 1 && BP >= 1 && TM >= 1 && TN >= 1 && BM % RPP == 0 && BN % RPP == 0, "bad tile");
    static_assert(BK % 16 == 0, "BK must be a multiple of the MFMA k (16)");

    // [stage][plane][row][LDW]
    __shared__ __attribute__((aligned(16))) uint32_t smem[2 * NP * (BM + BN) * LDW];
    uint32_t* As = smem;
    uint32_t* Bs = smem + 2 * NP * BM * LDW;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int kq = tid % KQ, lrow = tid / KQ;

    const int KHW = p.KH * p.KW;
    const int pd = p.KD >> 1, ph = p.KH >> 1, pw = p.KW >> 1;
    const uint32_t wrow = (uint32_t)p.KD * KHW * p.w_cin;
    const int DHW = p.D * p.H * p.W;

    // this thread's AP tile rows (output pixels)
    int rn[AP], rd[AP], rh[AP], rw[AP];
    bool rok[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int m = m0 + i * RPP + lrow;
        rok[i] = m < M;
        const RowCoord rc = decode_row(rok[i] ? m : 0, p.D, p.H, p.W);
        rn[i] = rc.n, rd[i] = rc.d, rh[i] = rc.h, rw[i] = rc.w;
    }
    // BatchNorm-on-load: one (a, b) pair per tile when every row of the workgroup's tile belongs to the same statistics group
    const int grp0 = (m0 / DHW) / p.pre_group;
    const bool grp_uniform = p.pre_a && ((min(m0 + BM, M) - 1) / DHW) / p.pre_group == grp0;

    // position of this thread's 4-channel group on the K axis, advanced incrementally (no divisions in the loop)
    const int nk_all = (Ktot + BK - 1) / BK;
    const int kt0 = blockIdx.z * kt_per_split;
    const int nk = min(nk_all, kt0 + kt_per_split);
    int k = kt0 * BK + kq * 4;
    int tap = k / p.Cin;
    int ci = k - tap * p.Cin;
    int kz = tap / KHW;
    int ky = (tap - kz * KHW) / p.KW;
    int kx = tap - kz * KHW - ky * p.KW;

    struct Stage {
        f32x4 a[AP], b[BP], pa, pb;
        unsigned vmask, bmask;
        int ci;
    };
    // Every load below is UNCONDITIONAL (invalid elements read a clamped, always-mapped address and are zeroed at store time):
    // a load inside a divergent `if` makes hipcc wait vmcnt(0) right behind it, which serialises the whole operand fetch.
    const int us = p.upsample ? 1 : 0;
    const int Hs = p.H >> us, Ws = p.W >> us;
    const float* pa_base = p.pre_a ? p.pre_a : p.x;
    const float* pb_base = p.pre_a ? p.pre_b : p.x;
    auto load = [&](Stage& s) {
        const bool kok = k < Ktot;
        const int dz = kz - pd, dy = ky - ph, dx = kx - pw;
        s.ci = ci;
        unsigned vm = 0;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int id = rd[i] + dz, ih = rh[i] + dy, iw = rw[i] + dx;
            const bool valid = rok[i] && kok && (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            const uint32_t off = ((((uint32_t)rn[i] * p.D + id) * Hs + (ih >> us)) * Ws + (iw >> us)) * p.Cin + ci;
            s.a[i] = *reinterpret_cast<const f32x4*>(p.x + (valid ? off : 0u));
            vm |= (valid ? 1u : 0u) << i;
        }
        s.vmask = vm;
        {
            const uint32_t g = (grp_uniform && kok) ? (uint32_t)grp0 * p.Cin + ci : 0u;
            s.pa = *reinterpret_cast<const f32x4*>(pa_base + g);
            s.pb = *reinterpret_cast<const f32x4*>(pb_base + g);
        }
        const uint32_t wk = kok ? (uint32_t)tap * p.w_cin + p.w_coff + ci : 0u;
        unsigned bm = 0;
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            const int co = n0 + i * RPP + lrow;
            const bool ok = co < p.Cout && kok;
            s.b[i] = *reinterpret_cast<const f32x4*>(p.w + (uint32_t)min(co, p.Cout - 1) * wrow + wk);
            bm |= (ok ? 1u : 0u) << i;
        }
        s.bmask = bm;
        // advance to the next k tile
        k += BK;
        ci += BK;
        while (ci >= p.Cin) {
            ci -= p.Cin;
            ++tap;
            if (++kx == p.KW) {
                kx = 0;
                if (++ky == p.KH) {
                    ky = 0;
                    ++kz;
                }
            }
        }
    };
    auto store = [&](const Stage& s, int buf) {
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            f32x4 v = s.a[i];
            if (p.pre_a) {
                f32x4 a = s.pa, b = s.pb;
                if (!grp_uniform) {  // tile straddles two statistics groups (never at the model's shapes): per-row lookup
                    const uint32_t g = (uint32_t)(rn[i] / p.pre_group) * p.Cin + s.ci;
                    a = *reinterpret_cast<const f32x4*>(p.pre_a + g);
                    b = *reinterpret_cast<const f32x4*>(p.pre_b + g);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaf(v[j], a[j], b[j]), 0.f);
            } else if (p.pre_relu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            v = ((s.vmask >> i) & 1u) ? v : zero4;
            u32x2 pl[NP];
            split_planes4<NP>(v, pl);
            uint32_t* dst = As + ((buf * NP) * BM + i * RPP + lrow) * LDW + kq * 2;
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<u32x2*>(dst + q * BM * LDW) = pl[q];
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            const f32x4 v = ((s.bmask >> i) & 1u) ? s.b[i] : zero4;
            u32x2 pl[NP];
            split_planes4<NP>(v, pl);
            uint32_t* dst = Bs + ((buf * NP) * BN + i * RPP + lrow) * LDW + kq * 2;
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<u32x2*>(dst + q * BN * LDW) = pl[q];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto mma = [&](int cur) {
        // fragment base of this lane: row (lane & 31) of the wave's first block, 8 consecutive k starting at (lane >> 5) * 8
        const uint32_t* Ab = As + ((cur * NP) * BM + wm * TM * 32 + (lane & 31)) * LDW + (lane >> 5) * 4;
        const uint32_t* Bb = Bs + ((cur * NP) * BN + wn * TN * 32 + (lane & 31)) * LDW + (lane >> 5) * 4;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8_t af[NP][TM], bf[NP][TN];
            __builtin_amdgcn_s_setprio(1);  // the co-resident workgroup is usually in its load / store phase: matrix pipe first
#pragma unroll
            for (int t = 0; t < NP; ++t) {  // (in the order the products consume them: schedule_split_products)
                const int qa = NP - 1 - t, qb = t;
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[qa][i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(Ab + (qa * BM + i * 32) * LDW + kk * 8));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bf[qb][j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(Bb + (qb * BN + j * 32) * LDW + kk * 8));
            }
            // term-major order: consecutive MFMAs hit different accumulators (no back-to-back dependent issue); small terms first
            for_each_product<NP>([&](auto qa, auto qb) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[qa][i], bf[qb][j], acc[i][j], 0, 0, 0);
            });
            schedule_split_products<NP, TM, TN>();
            __builtin_amdgcn_s_setprio(0);
        }
    };

    // Two register stages: while tile t is multiplied out of LDS, tile t+1 waits in registers and tile t+2 is in flight, so a
    // load has a whole iteration (and the other resident workgroup's MFMAs) to land before it is split and stored.
    Stage st0, st1;
    load(st0);
    load(st1);  // beyond the last tile k >= Ktot: harmless clamped loads, zero tiles
    store(st0, 0);
    __syncthreads();
    for (int kt = kt0; kt < nk; kt += 2) {
        load(st0);
        mma(0);
        store(st1, 1);
        __syncthreads();
        if (kt + 1 >= nk) break;
        load(st1);
        mma(1);
        store(st0, 0);
        __syncthreads();
    }

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

// ------------------------------------------------------------------------------------------------------------------------
// Weight gradient on the bf16 matrix cores:  G[co][k] = sum_m dY[m][co] * A[m][k]  over one slab of pixels m.
// Both MFMA operands need 8 consecutive *pixels* per lane (the reduction index), while HBM has channels contiguous: every
// thread therefore loads a 4-pixel x 4-channel block (four 16-byte loads), transposes it in registers and writes, per
// channel, 4 consecutive pixels as one 8-byte bf16 group -> LDS images [co][m] and [k][m] with the padded row stride of the
// forward kernel.  bias_grad (optional): column sums of dY ride along (the workgroups of the first k tile add them up).
// ------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void advance_row(RowCoord& r, int delta, int D, int H, int W) {
    r.w += delta;
    while (r.w >= W) {
        r.w -= W;
        if (++r.h == H) {
            r.h = 0;
            if (++r.d == D) {
                r.d = 0;
                ++r.n;
            }
        }
    }
}

template <int BI, int WI, int WJ, int NS>
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16_kernel(const dgmr_wgrad_args p, const int M, const int Ktot,
                                                                 const int rows_per_split, const int splits_per_group,
                                                                 const int rows_per_group) {
    constexpr int BJ = 128, BR = 32;
    constexpr int LDW = BR / 2 + 4;  // dwords per LDS row: 32 bf16 pixels + 16 bytes of padding
    constexpr int NP = planes_of<NS>::value;
    constexpr int TM = BI / WI / 32, TN = BJ / WJ / 32;
    static_assert(WI * WJ == 4 && TM >= 1 && TN >= 1, "bad tile");

    __shared__ __attribute__((aligned(16))) uint32_t smem[2 * NP * (BI + BJ) * LDW];
    uint32_t* Ys = smem;
    uint32_t* Xs = smem + 2 * NP * BI * LDW;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wi = wid / WJ, wj = wid % WJ;
    const int k0 = blockIdx.x * BJ, co0 = blockIdx.y * BI;
    const int grp = blockIdx.z / splits_per_group;
    const int r_begin = grp * rows_per_group + (blockIdx.z - grp * splits_per_group) * rows_per_split;
    const int r_end = min(min(M, (grp + 1) * rows_per_group), r_begin + rows_per_split);
    const int mg = tid & 7, cg = tid >> 3;  // this thread's 4-pixel group (0..7) and 4-channel group (0..31)

    const int KHW = p.KH * p.KW;
    const int pd = p.KD >> 1, ph = p.KH >> 1, pw = p.KW >> 1;
    const KPos kp = decode_k(k0 + cg * 4, Ktot, p.Cin, p.KW, KHW, pd, ph, pw);  // fixed for the whole kernel
    const int co = co0 + cg * 4;
    const bool y_on = cg * 4 < BI && co < p.Cout;

    RowCoord rc = decode_row(min(r_begin + mg * 4, M - 1), p.D, p.H, p.W);  // first pixel of this thread's group
    int m_first = r_begin + mg * 4;

    f32x4 ry[4], rx[4];
    unsigned xmask = 0, ymask = 0;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    // BatchNorm-on-load parameters: this thread's 4 channels never change and a slab lies inside one statistics group
    const int DHW = p.D * p.H * p.W;
    f32x4 bn_a = {1.f, 1.f, 1.f, 1.f}, bn_b = {0.f, 0.f, 0.f, 0.f};
    if (p.pre_a && kp.ok) {
        const uint32_t g = (uint32_t)((min(r_begin, M - 1) / DHW) / p.pre_group) * p.Cin + kp.ci;
        bn_a = *reinterpret_cast<const f32x4*>(p.pre_a + g);
        bn_b = *reinterpret_cast<const f32x4*>(p.pre_b + g);
    }
    const int us = p.upsample ? 1 : 0;
    const int Hs = p.H >> us, Ws = p.W >> us;
    const int co_c = min(co, p.Cout - 4);  // clamped (Cout % 4 == 0): invalid lanes read a mapped address and are zeroed
    // every load is unconditional (see conv_bf16_kernel): out-of-range elements read a clamped address, masks zero them later
    auto load = [&]() {
        RowCoord r = rc;
        xmask = 0;
        ymask = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m_first + j;
            const bool ok = m < r_end;
            ry[j] = *reinterpret_cast<const f32x4*>(p.dy + (size_t)(ok ? m : 0) * p.Cout + co_c);
            ymask |= ((ok && y_on) ? 1u : 0u) << j;
            const int id = r.d + kp.dz, ih = r.h + kp.dy, iw = r.w + kp.dx;
            const bool valid = ok && kp.ok && (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            const uint32_t off = ((((uint32_t)r.n * p.D + id) * Hs + (ih >> us)) * Ws + (iw >> us)) * p.Cin + kp.ci;
            rx[j] = *reinterpret_cast<const f32x4*>(p.x + (valid ? off : 0u));
            xmask |= (valid ? 1u : 0u) << j;
            advance_row(r, 1, p.D, p.H, p.W);
        }
        m_first += BR;
        advance_row(rc, BR, p.D, p.H, p.W);
    };
    auto store = [&](int buf) {
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ry[j] = ((ymask >> j) & 1u) ? ry[j] : zero4;
            bsum += ry[j];
            f32x4 v = rx[j];
            if (p.pre_a) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = fmaxf(fmaf(v[c], bn_a[c], bn_b[c]), 0.f);
            } else if (p.pre_relu) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
            }
            rx[j] = ((xmask >> j) & 1u) ? v : zero4;
        }
        // transpose: channel c of the 4 pixels -> one 8-byte group of 4 consecutive pixels
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            u32x2 pl[NP];
            if (cg * 4 < BI) {
                split_planes4<NP>((f32x4){ry[0][c], ry[1][c], ry[2][c], ry[3][c]}, pl);
                uint32_t* dst = Ys + ((buf * NP) * BI + cg * 4 + c) * LDW + mg * 2;
#pragma unroll
                for (int q = 0; q < NP; ++q) *reinterpret_cast<u32x2*>(dst + q * BI * LDW) = pl[q];
            }
            split_planes4<NP>((f32x4){rx[0][c], rx[1][c], rx[2][c], rx[3][c]}, pl);
            uint32_t* dst = Xs + ((buf * NP) * BJ + cg * 4 + c) * LDW + mg * 2;
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<u32x2*>(dst + q * BJ * LDW) = pl[q];
        }
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
        load();
        store(0);
    }
    __syncthreads();
    for (int it = 0; it < nr; ++it) {
        const int cur = it & 1;
        load();  // past the slab's end every element is masked
        const uint32_t* Yb = Ys + ((cur * NP) * BI + wi * TM * 32 + (lane & 31)) * LDW + (lane >> 5) * 4;
        const uint32_t* Xb = Xs + ((cur * NP) * BJ + wj * TN * 32 + (lane & 31)) * LDW + (lane >> 5) * 4;
#pragma unroll
        for (int kk = 0; kk < BR / 16; ++kk) {
            bf16x8_t yf[NP][TM], xf[NP][TN];
            __builtin_amdgcn_s_setprio(1);  // the co-resident workgroup is usually in its load / store phase: matrix pipe first
#pragma unroll
            for (int t = 0; t < NP; ++t) {  // (in the order the products consume them: schedule_split_products)
                const int qa = NP - 1 - t, qb = t;
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    yf[qa][i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(Yb + (qa * BI + i * 32) * LDW + kk * 8));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    xf[qb][j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(Xb + (qb * BJ + j * 32) * LDW + kk * 8));
            }
            // term-major order: consecutive MFMAs hit different accumulators (no back-to-back dependent issue); small terms first
            for_each_product<NP>([&](auto qa, auto qb) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[qa][i], xf[qb][j], acc[i][j], 0, 0, 0);
            });
            schedule_split_products<NP, TM, TN>();
            __builtin_amdgcn_s_setprio(0);
        }
        if (it + 1 < nr) store(cur ^ 1);
        __syncthreads();
    }

    float* out = p.partial + (size_t)blockIdx.z * p.Cout * Ktot;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c_o = co0 + wi * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (c_o >= p.Cout) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int k = k0 + wj * TN * 32 + j * 32 + (lane & 31);
                if (k < Ktot) out[(size_t)c_o * Ktot + k] = acc[i][j][r];
            }
        }
    // bias gradient: the 8 threads of a channel group (consecutive lanes) hold the partial column sums of this slab
    if (p.bias_grad && blockIdx.x == 0 && nr > 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = bsum[c];
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 4, 64);
            if (mg == 0 && y_on) atomicAdd(p.bias_grad + (size_t)blockIdx.z * p.bias_stride + co + c, v);  // (bias_stride = Cout: a row per slab, one writer)
        }
    }
}

}  // namespace

namespace {

// ------------------------------------------------------------------------------------------------------------------------
// 3x3 convolution with the INPUT WINDOW staged in LDS (forward and data gradient of the big feature maps).
//
// The implicit-GEMM kernels above gather every im2col element from L2 once per filter tap and spend most of their issue slots
// on address arithmetic and on splitting fp32 operands.  Here a workgroup owns TH x TW = 128 output pixels of one image and
// BN output channels and walks the input channels in chunks of 32:
//   * the (TH+2) x (TW+2) input halo of the chunk is fetched ONCE (fused BatchNorm/ReLU prologue, bf16 split) into LDS and all
//     nine taps read their shifted windows from it: 5.6x fewer activation loads and 9x less prologue / split arithmetic;
//   * weights arrive pre-split as bf16 planes [2][Cout][9][Cin] (dgmr_split_weights, once per optimiser step) and are copied
//     tap by tap through a two-stage LDS ring with no arithmetic at all;
//   * nearest-2x upsampling stages the half-resolution halo and folds the >>1 into the window address.
// ------------------------------------------------------------------------------------------------------------------------
// NSTAGE: weight stages in LDS (2: one barrier per tap; 1: two barriers, smaller footprint -> three workgroups per CU).
// LAZYA: fetch the next halo at the chunk boundary instead of holding it in registers under the nine taps (fewer live VGPRs).
// BM: output pixels per workgroup.  256 (TW = 32 or 16 only) gives every wave 64 pixels: fewer LDS fragment reads per MFMA
// ((TM + TN) * planes reads feed TM * TN * terms MFMAs), half the barriers per MFMA and a 1.33x instead of 1.59x halo.
template <int BN, int WM, int WN, int NS, int NSTAGE = 2, bool LAZYA = false, int BM = 128>
__global__ __launch_bounds__(256, (NSTAGE == 1 && BM == 128) ? 3 : 2) void conv3x3_win_kernel(const dgmr_conv_args p, const int tw_shift,
                                                                                             const int tiles_w, const int tiles_hw,
                                                                                             const int g_shift) {
    constexpr int CK = 32;
    constexpr int LOG_BM = BM == 256 ? 8 : 7;
    static_assert(BM == 128 || BM == 256, "BM");
    constexpr int LDW = CK / 2 + 4;  // 80-byte rows
    constexpr int NP = planes_of<NS>::value;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int AMAX = BM == 256 ? 10 * 34 : 6 * 34;  // halo pixels: 6 x 34 (TW = 32) or 10 x 18 (TW = 16); BM 256: 10 x 34 / 18 x 18
    constexpr int APASS = (AMAX * 8 + 255) / 256;     // 16-byte fp32 items of the halo per thread
    constexpr int BITEMS = BN * 4 * NP;               // 16-byte bf16 items of one weight stage
    constexpr int BPASS = (BITEMS + 255) / 256;
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "bad tile");

    __shared__ __attribute__((aligned(16))) uint32_t smem[NP * AMAX * LDW + NSTAGE * NP * BN * LDW];
    uint32_t* As = smem;                     // [plane][pixel][LDW]
    uint32_t* Bs = smem + NP * AMAX * LDW;   // [stage][plane][co][LDW]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    // A tile is 2^g_shift whole small images (8x8 maps: two images of 64 pixels, each with its own 10x10 halo) or, for g_shift
    // == 0, TH x TW pixels of one image.  Groups (1/sigma, BatchNorm statistics, masks) never split the images of a tile (host).
    const int TW = 1 << tw_shift, TH = (BM >> tw_shift) >> g_shift;  // rows per sub-tile
    const int sub_shift = LOG_BM - g_shift;                           // log2(pixels per sub-tile)
    const int tile = blockIdx.x;
    const int n = g_shift ? (tile << g_shift) : tile / tiles_hw;      // first image of the tile
    const int trem = g_shift ? 0 : tile - n * tiles_hw;
    const int th = trem / tiles_w;
    const int h0 = th * TH, w0 = (trem - th * tiles_w) * TW;
    const int n0 = blockIdx.y * BN;
    const int us = p.upsample ? 1 : 0;
    const int Hs = p.H >> us, Ws = p.W >> us;
    const int oh = (h0 - 1) >> us, ow = (w0 - 1) >> us;  // halo origin in input coordinates (arithmetic shift: -1 stays -1)
    const int HTw = (TW >> us) + 2;
    const int HP = ((TH >> us) + 2) * HTw;  // halo pixels per sub-tile
    const int npix = HP << g_shift;
    const int nchunks = (p.Cin + CK - 1) / CK;
    const int S = nchunks * 9;

    // ---- activation halo: per-thread item geometry is the same for every chunk ----
    const int cq = tid & 7;  // 4-channel group inside the chunk
    uint32_t a_goff[APASS];
    unsigned a_valid = 0;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
        const int pix = (tid >> 3) + i * 32;
        const int sub = pix / HP, prem = pix - sub * HP;
        const int lr = prem / HTw, lc = prem - lr * HTw;
        const int ih = oh + lr, iw = ow + lc;
        const bool ok = pix < npix && (unsigned)ih < (unsigned)Hs && (unsigned)iw < (unsigned)Ws;
        a_goff[i] = ok ? (((uint32_t)(n + sub) * Hs + ih) * Ws + iw) * p.Cin + cq * 4 : 0u;
        a_valid |= (ok ? 1u : 0u) << i;
    }
    const float* pa_base = p.pre_a ? p.pre_a : p.x;
    const float* pb_base = p.pre_a ? p.pre_b : p.x;
    const uint32_t grp_off = (uint32_t)(n / p.pre_group) * p.Cin;

    f32x4 ra[APASS], rpa, rpb;
    bool a_kok = false;
    auto issue_a = [&](int chunk) {
        const int cb = chunk * CK + cq * 4;
        a_kok = cb < p.Cin;
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            const bool ok = a_kok && ((a_valid >> i) & 1u);
            ra[i] = *reinterpret_cast<const f32x4*>(p.x + (ok ? a_goff[i] + chunk * CK : 0u));
        }
        const uint32_t g = a_kok ? grp_off + cb : 0u;
        rpa = *reinterpret_cast<const f32x4*>(pa_base + g);
        rpb = *reinterpret_cast<const f32x4*>(pb_base + g);
    };
    auto store_a = [&]() {
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            const int pix = (tid >> 3) + i * 32;
            f32x4 v = ra[i];
            if (p.pre_a) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaf(v[j], rpa[j], rpb[j]), 0.f);
            } else if (p.pre_relu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            v = (a_kok && ((a_valid >> i) & 1u)) ? v : zero4;
            u32x2 pl[NP];
            split_planes4<NP>(v, pl);
            if (pix < AMAX) {
                uint32_t* dst = As + pix * LDW + cq * 2;
#pragma unroll
                for (int q = 0; q < NP; ++q) *reinterpret_cast<u32x2*>(dst + q * AMAX * LDW) = pl[q];
            }
        }
    };

    // ---- weights: stage s = chunk * 9 + tap is a plain copy of pre-split bf16 ----
    const size_t plane_stride = (size_t)p.Cout * 9 * p.Cin;  // bf16 elements per plane
    u32x4 rb[BPASS];
    unsigned b_ok = 0;
    auto issue_b = [&](int s) {
        const int chunk = s / 9, tap = s - chunk * 9;
        b_ok = 0;
#pragma unroll
        for (int i = 0; i < BPASS; ++i) {
            const int item = tid + i * 256;
            const int plane = item / (BN * 4);
            const int r = item - plane * (BN * 4);
            const int co = n0 + (r >> 2), q = r & 3;
            const int ci = chunk * CK + q * 8;
            const bool ok = item < BITEMS && s < S && co < p.Cout && ci < p.Cin;
            const size_t off = ok ? plane * plane_stride + ((size_t)co * 9 + tap) * p.Cin + ci : 0;
            rb[i] = *reinterpret_cast<const u32x4*>(p.w_split + off);
            b_ok |= (ok ? 1u : 0u) << i;
        }
    };
    auto store_b = [&](int stage) {
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < BPASS; ++i) {
            const int item = tid + i * 256;
            if (item < BITEMS) {
                const int plane = item / (BN * 4);
                const int r = item - plane * (BN * 4);
                uint32_t* dst = Bs + ((stage * NP + plane) * BN + (r >> 2)) * LDW + (r & 3) * 4;
                *reinterpret_cast<u32x4*>(dst) = ((b_ok >> i) & 1u) ? rb[i] : z;
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // tile-local pixel of this lane in each of the wave's TM row blocks
    int prow[TM], pcol[TM], pbase[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int q = wm * TM * 32 + i * 32 + (lane & 31);
        prow[i] = (q >> tw_shift) & (TH - 1);
        pcol[i] = q & (TW - 1);
        pbase[i] = (q >> sub_shift) * HP;  // halo of the lane's image inside the tile
    }
    auto mma = [&](int tap, int stage) {
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const uint32_t* Ab[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int lr = ((h0 + prow[i] + dy) >> us) - oh;
            const int lc = ((w0 + pcol[i] + dx) >> us) - ow;
            Ab[i] = As + (pbase[i] + lr * HTw + lc) * LDW + (lane >> 5) * 4;
        }
        const uint32_t* Bb = Bs + ((stage * NP) * BN + wn * TN * 32 + (lane & 31)) * LDW + (lane >> 5) * 4;
#pragma unroll
        for (int kk = 0; kk < CK / 16; ++kk) {
            bf16x8_t af[NP][TM], bf[NP][TN];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[q][i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(Ab[i] + q * AMAX * LDW + kk * 8));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bf[q][j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(Bb + (q * BN + j * 32) * LDW + kk * 8));
            }
            // term-major order: consecutive MFMAs hit different accumulators (no back-to-back dependent issue); small terms first
            __builtin_amdgcn_s_setprio(1);  // the co-resident workgroup is usually in its load / store phase: matrix pipe first
            for_each_product<NP>([&](auto qa, auto qb) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[qa][i], bf[qb][j], acc[i][j], 0, 0, 0);
            });
            __builtin_amdgcn_s_setprio(0);
        }
    };

    issue_a(0);
    issue_b(0);
    store_a();
    store_b(0);
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        if (!LAZYA) issue_a(chunk + 1);  // next halo: in flight under the nine taps (clamped, zero beyond the last chunk)
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int s = chunk * 9 + tap;
            issue_b(s + 1);
            if (NSTAGE == 2) {
                mma(tap, s & 1);
                store_b((s + 1) & 1);
                if (tap == 8) {
                    __syncthreads();  // every wave is done with this chunk's halo
                    store_a();
                }
                __syncthreads();
            } else {  // one weight stage (smaller LDS footprint, three workgroups per CU): publish only after everyone has read
                mma(tap, 0);
                __syncthreads();
                store_b(0);
                if (tap == 8) {
                    if (LAZYA) issue_a(chunk + 1);
                    store_a();
                }
                __syncthreads();
            }
        }
    }

    // epilogue: lane = output channel, 16 pixels per MFMA block.  The whole tile lies in image n: scale and bias are hoisted, and
    // the common case (scale, bias, optional relu - no addend / residual / mask / gating) is a straight fma + store.
    const float sc = p.scale ? p.scale[n / p.scale_group] : 1.f;
    float bj[TN];
    int colj[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        colj[j] = n0 + wn * TN * 32 + j * 32 + (lane & 31);
        bj[j] = (p.bias && colj[j] < p.Cout) ? p.bias[colj[j]] : 0.f;
    }
    const bool simple = !p.addend && p.epi_mode == DGMR_EPI_PLAIN;  // scale, bias, relu, residual, relu/BN mask: straight line
    float maj[TN], mbj[TN];  // affine of the BatchNorm whose relu is being back-propagated through (data gradient), per column
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const bool on = p.mask_a && colj[j] < p.Cout;
        const size_t g = (size_t)(n / p.mask_group) * p.Cout + (on ? colj[j] : 0);
        maj[j] = on ? p.mask_a[g] : 1.f;
        mbj[j] = on ? p.mask_b[g] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int hh = h0 + ((q >> tw_shift) & (TH - 1)), ww = w0 + (q & (TW - 1)), ni = n + (q >> sub_shift);
            const int m = (ni * p.H + hh) * p.W + ww;
            float* yrow = p.y + (size_t)m * p.Cout;
            const size_t rbase = p.residual_up ? (((size_t)ni * (p.H >> 1) + (hh >> 1)) * (p.W >> 1) + (ww >> 1)) * p.Cout
                                               : (size_t)m * p.Cout;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (colj[j] >= p.Cout) continue;
                if (simple) {
                    float v = fmaf(acc[i][j][r], sc, bj[j]);
                    if (p.act_relu) v = fmaxf(v, 0.f);
                    if (p.residual) v += p.residual[rbase + colj[j]];
                    if (p.mask_src) v = fmaf(p.mask_src[(size_t)m * p.Cout + colj[j]], maj[j], mbj[j]) > 0.f ? v : 0.f;
                    yrow[colj[j]] = v;
                } else {
                    epilogue_store(p, acc[i][j][r], ni, colj[j], (size_t)m * p.Cout + colj[j], rbase + colj[j]);
                }
            }
        }
    }
}

// out[plane][i], plane k = bf16(w - plane 0 - ... - plane k-1), `planes` of them (2: bf16x3 / bf16, 3: bf16x6); i runs over
// (co, tap, ci) of the slice [w_coff, w_coff+Cin)
__global__ void split_weights_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int64_t total, int Cin, int w_cin,
                                     int w_coff, int planes, int64_t plane_stride) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total / 2; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i * 2;  // Cin is even: the pair never straddles a row
        const int64_t row = e / Cin;
        const int ci = (int)(e - row * Cin);
        float a = w[row * w_cin + w_coff + ci], b = w[row * w_cin + w_coff + ci + 1];
        for (int k = 0; k < planes; ++k) {
            const uint32_t q = pack_bf16(a, b);
            reinterpret_cast<uint32_t*>(out + (int64_t)k * plane_stride)[i] = q;
            a -= __uint_as_float(q << 16);
            b -= __uint_as_float(q & 0xffff0000u);
        }
    }
}

}  // namespace
