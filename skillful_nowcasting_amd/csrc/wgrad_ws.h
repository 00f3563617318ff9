// Weight gradient of a 3x3 convolution, WAVE-SPECIALISED (round 3): four loader waves keep a double-buffered LDS image of the next
// tile coming while three or four matrix waves multiply the current one.
//
//   G[co][tap][ci] = sum over pixels  dY[n,h,w,co] * pre(x)[n, h+dy, w+dx, ci]     (autograd of F.conv2d at every `spectral_norm(conv)`
//                                                                                   call of the reference)
//
// conv_wgrad_win_kernel (wgrad_win.h) does the same arithmetic with ONE set of waves: fetch the tile, transpose it 4 px x 4 ch in
// registers, store it [channel][pixel] through conflicted 8-byte LDS writes, barrier, 108 MFMAs per wave, barrier - nothing overlaps
// inside a workgroup and two of them share a CU: a CU finishes a 64-pixel tile every ~12 000 cycles where the matrix work is 3 456
// (profiles/r03_probe_wgrad_full_batch.log: 181 - 303 TF against 314 - 354 for the forward of the same layers).  Here
//   * the tiles stay PIXEL-MAJOR in LDS exactly as they arrive from HBM ([pixel][32 ci], [pixel][BI co], bf16 planes): the loaders only
//     apply the fused BatchNorm / ReLU prologue, split into planes and issue lane-linear ds_write_b64 - no register transposes;
//   * the matrix waves read their MFMA fragments with ds_read_b64_tr_b16, gfx950's transposing LDS read: within a 16-lane group lane i
//     supplies the address of 4 consecutive 16-bit elements of row i / 4 (segment i % 4) and receives column i of that 4 x 16 block
//     (decoded on the hardware: tools/probes/tr_read_probe.hip, profiles/r03_probe_ds_read_tr16_b64.log) - the reduction index (pixels)
//     becomes the fragment's k without any data movement, and the +-1 shifts of the taps are plain row offsets;
//   * LDS holds TWO tile images: while the matrix waves are on tile t the loaders store tile t + 1 (fetched two iterations earlier
//     into one of two register sets: the HBM latency has two whole matrix phases to pass) - ONE barrier per tile;
//   * matrix waves: MW = 3 - wave w owns filter row w (3 taps x BI / 32 output-channel blocks = up to 9 accumulators [32 co x 32 ci],
//     as before) - or MW = 4, one per SIMD, on 16 x 16 x 32 MFMAs (see the template); one workgroup of 7 / 8 waves per CU (84 KB of
//     LDS at BI = 96 in bf16x3);
//   * the two roles are two DISJOINT programs (`if (loader) {...} else {...}`, each with its own loop) that meet at the same
//     barriers - the hardware counts arriving waves, not code addresses.  In one shared loop with per-role bodies the compiler keeps
//     the loaders' 104 staging registers and the 144 accumulators live together (1 KB of scratch per lane) and merges the two paths'
//     s_waitcnt bookkeeping at every barrier, so that each store waits for nearly all loads in flight.
// Output: partial[slab][co][tap*Cin + ci], the layout dgmr_wgrad_reduce* consume; bias gradient as in wgrad_win.h.
// 3 x 3 x 3 convs: one launch per depth tap (kernel argument kd) over the N x D planes, each writing its 9 taps of the 27.
#pragma once
#include "conv_bf16.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// 8 bytes per lane, transposed across each 16-lane group; `p` points into LDS
__device__ __forceinline__ s16x4 lds_read_tr(const uint32_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
#else
    (void)p;
    return (s16x4){0, 0, 0, 0};
#endif
}
// MFMA fragment (8 consecutive k of this lane's row / column): k 0..3 from `p`, k 4..7 from four rows further
__device__ __forceinline__ bf16x8_t tr_fragment(const uint32_t* p, int row4_dwords) {
    const s16x4 lo = lds_read_tr(p), hi = lds_read_tr(p + row4_dwords);
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

constexpr int ws_gcd(int a, int b) { return b == 0 ? a : ws_gcd(b, a % b); }

// scheduling template "MFMA, its share of READS LDS reads" x M (the builtin wants literal counts)
template <int READS, int M, int I>
__device__ __forceinline__ void ws_interleave() {
    if constexpr (I < M) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        constexpr int N = (READS * (I + 1)) / M - (READS * I) / M;
        if constexpr (N > 0) __builtin_amdgcn_sched_group_barrier(0x100, N, 0);
        ws_interleave<READS, M, I + 1>();
    }
}

// MW = 3: three matrix waves, one filter row each, 32 x 32 x 16 MFMAs (9 accumulators [32 co x 32 ci]).
// MW = 4: FOUR matrix waves, one per SIMD, 16 x 16 x 32 MFMAs: wave (ci half, co half) owns 16 input channels x BI / 2 output channels
//         x all 9 taps = 9 x BI / 32 accumulators [16 co x 16 ci]; the LDS images are [16-channel block][pixel][16] (32-byte rows), so
//         that the 8 pixel rows a 32-lane group of a transposing read touches are 256 contiguous bytes at any tap shift.
// PHASE (MW = 4 only; round 4): the weight gradient of a conv behind a nearest-2x upsample, one output-pixel parity (py, px) per
//         launch.  Output pixel (2r + py, 2c + px) reads the input pixels (r + py - 1 + a, c + px - 1 + b), a, b in {0, 1}, under the
//         filter rows / columns that fold onto them (ky -> a = (ky + 1 - py) >> 1), so
//             dW[ky][kx] = sum over the four parities of  G[a(ky)][b(kx)],   G[a][b] = sum_pixels dY(2r + py, 2c + px) x(r + py - 1 + a, ...)
//         - four 2 x 2-tap gradients on the LOW-resolution map (p.H x p.W = the input's; dY is read with stride 2) instead of one
//         3 x 3-tap gradient on the upsampled one: 16 instead of 36 multiply steps per input pixel, as forward and data gradient
//         already run.  Every launch writes a complete 9-tap slab (each G into the taps it feeds); the four parities of a slab are
//         consecutive rows of `partial` (slab_mul = 4), which the reduction sums like any other slabs.
// PSPLIT (MW = 4, BI = 48; round 6): 48 output channels are three 16-channel blocks, which do not split over two wave halves - the
//         64-column tile ran the 48-channel layers (up_g4.last_conv_3x3 at 128 x 128, the fourth ConvGRU, the context stack's second
//         block, the temporal discriminator's 3-D blocks) with a quarter of its MFMAs on padding (PMC, round 5: 27.6 % pipe busy, the
//         third-largest kernel of the step).  Here the four matrix waves split the PIXELS instead: wave (ci half, pixel half) owns 16
//         input channels x all 48 output channels x 9 taps = 27 accumulators (the BI = 96 count) and multiplies ONE of the tile's
//         two 32-pixel steps - 81 instead of 108 MFMAs per wave and tile, none on padding.  The two pixel halves of a slab meet once,
//         after the last tile, through LDS (fixed order: half 0 + half 1).
template <int BI, int NS, int TWS, int MW = 3, bool PHASE = false, bool PSPLIT = false>
__global__ __launch_bounds__(MW * 64 + 256) void conv_wgrad_ws_kernel(const dgmr_wgrad_args p, const int tiles_w, const int tiles_hw,
                                                           const int tiles_per_split, const int splits_per_group,
                                                           const int tiles_per_group, const int dbg, const int kd, const int ph = 0,
                                                           const int slab_mul = 1) {
    static_assert(!PHASE || MW == 4, "phase mode: four matrix waves");
    const int ppy = PHASE ? ph >> 1 : 0, ppx = PHASE ? ph & 1 : 0;
    constexpr int NP = planes_of<NS>::value;
    constexpr int CK = 32, CB = BI / 32;
    constexpr int TW = 1 << TWS, TH = 64 >> TWS;      // 32 x 2 or 16 x 4 pixels
    constexpr int HTW = TW + 2, HTH = TH + 2;          // halo: columns w0-1 .. w0+TW, rows h0-1 .. h0+TH
    constexpr int HPIX = HTH * HTW;                    // 136 / 108 pixels
    constexpr int XROW = CK / 2;                       // dwords per halo pixel: 32 bf16
    constexpr int YROW = 48;                           // dwords per dY pixel: BI bf16 (64: padded to 192 B): the 4 rows of a transposing
                                                       // read's 32-lane group then start at banks 0 / 48 / 32 / 16 (x 16 banks each)
    constexpr int XPL = HPIX * XROW, YPL = MW == 4 ? 64 * (BI / 2) : 64 * YROW;  // dwords per plane
    constexpr int BUF = NP * (XPL + YPL);              // dwords per tile image
    constexpr int NL = 256;                            // loader threads (4 waves)
    constexpr int XITEMS = HPIX * 8, YQ = BI / 4, YITEMS = 64 * YQ;  // 16-byte fp32 items of a tile: [pixel][4-channel quad]
    constexpr int XP = (XITEMS + NL - 1) / NL, YP = YITEMS / NL;
    // a loader thread's dY items i, i + BSN, ... have the same channel quad (items advance by NL % YQ quads): 3 (BI 96) / 1 (BI 64)
    constexpr int BSN = NL % YQ == 0 ? 1 : YQ / ws_gcd(NL % YQ, YQ);
    static_assert(BI == 64 || BI == 96 || (BI == 48 && PSPLIT), "BI");
    static_assert(!PSPLIT || (MW == 4 && !PHASE && BI == 48 && NP <= 2), "pixel-split tile");
    constexpr int NACC4 = 9 * 3 * 4;                   // accumulator registers of a PSPLIT matrix wave
    constexpr int RED_DW = ((NL % (BI / 4) == 0 ? 1 : (BI / 4) / ws_gcd(NL % (BI / 4), BI / 4)) * NL * 4);  // dwords of the bias sums (`red`)
    constexpr int EXCH_DW = PSPLIT ? 2 * NACC4 * 64 : 0;  // the pixel halves' exchange: [ci half][register][lane]
    static_assert(TWS == 5 || TWS == 4, "tile width");
    static_assert(YITEMS % NL == 0 && YP % BSN == 0, "dY items per loader thread");

    __shared__ __attribute__((aligned(16))) uint32_t smem[(2 * BUF > RED_DW + EXCH_DW) ? 2 * BUF : RED_DW + EXCH_DW];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: the role branch and the loops below are uniform control flow)
    // workgroups go to the 8 XCDs round-robin by linear index, each XCD with its own L2: give every XCD a CONTIGUOUS range of logical
    // workgroups, so that the input-channel chunks (which all read the slab's dY) and the output-channel tiles (which all read its x)
    // of one slab run side by side under one L2 instead of fetching the slab from HBM once per XCD
    const int n_chunks = (p.Cin + CK - 1) / CK, n_cot = (p.Cout + BI - 1) / BI;
    const int total = gridDim.x, xcd = blockIdx.x & 7, per_xcd = total >> 3, rem = total & 7;
    const int wg = xcd * per_xcd + min(xcd, rem) + (blockIdx.x >> 3);
    const int chunk = wg % n_chunks, wg2 = wg / n_chunks;
    const int slab = wg2 / n_cot, co0 = (wg2 - slab * n_cot) * BI;
    const int grp = slab / splits_per_group;
    const int t_begin = grp * tiles_per_group + (slab - grp * splits_per_group) * tiles_per_split;
    const int t_end = min((grp + 1) * tiles_per_group, t_begin + tiles_per_split);
    const int nt = max(t_end - t_begin, 0);  // (0: a slab plan with more slabs than tiles - that slab's partial sums are zeros)
    const bool want_bias = p.bias_grad && chunk == 0;  // (uniform per workgroup)
    float* red = reinterpret_cast<float*>(smem);       // [slot][loader thread][4]: the loaders' bias sums, after the last tile

    if (wid >= MW) {
        // ================================================ loaders ================================================
        const int lt = tid - MW * 64;  // 0 .. 255
        const int xq = lt & 7;     // this thread's 4-channel quad of the 32-channel chunk: the same for all its X items
        const int xci = chunk * CK + xq * 4;
        const bool xc_ok = xci < p.Cin;
        const int us = (!PHASE && p.upsample) ? 1 : 0;
        const int Hs = p.H >> us, Ws = p.W >> us;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        // per-thread constants: every item's element offset from its tile's origin (the tile origin is a multiple of the tile size, so
        // floor((origin - 1 + r) / 2) of the fused x2 upsampling splits into origin / 2 + ((r - 1) >> 1)) and its edge membership
        // (bit i = item i is valid at all / lies in the halo's top / bottom / left / right edge; y_ok: co < Cout)
        int xrel[XP];
        uint32_t yrel[YP];
        unsigned m_all = 0, m_top = 0, m_bot = 0, m_left = 0, m_right = 0, y_ok = 0;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int idx = lt + i * NL;
            const int pix = idx >> 3;
            const int pr = pix / HTW, pc = pix - pr * HTW;  // (compile-time divisor)
            xrel[i] = (((pr - 1) >> us) * Ws + ((pc - 1) >> us)) * p.Cin + xci;
            const unsigned bit = 1u << i;
            m_all |= (idx < XITEMS && xc_ok) ? bit : 0u;
            m_top |= pr == 0 ? bit : 0u;
            m_bot |= pr == HTH - 1 ? bit : 0u;
            m_left |= pc == 0 ? bit : 0u;
            m_right |= pc == HTW - 1 ? bit : 0u;
        }
#pragma unroll
        for (int i = 0; i < YP; ++i) {
            const int idx = lt + i * NL;
            const int pix = idx / YQ, q = idx - pix * YQ;  // (compile-time divisor)
            const int co = co0 + q * 4;
            yrel[i] = PHASE ? (uint32_t)((pix >> TWS) * 4 * p.W + (pix & (TW - 1)) * 2) * p.Cout + min(co, p.Cout - 4)  // dY: 2 p.H x 2 p.W, stride 2
                            : (uint32_t)((pix >> TWS) * p.W + (pix & (TW - 1))) * p.Cout + min(co, p.Cout - 4);
            y_ok |= co < p.Cout ? 1u << i : 0u;
        }
        // position of the next tile to fetch (tiles are fetched in order: stepped, not divided) and how many lie beyond it
        int cn = t_begin / tiles_hw, ch0, cw0, left = nt - 1;  // (cn: image = sample x depth plane of a 3-D conv)
        {
            const int trem = t_begin - cn * tiles_hw;
            const int th = trem / tiles_w;
            ch0 = th * TH;
            cw0 = (trem - th * tiles_w) << TWS;
        }
        // 3 x 3 x 3 convs (the temporal discriminator's first blocks): one launch per depth tap kd.  dY plane d meets x plane d + kd - 1
        // of the same sample - a constant shift of the image index - and contributes nothing where that plane lies outside the volume
        const int Dp = p.D, dshift = p.KD == 3 ? kd - 1 : 0;
        int cd = cn % Dp;  // depth plane of image cn
        // prologue as one expression, max(a x + b, floor): a = 1, b = 0 where there is no BatchNorm, floor = -inf where there is no ReLU
        const float relu_floor = (p.pre_a || p.pre_relu) ? 0.f : -__builtin_inff();
        // two staging sets (tile parity): the tile's X and dY items, its BatchNorm scale / shift, the validity bits of its X items
        f32x4 sx[2][XP], sy[2][YP], sa[2], sb[2], bsum[BSN];
        unsigned xm[2] = {0, 0};
#pragma unroll
        for (int e = 0; e < 2; ++e) sa[e] = (f32x4){1.f, 1.f, 1.f, 1.f}, sb[e] = zero4;
#pragma unroll
        for (int i = 0; i < BSN; ++i) bsum[i] = zero4;

        // issue the global loads of the next tile into staging set E; every load unconditional on a clamped address - and every CALL
        // unconditional: the compiler counts the loads in flight per control-flow path and waits for the minimum over all paths, so
        // one `if (more tiles) issue()` makes every store wait for vmcnt(0), i.e. for the set issued last as well (the prefetch depth
        // of 2 then behaves as 1).  Past the slab's last tile the position stays and that tile is fetched again, into a set that is
        // never stored
        auto issue = [&](auto set_c) {
            constexpr int E = decltype(set_c)::value;
            const uint32_t xb = (((uint32_t)(cn + dshift) * Hs + (ch0 >> us)) * Ws + (cw0 >> us)) * p.Cin;  // (wraps for image -1: masked)
            const bool plane_ok = (unsigned)(cd + dshift) < (unsigned)Dp;
            const unsigned kill = (ch0 == 0 ? m_top : 0u) | (ch0 + TH >= p.H ? m_bot : 0u) | (cw0 == 0 ? m_left : 0u) | (cw0 + TW >= p.W ? m_right : 0u);
            const unsigned m = plane_ok ? m_all & ~kill : 0u;
            xm[E] = m;
#pragma unroll
            for (int i = 0; i < XP; ++i) sx[E][i] = *reinterpret_cast<const f32x4*>(p.x + (((m >> i) & 1u) ? xb + (uint32_t)xrel[i] : 0u));
            if (p.pre_a) {
                const uint32_t g = xc_ok ? (uint32_t)(cn / Dp / p.pre_group) * p.Cin + xci : 0u;
                sa[E] = *reinterpret_cast<const f32x4*>(p.pre_a + g);
                sb[E] = *reinterpret_cast<const f32x4*>(p.pre_b + g);
            }
            const float* yb = PHASE ? p.dy + (((size_t)cn * 2 * p.H + 2 * ch0 + ppy) * 2 * p.W + 2 * cw0 + ppx) * p.Cout
                                    : p.dy + (((size_t)cn * p.H + ch0) * p.W + cw0) * p.Cout;
#pragma unroll
            for (int i = 0; i < YP; ++i) sy[E][i] = *reinterpret_cast<const f32x4*>(yb + yrel[i]);
            if (left > 0) {
                --left;
                cw0 += TW;
                if (cw0 >= p.W) {
                    cw0 = 0;
                    ch0 += TH;
                    if (ch0 >= p.H) {
                        ch0 = 0;
                        ++cn;
                        cd = cd + 1 == Dp ? 0 : cd + 1;
                    }
                }
            }
        };
        // prologue, split into planes, lane-linear 8-byte LDS stores: X[plane][halo pixel][32 ci], dY[plane][pixel][BI co]
        // (tile parity = staging set = LDS image; !live: past the slab's end, nobody reads what is stored)
        auto store = [&](auto set_c, bool live) {
            constexpr int E = decltype(set_c)::value;
            // MW = 3: X[pixel][32 ci] = item * 2 dwords;  MW = 4: X[ci half][pixel][16 ci], dY[co block of 16][pixel][16 co]
            uint32_t* Xs = smem + E * BUF + (MW == 4 ? (xq >> 2) * (HPIX * 8) + (lt >> 3) * 8 + (xq & 3) * 2 : lt * 2);
            uint32_t* Ys = smem + E * BUF + NP * XPL;
            const unsigned m = xm[E];
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                if (i == XP - 1 && XITEMS % NL != 0 && lt + i * NL >= XITEMS) continue;
                f32x4 v = sx[E][i];
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = fmaxf(fmaf(v[c], sa[E][c], sb[E][c]), relu_floor);
                v = ((m >> i) & 1u) ? v : zero4;
                u32x2 pl[NP];
                split_planes4<NP>(v, pl);
#pragma unroll
                for (int q = 0; q < NP; ++q) *reinterpret_cast<u32x2*>(Xs + q * XPL + i * (MW == 4 ? NL : NL * 2)) = pl[q];  // (NL items = 32 pixels further)
            }
#pragma unroll
            for (int i = 0; i < YP; ++i) {
                const int idx = lt + i * NL;
                const int pix = idx / YQ, q = idx - pix * YQ;
                const f32x4 v = ((y_ok >> i) & 1u) ? sy[E][i] : zero4;
                if (want_bias && live) bsum[i % BSN] += v;  // (item i + BSN of this thread has the same channel quad)
                u32x2 pl[NP];
                split_planes4<NP>(v, pl);
#pragma unroll
                for (int k = 0; k < NP; ++k)
                    *reinterpret_cast<u32x2*>(Ys + k * YPL + (MW == 4 ? (q >> 2) * (64 * 8) + pix * 8 + (q & 3) * 2 : pix * YROW + q * 2)) = pl[k];
            }
        };
        // iteration t: the matrix waves multiply image t & 1 while the loaders store tile t + 1 (its loads were issued TWO matrix
        // phases earlier, into staging set (t + 1) & 1) into the other image and issue the loads of tile t + 3 into the set just freed
        using set0 = std::integral_constant<int, 0>;
        using set1 = std::integral_constant<int, 1>;
        if (nt > 0) {  // (the loop INSIDE the condition: a path that reaches it with no loads in flight would make its waits conservative)
            issue(set0{});
            issue(set1{});
            store(set0{}, true);
            issue(set0{});
            __syncthreads();
            // (tiles in pairs with ONE loop exit: a `break` between the halves becomes a second edge into the loop header in the
            //  structurised control flow, with the staging sets in the opposite order - and the waits turn conservative again; with an
            //  odd tile count the last half-iteration stores and fetches for nobody)
            for (int t = 0; t < nt; t += 2) {
                store(set1{}, t + 1 < nt);
                issue(set1{});
                __syncthreads();
                store(set0{}, t + 2 < nt);
                issue(set0{});
                __syncthreads();
            }
        } else {
            __syncthreads();
        }
        // ---- bias gradient (first input-channel chunk only): every loader thread's column sums of dY go to LDS (the tile images are
        //      free now); the matrix waves add them up in a fixed order ----
        if (want_bias) {
#pragma unroll
            for (int i = 0; i < BSN; ++i) *reinterpret_cast<f32x4*>(red + (i * NL + lt) * 4) = bsum[i];
        }
        if (want_bias || PSPLIT) __syncthreads();  // (PSPLIT: the matrix waves' pixel halves meet at this barrier too)
    } else {
        // ============================================= matrix waves =============================================
        if constexpr (MW == 4) {
            constexpr int CBH = PSPLIT ? 3 : BI / 32;  // 16-channel output blocks of this wave (PSPLIT: all three; else its half's)
            const int cih = wid & 1, coh = PSPLIT ? 0 : wid >> 1;
            const int psh = PSPLIT ? wid >> 1 : 0;  // PSPLIT: which 32-pixel step of every tile this wave multiplies
            constexpr int NT = PHASE ? 4 : 9;  // taps a wave accumulates
            f32x4 acc[NT][CBH];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int c = 0; c < CBH; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // a 32-pixel step of the reduction: operand lane l holds k = 8 (l >> 4) ... + 7 for row / column l & 15; the two transposing
            // reads of a fragment deliver k 0..3 and 4..7 of each 16-lane group g.  Which PIXEL a k stands for is free as long as both
            // operands agree: pixel = 16 (g >> 1) + 8 rd + 4 (g & 1) + j, so that the lanes 0 - 31 of one read touch 8 consecutive pixels
            const int tg = lane >> 4, ti = lane & 15;
            const int py_lane = 16 * (tg >> 1) + 4 * (tg & 1) + (ti >> 2);
            const int px_lane = (TWS == 5 ? 16 * (tg >> 1) : (tg >> 1) * HTW) + 4 * (tg & 1) + (ti >> 2);  // (16-wide tiles: 2 rows per step)
            const int seg = (ti & 3) * 2;
            auto compute = [&](int buf) {
                const uint32_t* Xs = smem + buf * BUF + cih * (HPIX * 8) + px_lane * 8 + seg;
                const uint32_t* Ys = smem + buf * BUF + NP * XPL + coh * CBH * (64 * 8) + py_lane * 8 + seg;
                auto fetch_x = [&](int ks, int pl, bf16x8_t (&dst)[NT]) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        // halo row / column of tap t: the 3 x 3 taps, or (phase) the 2 x 2 window that starts at row ppy, column ppx
                        const int hp = PHASE ? ((TWS == 5 ? ks : 2 * ks) + ppy + (t >> 1)) * HTW + ppx + (t & 1)
                                             : ((TWS == 5 ? ks : 2 * ks) + t / 3) * HTW + t % 3;
                        dst[t] = tr_fragment(Xs + pl * XPL + hp * 8, 8 * 8);
                    }
                };
                auto fetch_y = [&](int ks, int pl, bf16x8_t (&dst)[CBH]) {
#pragma unroll
                    for (int c = 0; c < CBH; ++c) dst[c] = tr_fragment(Ys + pl * YPL + c * (64 * 8) + ks * 32 * 8, 8 * 8);
                };
                auto mm = [&](const bf16x8_t (&y)[CBH], const bf16x8_t (&x)[NT]) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int c = 0; c < CBH; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y[c], x[t], acc[t][c], 0, 0, 0);
                };
                constexpr int M = NT * CBH, RX = 2 * NT, RY = CBH * 2, RX0 = PHASE ? 4 : 6;
                if constexpr (PSPLIT) {
                    // one 32-pixel step per wave: hi.lo + lo.hi + hi.hi with the second product's fragments read in the first one's shadow
                    if constexpr (NP == 2) {
                        bf16x8_t y0[CBH], y1[CBH], x0[NT], x1[NT];
                        fetch_y(psh, 0, y0), fetch_x(psh, 1, x1);
                        __builtin_amdgcn_sched_group_barrier(0x100, RY + RX0, 0);
                        fetch_y(psh, 1, y1), fetch_x(psh, 0, x0);
                        mm(y0, x1);  // hi(dY) . lo(x)
                        ws_interleave<RX - RX0 + RX + RY, M, 0>();
                        mm(y1, x0);  // lo(dY) . hi(x)
                        __builtin_amdgcn_sched_group_barrier(0x008, M, 0);
                        mm(y0, x0);  // hi . hi
                        __builtin_amdgcn_sched_group_barrier(0x008, M, 0);
                    } else {
                        bf16x8_t xf[NT], yf[CBH];
                        fetch_y(psh, 0, yf), fetch_x(psh, 0, xf);
                        mm(yf, xf);
                    }
                } else if constexpr (NP == 2) {  // (product order and fragment rotation as in the three-wave variant below)
                    bf16x8_t y0[2][CBH], x0[NT], x1[NT], y1[CBH];
                    fetch_y(0, 0, y0[0]), fetch_x(0, 1, x1);
                    __builtin_amdgcn_sched_group_barrier(0x100, RY + RX0, 0);  // dY and the first filter row of x: the first MFMAs can start
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const int e = ks & 1;
                        fetch_y(ks, 1, y1), fetch_x(ks, 0, x0);
                        mm(y0[e], x1);  // hi(dY) . lo(x)
                        if (ks == 0) ws_interleave<RX - RX0 + RX + RY, M, 0>();
                        else ws_interleave<RX + RY, M, 0>();
                        if (ks < 1) fetch_x(ks + 1, 1, x1), fetch_y(ks + 1, 0, y0[e ^ 1]);
                        mm(y1, x0);  // lo(dY) . hi(x)
                        if (ks < 1) ws_interleave<RX + RY, M, 0>();
                        else __builtin_amdgcn_sched_group_barrier(0x008, M, 0);
                        mm(y0[e], x0);  // hi . hi
                        __builtin_amdgcn_sched_group_barrier(0x008, M, 0);
                    }
                } else {
                    static_assert(NP == 1, "four matrix waves: bf16 and bf16x3 only (bf16x6 needs 144 fragment registers)");
                    bf16x8_t xf[2][NT], yf[2][CBH];
                    fetch_y(0, 0, yf[0]), fetch_x(0, 0, xf[0]);
                    fetch_y(1, 0, yf[1]), fetch_x(1, 0, xf[1]);
                    mm(yf[0], xf[0]);
                    mm(yf[1], xf[1]);
                }
            };
            __builtin_amdgcn_s_setprio(1);
            __syncthreads();
            for (int t = 0; t < nt; t += 2) {
                if (!(dbg & 2)) compute(0);
                __syncthreads();
                if (t + 1 < nt && !(dbg & 2)) compute(1);
                __syncthreads();
            }
            if constexpr (PSPLIT) {
                // the slab's two pixel halves: half 1 hands its accumulators over through LDS (the tile images are free now; behind `red`),
                // half 0 adds them - a fixed order - and writes the slab
                float* ex = reinterpret_cast<float*>(smem) + RED_DW + cih * (NACC4 * 64) + lane;
                if (psh == 1) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int c = 0; c < CBH; ++c)
#pragma unroll
                            for (int r = 0; r < 4; ++r) ex[((t * CBH + c) * 4 + r) * 64] = acc[t][c][r];
                }
                __syncthreads();  // (the loaders' bias sums are published by the same barrier)
                if (psh == 0) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int c = 0; c < CBH; ++c)
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[t][c][r] += ex[((t * CBH + c) * 4 + r) * 64];
                }
            }
            // ---- partial[slab][co][tap*Cin + ci]: lane = input channel (16 per wave), 4 output channels per lane and block ----
            const int Ktot = 9 * p.KD * p.Cin, tap0 = (p.KD == 3 ? kd : 0) * 9;
            float* out = p.partial + ((size_t)slab * slab_mul + (PHASE ? ph : 0)) * p.Cout * Ktot;
            const int ci = chunk * CK + cih * 16 + (lane & 15);
            if (ci < p.Cin && psh == 0) {
#pragma unroll
                for (int c = 0; c < CBH; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = co0 + (coh * CBH + c) * 16 + 4 * (lane >> 4) + r;
                        if (co >= p.Cout) continue;
#pragma unroll
                        for (int t = 0; t < 9; ++t) {
                            float v;
                            if constexpr (PHASE) {  // filter tap (ky, kx) is fed by window tap (a, b) = ((ky + 1 - py) >> 1, (kx + 1 - px) >> 1)
                                const int a = (t / 3 + 1 - ppy) >> 1, b = (t % 3 + 1 - ppx) >> 1;
                                const float va0 = b ? acc[1][c][r] : acc[0][c][r], va1 = b ? acc[3][c][r] : acc[2][c][r];
                                v = a ? va1 : va0;
                            } else {
                                v = acc[t][c][r];
                            }
                            out[(size_t)co * Ktot + (tap0 + t) * p.Cin + ci] = v;
                        }
                    }
            }
        } else {
            f32x16 acc[CB * 3];
    #pragma unroll
            for (int a = 0; a < CB * 3; ++a)
    #pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
            // transposing-read geometry of this lane: 16-lane group g, lane i of it supplies row (i >> 2) of the 4 x 16 block, segment (i & 3)
            const int tg = lane >> 4, ti = lane & 15;
            const int k_lane = 8 * (tg >> 1) + (ti >> 2);      // pixel (inside a 16-pixel step) whose row this lane addresses for k 0..3
            const int col_lane = 16 * (tg & 1) + 4 * (ti & 3);  // first of the 4 channels (inside a 32-channel block) it addresses
            auto compute = [&](int buf) {
                const uint32_t* Xs = smem + buf * BUF;
                const uint32_t* Ys = Xs + NP * XPL;
                // 16-pixel step kk = tile pixels [16 kk, 16 kk + 16): tile row (16 kk) >> TWS, first column (16 kk) & (TW - 1); filter row
                // wid reads halo row (tile row + wid), tap d halo column (column + d)
                auto fetch_x = [&](int kk, int pl, bf16x8_t (&dst)[3]) {
                    const int hp0 = (((kk * 16) >> TWS) + wid) * HTW + ((kk * 16) & (TW - 1));
                    const uint32_t* xb = Xs + pl * XPL + (hp0 + k_lane) * XROW + (col_lane >> 1);
    #pragma unroll
                    for (int d = 0; d < 3; ++d) dst[d] = tr_fragment(xb + d * XROW, 4 * XROW);
                };
                auto fetch_y = [&](int kk, int pl, bf16x8_t (&dst)[CB]) {
                    const uint32_t* yb = Ys + pl * YPL + (kk * 16 + k_lane) * YROW + (col_lane >> 1);
    #pragma unroll
                    for (int c = 0; c < CB; ++c) dst[c] = tr_fragment(yb + c * 16, 4 * YROW);
                };
                auto mm = [&](const bf16x8_t (&y)[CB], const bf16x8_t (&x)[3]) {
    #pragma unroll
                    for (int c = 0; c < CB; ++c)
    #pragma unroll
                        for (int d = 0; d < 3; ++d) acc[c * 3 + d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y[c], x[d], acc[c * 3 + d], 0, 0, 0);
                };
                // instruction order asked of the scheduler for one product with `reads` transposing LDS reads to hide: the reads spread
                // over the product's MFMAs instead of clustered before them.  A wave issues in order and an MFMA occupies the matrix pipe
                // for 32 cycles, so a read placed between two MFMAs issues in the first one's shadow; a cluster of reads ahead of the
                // MFMAs is paid in full - and ONE wave per SIMD gets 8-byte LDS reads out at a fraction of the LDS rate (~10 cycles
                // each, MI355X_MICROARCH.md, LDS), which is what the matrix waves have
                auto interleave = [&](auto reads_c) { ws_interleave<decltype(reads_c)::value, CB * 3, 0>(); };
                if constexpr (NP == 2) {
                    // bf16x3 = three products hi.lo + lo.hi + hi.hi per step.  Every fragment group is read ONE product before the product
                    // that needs it, into registers whose last reader has just finished - 5 fragment groups (60 registers) live at the
                    // peak instead of the 8 of two whole sets, beside the 144 accumulator registers:
                    //   product      operands            reads in its shadow
                    //   hi.lo        y0[e], x1           y1, x0       (this step's other two products)
                    //   lo.hi        y1,    x0           x1', y0[e^1] (the next step's first product; x1 is free since hi.lo)
                    //   hi.hi        y0[e], x0           -
                    constexpr int RX = 3 * 2, RY = CB * 2;  // LDS reads per X / dY fragment group
                    using R = std::integral_constant<int, RX + RY>;
                    bf16x8_t y0[2][CB], x0[3], x1[3], y1[CB];
                    fetch_y(0, 0, y0[0]), fetch_x(0, 1, x1);
                    __builtin_amdgcn_sched_group_barrier(0x100, RX + RY, 0);
    #pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const int e = kk & 1;
                        fetch_y(kk, 1, y1), fetch_x(kk, 0, x0);
                        mm(y0[e], x1);  // hi(dY) . lo(x)
                        interleave(R{});
                        if (kk < 3) fetch_x(kk + 1, 1, x1), fetch_y(kk + 1, 0, y0[e ^ 1]);
                        mm(y1, x0);  // lo(dY) . hi(x)
                        if (kk < 3) interleave(R{});
                        else __builtin_amdgcn_sched_group_barrier(0x008, CB * 3, 0);
                        mm(y0[e], x0);  // hi . hi
                        __builtin_amdgcn_sched_group_barrier(0x008, CB * 3, 0);
                    }
                } else if constexpr (NP == 1) {
                    bf16x8_t xf[2][3], yf[2][CB];
                    fetch_x(0, 0, xf[0]), fetch_y(0, 0, yf[0]);
    #pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        if (kk < 3) fetch_x(kk + 1, 0, xf[(kk + 1) & 1]), fetch_y(kk + 1, 0, yf[(kk + 1) & 1]);
                        mm(yf[kk & 1], xf[kk & 1]);
                    }
                } else {  // bf16x6: six products per step (54 MFMAs) behind one fragment set of 72 registers
    #pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        bf16x8_t xf[NP][3], yf[NP][CB];
    #pragma unroll
                        for (int pl = 0; pl < NP; ++pl) fetch_x(kk, pl, xf[pl]), fetch_y(kk, pl, yf[pl]);
                        for_each_product<NP>([&](auto qa, auto qb) { mm(yf[qa], xf[qb]); });
                    }
                }
            };
            __builtin_amdgcn_s_setprio(1);  // (in the step, beside the other streams' kernels: no difference with or without, 990.5 / 987.5 vs 990.3 / 992.8 ms)
            __syncthreads();
            for (int t = 0; t < nt; t += 2) {
                if (!(dbg & 2)) compute(0);  // (dgmr_debug_flags 16 -> dbg 2: timing probe without the matrix work)
                __syncthreads();
                if (t + 1 < nt && !(dbg & 2)) compute(1);
                __syncthreads();
            }
            // ---- partial[slab][co][tap*Cin + ci]: lane = input channel, 16 output channels per MFMA block ----
            const int Ktot = 9 * p.KD * p.Cin, tap0 = (p.KD == 3 ? kd : 0) * 9;  // K order (kd, kh, kw, ci)
            float* out = p.partial + (size_t)slab * p.Cout * Ktot;
            const int ci = chunk * CK + (lane & 31);
            if (ci < p.Cin) {
    #pragma unroll
                for (int c = 0; c < CB; ++c)
    #pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = co0 + c * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        if (co >= p.Cout) continue;
    #pragma unroll
                        for (int d = 0; d < 3; ++d) out[(size_t)co * Ktot + (tap0 + wid * 3 + d) * p.Cin + ci] = acc[c * 3 + d][r];
                    }
            }
        }
        if (want_bias) {  // (deterministic within the workgroup; across workgroups one float atomic per channel - or, with
                          //  dgmr_wgrad_args.bias_partial, a row of its own per slab that dgmr_conv_wgrad adds up in order)
            if (!PSPLIT) __syncthreads();  // the loaders have written their sums: slot i of thread lt covers channel quad (lt + i NL) mod YQ
            if (tid < BI && co0 + tid < p.Cout) {
                const int q = tid >> 2, comp = tid & 3;
                float total = 0.f;
#pragma unroll
                for (int i = 0; i < BSN; ++i)
                    for (int l = ((q - i * NL) % YQ + YQ) % YQ; l < NL; l += YQ) total += red[(i * NL + l) * 4 + comp];
                atomicAdd(p.bias_grad + (size_t)slab * p.bias_stride + co0 + tid, total);  // (bias_stride = Cout: a row per slab, ONE writer per element)
            }
        }
    }
}

}  // namespace
