// 3x3 convolution, forward and data gradient, WAVE-SPECIALISED and persistent (round 4).
//
// conv3x3_glds_kernel (conv_win_glds.h) gives every wave every job: stage the halo of a 32-channel chunk (global loads, BatchNorm /
// ReLU prologue, bf16 split, ds_write), multiply nine taps, run the epilogue - and relies on two or three co-resident workgroups
// to overlap them.  Measured (profiles/r03_probe_window_phases_*): the matrix loop alone runs at the rate the part sustains under
// its power limit, a whole launch at 70 - 75 % of it; staging is 7 - 15 % of a launch, the epilogue 12 - 38 %.  Here the jobs belong
// to different waves of ONE workgroup per CU that walks a contiguous run of (tile, phase, output-channel block) items:
//   * 4 x WN MATRIX waves (two per SIMD at WN = 2): LDS-DMA of the next tap's weights, fragment reads, MFMAs - nothing else.  At
//     the last tap of an item they park the accumulators in LDS (lane-linear 16-byte stores) and start the next item at once.
//   * 4 LOADER waves (one per SIMD): while chunk s is multiplied they write the halo of chunk s + 1 into the other A buffer - from
//     registers loaded earlier (nine-tap steps: one register set, loaded at the first tap and stored from the third on; four-tap
//     steps: two sets, each loaded a whole step ahead) - and run the EPILOGUE of the PREVIOUS item out of the parked accumulators,
//     in the accumulators' own layout (lane = output channel: per-column constants are one value per lane, the BatchNorm sums are
//     in-lane adds plus one or two lane exchanges; only the result is transposed across the quad for the 16-byte store).  Its fused
//     operand (residual or relu-mask source) arrives through a small ring of registers loaded D taps ahead, per-column constants
//     through a parameter block in LDS, so that no loader instruction waits for a load it has just issued (every load is
//     unconditional on a clamped address and goes through an explicit global address space - a pointer SELECTED among kernel
//     arguments turns into flat_load, which is waited for with vmcnt(0): exact s_waitcnt counts, the lesson of wgrad_ws.h).
//   * The hardware has ONE barrier per workgroup, and a weight stage shared by the matrix waves needs one per tap: all waves meet
//     at every tap.  The loaders' work is therefore cut into per-tap slices (a few halo items, UPS epilogue units, one ring load);
//     both roles are disjoint programs with the same loop skeleton and so the same barrier count by construction.
// Arithmetic per output element is the glds kernel's (same MFMA sequence, same epilogue expressions): y is bit-identical to
// conv3x3_glds_kernel<BN, 4, 1, NS, 128, false, M16> (dgmr_conv_tune window = 6; tests/test_gpu_kernels.py, tools/ws_check.py); the
// BatchNorm partial sums are added in another order (1e-8 apart).
// Scope: 2-D 3x3 (plain, phase, pooled modes), epi_mode PLAIN with the 16-byte epilogue, at most one fused epilogue operand.
// Status (round 4): correct, faster than the one-role kernels on launches of at most one workgroup per CU, 10 - 25 %
// slower than the 256-pixel-tile one-role kernel on the big launches - opt-in (DGMR_WS_AUTO=1 / dgmr_conv_tune window = 7).
#pragma once
#include "conv_win_glds.h"

namespace {

struct WsGeom {
    int tile, ph, n0, n, h0, w0;
};

// Loads through a pointer that was SELECTED among kernel arguments (residual or mask source, bias or mask_a or ...): hipcc loses
// the address space in the select and emits flat_load, which counts on vmcnt AND lgkmcnt and is waited for with vmcnt(0) - every
// counted wait of the loaders' pipeline collapses.  The explicit global address space keeps them global_load.
typedef __attribute__((address_space(1))) const float ws_gfloat;
typedef __attribute__((address_space(1))) const f32x4 ws_gfloat4;
__device__ __forceinline__ float ws_gload(const float* p, uint32_t off) { return ((ws_gfloat*)p)[off]; }
__device__ __forceinline__ f32x4 ws_gload4(const float* p, uint32_t off) { return *((ws_gfloat4*)(p + off)); }

// MODE 0: plain 3x3 (9 taps per 32-channel chunk)   1: phase (forward of an upsampling conv: 4 taps, item = one output-pixel parity)
//      2: pooled (its data gradient: four parity planes x chunks, 4 taps each)
template <int MODE>
struct ws_mode {
    static constexpr int T = MODE == 0 ? 9 : 4;       // taps (= barriers) per step
    // halo register sets of a loader thread = steps covered by the loaders' statically unrolled loop body.  Nine taps: ONE set, loaded
    // at the first tap of a step and stored from its third tap on (two taps for the loads to land, then one item per tap); four
    // taps: TWO sets, each loaded a whole step before it is stored.  (Every slice of the body is its own copy of the epilogue code:
    // 18 copies for two 9-tap steps would be 70 KB of instructions.)
    static constexpr int NSETS = MODE == 0 ? 1 : 2;
    static constexpr int RING = MODE == 0 ? 9 : 4;    // epilogue-operand ring: slots (must divide NSETS * T)
    static constexpr int D = RING - 1;                // a unit's operand is loaded D taps before the unit runs
    static constexpr int UPSMAX = MODE == 1 ? 2 : 1;  // epilogue units per tap, at most (a phase item has as few as 8 taps)
};

template <int BN, int WN, int NS, bool M16, bool EOP, int MODE>
__global__ __launch_bounds__(64 * (4 * WN + 4)) void conv3x3_ws_kernel(const dgmr_conv_args p, const int tw_shift, const int tiles_w,
                                                                        const int tiles_hw, const int g_shift, const int n_tiles,
                                                                        const int n_nb, const int ups) {
    constexpr int BM = 128, WM = 4, CK = 32, LOG_BM = 7, ROW = CK / 2;
    constexpr int NP = planes_of<NS>::value;
    constexpr int MB = M16 ? 16 : 32, RPB = M16 ? 4 : 16, NG = RPB / 4;
    constexpr int TM = BM / WM / MB, TN = BN / WN / MB;
    constexpr int NMW = WM * WN, NMT = 64 * NMW;           // matrix waves / threads
    constexpr int AMAX = 6 * 34, APASS = (AMAX * 8 + 255) / 256;
    constexpr int BUNITS = BN * 4 * NP, BPASS = (BUNITS + NMT - 1) / NMT;
    constexpr int BSTAGE = NP * BN * ROW, ASIZE = NP * AMAX * ROW;  // dwords
    constexpr int GSZ = TM * NG;                           // epilogue units per (wn, column block): the rows one statistics sum covers
    constexpr int NU = WN * TN * GSZ;                      // epilogue units (one f32x4 per lane each) per loader wave and item
    constexpr int T = ws_mode<MODE>::T, RING = ws_mode<MODE>::RING, D = ws_mode<MODE>::D, NSETS = ws_mode<MODE>::NSETS;
    constexpr int UPSMAX = ws_mode<MODE>::UPSMAX;
    constexpr int CSIZE = BM * BN, REDSIZE = WM * BN * 2, PARSIZE = 3 * BN + 4;
    static_assert(TM >= 1 && TN >= 1 && BUNITS % 64 == 0 && (NSETS * T) % RING == 0, "bad tile");
    typedef float accv_t __attribute__((ext_vector_type(RPB)));

    __shared__ __attribute__((aligned(16))) uint32_t smem[2 * ASIZE + 2 * BSTAGE + CSIZE + REDSIZE + PARSIZE];
    uint32_t* As = smem;                                   // [buffer][plane][pixel][ROW]
    uint32_t* Bs = smem + 2 * ASIZE;                       // [stage][plane][co][ROW]
    float* Cs = reinterpret_cast<float*>(smem + 2 * ASIZE + 2 * BSTAGE);  // [unit][lane][4]: the parked accumulators of one item
    float* red = Cs + CSIZE;                               // [WM][BN][2]
    float* par = red + REDSIZE;                            // bias [BN] | mask_a [BN] | mask_b [BN] | 1/sigma

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int TW = 1 << tw_shift, TH = (BM >> tw_shift) >> g_shift;
    const int sub_shift = LOG_BM - g_shift;
    const int HTw = halo_row_stride(TW + 2), HP = (TH + 2) * HTw, npix = HP << g_shift;  // (rows of 16-wide tiles 20 apart: conv_win_glds.h)
    const int nchunks = (p.Cin + CK - 1) / CK;
    const int spi = MODE == 2 ? 4 * nchunks : nchunks;     // steps per item
    const int nsl = spi * T;                               // slices (= barriers) per item
    constexpr int NPH = MODE == 1 ? 4 : 1;
    // this workgroup's run of items: item = (tile * NPH + phase) * n_nb + output-channel block
    const int n_items = n_tiles * NPH * n_nb;
    const int per = n_items / (int)gridDim.x, rem = n_items - per * (int)gridDim.x;
    const int item0 = (int)blockIdx.x * per + min((int)blockIdx.x, rem);
    const int cnt = per + ((int)blockIdx.x < rem ? 1 : 0);
    const int S = cnt * spi;                               // steps of this workgroup
    auto decode = [&](int item) {
        WsGeom g;
        const int nb = item % n_nb, t2 = item / n_nb;
        g.ph = MODE == 1 ? (t2 & 3) : 0;
        g.tile = MODE == 1 ? (t2 >> 2) : t2;
        g.n0 = nb * BN;
        g.n = g_shift ? (g.tile << g_shift) : g.tile / tiles_hw;
        const int trem = g_shift ? 0 : g.tile - g.n * tiles_hw;
        const int th = trem / tiles_w;
        g.h0 = th * TH;
        g.w0 = (trem - th * tiles_w) * TW;
        return g;
    };
    const int taps_w = MODE == 0 ? 9 : (MODE == 1 ? 4 : 16);  // taps per output channel in the weight tensor
    const size_t plane_stride = MODE == 0 ? (size_t)p.Cout * 9 * p.Cin : (size_t)p.Cout * 16 * p.Cin;  // bf16 elements per plane
    const bool has_tail = (p.Cin & (CK - 1)) != 0;
    const int c_last = (nchunks - 1) * CK;

    if (wid < NMW) {
        // =============================================== matrix waves ===============================================
        const int wm = wid / WN, wn = wid % WN;
        accv_t acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < RPB; ++r) acc[i][j][r] = 0.f;
        // halo pixel of this lane's rows under tap (dy, dx) = rp0 + dy * HTw + dx  (no upsampling here: the window is linear)
        int rp0[TM], cp0[TM];  // (cp0: the halo column under tap dx = 0 - what the activation rows are swizzled by)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int q = wm * TM * MB + i * MB + (lane & (MB - 1));
            cp0[i] = q & (TW - 1);
            rp0[i] = (q >> sub_shift) * HP + ((q >> tw_shift) & (TH - 1)) * HTw + cp0[i];
        }
        const int kg = M16 ? lane >> 4 : lane >> 5;
        const int bsw = lds_swz<M16>(lane);
        const uint32_t* Bb0 = Bs + (wn * TN * MB + (lane & (MB - 1))) * ROW;
        const bool tail16 = has_tail && (p.Cin & (CK - 1)) <= 16;
        auto mma = [&](int dyi, int dxi, int abuf, int stage, bool half) {
            const uint32_t* Ab[TM];
            int asw[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int pix = rp0[i] + dyi * HTw + dxi;
                Ab[i] = As + abuf * ASIZE + pix * ROW;
                asw[i] = lds_swz<M16>(cp0[i] + dxi);
            }
            const uint32_t* Bb = Bb0 + stage * BSTAGE;
#pragma unroll
            for (int kk = 0; kk < (M16 ? 1 : CK / 16); ++kk) {
                if (kk == 1 && half) break;  // (wave-uniform)
                const int ks = M16 ? kg : kk * 2 + kg;
                bf16x8_t af[NP][TM], bf[NP][TN];
                const int ob = (ks ^ bsw) << 2;
#pragma unroll
                for (int q = 0; q < NP; ++q) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        af[q][i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(Ab[i] + q * AMAX * ROW + ((ks ^ asw[i]) << 2)));
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        bf[q][j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(Bb + (q * BN + j * MB) * ROW + ob));
                }
                __builtin_amdgcn_s_setprio(1);
                for_each_product<NP>([&](auto qa, auto qb) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j] = mfma_blk<M16>(af[qa][i], bf[qb][j], acc[i][j]);
                });
                __builtin_amdgcn_s_setprio(0);
            }
        };
        // ---- weight cursor: one tap ahead of the multiplication ----
        uint32_t b_off[BPASS], b_tail[BPASS];
        auto b_setup = [&](int rel) {
            const WsGeom g = decode(item0 + rel);
#pragma unroll
            for (int i = 0; i < BPASS; ++i) {
                const int u = min(tid + i * NMT, BUNITS - 1);
                const int plane = u / (BN * 4);
                const int r = (u >> 2) % BN;
                const int ch = ((u & 3) ^ lds_swz<M16>(r)) * 8;
                const uint32_t row = (uint32_t)(plane * plane_stride) + (uint32_t)(g.ph * p.Cout + min(g.n0 + r, p.Cout - 1)) * (uint32_t)taps_w * p.Cin;
                b_off[i] = row + ch;
                b_tail[i] = row + (c_last + ch < p.Cin ? c_last + ch : 0);
            }
        };
        int bq_rel = 0, bq_step = 0, bq_tap = 0, bq_chunk = 0, bq_pl = 0;  // (bq_step = bq_pl * nchunks + bq_chunk in the pooled mode)
        bool bq_on = true;
        auto dma_next = [&](int stage) {
            if (!bq_on) return;
            const bool tail = has_tail && bq_chunk == nchunks - 1;
            const int wt = MODE == 2 ? bq_pl * 4 + bq_tap : bq_tap;
            const uint16_t* base = p.w_split + ((size_t)wt * p.Cin + (tail ? 0 : bq_chunk * CK));
#pragma unroll
            for (int i = 0; i < BPASS; ++i)
                if (i * NMT + wid * 64 < BUNITS) lds_dma16(base + (tail ? b_tail[i] : b_off[i]), Bs + stage * BSTAGE + (i * NMT + wid * 64) * 4);
            if (++bq_tap == T) {
                bq_tap = 0;
                ++bq_step;
                if (++bq_chunk == nchunks) {
                    bq_chunk = 0;
                    ++bq_pl;
                }
                if (bq_step == spi) {
                    bq_step = 0;
                    bq_pl = 0;
                    if (++bq_rel == cnt) bq_on = false;
                    else b_setup(bq_rel);
                }
            }
        };
        b_setup(0);
        dma_next(0);
        dma_drain();
        __syncthreads();  // barrier 0: the loaders have staged the first halo
        int gt = 0, sidx = 0;
        for (int it = 0; it < cnt; ++it) {
            const WsGeom g = decode(item0 + it);
            const int py = g.ph >> 1, px = g.ph & 1;
            int chunk = 0, pl = 0;
            for (int s = 0; s < spi; ++s, ++sidx) {
                const int abuf = sidx & 1;
                const bool half = tail16 && chunk == nchunks - 1;
                const int pp = pl >> 1, qq = pl & 1;
#pragma unroll
                for (int tap = 0; tap < T; ++tap) {
                    const int st = gt & 1;
                    dma_next(st ^ 1);
                    if (MODE == 0) mma(tap / 3, tap % 3, abuf, st, half);
                    else if (MODE == 1) mma((tap >> 1) + py, (tap & 1) + px, abuf, st, half);
                    else mma((tap >> 1) + 1 - pp, (tap & 1) + 1 - qq, abuf, st, half);
                    if (tap == T - 1 && s == spi - 1) {
                        // park the item's sums for the loaders (who finished reading the previous item's at least one barrier ago)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
#pragma unroll
                                for (int gq = 0; gq < NG; ++gq) {
                                    const f32x4 v = {acc[i][j][4 * gq + 0], acc[i][j][4 * gq + 1], acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]};
                                    *reinterpret_cast<f32x4*>(Cs + ((wm * NU + (wn * TN + j) * GSZ + i * NG + gq) * 64 + lane) * 4) = v;
#pragma unroll
                                    for (int r = 0; r < 4; ++r) acc[i][j][4 * gq + r] = 0.f;
                                }
                    }
                    dma_drain();
                    __syncthreads();
                    ++gt;
                }
                if (++chunk == nchunks) {
                    chunk = 0;
                    ++pl;
                }
            }
        }
        __syncthreads();  // the loaders' last epilogue: its statistics meet in LDS
        return;
    }

    // ==================================================== loaders ====================================================
    const int lw = wid - NMW, lt = tid - NMT;  // 0..3, 0..255
    // measurement switches (dgmr_debug_flags 1 / 2, tools/ws_check.py --dbg=): 1 = the loaders skip the epilogue (nothing is stored),
    // 2 = they skip the halo staging (the matrix waves multiply stale LDS contents).  Outputs are garbage; only the time means anything.
    const int dbg = p.reserved1 & 3;
    const int cq = lt & 7;
    // (wave-uniform reciprocals, parked in scalar registers)
    const float inv_hp = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, 1.f / (float)HP)));
    const float inv_htw = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, 1.f / (float)HTw)));
    const float* pa_base = p.pre_a ? p.pre_a : p.x;
    const float* pb_base = p.pre_a ? p.pre_b : p.x;
    const int Hs = p.H, Ws = p.W;  // the map the window lies on (pooled mode: one parity plane of the 2 Hs x 2 Ws input)

    // ---- halo fetch cursor: two to three steps ahead of the multiplication ----
    uint32_t a_goff[APASS];
    unsigned a_valid = 0;
    uint32_t f_grp = 0;
    unsigned a_swz = 0;  // 2 bits per halo item of this thread: lds_swz of its halo column (the same for every tile)
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
        const int pix = (lt >> 3) + i * 32;
        const int sub = (int)(((float)pix + 0.5f) * inv_hp), prem = pix - sub * HP;
        const int lr = (int)(((float)prem + 0.5f) * inv_htw), lc = prem - lr * HTw;
        a_swz |= (unsigned)lds_swz<M16>(lc) << (2 * i);
    }
    int f_rel = 0, f_step = 0, f_chunk = 0, f_pl = 0;
    auto f_setup = [&](int rel) {
        const WsGeom g = decode(item0 + rel);
        const int oh = g.h0 - 1, ow = g.w0 - 1;
        a_valid = 0;
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            const int pix = (lt >> 3) + i * 32;
            const int sub = (int)(((float)pix + 0.5f) * inv_hp), prem = pix - sub * HP;
            const int lr = (int)(((float)prem + 0.5f) * inv_htw), lc = prem - lr * HTw;
            const int ih = oh + lr, iw = ow + lc;
            const bool ok = pix < npix && (unsigned)ih < (unsigned)Hs && (unsigned)iw < (unsigned)Ws;
            a_goff[i] = !ok ? 0u
                        : MODE == 2 ? (((uint32_t)(g.n + sub) * 2 * Hs + 2 * ih) * 2 * Ws + 2 * iw) * p.Cin + cq * 4
                                    : (((uint32_t)(g.n + sub) * Hs + ih) * Ws + iw) * p.Cin + cq * 4;
            a_valid |= (ok ? 1u : 0u) << i;
        }
        f_grp = (uint32_t)(g.n / p.pre_group) * p.Cin;
    };
    f32x4 ra[NSETS][APASS], rpa[NSETS], rpb[NSETS];
    unsigned xm[NSETS] = {};
    // the loads of the cursor's step into register set E (all unconditional), then the cursor moves on; past the last step it stays
    // (that step is fetched again, into a set that is never stored)
    auto issue = [&](auto set_c) {
        constexpr int E = decltype(set_c)::value;
        const int cb = f_chunk * CK + cq * 4;
        const bool kok = cb < p.Cin;
        const unsigned valid = kok ? a_valid : 0u;
        const uint32_t shift = f_chunk * CK + (MODE == 2 ? (uint32_t)(((f_pl >> 1) * 2 * Ws + (f_pl & 1)) * p.Cin) : 0u);
        xm[E] = valid;
#pragma unroll
        for (int i = 0; i < APASS; ++i) ra[E][i] = *reinterpret_cast<const f32x4*>(p.x + (((valid >> i) & 1u) ? a_goff[i] + shift : 0u));
        rpa[E] = ws_gload4(pa_base, (p.pre_a && kok) ? f_grp + cb : 0u);
        rpb[E] = ws_gload4(pb_base, (p.pre_a && kok) ? f_grp + cb : 0u);
        if (f_rel * spi + f_step + 1 < S) {
            ++f_step;
            if (++f_chunk == nchunks) {
                f_chunk = 0;
                ++f_pl;
            }
            if (f_step == spi) {
                f_step = 0;
                f_pl = 0;
                f_setup(++f_rel);
            }
        }
    };
    // prologue, bf16 split, swizzled ds_write of items [I0, I1) of set E into A buffer `buf` (conv3x3_glds_kernel's stage_a)
    auto store_items = [&](auto set_c, auto i0_c, auto i1_c, int buf) {
        constexpr int E = decltype(set_c)::value, I0 = decltype(i0_c)::value, I1 = decltype(i1_c)::value;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        const unsigned valid = xm[E];
#pragma unroll
        for (int i = I0; i < I1; ++i) {
            const int pix = (lt >> 3) + i * 32;
            f32x4 v = ra[E][i];
            if (p.pre_a) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaf(v[j], rpa[E][j], rpb[E][j]), 0.f);
            } else if (p.pre_relu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            v = ((valid >> i) & 1u) ? v : zero4;
            u32x2 pl[NP];
            split_planes4<NP>(v, pl);
            if (pix < AMAX) {
                uint32_t* dst = As + buf * ASIZE + pix * ROW + (((cq >> 1) ^ ((a_swz >> (2 * i)) & 3)) << 2) + (cq & 1) * 2;
#pragma unroll
                for (int q = 0; q < NP; ++q) *reinterpret_cast<u32x2*>(dst + q * AMAX * ROW) = pl[q];
            }
        }
    };

    // ---- epilogue of the previous item, one "unit" (the four accumulator registers of one lane: 4 consecutive pixels x 1 column) at
    //      a time.  The arithmetic and the statistics run in the accumulators' own layout (lane = output channel: per-column
    //      constants are one value per lane, the column sums are in-lane adds plus one or two lane swaps); only the result is
    //      transposed across the quad for the 16-byte store.  The glds kernel transposes first and pays a 4-step butterfly on eight
    //      values per column block for the same sums: 40 of its 110 VALU instructions per unit.  Per element the expressions are
    //      the same, so y is bit-identical; the statistics are summed in another order. ----
    const int pshift = MODE == 1 ? 1 : 0, oH = p.H << pshift, oW = p.W << pshift;
    const int j4 = lane & 3, q4 = (lane & (MB - 1)) >> 2, rsel = M16 ? lane >> 4 : lane >> 5, col_l = lane & (MB - 1);
    const bool want_stats = p.stats_out != nullptr;
    const float* eop_base = EOP ? (p.residual ? p.residual : p.mask_src) : nullptr;
    const bool eop_res = p.residual != nullptr;
    const uint32_t pstride = (uint32_t)p.Cout << pshift;  // elements between two consecutive rows of a unit
    // lane constants per row group r2 = i * NG + gq of a column block: element offsets (without the item's base and the block's
    // first column) of the unit's first row in the accumulator layout / of this lane's row after the transposition / in a
    // half-resolution residual
    // (scalars, not arrays: a run-time index into a four-element array sends it to scratch memory, and scratch loads count on vmcnt)
    auto lane_pix = [&](int r2, bool half_res) {
        const int i = r2 / NG, gq = r2 - i * NG;
        const int q = lw * TM * MB + (M16 ? i * 16 + 4 * rsel : i * 32 + 8 * gq + 4 * rsel);
        const int qs = q >> sub_shift, qh = (q >> tw_shift) & (TH - 1), qw = q & (TW - 1);
        return half_res ? (uint32_t)((qs * (oH >> 1) + (qh >> 1)) * (oW >> 1) + (qw >> 1)) : (uint32_t)((qs * oH + (qh << pshift)) * oW + (qw << pshift));
    };
    const uint32_t ln0 = lane_pix(0, false) * p.Cout + col_l, ln1 = lane_pix(1 % GSZ, false) * p.Cout + col_l;
    const uint32_t ln2 = lane_pix(2 % GSZ, false) * p.Cout + col_l, ln3 = lane_pix(3 % GSZ, false) * p.Cout + col_l;
    const uint32_t lr0 = lane_pix(0, true) * p.Cout + col_l, lr1 = lane_pix(1 % GSZ, true) * p.Cout + col_l;  // (plain mode only)
    const uint32_t lr2 = lane_pix(2 % GSZ, true) * p.Cout + col_l, lr3 = lane_pix(3 % GSZ, true) * p.Cout + col_l;
    const uint32_t ls_delta = (uint32_t)(j4 << pshift) * p.Cout + 4 * q4 - col_l;  // transposed (store) position = accumulator position + this
    auto pick_ln = [&](int r2) { return r2 == 0 ? ln0 : (r2 == 1 ? ln1 : (r2 == 2 ? ln2 : ln3)); };
    auto pick_lr = [&](int r2) { return r2 == 0 ? lr0 : (r2 == 1 ? lr1 : (r2 == 2 ? lr2 : lr3)); };
    // the parked item: element offset of its first output pixel's column n0 (full / half resolution), its row of statistics partials
    uint32_t eg_off = 0, eg_roff = 0;
    int eg_n0 = 0, eg_srow = 0, eg_n = 0;
    auto set_eg = [&](const WsGeom& g) {
        const int hh = (g.h0 << pshift) + (g.ph >> 1), ww = (g.w0 << pshift) + (g.ph & 1);
        eg_off = (uint32_t)((g.n * oH + hh) * oW + ww) * p.Cout + g.n0;
        eg_roff = (uint32_t)((g.n * (oH >> 1) + (hh >> 1)) * (oW >> 1) + (ww >> 1)) * p.Cout + g.n0;
        eg_n0 = g.n0;
        eg_n = g.n;
        eg_srow = MODE == 1 ? g.tile * 4 + g.ph : g.tile;
    };
    bool e_on = false;
    int sigma = 0;              // slice inside the current item
    const int esl = (NU + ups - 1) / ups;
    float s0 = 0.f, s1 = 0.f;
    f32x4 ring[RING][UPSMAX];
    // the fused operand of unit u in the accumulator layout: four 4-byte loads, one per row
    auto load_eop = [&](int u, bool on) {
        const int grp = u / GSZ, r2 = u - grp * GSZ;
        const bool col_ok = eg_n0 + grp * MB + col_l < p.Cout;
        f32x4 v;
        const bool half = MODE == 0 && eop_res && p.residual_up;  // (a half-resolution residual exists in the plain mode only)
        const uint32_t b0 = (half ? eg_roff + pick_lr(r2) : eg_off + pick_ln(r2)) + grp * MB;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t off = half ? b0 + (uint32_t)(r >> 1) * p.Cout : b0 + (uint32_t)r * pstride;
            v[r] = ws_gload(eop_base, (on && col_ok) ? off : 0u);
        }
        return v;
    };
    auto run_unit = [&](int u, f32x4 eo) {
        const int grp = u / GSZ, r2 = u - grp * GSZ;
        const int cl = grp * MB + col_l;  // this lane's column inside the item's block
        const f32x4 c4 = *reinterpret_cast<const f32x4*>(Cs + ((lw * NU + u) * 64 + lane) * 4);
        const float bv = par[cl], mav = par[BN + cl], mbv = par[2 * BN + cl], sc = par[3 * BN];
        f32x4 o;
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = fmaf(c4[r], sc, bv);
            if (p.act_relu) v = fmaxf(v, 0.f);
            if (EOP && eop_res) v += eo[r];
            if (EOP && !eop_res) v = fmaf(eo[r], mav, mbv) > 0.f ? v : 0.f;
            o[r] = v;
            t0 += v;
            t1 = fmaf(v, (EOP && !eop_res) ? eo[r] : v, t1);
        }
        const f32x4 ot = quad_transpose(o[0], o[1], o[2], o[3], lane);  // -> 4 consecutive channels of this lane's row
        if (eg_n0 + grp * MB + 4 * q4 < p.Cout) *reinterpret_cast<f32x4*>(p.y + eg_off + pick_ln(r2) + ls_delta + grp * MB) = ot;
        if (want_stats) {
            s0 += t0;
            s1 += t1;
            if (r2 == GSZ - 1) {  // last row group of this column block: fold the lane groups that hold other rows of the same column
                float a = s0, b = s1;
                a += __shfl_xor(a, 32, 64);
                b += __shfl_xor(b, 32, 64);
                if (M16) {
                    a += __shfl_xor(a, 16, 64);
                    b += __shfl_xor(b, 16, 64);
                }
                if (lane < MB) {
                    red[(lw * BN + cl) * 2 + 0] = a;
                    red[(lw * BN + cl) * 2 + 1] = b;
                }
                s0 = 0.f;
                s1 = 0.f;
            }
        }
    };
    auto stats_final = [&]() {  // after a barrier behind the last unit: the four loader waves' row sums meet here, one row of partials per tile
        if (!want_stats) return;
        for (int idx = lt; idx < BN * 2; idx += 256) {
            const int cl = idx >> 1, which = idx & 1;
            float v = 0.f;
#pragma unroll
            for (int qq = 0; qq < WM; ++qq) v += red[(qq * BN + cl) * 2 + which];
            if (eg_n0 + cl < p.Cout) p.stats_out[((size_t)eg_srow * 2 + which) * p.Cout + eg_n0 + cl] = v;
        }
    };
    // per-column constants of the item being multiplied, read at the last tap of every step (unconditionally) and published to LDS
    // at the item boundary: thread t < 3 BN / 4 holds four of bias | mask_a | mask_b, thread 3 BN / 4 the 1 / sigma of the tile's sample
    f32x4 pv = {0.f, 0.f, 0.f, 0.f};
    float psc = 1.f;
    int c_rel = 0, c_step = 0;  // the item / step being multiplied
    WsGeom cg = decode(item0);
    const int par_which = lt / (BN / 4), par_c4 = (lt - par_which * (BN / 4)) * 4;
    auto load_params = [&]() {  // (two unconditional loads; what they mean is sorted out when they are published, a tap later)
        const int col = min(cg.n0 + par_c4, p.Cout - 4);
        const float* src = p.x;
        uint32_t off = 0;
        if (par_which == 0 && p.bias) src = p.bias, off = col;
        if (par_which == 1 && p.mask_a) src = p.mask_a, off = (uint32_t)(cg.n / p.mask_group) * p.Cout + col;
        if (par_which == 2 && p.mask_a) src = p.mask_b, off = (uint32_t)(cg.n / p.mask_group) * p.Cout + col;
        pv = ws_gload4(src, off);
        psc = ws_gload(p.scale ? p.scale : p.x, p.scale ? (uint32_t)(cg.n / p.scale_group) : 0u);
    };
    auto publish_params = [&]() {
        f32x4 w4 = pv;
        if (par_which == 0 && !p.bias) w4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (par_which == 1 && !p.mask_a) w4 = (f32x4){1.f, 1.f, 1.f, 1.f};
        if (par_which == 2 && !p.mask_a) w4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (par_which < 3) *reinterpret_cast<f32x4*>(par + par_which * BN + par_c4) = w4;
        else if (lt == 3 * (BN / 4)) par[3 * BN] = p.scale ? psc : 1.f;
    };
    // one tap's share of the epilogue; P: the slice's position in the statically unrolled body (names the ring slot)
    auto epi_slice = [&](auto p_c) {
        constexpr int P = decltype(p_c)::value;
        if (e_on) {
            const int pos = sigma - D;
            if (pos >= 0 && pos < esl) {
#pragma unroll
                for (int k = 0; k < UPSMAX; ++k) {
                    const int u = pos * ups + k;
                    if (k < ups && u < NU) run_unit(u, ring[P % RING][k]);
                }
            } else if (pos == esl) {
                stats_final();
            }
        }
        if (EOP) {  // operands of the units D taps from now (clamped when there are none: the loads stay unconditional)
#pragma unroll
            for (int k = 0; k < UPSMAX; ++k) {
                const int u = sigma * ups + k;
                const bool on = e_on && sigma < esl && k < ups && u < NU;
                ring[(P + D) % RING][k] = load_eop(on ? u : 0, on);
            }
        }
    };
    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;
    // one tap of the loaders' program; H: the step's position in the statically unrolled body, t: the tap
    auto slice = [&](auto h_c, auto t_c, int step_idx) {
        constexpr int H = decltype(h_c)::value, t = decltype(t_c)::value;
        const bool more = step_idx + 1 < S;
        const int nbuf = (step_idx + 1) & 1;  // the A buffer of the next step
        if (sigma == 0 && e_on) publish_params();
        if (!(dbg & 1)) epi_slice(std::integral_constant<int, H * T + t>{});
        if constexpr (NSETS == 1) {
            // taps 2 .. T-1: one halo item each (T - 2 >= APASS), loaded at tap 0 of this very step
            if (t == 0 && !(dbg & 2)) issue(c0{});
            if (t >= 2 && t - 2 < APASS && more && !(dbg & 2)) store_items(c0{}, std::integral_constant<int, (t >= 2 ? t - 2 : 0)>{}, std::integral_constant<int, (t >= 2 ? t - 1 : 0)>{}, nbuf);
            if (t == T - 1) load_params();
        } else {
            using NE = std::integral_constant<int, (H + 1) & 1>;  // the set that holds the next step's halo
            if (more && !(dbg & 2)) store_items(NE{}, std::integral_constant<int, (t * APASS) / T>{}, std::integral_constant<int, ((t + 1) * APASS) / T>{}, nbuf);
            if (t == T - 1) {
                load_params();
                if (!(dbg & 2)) issue(NE{});
            }
        }
        if (t == T - 1) {  // the step being multiplied ends with this tap
            if (++c_step == spi) {
                c_step = 0;
                set_eg(cg);
                e_on = true;
                if (++c_rel < cnt) cg = decode(item0 + c_rel);
            }
        }
        if (++sigma == nsl) sigma = 0;
        __syncthreads();
    };
    auto step_body = [&](auto h_c, int step_idx) {
        slice(h_c, std::integral_constant<int, 0>{}, step_idx);
        slice(h_c, std::integral_constant<int, 1>{}, step_idx);
        slice(h_c, std::integral_constant<int, 2>{}, step_idx);
        slice(h_c, std::integral_constant<int, 3>{}, step_idx);
        if constexpr (T == 9) {
            slice(h_c, std::integral_constant<int, 4>{}, step_idx);
            slice(h_c, std::integral_constant<int, 5>{}, step_idx);
            slice(h_c, std::integral_constant<int, 6>{}, step_idx);
            slice(h_c, std::integral_constant<int, 7>{}, step_idx);
            slice(h_c, std::integral_constant<int, 8>{}, step_idx);
        }
    };
    static_assert(NSETS == 2 || T - 2 >= APASS, "one halo item per tap");
    f_setup(0);
    issue(c0{});  // step 0
    if constexpr (NSETS == 2) issue(c1{});  // step 1
    store_items(c0{}, std::integral_constant<int, 0>{}, std::integral_constant<int, APASS>{}, 0);
    if constexpr (NSETS == 2) issue(c0{});  // step 2
#pragma unroll
    for (int k = 0; k < UPSMAX; ++k)
#pragma unroll
        for (int r = 0; r < RING; ++r) ring[r][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();  // barrier 0
    if constexpr (NSETS == 1) {
        for (int sidx = 0; sidx < S; ++sidx) step_body(c0{}, sidx);
    } else {
        for (int sidx = 0; sidx < S; sidx += 2) {
            step_body(c0{}, sidx);
            if (sidx + 1 < S) step_body(c1{}, sidx + 1);
        }
    }
    // ---- the last item's epilogue: nothing left to overlap with.  The loader waves cannot meet without the matrix waves, so every
    //      wave writes the WHOLE parameter block itself (all four write identical values) and reads it behind its own LDS wait ----
    for (int idx = lane; idx < 3 * (BN / 4) + 1; idx += 64) {
        const int which = idx / (BN / 4), c4i = (idx - which * (BN / 4)) * 4;
        const int col = min(eg_n0 + c4i, p.Cout - 4);
        f32x4 w4 = {0.f, 0.f, 0.f, 0.f};
        if (which == 0 && p.bias) w4 = *reinterpret_cast<const f32x4*>(p.bias + col);
        if (which == 1) w4 = p.mask_a ? *reinterpret_cast<const f32x4*>(p.mask_a + (size_t)(eg_n / p.mask_group) * p.Cout + col) : (f32x4){1.f, 1.f, 1.f, 1.f};
        if (which == 2 && p.mask_a) w4 = *reinterpret_cast<const f32x4*>(p.mask_b + (size_t)(eg_n / p.mask_group) * p.Cout + col);
        if (which < 3) *reinterpret_cast<f32x4*>(par + which * BN + c4i) = w4;
        else par[3 * BN] = p.scale ? p.scale[eg_n / p.scale_group] : 1.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    for (int u = 0; u < NU && !(dbg & 1); ++u) {
        f32x4 eo = {0.f, 0.f, 0.f, 0.f};
        if (EOP) eo = load_eop(u, true);
        run_unit(u, eo);
    }
    __syncthreads();
    stats_final();
}

}  // namespace
