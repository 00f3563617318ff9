// 3x3 convolution with the input window in LDS, WEIGHT STAGES FILLED BY LDS-DMA (forward and data gradient of the big maps).
//
// conv3x3_win_kernel copies every weight stage HBM/L2 -> VGPRs -> ds_write_b128 -> LDS.  On gfx950 a 16-byte LDS store moves its
// address and data registers at 13 cycles per wave-instruction (79 B/clk): the nine stage copies of a 32-channel chunk keep the
// LDS busy almost as long as all fragment reads together, and LDS time, not the matrix pipe, bounds that kernel.  Here
//   * the weight stage of tap s+1 is written by global_load_lds_dwordx4 (no registers, no LDS store instruction) into the second
//     of two stage buffers while tap s is multiplied: ONE barrier per tap instead of two;
//   * LDS rows are the bare 64 bytes of 32 bf16 (no padding - the DMA image is lane-linear): the 16-byte k-slot s of row r lives
//     at slot s ^ ((r >> 2) & 3), which keeps the 16 rows of a ds_read_b128 lane group on 16 distinct bank slots; the weight rows
//     are swizzled through the per-lane SOURCE address of the DMA, the activation rows at their ds_write;
//   * 26 KB of activations + 2 x 12 KB of weights (96 output channels, bf16x3) = three workgroups per CU.
// 3x3x3 convolutions (the temporal discriminator's 3-D blocks, discriminators.py:189-205) run through the same kernel: a tile lies
// in one depth plane, every 32-channel chunk is walked as three groups of nine taps, and the halo of input plane d + kd - 1 is
// restaged per group (all zeros when that plane is outside the volume).
// A DMA cannot zero-fill: output channels beyond Cout read the last row (their columns are never stored) and input channels
// beyond Cin read channel group 0 (their activations are exact zeros).  Everything else is conv3x3_win_kernel's.
#pragma once
#include "conv_bf16.h"

namespace {

// 16 bytes per lane, global -> LDS at (wave-uniform) lds + 16 * lane, no registers in between (global_load_lds_dwordx4).
// The builtin exists in the device pass only; the host pass needs the kernel body just to emit its launch stub.
__device__ __forceinline__ void lds_dma16(const void* gptr, uint32_t* lds) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(gptr, lds, 16, 0, 0);
#else
    (void)gptr;
    (void)lds;
#endif
}

// An LDS-DMA is a pending LDS write on the issuing wave's VM counter and nothing else orders it: before the barrier that publishes a
// stage to the other waves every wave drains its own DMAs EXPLICITLY (hipcc currently happens to emit this wait in front of
// __syncthreads(), but the memory model does not oblige it to; tests/test_abi.py greps the disassembly for it).
__device__ __forceinline__ void dma_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// BM: output pixels per workgroup.  256 (8 x 32 or 16 x 16 pixels of one image) gives every wave twice the pixels per weight
// fragment: (TM + TN) * planes LDS reads feed TM * TN * terms MFMAs, half the barriers per MFMA, a 1.33x instead of 1.59x halo - at
// two workgroups per CU instead of three (68 KB of LDS, ~170 registers).
template <bool M16, typename ACC>
__device__ __forceinline__ ACC mfma_blk(const bf16x8_t& a, const bf16x8_t& b, const ACC& c) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (M16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#else
    return c;
#endif
}

// value of lane ^ 1 / lane ^ 2 (inside a quad of lanes: DPP quad_perm [1,0,3,2] / [2,3,0,1] - a VALU move, no LDS traffic)
__device__ __forceinline__ float quad_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
}
__device__ __forceinline__ float quad_xor2(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
}
// 4 x 4 transpose across a quad of lanes: lane c (= column c of the quad) holds rows 0..3 of its column in a0..a3; on return lane j
// holds columns 0..3 of row j.  Two exchange stages (with lane ^ 1, then lane ^ 2), 4 DPP moves + 8 selects.
__device__ __forceinline__ f32x4 quad_transpose(float a0, float a1, float a2, float a3, int lane) {
    const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
    const float r01 = quad_xor1(b0 ? a0 : a1), r23 = quad_xor1(b0 ? a2 : a3);
    a0 = b0 ? r01 : a0;
    a1 = b0 ? a1 : r01;
    a2 = b0 ? r23 : a2;
    a3 = b0 ? a3 : r23;
    const float r02 = quad_xor2(b1 ? a0 : a2), r13 = quad_xor2(b1 ? a1 : a3);
    a0 = b1 ? r02 : a0;
    a2 = b1 ? a2 : r02;
    a1 = b1 ? r13 : a1;
    a3 = b1 ? a3 : r13;
    return (f32x4){a0, a1, a2, a3};
}

// Swizzle of the bare 64-byte LDS rows: logical 16-byte k-slot s of row r lives at slot s ^ lds_swz<M16>(r).
//   32 x 32 blocks (fragment = row lane & 31, k-slot 2 kk + (lane >> 5)):  (r >> 2) & 3  - conflict-free for any first row.
//   16 x 16 blocks (fragment = row lane & 15, k-slot lane >> 4): the four k-slots of a row group are read by ONE instruction, and
//   gfx950's 16-lane ds_read_b128 groups ({0-3, 12-15, 20-27}, ...) mix rows 0-3 / 12-15 at one k-slot with rows 4-11 at the next:
//   under (r >> 2) & 3 every group collides 2-way for 14 of 16 first rows (PMC, round 4: 44 % of the LDS cycles of an M16 kernel were
//   bank conflicts, the 32 x 32 kernels have none).  2 * ((r >> 2) & 1) is conflict-free for every first row
//   (tests/test_lds_layouts.py enumerates both).
template <bool M16>
__device__ __forceinline__ int lds_swz(int r) {
    return M16 ? ((r >> 2) & 1) * 2 : (r >> 2) & 3;
}
// Activation rows are swizzled by their halo COLUMN, not by their pixel index, and the halo rows of 16-pixel-wide tiles are 20 pixels
// apart instead of 18: the 32 rows of a 32 x 32 block's fragment are then two tile rows whose columns line up modulo 4, and the
// fragment reads are conflict-free for every tap shift (by pixel index they collided 2-way on every 16-wide map - PMC, round 4: 31 %
// of the LDS cycles of the 128-column kernel on 16 x 16 maps - whatever the row stride; tests/test_lds_layouts.py enumerates
// tile widths, strides and shifts).  8-pixel-wide tiles (two 8 x 8 images per tile) stay 2-way conflicted (3-way by pixel index).
__host__ __device__ constexpr int halo_row_stride(int halo_cols) { return halo_cols == 18 ? 20 : halo_cols; }

// PRIV: every wave streams ITS OWN 32 x TN output-channel slice of the weights into a private double buffer and nothing but the
// activation halo is shared: no barrier between taps (one pair per 32-channel chunk, when the halo is replaced), the waves of a
// workgroup drift apart and the SIMDs interleave them freely - the per-tap barrier made every workgroup wait for its slowest SIMD
// nine times per chunk (PMC: 37 % of the wave cycles parked, matrix pipe 50 % busy).
// M16: the wave's tile is made of 16 x 16 blocks (v_mfma_f32_16x16x32_bf16: one instruction per 32-channel chunk and block) instead
// of 32 x 32 ones: 48 output channels are three blocks - the 64-column tile spent a quarter of its matrix work on padding.
// PAIR (round 5, phase mode only): ONE workgroup computes BOTH column parities (px = 0, 1) of a row parity py from one staged halo - two
// accumulator sets, eight taps per 32-channel chunk instead of four.  The four-workgroups-per-tile scheme staged (fetched, BatchNorm +
// relu'd, split, LDS-stored) every halo four times for four taps each: 9.1 VALU instructions per MFMA against 5.0 in the plain mode,
// matrix pipe 29 - 33 % busy (PMC, profiles/r05_pmc_classes.json).  128-pixel tiles: the doubled accumulators take the registers the
// 256-pixel tile's second block row had.
template <int BN, int WM, int WN, int NS, int BM = 128, bool PRIV = false, bool M16 = false, bool PAIR = false>
__global__ __launch_bounds__(256, (BN == 128 || BM == 256 || PRIV || NS == 6 || (M16 && BN >= 96) || PAIR) ? 2 : 3) void conv3x3_glds_kernel(const dgmr_conv_args p, const int tw_shift,
                                                                                          const int tiles_w, const int tiles_hw,
                                                                                          const int g_shift) {
    constexpr int CK = 32;
    constexpr int LOG_BM = BM == 256 ? 8 : 7;
    static_assert(BM == 128 || BM == 256, "BM");
    constexpr int ROW = CK / 2;  // dwords per LDS row
    constexpr int NP = planes_of<NS>::value;
    constexpr int MB = M16 ? 16 : 32;   // edge of an MFMA block
    constexpr int RPB = M16 ? 4 : 16;   // accumulator registers per block
    constexpr int TM = BM / WM / MB, TN = BN / WN / MB;
    static_assert(!M16 || (!PRIV && BN % 16 == 0), "M16 tile");
    typedef float accv_t __attribute__((ext_vector_type(RPB)));
    constexpr int AMAX = BM == 256 ? 10 * 34 : 6 * 34;  // halo pixels: 6 x 34 / 10 x 20 (BM 128: rows of 16-wide tiles 20 apart), 10 x 34 / 18 x 18 (BM 256)
    constexpr int APASS = (AMAX * 8 + 255) / 256;
    constexpr int BUNITS = BN * 4 * NP;  // 16-byte units of one weight stage
    // PRIV: a wave's slice is TN x 32 rows per plane = TN * 2 DMA instructions (16 rows each) per plane
    constexpr int BPASS = PRIV ? TN * 2 * NP : (BUNITS + 255) / 256;  // plain bf16 at 96 channels: 384 units = one full pass + waves 0, 1 of a second
    constexpr int WSLICE = NP * TN * 32 * ROW;                        // dwords of one wave's private slice of a stage
    constexpr int BSTAGE = PRIV ? 4 * WSLICE : NP * BN * ROW;         // dwords of one stage
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1 && BUNITS % 64 == 0, "bad tile");
    static_assert(!PAIR || (BM == 128 && !PRIV), "PAIR tile");
    constexpr int NACC = PAIR ? 2 : 1;

    __shared__ __attribute__((aligned(16))) uint32_t smem[NP * AMAX * ROW + 2 * BSTAGE];
    uint32_t* As = smem;                     // [plane][pixel][ROW]
    uint32_t* Bs = smem + NP * AMAX * ROW;   // [stage][plane][co][ROW]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int TW = 1 << tw_shift, TH = (BM >> tw_shift) >> g_shift;
    const int sub_shift = LOG_BM - g_shift;
    // phase mode: blockIdx.x = 4 x tiles.  The four phases of a tile read the same input halo: they get workgroup ids 8 apart, i.e.
    // the SAME XCD (ids are dealt round-robin over the 8 XCDs) back to back, so three of the four halo fetches hit that XCD's L2
    // (PMC before: 5.2x the input read from the fabric; phases as blockIdx.y put them a whole grid row apart)
    const bool phase_pre = p.reserved0 == 1;
    const int bid = blockIdx.x;
    // (PAIR: blockIdx.x = 2 x tiles, one workgroup per row parity; ph_pre is then the px = 0 phase of the pair)
    const bool xcd_map = phase_pre && (gridDim.x & (PAIR ? 15 : 31)) == 0;
    const int ph_pre = !phase_pre ? 0 : PAIR ? 2 * (xcd_map ? (bid >> 3) & 1 : bid & 1) : (xcd_map ? (bid >> 3) & 3 : bid & 3);
    const int tile = !phase_pre ? bid : PAIR ? (xcd_map ? ((bid >> 4) << 3) | (bid & 7) : bid >> 1) : (xcd_map ? ((bid >> 5) << 3) | (bid & 7) : bid >> 2);
    const int n = g_shift ? (tile << g_shift) : tile / tiles_hw;  // first image of the tile; 3-D: depth plane (sample * D + d)
    const int KD = p.KD;                                           // 1, or 3 (then g_shift == 0)
    const int smp = KD == 3 ? n / p.D : n;                         // sample of the tile's first image: statistics / sigma groups
    const int dpl = KD == 3 ? n - smp * p.D : 0;
    const int trem = g_shift ? 0 : tile - n * tiles_hw;
    const int th = trem / tiles_w;
    const int h0 = th * TH, w0 = (trem - th * tiles_w) * TW;
    // Phase mode (p.reserved0 = 1, set by the library for `upsample` convs that come with w_phase): the map (p.H x p.W) is the
    // LOW-resolution input; output pixel (2h + py, 2w + px) of the nearest-2x-upsampled conv reads the 2 x 2 input pixels
    // (h + py - 1 + a, w + px - 1 + b) with the tap sums of w_phase - four of nine MFMA steps per output, a halo staged once per
    // 128 INPUT pixels.
    // Pooled mode (p.reserved0 = 2, the data gradient of such a conv: 3x3 conv of the (2 p.H x 2 p.W) input followed by a 2 x 2 sum
    // pool = 4 x 4 stride-2 conv): the input is walked as its four pixel-parity planes (strided views, staged like depth planes), each
    // with the 2 x 2 taps of w_phase ([16 taps] = plane * 4 + tap) - 16 instead of 36 multiply steps per output pixel, and the
    // full-resolution gradient is never written.
    const bool phase = p.reserved0 == 1, pooled = p.reserved0 == 2;
    const int ph = ph_pre, py = ph >> 1, px = ph & 1;
    const int n0 = (int)blockIdx.y * BN;
    const int us = p.upsample ? 1 : 0;
    const int Hs = p.H >> us, Ws = p.W >> us;
    const int oh = (h0 - 1) >> us, ow = (w0 - 1) >> us;
    // (256-pixel tiles on 16-wide maps keep 18: 18 x 20 pixels would need a twelfth staging register set in a kernel that sits at 243
    //  registers - and no layer of the step runs them: 16 x 16 maps carry 384 channels = the 128-column kernel, 128-pixel tiles)
    const int HTw = BM == 128 ? halo_row_stride((TW >> us) + 2) : (TW >> us) + 2;
    const int HP = ((TH >> us) + 2) * HTw;
    const int npix = HP << g_shift;
    const int nchunks = (p.Cin + CK - 1) / CK;
    const int taps = phase ? 4 : (pooled ? 16 * KD : 9 * KD);

    // ---- activation halo (registers -> prologue -> split -> swizzled ds_write), as conv3x3_win_kernel ----
    const int cq = tid & 7;
    uint32_t a_goff[APASS];
    unsigned a_valid = 0;
    // (the 256-pixel-tile kernels keep the pixel-index swizzle: they run 32-wide tiles, where it is conflict-free, and the dominant one has
    //  no register to spare for anything else)
    constexpr bool BYCOL = BM == 128;
    unsigned a_swz = 0;  // 2 bits per item: lds_swz of the item's halo column
    // (pix / HP and prem / HTw by reciprocal multiplication: an integer division by a run-time divisor is ~35 VALU instructions, and
    //  the 2 x APASS of them were a third of a tile's set-up.  Exact: pix + 0.5 is at least 0.5 / HP = 1.5e-3 away from every multiple
    //  of HP in relative terms, the float error is 1e-6.)
    const float inv_hp = 1.f / (float)HP, inv_htw = 1.f / (float)HTw;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
        const int pix = (tid >> 3) + i * 32;
        const int sub = (int)(((float)pix + 0.5f) * inv_hp), prem = pix - sub * HP;
        const int lr = (int)(((float)prem + 0.5f) * inv_htw), lc = prem - lr * HTw;
        const int ih = oh + lr, iw = ow + lc;
        const bool ok = pix < npix && (unsigned)ih < (unsigned)Hs && (unsigned)iw < (unsigned)Ws;
        a_goff[i] = !ok ? 0u
                    : pooled ? (((uint32_t)(n + sub) * 2 * Hs + 2 * ih) * 2 * Ws + 2 * iw) * p.Cin + cq * 4  // pixel (2 ih, 2 iw): plane (0, 0)
                             : (((uint32_t)(n + sub) * Hs + ih) * Ws + iw) * p.Cin + cq * 4;
        a_valid |= (ok ? 1u : 0u) << i;
        a_swz |= (unsigned)lds_swz<M16>(BYCOL ? lc : pix) << (2 * i);
    }
    const float* pa_base = p.pre_a ? p.pre_a : p.x;
    const float* pb_base = p.pre_a ? p.pre_b : p.x;
    const uint32_t grp_off = (uint32_t)(smp / p.pre_group) * p.Cin;
    const uint32_t plane_elems = (uint32_t)Hs * Ws * p.Cin * (pooled ? 4u : 1u);  // one depth plane of the input (pooled: at full resolution)

    // measurement switches (dgmr_debug_flags, tools/conv_bench.py --dbg=): 1 = return before the epilogue, 2 = stage only the first halo -
    // the MFMA loop then runs on stale LDS contents; results are garbage, only the timing is meaningful.  0 in every product launch.
    const int dbg = p.reserved1;
    bool staged_once = false;
    // fetch, transform and store one 32-channel halo (latency covered by the other workgroups); kd: depth tap of a 3-D conv
    auto stage_a = [&](int chunk, int kd, uint32_t view_off = 0u) {  // view_off: element offset of a parity plane (pooled mode)
        if ((dbg & 2) && staged_once) return;
        staged_once = true;
        f32x4 ra[APASS];
        const int cb = chunk * CK + cq * 4;
        const int dz = KD == 3 ? kd - 1 : 0;
        const bool kok = cb < p.Cin && (unsigned)(dpl + dz) < (unsigned)p.D;
        const unsigned valid = kok ? a_valid : 0u;
        const uint32_t shift = chunk * CK + dz * (int)plane_elems + view_off;  // wraps consistently for dz = -1
#pragma unroll
        for (int i = 0; i < APASS; ++i)
            ra[i] = *reinterpret_cast<const f32x4*>(p.x + (((valid >> i) & 1u) ? a_goff[i] + shift : 0u));
        const f32x4 rpa = *reinterpret_cast<const f32x4*>(pa_base + ((p.pre_a && kok) ? grp_off + cb : 0u));
        const f32x4 rpb = *reinterpret_cast<const f32x4*>(pb_base + ((p.pre_a && kok) ? grp_off + cb : 0u));
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
            v = ((valid >> i) & 1u) ? v : zero4;
            u32x2 pl[NP];
            split_planes4<NP>(v, pl);
            if (pix < AMAX) {
                uint32_t* dst = As + pix * ROW + (((cq >> 1) ^ (BYCOL ? (int)((a_swz >> (2 * i)) & 3) : lds_swz<M16>(pix))) << 2) + (cq & 1) * 2;
#pragma unroll
                for (int q = 0; q < NP; ++q) *reinterpret_cast<u32x2*>(dst + q * AMAX * ROW) = pl[q];
            }
        }
    };

    // ---- weights: stage s = chunk * 9 + tap, written by LDS-DMA; unit u of a stage = 16 bytes at LDS offset 16 u ----
    const size_t plane_stride = phase ? (size_t)p.Cout * 16 * p.Cin : (size_t)p.Cout * (pooled ? 16 : 9) * p.KD * p.Cin;  // bf16 elements per plane
    // per-lane element offsets of the row / k-slot this lane fills (32 bits: a weight plane is < 2^31 elements); the tap / chunk part of
    // the address is wave-uniform and goes into the scalar base of global_load_lds.  b_tail: the same with k-slots beyond Cin
    // redirected to channel group 0 (only the last chunk of a Cin % 32 != 0 layer uses it)
    uint32_t b_off[BPASS], b_tail[BPASS];
    const int c_last = (nchunks - 1) * CK;
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
        const int u = min(tid + i * 256, BUNITS - 1);  // (lanes of a wave beyond the stage never issue: see dma_b)
        const int plane = PRIV ? i / (TN * 2) : u / (BN * 4);
        const int r = PRIV ? wn * TN * 32 + (i % (TN * 2)) * 16 + (lane >> 2) : (u >> 2) % BN;
        const int ch = (((PRIV ? lane : u) & 3) ^ lds_swz<M16>(r)) * 8;  // first channel (inside a chunk) of the logical k-slot held by physical slot u & 3
        const uint32_t row = (uint32_t)(plane * plane_stride) + (uint32_t)(ph * p.Cout + min(n0 + r, p.Cout - 1)) * (uint32_t)taps * p.Cin;
        b_off[i] = row + ch;
        b_tail[i] = row + (c_last + ch < p.Cin ? c_last + ch : 0);  // absolute channel: the tail's scalar base has no chunk offset
    }
    const bool has_tail = (p.Cin & (CK - 1)) != 0;
    const size_t phase_rows = (size_t)p.Cout * (size_t)taps * p.Cin;  // elements between the weight rows of two consecutive phases (PAIR)
    auto dma_b = [&](int chunk, int tap, int stage, int pxo = 0) {  // tap: 0 .. 9 KD - 1; pxo (PAIR): 1 = the px = 1 phase's tap sums
        const bool tail = has_tail && chunk == nchunks - 1;
        const uint16_t* base = p.w_split + ((size_t)tap * p.Cin + (tail ? 0 : chunk * CK)) + (PAIR ? (size_t)pxo * phase_rows : 0);
#pragma unroll
        for (int i = 0; i < BPASS; ++i) {
            if (PRIV)  // 16 rows x 64 bytes of this wave's own slice: [plane][TN * 32 rows][ROW]
                lds_dma16(base + (tail ? b_tail[i] : b_off[i]), Bs + stage * BSTAGE + wid * WSLICE + i * 16 * ROW);
            else if (i * 256 + wid * 64 < BUNITS)  // wave-uniform: a wave fills 64 consecutive units
                lds_dma16(base + (tail ? b_tail[i] : b_off[i]), Bs + stage * BSTAGE + (i * 256 + wid * 64) * 4);
        }
    };

    accv_t accs[NACC][TM][TN];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < RPB; ++r) accs[a][i][j][r] = 0.f;

    // halo pixel of this lane under each filter row / column: pix = rowpix[dy] + colpix[dx] (the taps are unrolled below)
    int rowpix[TM][3], colpix[TM][3];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int q = wm * TM * MB + i * MB + (lane & (MB - 1));
        const int prow = (q >> tw_shift) & (TH - 1), pcol = q & (TW - 1), pbase = (q >> sub_shift) * HP;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            rowpix[i][d] = pbase + (((h0 + prow + d + py - 1) >> us) - oh) * HTw;  // (py = px = 0 outside the phase mode;
            colpix[i][d] = ((w0 + pcol + d + px - 1) >> us) - ow;                  //  there d = 0, 1 are its two taps per axis)
        }
    }
    const int kg = M16 ? lane >> 4 : lane >> 5;  // this lane's 16-byte k-slot inside a 16-channel step (M16: inside the 32-channel chunk)
    const int bsw = lds_swz<M16>(lane);  // swizzle of this lane's weight rows (row = MB j + (lane & (MB - 1)): block offsets are multiples of 16)
    const uint32_t* Bb0 = PRIV ? Bs + wid * WSLICE + (lane & 31) * ROW : Bs + (wn * TN * MB + (lane & (MB - 1))) * ROW;
    constexpr int BPLANE = PRIV ? TN * 32 : BN;  // rows between the hi and the lo plane of a stage
    // half: the chunk holds <= 16 real channels (Cin = 48, 144: the last chunk) - its second 16-channel step is all zeros, skipped
    const bool tail16 = (p.Cin & (CK - 1)) != 0 && (p.Cin & (CK - 1)) <= 16;
    auto mma_sel = [&](int dyi, int dxi, int stage, bool half, auto acc_sel) {  // acc_sel: which accumulator set (PAIR: the column parity)
        accv_t (&acc)[TM][TN] = accs[decltype(acc_sel)::value];
        const uint32_t* Ab[TM];
        int asw[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int pix = rowpix[i][dyi] + colpix[i][dxi];
            Ab[i] = As + pix * ROW;
            asw[i] = lds_swz<M16>(BYCOL ? colpix[i][dxi] : pix);
        }
        const uint32_t* Bb = Bb0 + stage * BSTAGE;
#pragma unroll
        for (int kk = 0; kk < (M16 ? 1 : CK / 16); ++kk) {
            if (kk == 1 && half) break;  // (wave-uniform)
            const int ks = M16 ? kg : kk * 2 + kg;  // logical 16-byte k-slot of this lane's fragment
            bf16x8_t af[NP][TM], bf[NP][TN];
            const int ob = (ks ^ bsw) << 2;
            __builtin_amdgcn_s_setprio(1);  // (in front of the reads: s_setprio bounds a scheduling region)
            // fragments in the order the products consume them (for_each_product: lowest planes' product first = A's LAST plane with B's
            // first): LDS reads return in order, so the first MFMAs can issue while the later fragments are still on their way - read in
            // plane order they all had to land first (`s_waitcnt lgkmcnt(0)` in front of every 16-channel step; conv_bf16.h)
#pragma unroll
            for (int t = 0; t < NP; ++t) {
                const int qa = NP - 1 - t, qb = t;
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[qa][i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(Ab[i] + qa * AMAX * ROW + ((ks ^ asw[i]) << 2)));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bf[qb][j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(Bb + (qb * BPLANE + j * MB) * ROW + ob));
            }
            for_each_product<NP>([&](auto qa, auto qb) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma_blk<M16>(af[qa][i], bf[qb][j], acc[i][j]);
            });
            schedule_split_products<NP, TM, TN>();  // (conv_bf16.h: the reads of the next plane pair between the MFMAs of a product)
            __builtin_amdgcn_s_setprio(0);
        }
    };
    auto mma = [&](int dyi, int dxi, int stage, bool half) { mma_sel(dyi, dxi, stage, half, std::integral_constant<int, 0>{}); };

    dma_b(0, 0, 0);
    stage_a(0, 0);
    dma_drain();
    __syncthreads();
    const int ngroups = (phase || pooled) ? 0 : nchunks * KD;  // groups of nine taps: (chunk, kd)
    if (pooled) {
        // groups of four taps: (parity plane, chunk [, depth tap]).  Input row 2r - 1 + u, u = 0..3, of output row r: even rows (plane
        // bit 0) are u = 1, 3 = plane rows r, r + 1; odd rows u = 0, 2 = plane rows r - 1, r - hence the window row  a + 1 - parity  of
        // tap a.  KD = 3 (a DBlock's 3x3x3 conv followed by the 2 x 2 spatial part of its AvgPool3d): the same over the three depth
        // planes, 16 tap sums each ([48 taps] = kd * 16 + plane * 4 + tap), walked as virtual chunks vc = chunk * 3 + kd like the plain
        // 3-D loop below; the depth pair average is a separate streaming pass over the small map (dgmr_pool_depth2)
        const int nvc = nchunks * KD;
#pragma unroll
        for (int pl = 0; pl < 4; ++pl) {
            const int pp = pl >> 1, qq = pl & 1;
#pragma unroll 1
            for (int vc = 0; vc < nvc; ++vc) {
                const int chunk = KD == 3 ? vc / 3 : vc, kd = KD == 3 ? vc - chunk * 3 : 0;
                const bool last_vc = vc + 1 == nvc;
                const bool more = !(pl == 3 && last_vc);
                const int nv = last_vc ? 0 : vc + 1, npl = last_vc ? pl + 1 : pl;
                const int nchunk = KD == 3 ? nv / 3 : nv, nkd = KD == 3 ? nv - nchunk * 3 : 0;
#pragma unroll
                for (int tap = 0; tap < 4; ++tap) {
                    const int st = tap & 1;
                    if (tap < 3) dma_b(chunk, kd * 16 + pl * 4 + tap + 1, st ^ 1);
                    else if (more) dma_b(nchunk, nkd * 16 + npl * 4, st ^ 1);
                    mma((tap >> 1) + 1 - pp, (tap & 1) + 1 - qq, st, tail16 && chunk == nchunks - 1);
                    if (tap == 3 && more) {
                        __syncthreads();
                        stage_a(nchunk, nkd, (uint32_t)(((npl >> 1) * 2 * Ws + (npl & 1)) * p.Cin));
                    }
                    dma_drain();
                    if (!PRIV || (tap == 3 && more)) __syncthreads();
                }
            }
        }
    }
    if (phase && PAIR) {
        if constexpr (PAIR) {
#pragma unroll 1
            for (int chunk = 0; chunk < nchunks; ++chunk) {  // eight taps per chunk (px = 0: taps 0 .. 3, px = 1: taps 0 .. 3), one halo
                const bool more = chunk + 1 < nchunks;
                const bool half = tail16 && !more;
                auto step = [&](auto vc) {
                    constexpr int v = decltype(vc)::value, pxo = v >> 2, tap = v & 3, st = v & 1;
                    if (v < 7) dma_b(chunk, (v + 1) & 3, st ^ 1, (v + 1) >> 2);
                    else if (more) dma_b(chunk + 1, 0, st ^ 1, 0);
                    mma_sel(tap >> 1, (tap & 1) + pxo, st, half, std::integral_constant<int, pxo>{});
                    if (v == 7 && more) {
                        __syncthreads();
                        stage_a(chunk + 1, 0);
                    }
                    dma_drain();
                    __syncthreads();
                };
                step(std::integral_constant<int, 0>{});
                step(std::integral_constant<int, 1>{});
                step(std::integral_constant<int, 2>{});
                step(std::integral_constant<int, 3>{});
                step(std::integral_constant<int, 4>{});
                step(std::integral_constant<int, 5>{});
                step(std::integral_constant<int, 6>{});
                step(std::integral_constant<int, 7>{});
            }
        }
    } else if (phase) {
#pragma unroll 1
        for (int chunk = 0; chunk < nchunks; ++chunk) {  // four taps per chunk: the stage parity restarts with every chunk
            const bool more = chunk + 1 < nchunks;
#pragma unroll
            for (int tap = 0; tap < 4; ++tap) {
                const int st = tap & 1;
                if (tap < 3) dma_b(chunk, tap + 1, st ^ 1);
                else if (more) dma_b(chunk + 1, 0, st ^ 1);
                mma(tap >> 1, tap & 1, st, tail16 && !more);
                if (tap == 3 && more) {
                    __syncthreads();
                    stage_a(chunk + 1, 0);
                }
                dma_drain();
                if (!PRIV || (tap == 3 && more)) __syncthreads();
            }
        }
    }
#pragma unroll 1
    for (int g = 0; g < ngroups; ++g) {
        const int chunk = KD == 3 ? g / 3 : g;
        const int kd = KD == 3 ? g - chunk * 3 : 0;
        const bool more = g + 1 < ngroups;
        const int nchunk = (KD == 3 && kd < 2) ? chunk : chunk + 1, nkd = (KD == 3 && kd < 2) ? kd + 1 : 0;  // the next group
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int st = (g + tap) & 1;  // stage of s = 9 g + tap
            // the next stage's DMA is in flight under this tap's MFMAs
            if (tap < 8) dma_b(chunk, kd * 9 + tap + 1, st ^ 1);
            else if (more) dma_b(nchunk, nkd * 9, st ^ 1);
            mma(tap / 3, tap % 3, st, tail16 && chunk == nchunks - 1);
            if (tap == 8 && more) {
                __syncthreads();  // every wave is done with this group's halo
                stage_a(nchunk, nkd);
            }
            dma_drain();  // the stage written under this tap's MFMAs is read by every wave after the barrier
            if (!PRIV || (tap == 8 && more)) __syncthreads();
        }
    }

    // the epilogue of ONE accumulator set (PAIR: called once per column parity; ph / px shadow the workgroup's own)
    auto epilogue = [&](accv_t (&acc)[TM][TN], const int ph, const int px) {
        // ---- epilogue (conv3x3_win_kernel's) ----
        if (dbg & 1) {  // (the accumulators must stay live: a store that never happens for finite sums)
            float t = 0.f;
    #pragma unroll
            for (int i = 0; i < TM; ++i)
    #pragma unroll
                for (int j = 0; j < TN; ++j) t += acc[i][j][0];
            if (t == 123456.789f) p.y[0] = t;
            return;
        }
        const float sc = p.scale ? p.scale[smp / p.scale_group] : 1.f;
        const int emode = p.epi_mode;
        const int pshift = phase ? 1 : 0, oH = p.H << pshift, oW = p.W << pshift;  // the output map (phase mode: twice the input's)
        // ---- 16-byte epilogue (p.reserved1 & 4: Cout % 4 == 0 and every tensor it touches is 16-byte aligned; set by the library) ----
        // The accumulator blocks hold one COLUMN per lane (16 rows of a 32 x 32 block in 16 registers): a lane-per-column epilogue issues
        // one 4-byte store (and one 4-byte load per fused operand) per row and block - 96 stores per lane on a 256 x 96 tile, which is
        // what the epilogue's time went into (measured by switching it off, tools/r3_probe.sh: 12 ... 38 % of a launch).  Here every 4 x 4
        // patch (4 consecutive rows in 4 registers x the 4 lanes of a quad) is transposed across the quad with DPP moves, after which a
        // lane holds 4 consecutive CHANNELS of one pixel: a quarter of the memory instructions, each 16 bytes wide.  Same arithmetic per
        // element, in the same order, as the lane-per-column path below (kept for Cout % 4 != 0 / unaligned views): bit-identical outputs.
        if (p.reserved1 & 4) {
            const int j4 = lane & 3;                                  // row of the 4 x 4 patch this lane ends up with
            const int q4 = (lane & (MB - 1)) >> 2;                    // its column quad inside a block
            const int rsel = M16 ? lane >> 4 : lane >> 5;             // which rows of the block this lane group holds
            constexpr int NG = RPB / 4;                               // 4-row register groups per block
            // ConvGRU epilogues: fetch a row group's operands ahead of its patches (128-pixel tiles: what the recurrent steps run; the
            // 256-pixel tile with 96 columns has no registers for it and no ConvGRU launch of the step uses it)
            constexpr bool GRU_AHEAD = BM == 128;
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f}, one4 = {1.f, 1.f, 1.f, 1.f};
            // output pixel (row of the [pixels][Cout] matrix) of this lane in each of its TM x NG row groups; the residual's when it is
            // at half resolution
            int mpix[TM][NG], rpix[TM][NG];
    #pragma unroll
            for (int i = 0; i < TM; ++i)
    #pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int q = M16 ? wm * TM * 16 + i * 16 + 4 * rsel + j4 : wm * TM * 32 + i * 32 + j4 + 8 * g + 4 * rsel;
                    const int ni = n + (q >> sub_shift);
                    const int hh = ((h0 + ((q >> tw_shift) & (TH - 1))) << pshift) + py, ww = ((w0 + (q & (TW - 1))) << pshift) + px;
                    mpix[i][g] = (ni * oH + hh) * oW + ww;
                    rpix[i][g] = p.residual_up ? (ni * (oH >> 1) + (hh >> 1)) * (oW >> 1) + (ww >> 1) : mpix[i][g];
                }
            const bool want_stats_v = p.stats_out != nullptr && emode == DGMR_EPI_PLAIN;
            float* red = reinterpret_cast<float*>(smem);  // [WM][BN][2]
            if (want_stats_v) __syncthreads();  // (wave-uniform) every wave is done with the operand images before `red` overwrites them
    #pragma unroll
            for (int j = 0; j < TN; ++j) {  // one column block at a time: its per-column operands and sums stay in a few registers
                const int col4 = n0 + wn * TN * MB + j * MB + 4 * q4;
                const bool cok = col4 < p.Cout;  // (Cout % 4 == 0: the whole quad of columns is in or out)
                if (emode == DGMR_EPI_GRU_GATES2) {
                    // read and update gate of a ConvGRU step in one launch: columns [0, C) are the read gate's (pre_out, y = sigmoid * h),
                    // [C, 2C) the update gate's (y2 = pre-activation); every tensor has C channels per pixel (C % 4 == 0: no quad straddles)
                    const int C = p.gru_split;
                    const bool is2 = col4 >= C;
                    const int c0 = cok ? (is2 ? col4 - C : col4) : 0;
                    const float* bp = is2 ? p.bias2 : p.bias;
                    const float* ap = is2 ? p.addend2 : p.addend;
                    const f32x4 b4 = bp ? *reinterpret_cast<const f32x4*>(bp + c0) : zero4;
                    const float scj = is2 ? (p.scale2 ? p.scale2[smp / p.scale_group] : 1.f) : sc;
    #pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        // (operands of the row group's four patches first, then the patches: see the plain path below)
                        f32x4 ad[NG], hvs[NG];
                        if constexpr (GRU_AHEAD) {
    #pragma unroll
                            for (int g = 0; g < NG; ++g) ad[g] = ap ? *reinterpret_cast<const f32x4*>(ap + ((size_t)mpix[i][g] * C + c0)) : zero4;
    #pragma unroll
                            for (int g = 0; g < NG; ++g) hvs[g] = is2 ? zero4 : *reinterpret_cast<const f32x4*>(p.gru_h + ((size_t)mpix[i][g] * C + c0));
                        }
    #pragma unroll
                        for (int g = 0; g < NG; ++g) {
                            const size_t off = (size_t)mpix[i][g] * C + c0;
                            if constexpr (!GRU_AHEAD) {
                                ad[g] = ap ? *reinterpret_cast<const f32x4*>(ap + off) : zero4;
                                hvs[g] = is2 ? zero4 : *reinterpret_cast<const f32x4*>(p.gru_h + off);
                            }
                            f32x4 v = quad_transpose(acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3], lane);
                            if (ap) v += ad[g];
    #pragma unroll
                            for (int c = 0; c < 4; ++c) v[c] = fmaf(v[c], scj, b4[c]);
                            if (!cok) continue;
                            if (is2) {
                                *reinterpret_cast<f32x4*>(p.y2 + off) = v;
                            } else {
                                f32x4 o;
    #pragma unroll
                                for (int c = 0; c < 4; ++c) o[c] = sigmoid_(v[c]) * hvs[g][c];
                                if (p.pre_out) *reinterpret_cast<f32x4*>(p.pre_out + off) = v;
                                *reinterpret_cast<f32x4*>(p.y + off) = o;
                            }
                        }
                    }
                    continue;
                }
                const int cc = cok ? col4 : 0;
                const f32x4 b4 = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + cc) : zero4;
                f32x4 ma4 = one4, mb4 = zero4;
                if (p.mask_a) {
                    const size_t g = (size_t)(smp / p.mask_group) * p.Cout + cc;
                    ma4 = *reinterpret_cast<const f32x4*>(p.mask_a + g);
                    mb4 = *reinterpret_cast<const f32x4*>(p.mask_b + g);
                }
                f32x4 s0 = zero4, s1 = zero4;
                // (round 6) The fused operands of a row group are fetched UP FRONT, four patches at a time, and the patches are then finished
                // and stored with no memory wait in between.  The loop of rounds 3 - 5 loaded each patch's residual / mask source inside the
                // patch and the compiler's one `s_waitcnt vmcnt(0)` per patch - placed at the join of the operand branches, so it ran even
                // when a launch had no operand at all - also waited for the PREVIOUS patch's store to be acknowledged: 24 store round trips
                // per wave and tile, one after the other; that serial chain, not the transposes, was the "12 - 38 % of a launch" the epilogue
                // cost (tools/isa_outline.py; profiles/r06_epilogue_hoisted_loads_ab.log).  Same arithmetic per element, in the same
                // order: bit-identical outputs.
                if (emode == DGMR_EPI_PLAIN && !p.addend && !p.residual && !p.mask_src) {
                    // no fused operand (the first convs of the G-blocks, every launch of a no-grad forward that has no shortcut to add): a
                    // path WITHOUT any load, hence without any wait - the 24 stores of a wave stream out back to back
    #pragma unroll
                    for (int i = 0; i < TM; ++i) {
    #pragma unroll
                        for (int g = 0; g < NG; ++g) {
                            const size_t off = (size_t)mpix[i][g] * p.Cout + cc;
                            f32x4 o = quad_transpose(acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3], lane);
    #pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                o[c] = fmaf(o[c], sc, b4[c]);
                                if (p.act_relu) o[c] = fmaxf(o[c], 0.f);
                            }
                            if (cok) *reinterpret_cast<f32x4*>(p.y + off) = o;
                            if (want_stats_v) {
    #pragma unroll
                                for (int c = 0; c < 4; ++c) {
                                    s0[c] += o[c];
                                    s1[c] = fmaf(o[c], o[c], s1[c]);
                                }
                            }
                        }
                    }
                } else if (emode == DGMR_EPI_PLAIN) {
    #pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        f32x4 ad[NG], rs[NG], ms[NG];
    #pragma unroll
                        for (int g = 0; g < NG; ++g) ad[g] = p.addend ? *reinterpret_cast<const f32x4*>(p.addend + ((size_t)mpix[i][g] * p.Cout + cc)) : zero4;
    #pragma unroll
                        for (int g = 0; g < NG; ++g) rs[g] = p.residual ? *reinterpret_cast<const f32x4*>(p.residual + (size_t)rpix[i][g] * p.Cout + cc) : zero4;
    #pragma unroll
                        for (int g = 0; g < NG; ++g) ms[g] = p.mask_src ? *reinterpret_cast<const f32x4*>(p.mask_src + ((size_t)mpix[i][g] * p.Cout + cc)) : zero4;
    #pragma unroll
                        for (int g = 0; g < NG; ++g) {
                            const size_t off = (size_t)mpix[i][g] * p.Cout + cc;
                            f32x4 v = quad_transpose(acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3], lane);
                            if (p.addend) v += ad[g];
    #pragma unroll
                            for (int c = 0; c < 4; ++c) v[c] = fmaf(v[c], sc, b4[c]);
                            f32x4 o = v;
    #pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                if (p.act_relu) o[c] = fmaxf(o[c], 0.f);
                                if (p.residual) o[c] += rs[g][c];
                                if (p.mask_src) o[c] = fmaf(ms[g][c], ma4[c], mb4[c]) > 0.f ? o[c] : 0.f;
                            }
                            if (cok) *reinterpret_cast<f32x4*>(p.y + off) = o;
                            if (want_stats_v) {
    #pragma unroll
                                for (int c = 0; c < 4; ++c) {
                                    s0[c] += o[c];
                                    s1[c] = fmaf(o[c], p.mask_src ? ms[g][c] : o[c], s1[c]);
                                }
                            }
                        }
                    }
                } else {
    #pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        // ConvGRU step: pre_out = v; gate: y = sigmoid(v) * h; blend: y = s*h + (1-s)*relu(v), s = sigmoid(pu)
                        f32x4 ad[NG], hvs[NG], pvs[NG];
                        if constexpr (GRU_AHEAD) {
    #pragma unroll
                            for (int g = 0; g < NG; ++g) ad[g] = p.addend ? *reinterpret_cast<const f32x4*>(p.addend + ((size_t)mpix[i][g] * p.Cout + cc)) : zero4;
    #pragma unroll
                            for (int g = 0; g < NG; ++g) hvs[g] = *reinterpret_cast<const f32x4*>(p.gru_h + ((size_t)mpix[i][g] * p.Cout + cc));
    #pragma unroll
                            for (int g = 0; g < NG; ++g) pvs[g] = emode == DGMR_EPI_GRU_BLEND ? *reinterpret_cast<const f32x4*>(p.gru_pu + ((size_t)mpix[i][g] * p.Cout + cc)) : zero4;
                        }
    #pragma unroll
                        for (int g = 0; g < NG; ++g) {
                            const size_t off = (size_t)mpix[i][g] * p.Cout + cc;
                            if constexpr (!GRU_AHEAD) {
                                ad[g] = p.addend ? *reinterpret_cast<const f32x4*>(p.addend + off) : zero4;
                                hvs[g] = *reinterpret_cast<const f32x4*>(p.gru_h + off);
                                pvs[g] = emode == DGMR_EPI_GRU_BLEND ? *reinterpret_cast<const f32x4*>(p.gru_pu + off) : zero4;
                            }
                            f32x4 v = quad_transpose(acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3], lane);
                            if (p.addend) v += ad[g];
    #pragma unroll
                            for (int c = 0; c < 4; ++c) v[c] = fmaf(v[c], sc, b4[c]);
                            const f32x4 hv = hvs[g], pv = pvs[g];
                            f32x4 o;
    #pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                if (emode == DGMR_EPI_GRU_BLEND) {
                                    const float sg = sigmoid_(pv[c]);
                                    o[c] = sg * hv[c] + (1.f - sg) * fmaxf(v[c], 0.f);
                                } else {
                                    o[c] = sigmoid_(v[c]) * hv[c];
                                }
                            }
                            if (cok) {
                                if (p.pre_out) *reinterpret_cast<f32x4*>(p.pre_out + off) = v;
                                *reinterpret_cast<f32x4*>(p.y + off) = o;
                            }
                        }
                    }
                }
                if (want_stats_v) {  // per-column sums: fold the 4 rows of the quad and the row groups of the wave (the WM waves: below)
    #pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float a = s0[c], b = s1[c];
                        a += quad_xor1(a);
                        b += quad_xor1(b);
                        a += quad_xor2(a);
                        b += quad_xor2(b);
                        a += __shfl_xor(a, 32, 64);
                        b += __shfl_xor(b, 32, 64);
                        if (M16) {
                            a += __shfl_xor(a, 16, 64);
                            b += __shfl_xor(b, 16, 64);
                        }
                        if ((lane & (M16 ? 0x33 : 0x23)) == 0) {
                            const int cl = wn * TN * MB + j * MB + 4 * q4 + c;
                            red[(wm * BN + cl) * 2 + 0] = a;
                            red[(wm * BN + cl) * 2 + 1] = b;
                        }
                    }
                }
            }
            if (want_stats_v) {
                __syncthreads();
                for (int idx = tid; idx < BN * 2; idx += 256) {
                    const int cl = idx >> 1, which = idx & 1;
                    float v = 0.f;
    #pragma unroll
                    for (int qq = 0; qq < WM; ++qq) v += red[(qq * BN + cl) * 2 + which];
                    const size_t srow = phase ? (size_t)tile * 4 + ph : (size_t)tile;  // (a tile's four phases: consecutive rows)
                    if (n0 + cl < p.Cout) p.stats_out[(srow * 2 + which) * p.Cout + n0 + cl] = v;
                }
            }
            // tail probe (dgmr_debug_flags 64 / 128: sleep ~3.4 / ~6.8 us after the last store was ISSUED): a wave cannot retire before its
            // stores are acknowledged; if a launch does not get slower with the sleep, that wait is at least as long
            if (dbg & 64) __builtin_amdgcn_s_sleep(127);
            if (dbg & 128) {
                __builtin_amdgcn_s_sleep(127);
                __builtin_amdgcn_s_sleep(127);
            }
            return;
        }
        float bj[TN];
        int colj[TN];
    #pragma unroll
        for (int j = 0; j < TN; ++j) {
            colj[j] = n0 + wn * TN * MB + j * MB + (lane & (MB - 1));
            bj[j] = (p.bias && colj[j] < p.Cout) ? p.bias[colj[j]] : 0.f;
        }
        float maj[TN], mbj[TN];  // affine of the BatchNorm whose relu is being back-propagated through (data gradient), per column
    #pragma unroll
        for (int j = 0; j < TN; ++j) {
            const bool on = p.mask_a && colj[j] < p.Cout;
            const size_t g = (size_t)(smp / p.mask_group) * p.Cout + (on ? colj[j] : 0);
            maj[j] = on ? p.mask_a[g] : 1.f;
            mbj[j] = on ? p.mask_b[g] : 0.f;
        }
        // Every variant is straight-line per output row: the loads of a row (addend, ConvGRU state, residual, mask source) are issued
        // together, unconditionally, on clamped addresses.  (The element-wise generic epilogue with its per-element divisions and
        // dependent loads made a ConvGRU step conv spend as long in its epilogue as in its 18 taps.)
        const int cmax = p.Cout - 1;
        // BatchNorm statistics of the OUTPUT for the next layer, taken here (stats_out): per column sum y and sum y^2 (data gradient
        // through relu(BatchNorm(x)), i.e. with mask_src: sum y and sum y * x, the two sums of BatchNorm's backward) over this lane's
        // 16 TM rows, folded over the two lane halves, the WM waves (LDS) and written as ONE row of partials per workgroup tile
        float st0[TN], st1[TN];
    #pragma unroll
        for (int j = 0; j < TN; ++j) st0[j] = st1[j] = 0.f;
        const bool want_stats = p.stats_out != nullptr && emode == DGMR_EPI_PLAIN;
    #pragma unroll
        for (int i = 0; i < TM; ++i) {
    #pragma unroll
            for (int r = 0; r < RPB; ++r) {
                // accumulator register r of this lane: block row (r & 3) + 8 (r >> 2) + 4 (lane >> 5) [32 x 32] / 4 (lane >> 4) + r [16 x 16]
                const int q = M16 ? wm * TM * 16 + i * 16 + 4 * (lane >> 4) + r
                                  : wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int ni = n + (q >> sub_shift);
                const int hh = ((h0 + ((q >> tw_shift) & (TH - 1))) << pshift) + py, ww = ((w0 + (q & (TW - 1))) << pshift) + px;
                const size_t mrow = (size_t)((ni * oH + hh) * oW + ww) * p.Cout;
                const size_t rrow = p.residual_up ? (((size_t)ni * (oH >> 1) + (hh >> 1)) * (oW >> 1) + (ww >> 1)) * p.Cout : mrow;
                float v[TN];
    #pragma unroll
                for (int j = 0; j < TN; ++j) v[j] = acc[i][j][r];
                if (p.addend) {
    #pragma unroll
                    for (int j = 0; j < TN; ++j) v[j] += p.addend[mrow + min(colj[j], cmax)];
                }
    #pragma unroll
                for (int j = 0; j < TN; ++j) v[j] = fmaf(v[j], sc, bj[j]);
                if (emode == DGMR_EPI_PLAIN) {
                    float rs[TN], ms[TN];
                    if (p.residual) {
    #pragma unroll
                        for (int j = 0; j < TN; ++j) rs[j] = p.residual[rrow + min(colj[j], cmax)];
                    }
                    if (p.mask_src) {
    #pragma unroll
                        for (int j = 0; j < TN; ++j) ms[j] = p.mask_src[mrow + min(colj[j], cmax)];
                    }
    #pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        float o = v[j];
                        if (p.act_relu) o = fmaxf(o, 0.f);
                        if (p.residual) o += rs[j];
                        if (p.mask_src) o = fmaf(ms[j], maj[j], mbj[j]) > 0.f ? o : 0.f;
                        if (colj[j] < p.Cout) p.y[mrow + colj[j]] = o;
                        if (want_stats) {
                            st0[j] += o;
                            st1[j] = fmaf(o, p.mask_src ? ms[j] : o, st1[j]);
                        }
                    }
                } else {  // ConvGRU step: pre_out = v; gate: y = sigmoid(v) * h; blend: y = s*h + (1-s)*relu(v), s = sigmoid(pu)
                    float hv[TN], pv[TN];
    #pragma unroll
                    for (int j = 0; j < TN; ++j) hv[j] = p.gru_h[mrow + min(colj[j], cmax)];
                    if (emode == DGMR_EPI_GRU_BLEND) {
    #pragma unroll
                        for (int j = 0; j < TN; ++j) pv[j] = p.gru_pu[mrow + min(colj[j], cmax)];
                    }
    #pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        float o;
                        if (emode == DGMR_EPI_GRU_BLEND) {
                            const float sg = sigmoid_(pv[j]);
                            o = sg * hv[j] + (1.f - sg) * fmaxf(v[j], 0.f);
                        } else {
                            o = sigmoid_(v[j]) * hv[j];
                        }
                        if (colj[j] < p.Cout) {
                            if (p.pre_out) p.pre_out[mrow + colj[j]] = v[j];
                            p.y[mrow + colj[j]] = o;
                        }
                    }
                }
            }
        }
        if (want_stats) {  // (wave-uniform)
            if (PRIV) __syncthreads();  // no barrier since the last halo: other waves may still be reading the LDS
            float* red = reinterpret_cast<float*>(smem);  // [WM][BN][2]
    #pragma unroll
            for (int j = 0; j < TN; ++j) {
                st0[j] += __shfl_xor(st0[j], 32, 64);
                st1[j] += __shfl_xor(st1[j], 32, 64);
                if (M16) {  // four lane groups hold the same column
                    st0[j] += __shfl_xor(st0[j], 16, 64);
                    st1[j] += __shfl_xor(st1[j], 16, 64);
                }
                if (lane < MB) {
                    const int cl = wn * TN * MB + j * MB + lane;
                    red[(wm * BN + cl) * 2 + 0] = st0[j];
                    red[(wm * BN + cl) * 2 + 1] = st1[j];
                }
            }
            __syncthreads();
            for (int idx = tid; idx < BN * 2; idx += 256) {
                const int cl = idx >> 1, which = idx & 1;
                float v = 0.f;
    #pragma unroll
                for (int q = 0; q < WM; ++q) v += red[(q * BN + cl) * 2 + which];
                const size_t srow = phase ? (size_t)tile * 4 + ph : (size_t)tile;  // (a tile's four phases: consecutive rows)
                if (n0 + cl < p.Cout) p.stats_out[(srow * 2 + which) * p.Cout + n0 + cl] = v;
            }
        }
    };
    if constexpr (PAIR) {
        epilogue(accs[0], ph, 0);
        __syncthreads();  // (the statistics rows of the first parity have been read out of `red`)
        epilogue(accs[1], ph + 1, 1);
    } else {
        epilogue(accs[0], ph, px);
    }
}

}  // namespace
