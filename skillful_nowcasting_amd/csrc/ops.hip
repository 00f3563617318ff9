// HBM-bound kernels of the DGMR step for gfx950: spectral-norm power iteration, BatchNorm statistics and
// backward, pooling / space-to-depth layout moves, ConvGRU gating, latent attention, discriminator heads,
// losses and Adam.  All of them are bandwidth- or latency-bound (no MFMA): coalesced, 16-byte vectorised
// where the layout allows, per-channel reductions accumulate in double so that E[x^2]-E[x]^2 stays accurate.
#include <stdarg.h>

#include <algorithm>
#include <cmath>

#include "common.h"

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void dgmr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* dgmr_last_error(void) { return g_err; }
extern "C" int dgmr_abi_version(void) { return DGMR_ABI_VERSION; }
int g_deterministic = 0;
extern "C" int dgmr_set_deterministic(int on) {
    g_deterministic = on ? 1 : 0;
    return 0;
}
extern "C" int dgmr_get_deterministic(void) { return g_deterministic; }

namespace {

constexpr int EW_THREADS = 256;
inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
inline int ew_blocks(int64_t n_items) { return (int)std::min<int64_t>((n_items + EW_THREADS - 1) / EW_THREADS, 256 * 16); }

#define GRID_STRIDE(i, n) \
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// ------------------------------------------------------------------------------------------------
// spectral norm
// ------------------------------------------------------------------------------------------------
// t[i] = sum_k w[i][k] * v[perm(k)].  (Round 5: the norms / dots over t that used to meet in float atomics on `scratch` are recomputed
// by their consumers in a fixed order - the first, traced forward of a module went through these kernels and was the last source of
// run-to-run differences; `scratch` stays in the signatures and is left untouched.)
__global__ void sn_rows_kernel(const float* __restrict__ w, const float* __restrict__ v, float* __restrict__ t, int K, int Cin, int taps) {
    __shared__ float red[32];
    const int i = blockIdx.x;
    const float* wr = w + (size_t)i * K;
    float s = 0.f;
    for (int k = threadIdx.x * 4; k < K; k += blockDim.x * 4) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + k);
        const int tp = k / Cin, ci = k - tp * Cin;
#pragma unroll
        for (int j = 0; j < 4; ++j) s = fmaf(wv[j], v[(size_t)(ci + j) * taps + tp], s);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) t[i] = s;
}

// sum_i a[i] * b[i] over n values, the same fixed order in every workgroup that calls it (strided per thread, then block_sum)
__device__ __forceinline__ float sn_dot_fixed(const float* __restrict__ a, const float* __restrict__ b, int n, float* red) {
    float q = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) q = fmaf(a[i], b[i], q);
    return block_sum(q, red);
}

// s[k] = sum_i w[i][k] * t[i] / max(||t||, eps); block 0 also writes u = t / max(||t||, eps); scratch[1] += s^2
__global__ void sn_cols_kernel(const float* __restrict__ w, const float* __restrict__ t, float* __restrict__ s_out,
                               float* __restrict__ u, float* __restrict__ u_save, float* __restrict__ scratch, int Cout, int K,
                               float eps) {
    __shared__ float part[4][64];
    __shared__ float red[32];
    const float inv = 1.f / fmaxf(sqrtf(sn_dot_fixed(t, t, Cout, red)), eps);
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + c;
    float s = 0.f;
    if (k < K)
        for (int i = rg; i < Cout; i += 4) s = fmaf(w[(size_t)i * K + k], t[i] * inv, s);
    part[rg][c] = s;
    __syncthreads();
    if (rg == 0 && k < K) s_out[k] = part[0][c] + part[1][c] + part[2][c] + part[3][c];
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < Cout; i += blockDim.x) {
            const float ui = t[i] * inv;
            u[i] = ui;
            if (u_save) u_save[i] = ui;
        }
}

// v[perm(k)] = s[k]/max(||s||,eps); inv_sigma = max(||s||,eps)/||s||^2   (sigma = u^T W v = ||s||^2 / max(||s||, eps))
__global__ void sn_finish_train_kernel(const float* __restrict__ s, float* __restrict__ v, float* __restrict__ v_save,
                                       float* __restrict__ inv_sigma, const float* __restrict__ scratch, int K, int Cin, int taps,
                                       float eps) {
    __shared__ float red[32];
    const float n2 = sn_dot_fixed(s, s, K, red);  // ||s||^2, recomputed by every workgroup in the same order
    const float d = fmaxf(sqrtf(n2), eps);
    const float inv = 1.f / d;
    GRID_STRIDE(k, K) {
        const int tp = k / Cin, ci = k - tp * Cin;
        const float vk = s[k] * inv;
        const size_t j = (size_t)ci * taps + tp;
        v[j] = vk;
        if (v_save) v_save[j] = vk;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) inv_sigma[0] = d / n2;
}

__global__ void sn_finish_eval_kernel(const float* __restrict__ u, const float* __restrict__ v, float* __restrict__ u_save,
                                      float* __restrict__ v_save, float* __restrict__ inv_sigma,
                                      const float* __restrict__ t, int Cout, int K) {
    __shared__ float red[32];
    const float sigma = sn_dot_fixed(u, t, Cout, red);  // u^T W v
    GRID_STRIDE(i, (int64_t)Cout + K) {
        if (i < Cout) {
            if (u_save) u_save[i] = u[i];
        } else if (v_save)
            v_save[i - Cout] = v[i - Cout];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) inv_sigma[0] = 1.f / sigma;
}


// ---- T calls of one spectral-norm module in one go ---------------------------------------------------------
// The reference runs one power iteration per module CALL (parametrizations.py:512-513); a sampler / discriminator
// conv is called once per forecast step / frame, i.e. T dependent iterations on the same W.  With A = W W^T
// ([Cout][Cout], cached per weight version) the chain never touches W again:
//     s_t = W^T u_t,  n_t^2 = u_t^T A u_t,  d_t = max(n_t, eps),  v_t = s_t / d_t,  sigma_t = n_t^2 / d_t,
//     u_{t+1} = W v_t / max(||W v_t||, eps) = (A u_t / d_t) / max(||A u_t|| / d_t, eps)
// so W is read twice per forward (t0 = W v_0 and S = W^T [u_1..u_T]) instead of 2T times.
constexpr int SN_TMAX = 32;

// one workgroup: u_1 from t0, then T mat-vecs with A.  u_hist [T][Cout], dnorm [T], inv_sigma [T]; u <- u_T.
__global__ __launch_bounds__(1024) void sn_gram_chain_kernel(const float* __restrict__ A, const float* __restrict__ t0,
                                                             const float* __restrict__ scratch, float* __restrict__ u,
                                                             float* __restrict__ u_hist, float* __restrict__ dnorm,
                                                             float* __restrict__ inv_sigma, int Cout, int T, float eps) {
    extern __shared__ float sh[];  // ucur[Cout] | y[Cout] | red[32]
    float* ucur = sh;
    float* y = sh + Cout;
    float* red = sh + 2 * Cout;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    {
        const float inv = 1.f / fmaxf(sqrtf(sn_dot_fixed(t0, t0, Cout, red)), eps);  // (was scratch[0]: float atomics of sn_rows_kernel)
        for (int i = threadIdx.x; i < Cout; i += blockDim.x) ucur[i] = t0[i] * inv;
    }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        float part = 0.f;  // this wave's share of u^T A u
        float part2 = 0.f;  // and of ||A u||^2
        for (int i = wid; i < Cout; i += nw) {
            const float* ar = A + (size_t)i * Cout;
            float s = 0.f;
            for (int j = lane; j < Cout; j += 64) s = fmaf(ar[j], ucur[j], s);
            s = wave_sum(s);
            if (lane == 0) {
                y[i] = s;
                part = fmaf(ucur[i], s, part);
                part2 = fmaf(s, s, part2);
            }
        }
        // block reductions of part / part2 (lane 0 of each wave holds them)
        __syncthreads();
        if (lane == 0) {
            red[wid] = part;
            red[16 + wid] = part2;
        }
        __syncthreads();
        float n2 = 0.f, y2 = 0.f;
        for (int k = 0; k < nw; ++k) {
            n2 += red[k];
            y2 += red[16 + k];
        }
        const float d = fmaxf(sqrtf(fmaxf(n2, 0.f)), eps);
        for (int i = threadIdx.x; i < Cout; i += blockDim.x) u_hist[(size_t)t * Cout + i] = ucur[i];
        if (threadIdx.x == 0) {
            dnorm[t] = d;
            inv_sigma[t] = d / n2;
        }
        __syncthreads();
        if (t + 1 < T) {
            // u_{t+1} = (y / d) / max(||y|| / d, eps)
            const float invd = 1.f / d;
            const float inv = invd / fmaxf(sqrtf(y2) * invd, eps);
            for (int i = threadIdx.x; i < Cout; i += blockDim.x) ucur[i] = y[i] * inv;
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < Cout; i += blockDim.x) u[i] = ucur[i];
}

// S[t][k] = sum_i W[i][k] u_hist[t][i] for all t in one pass over W; v_hist[t][perm(k)] = S / d_t; v <- v_hist[T-1].
__global__ void sn_cols_seq_kernel(const float* __restrict__ w, const float* __restrict__ u_hist, const float* __restrict__ dnorm,
                                   float* __restrict__ v, float* __restrict__ v_hist, int Cout, int K, int Cin, int taps, int T) {
    __shared__ float part[4][SN_TMAX][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + c;
    float acc[SN_TMAX];
#pragma unroll
    for (int t = 0; t < SN_TMAX; ++t) acc[t] = 0.f;
    if (k < K)
        for (int i = rg; i < Cout; i += 4) {
            const float wv = w[(size_t)i * K + k];
#pragma unroll
            for (int t = 0; t < SN_TMAX; ++t)
                if (t < T) acc[t] = fmaf(wv, u_hist[(size_t)t * Cout + i], acc[t]);
        }
#pragma unroll
    for (int t = 0; t < SN_TMAX; ++t) part[rg][t][c] = acc[t];
    __syncthreads();
    if (k < K) {
        const int tp = k / Cin, ci = k - tp * Cin;
        const size_t j = (size_t)ci * taps + tp;
        for (int t = rg; t < T; t += 4) {
            const float s = (part[0][t][c] + part[1][t][c] + part[2][t][c] + part[3][t][c]) / dnorm[t];
            v_hist[(size_t)t * K + j] = s;
            if (t == T - 1) v[j] = s;
        }
    }
}

// ---- the same for MANY modules in three launches (all spectral-norm work of one generator / discriminator forward) ----
// Power iterations do not depend on activations, so a forward draws every module's call sequence up front: block -> module
// through the descriptors' block prefix sums; the chains of all modules run concurrently, one workgroup each.
__device__ __forceinline__ int sn_find(const dgmr_sn_desc* __restrict__ d, int n, int blk, bool rows) {
    int m = 0;
    while (m + 1 < n && blk >= (rows ? d[m + 1].row_block0 : d[m + 1].col_block0)) ++m;
    return m;
}

__global__ void sn_rows_multi_kernel(const dgmr_sn_desc* __restrict__ descs, int n, float* __restrict__ arena) {
    __shared__ float red[32];
    const int m = sn_find(descs, n, blockIdx.x, true);
    const dgmr_sn_desc d = descs[m];
    const int i = blockIdx.x - d.row_block0;
    const int K = d.Cin * d.taps;
    const float* wr = d.w + (size_t)i * K;
    float s = 0.f;
    for (int k = threadIdx.x * 4; k < K; k += blockDim.x * 4) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + k);
        const int tp = k / d.Cin, ci = k - tp * d.Cin;
#pragma unroll
        for (int j = 0; j < 4; ++j) s = fmaf(wv[j], d.v[(size_t)(ci + j) * d.taps + tp], s);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) arena[d.tmp_off + i] = s;
}

// One power iteration of EVERY module per launch.  The chain u_{t+1} ~ (W W^T) u_t is sequential in t, but one workgroup per module
// (the round-1 design) read a 1536 x 1536 Gram matrix through a single CU at ~30 GB/s: 0.3 ms per iteration, 1.4 ms per
// discriminator forward, all on the step's critical path.  Here launch t computes y_t = A u_t for all modules with 32 rows of A per
// workgroup (hundreds of workgroups, every A from L2 at full rate); the quantities that need the WHOLE vector - ||y||, u^T y - are
// recomputed at the start of launch t+1 by every workgroup of the module from y_{t-1} and u_{t-1} (a few thousand fmas, identical
// in every workgroup: same data, same reduction order), so no cross-workgroup hand-off exists inside a launch and the stream
// order is the only synchronisation.  ~6 us per launch instead of 80 ... 300 us per iteration.
//   t = 0:        u_0 = t0 / max(||t0||, eps)                               (t0 = W v, sn_rows_multi_kernel)
//   0 < t < T:    finish call t-1 (dnorm, 1/sigma) from (u_{t-1}, y_{t-1}); u_t = (y/d) / max(||y||/d, eps)
//   t = T:        finish call T-1; module's u <- u_{T-1}
// tmp region of a module: t0[Cout] | dnorm[T] | y[2][Cout] (double-buffered: launch t still reads y_{t-1} while writing y_t).
constexpr int SN_RB = 32;

__device__ __forceinline__ int sn_find_iter(const dgmr_sn_desc* __restrict__ d, int n, int blk) {
    int m = 0;
    while (m + 1 < n && blk >= d[m + 1].iter_block0) ++m;
    return m;
}

__global__ __launch_bounds__(256) void sn_iter_multi_kernel(const dgmr_sn_desc* __restrict__ descs, int n, float* __restrict__ arena,
                                                            int t) {
    extern __shared__ float sh[];  // u[Cout] | red[64]
    const int m = sn_find_iter(descs, n, blockIdx.x);
    const dgmr_sn_desc d = descs[m];
    const int Cout = d.Cout, T = d.T;
    if (t > T) return;
    const int rb = blockIdx.x - d.iter_block0;
    const float eps = d.eps;
    float* tmp = arena + d.tmp_off;
    const float* t0 = tmp;
    float* dnorm = tmp + Cout;
    float* Y = tmp + Cout + T;
    float* u_hist = arena + d.u_hist_off;
    float* inv_sigma = arena + d.inv_sigma_off;
    float* u = sh;
    float* red = sh + Cout;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (t == 0) {
        float q = 0.f;
        for (int i = threadIdx.x; i < Cout; i += blockDim.x) q = fmaf(t0[i], t0[i], q);
        q = block_sum(q, red);
        const float inv = 1.f / fmaxf(sqrtf(q), eps);
        for (int i = threadIdx.x; i < Cout; i += blockDim.x) u[i] = t0[i] * inv;
    } else {
        const int sp = d.perm ? d.perm[t - 1] : t - 1;
        const float* __restrict__ y = Y + ((t - 1) & 1) * Cout;
        const float* __restrict__ up = u_hist + (size_t)sp * Cout;
        float a = 0.f, b = 0.f;
        for (int i = threadIdx.x; i < Cout; i += blockDim.x) {
            const float yi = y[i];
            a = fmaf(up[i], yi, a);
            b = fmaf(yi, yi, b);
        }
        const float n2 = block_sum(a, red);
        const float y2 = block_sum(b, red + 32);
        const float dd = fmaxf(sqrtf(fmaxf(n2, 0.f)), eps);
        if (rb == 0 && threadIdx.x == 0) {
            dnorm[t - 1] = dd;
            inv_sigma[sp] = dd / n2;
        }
        if (t == T) {
            if (rb == 0)
                for (int i = threadIdx.x; i < Cout; i += blockDim.x) d.u[i] = up[i];
            return;
        }
        const float invd = 1.f / dd;
        const float inv = invd / fmaxf(sqrtf(y2) * invd, eps);
        for (int i = threadIdx.x; i < Cout; i += blockDim.x) u[i] = y[i] * inv;
    }
    __syncthreads();
    const int st = d.perm ? d.perm[t] : t;
    if (rb == 0)
        for (int i = threadIdx.x; i < Cout; i += blockDim.x) u_hist[(size_t)st * Cout + i] = u[i];
    // y_t[rows of this workgroup] = A[rows] . u_t : each wave takes rows r0 + wid, + nw, ...; four rows in flight per wave
    const float* __restrict__ A = d.gram;
    float* yo = Y + (t & 1) * Cout;
    const int r0 = rb * SN_RB, r1 = min(Cout, r0 + SN_RB);
    for (int i0 = r0 + wid; i0 < r1; i0 += 4 * nw) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j = lane; j < Cout; j += 64) {
            const float uj = u[j];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + r * nw;
                if (i < r1) s[r] = fmaf(A[(size_t)i * Cout + j], uj, s[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + r * nw;
            const float v = wave_sum(s[r]);
            if (lane == 0 && i < r1) yo[i] = v;
        }
    }
}

// (round 6) The row loop walks tiles of 64 rows: the tile's u values of all SN_TMAX calls are staged in LDS once ([row][call], read back
// as broadcast ds_read_b128) and the thread's 16 weights of the tile are fetched up front.  Before, every (row, call) pair cost a flat
// load through 64-bit vector address arithmetic and a branch - ten instructions per multiply-add, 1.0 - 1.2 ms per launch.  The sums are
// formed in the same order (rows rg, rg + 4, ... per call): bit-identical results.
__global__ __launch_bounds__(256) void sn_cols_multi_kernel(const dgmr_sn_desc* __restrict__ descs, int n, float* __restrict__ arena) {
    constexpr int RT = 64, US = SN_TMAX + 4;  // rows per tile; LDS row stride in floats (16-byte aligned rows)
    __shared__ float part[4][SN_TMAX][64];
    __shared__ __attribute__((aligned(16))) float ut[RT * US];
    const int m = sn_find(descs, n, blockIdx.x, false);
    const dgmr_sn_desc d = descs[m];
    const int Cout = d.Cout, T = d.T, K = d.Cin * d.taps;
    const float* __restrict__ u_hist = arena + d.u_hist_off;
    const float* __restrict__ dnorm = arena + d.tmp_off + Cout;
    float* v_hist = arena + d.v_hist_off;
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int k = (blockIdx.x - d.col_block0) * 64 + c;
    const bool kok = k < K;
    // sequences longer than SN_TMAX calls (batched generator draws: draws x forecast steps) go in chunks; W is re-read from L2
    for (int tc = 0; tc < T; tc += SN_TMAX) {
        const int Tc = min(SN_TMAX, T - tc);
        float acc[SN_TMAX];
#pragma unroll
        for (int t = 0; t < SN_TMAX; ++t) acc[t] = 0.f;
        for (int i0 = 0; i0 < Cout; i0 += RT) {
            __syncthreads();  // the previous tile (and the previous chunk's partial sums) have been consumed
            // calls t = rg, rg + 4, ...: 64 consecutive rows per wave and call (calls beyond the chunk read call tc: never used)
#pragma unroll
            for (int j = 0; j < SN_TMAX / 4; ++j) {
                const int t = rg + 4 * j;
                const int sl = t < Tc ? (d.perm ? d.perm[tc + t] : tc + t) : (d.perm ? d.perm[tc] : tc);
                ut[c * US + t] = i0 + c < Cout ? u_hist[(size_t)sl * Cout + i0 + c] : 0.f;
            }
            float wv[RT / 4];
#pragma unroll
            for (int r = 0; r < RT / 4; ++r) {
                const int i = i0 + rg + 4 * r;
                wv[r] = (kok && i < Cout) ? d.w[(size_t)i * K + k] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < RT / 4; ++r) {
                if (i0 + rg + 4 * r >= Cout) break;  // (wave-uniform)
                const float4* up = reinterpret_cast<const float4*>(ut + (rg + 4 * r) * US);
#pragma unroll
                for (int q = 0; q < SN_TMAX / 4; ++q) {
                    const float4 u4 = up[q];
                    acc[4 * q + 0] = fmaf(wv[r], u4.x, acc[4 * q + 0]);
                    acc[4 * q + 1] = fmaf(wv[r], u4.y, acc[4 * q + 1]);
                    acc[4 * q + 2] = fmaf(wv[r], u4.z, acc[4 * q + 2]);
                    acc[4 * q + 3] = fmaf(wv[r], u4.w, acc[4 * q + 3]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < SN_TMAX; ++t) part[rg][t][c] = acc[t];
        __syncthreads();
        if (kok) {
            const int tp = k / d.Cin, ci = k - tp * d.Cin;
            const size_t j = (size_t)ci * d.taps + tp;
            for (int t = rg; t < Tc; t += 4) {
                const float sv = (part[0][t][c] + part[1][t][c] + part[2][t][c] + part[3][t][c]) / dnorm[tc + t];
                const int sl = d.perm ? d.perm[tc + t] : tc + t;
                v_hist[(size_t)sl * K + j] = sv;
                if (tc + t == T - 1) d.v[j] = sv;
            }
        }
    }
}

__global__ void zero_kernel(float* p, int n) {
    if (threadIdx.x < n) p[threadIdx.x] = 0.f;
}

// ------------------------------------------------------------------------------------------------
// per-channel reductions over [G][R][C]
// ------------------------------------------------------------------------------------------------
// F: void(int64_t elem_index, int g, int c, double& s0, double& s1)
// Accumulation is in double from the first element on.  Batch variance is formed as E[x^2] - mean^2, which cancels by mean^2 / var:
// the discriminator heads normalise relu-sum features whose spread across the 8 ... 32 samples of a call is ~1 % of their mean
// (mean^2 / var ~ 1e4), and float partial sums (6e-8 x 1e4 = 6e-4 on rstd) put a 1e-3 ... 1e-2 error on every gradient behind the
// head (found by tests/test_gpu_stages.py::test_temporal_discriminator_backward_stages).  These kernels are HBM-bound either way.
template <class F>
__device__ __forceinline__ void chan_reduce2(F f, int64_t R, int C, double* __restrict__ out /* [G][2][C] */, int det = 0) {
    __shared__ double l0[256], l1[256];
    const int g = blockIdx.y;
    const int64_t rows_per_block = (R + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = blockIdx.x * rows_per_block, r1 = min(R, r0 + rows_per_block);
    for (int cb = 0; cb < C; cb += 256) {
        const int Wd = min(256, C - cb);
        const int RL = 256 / Wd;
        const int c = cb + threadIdx.x % Wd, rl = threadIdx.x / Wd;
        double s0 = 0.0, s1 = 0.0;
        if (rl < RL)
            for (int64_t r = r0 + rl; r < r1; r += RL) f(((int64_t)g * R + r) * C + c, g, c, s0, s1);
        l0[threadIdx.x] = s0;
        l1[threadIdx.x] = s1;
        __syncthreads();
        if (threadIdx.x < Wd) {
            double a0 = 0.0, a1 = 0.0;
            for (int k = 0; k < RL; ++k) {
                a0 += l0[threadIdx.x + k * Wd];
                a1 += l1[threadIdx.x + k * Wd];
            }
            if (det) {  // deterministic mode: row 1 + blockIdx.x of the caller's buffer; reduce_rows_finish_kernel adds the rows up in order
                double* row = out + (size_t)(1 + blockIdx.x) * gridDim.y * 2 * C;
                row[((size_t)g * 2 + 0) * C + c] = a0;
                row[((size_t)g * 2 + 1) * C + c] = a1;
            } else {
                atomicAdd(out + ((size_t)g * 2 + 0) * C + c, a0);
                atomicAdd(out + ((size_t)g * 2 + 1) * C + c, a1);
            }
        }
        __syncthreads();
    }
}

// out[j] += sum over rows b = 0 .. nb-1 of out[(1 + b) * n + j], in that order (deterministic mode of the per-channel reductions)
__global__ void reduce_rows_finish_kernel(double* __restrict__ out, int nb, int64_t n) {
    GRID_STRIDE(j, n) {
        double a = 0.0;
        for (int b = 0; b < nb; ++b) a += out[(size_t)(1 + b) * n + j];
        out[j] += a;
    }
}

__global__ void bn_stats_kernel(const float* __restrict__ x, double* __restrict__ sums, int64_t R, int C, int det) {
    chan_reduce2(
        [&](int64_t i, int, int, double& s0, double& s1) {
            const double v = (double)x[i];
            s0 += v;
            s1 = fma(v, v, s1);
        },
        R, C, sums, det);
}

// deterministic bn_partial_reduce: one workgroup per (32 columns, group); 8 row lanes sum rows r = lane (mod 8) in order, then in lane order
__global__ __launch_bounds__(256) void bn_partial_reduce_det_kernel(const float* __restrict__ partials, double* __restrict__ sums, int64_t R, int C2) {
    __shared__ double part[8][32];
    const int g = blockIdx.y, col = threadIdx.x & 31, lane = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + col;
    double a = 0.0;
    if (j < C2)
        for (int64_t r = lane; r < R; r += 8) a += (double)partials[((size_t)g * R + r) * C2 + j];
    part[lane][col] = a;
    __syncthreads();
    if (lane == 0 && j < C2) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += part[k][col];
        sums[(size_t)g * C2 + j] += t;
    }
}

// sums[g][k][c] += sum_r partials[(g*R + r)][k][c]: blockIdx.y = g, blockIdx.z = slice of the rows; one thread per (k, c)
__global__ void bn_partial_reduce_kernel(const float* __restrict__ partials, double* __restrict__ sums, int64_t R, int C2) {
    const int g = blockIdx.y;
    const int64_t per = (R + gridDim.z - 1) / gridDim.z;
    const int64_t r0 = blockIdx.z * per, r1 = min(R, r0 + per);
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < C2; j += gridDim.x * blockDim.x) {
        double a = 0.0;
        for (int64_t r = r0; r < r1; ++r) a += (double)partials[((size_t)g * R + r) * C2 + j];
        atomicAdd(sums + (size_t)g * C2 + j, a);
    }
}

__global__ void bn_bwd_center_kernel(double* __restrict__ sums, const float* __restrict__ mean, const float* __restrict__ rstd, int GC,
                                     int C) {
    GRID_STRIDE(i, GC) {
        const int64_t g = i / C, c = i - g * C;
        double* s = sums + g * 2 * C;
        s[C + c] = (double)rstd[i] * (s[C + c] - (double)mean[i] * s[c]);
    }
}

__global__ void bn_bwd_reduce_kernel(const float* __restrict__ gy, const float* __restrict__ x, const float* __restrict__ mean,
                                     const float* __restrict__ rstd, double* __restrict__ sums, int64_t R, int C, int det) {
    chan_reduce2(
        [&](int64_t i, int g, int c, double& s0, double& s1) {
            const float gv = gy[i];
            const float xh = (x[i] - mean[(size_t)g * C + c]) * rstd[(size_t)g * C + c];
            s0 += (double)gv;
            s1 = fma((double)gv, (double)xh, s1);
        },
        R, C, sums, det);
}

__global__ void colsum_kernel(const float* __restrict__ x, double* __restrict__ sums, int64_t R, int C, int det) {
    chan_reduce2([&](int64_t i, int, int, double& s0, double&) { s0 += (double)x[i]; }, R, C, sums, det);
}

// out[c] (+)= sum_r x[r][c] for FEW rows and MANY columns (one thread per 4 columns; colsum_kernel is for the opposite shape)
__global__ void sum_rows_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ out, int R, int64_t C4, int accumulate) {
    GRID_STRIDE(c, C4) {
        f32x4 s = accumulate ? out[c] : (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < R; ++r) s += x[(int64_t)r * C4 + c];
        out[c] = s;
    }
}

// out[g][c] = sum_r w[(g*R + r) / rows_per_w] * x[g][r][c]  (w == nullptr: plain sums); blockIdx.y = g, one thread per 4 columns
__global__ void group_rowsum_kernel(const f32x4* __restrict__ x, const float* __restrict__ w, f32x4* __restrict__ out, int R, int64_t C4,
                                    int rows_per_w, int w_stride) {
    const int g = blockIdx.y;
    x += (int64_t)g * R * C4;
    out += (int64_t)g * C4;
    const int64_t wblock = C4 / w_stride;  // columns (in float4) that share one weight column
    GRID_STRIDE(c, C4) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        const int64_t wc = w_stride > 1 ? c / wblock : 0;
        for (int r = 0; r < R; ++r) {
            const float f = w ? w[(((int64_t)g * R + r) / rows_per_w) * w_stride + wc] : 1.f;
            s += f * x[(int64_t)r * C4 + c];
        }
        out[c] = s;
    }
}

// dst[(k*repeat + r)][:] = src[k][:]: every block of C4 float4 repeated `repeat` times in place ('k ... -> (k repeat) ...')
__global__ void repeat_interleave_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int64_t C4, int repeat, int64_t total4) {
    GRID_STRIDE(i, total4) {
        const int64_t row = i / C4;
        dst[i] = src[(row / repeat) * C4 + (i - row * C4)];
    }
}

// dst[i][c] = src[c] for i < repeat  (einops 'b ... -> (repeat b) ...' with b == 1)
__global__ void repeat_rows_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int64_t C4, int64_t total4) {
    GRID_STRIDE(i, total4) dst[i] = src[i % C4];
}

__global__ void colsum_finish_kernel(const double* __restrict__ sums, float* __restrict__ out, int C, int accumulate) {
    GRID_STRIDE(c, C) out[c] = (accumulate ? out[c] : 0.f) + (float)sums[c];
}

// (round 6) 64 channels x 16 group lanes per workgroup: the per-(group, channel) part - two double divisions, a double square root and a
// reciprocal: ~300 cycles - runs 16 groups at a time; only the running-statistics recurrence (two float multiply-adds per call, in the
// reference's call order) is sequential, one lane per channel reading (float) mean / unbiased variance back from LDS.  Before: one thread
// per channel walked all G groups (G = draws x steps, up to 108) - 29 us per launch, 103 launches per training step, on the forward
// passes' critical path.  Every expression is written in the contracted form the compiler had chosen for the old loop: bit-identical.
constexpr int BNF_C = 64, BNF_L = 16, BNF_Q = 64;  // channels per workgroup, group lanes, groups per LDS chunk
__global__ __launch_bounds__(BNF_C* BNF_L) void bn_finalize_kernel(const double* __restrict__ sums, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, float* __restrict__ rm,
                                                                   float* __restrict__ rv, int64_t* __restrict__ nbt,
                                                                   float* __restrict__ a, float* __restrict__ b,
                                                                   float* __restrict__ save_mean, float* __restrict__ save_rstd, int G,
                                                                   int64_t R, int C, float eps, float momentum,
                                                                   const int32_t* __restrict__ order) {
    __shared__ float s_mean[BNF_Q][BNF_C], s_unb[BNF_Q][BNF_C];
    const int cl = threadIdx.x & (BNF_C - 1), lane = threadIdx.x / BNF_C;
    const int c = blockIdx.x * BNF_C + cl;
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt && sums) nbt[0] += G;
    const bool cok = c < C;
    const float gm = (cok && gamma) ? gamma[c] : 1.f, bt = (cok && beta) ? beta[c] : 0.f;
    if (!sums) {  // eval: running statistics
        if (lane == 0 && cok) {
            const float mean = rm[c], rstd = 1.f / sqrtf(rv[c] + eps);
            a[c] = gm * rstd;
            b[c] = bt - mean * gm * rstd;
            if (save_mean) {
                save_mean[c] = mean;
                save_rstd[c] = rstd;
            }
        }
        return;
    }
    float rmc = 0.f, rvc = 0.f;
    if (lane == 0 && cok) {
        rmc = rm[c];
        rvc = rv[c];
    }
    const double Rd = (double)R, bessel = R > 1 ? (double)R / (double)(R - 1) : 1.0;
    const float keep = 1.f - momentum;
    for (int q0 = 0; q0 < G; q0 += BNF_Q) {
        const int nq = min(BNF_Q, G - q0);
        for (int qq = lane; qq < nq; qq += BNF_L) {
            // the q-th call of the module (the order in which the reference updates its running statistics) is group order[q] of the batch
            const int g = order ? order[q0 + qq] : q0 + qq;
            if (cok) {
                const double mean = sums[((size_t)g * 2 + 0) * C + c] / Rd;
                double var = fma(-mean, mean, sums[((size_t)g * 2 + 1) * C + c] / Rd);
                if (var < 0.0) var = 0.0;
                const float rstd = (float)(1.0 / sqrt(var + (double)eps));
                const float av = gm * rstd, mf = (float)mean;
                a[(size_t)g * C + c] = av;
                b[(size_t)g * C + c] = __fmaf_rn(-av, mf, bt);
                if (save_mean) {
                    save_mean[(size_t)g * C + c] = mf;
                    save_rstd[(size_t)g * C + c] = rstd;
                }
                s_mean[qq][cl] = mf;
                s_unb[qq][cl] = (float)(R > 1 ? bessel * var : var);
            }
        }
        __syncthreads();
        if (lane == 0 && cok)
            for (int qq = 0; qq < nq; ++qq) {
                rmc = __fmaf_rn(keep, rmc, __fmul_rn(momentum, s_mean[qq][cl]));
                rvc = __fmaf_rn(keep, rvc, __fmul_rn(momentum, s_unb[qq][cl]));
            }
        __syncthreads();
    }
    if (lane == 0 && cok) {
        rm[c] = rmc;
        rv[c] = rvc;
    }
}

__global__ void bn_bwd_apply_kernel(const float* __restrict__ gy, const float* __restrict__ x, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                    const double* __restrict__ sums, const float* __restrict__ dx_add, float* __restrict__ dx,
                                    int64_t R, int C, int train) {
    const int g = blockIdx.y;
    const int64_t n = R * C;
    const float invR = 1.f / (float)R;
    GRID_STRIDE(i, n) {
        const int c = i % C;
        const size_t gc = (size_t)g * C + c;
        const size_t idx = (size_t)g * n + i;
        const float gm = gamma ? gamma[c] : 1.f;
        const float rs = rstd[gc];
        float v = gy[idx];
        if (train) {
            const float xh = (x[idx] - mean[gc]) * rs;
            v = v - (float)sums[((size_t)g * 2 + 0) * C + c] * invR - xh * (float)sums[((size_t)g * 2 + 1) * C + c] * invR;
        }
        v *= gm * rs;
        if (dx_add) v += dx_add[idx];
        dx[idx] = v;
    }
}

// The same, 16 bytes per lane: the grid stride is a multiple of C / 4, so a thread's channel quad - and its five coefficient quads -
// never change over its rows (the scalar kernel spends a 64-bit modulo and seven dependent loads per element: 4.3 TB/s of HBM
// traffic where a streaming kernel reaches ~6).  Same expression per element as above.
__global__ __launch_bounds__(256) void bn_bwd_apply4_kernel(const f32x4* __restrict__ gy, const f32x4* __restrict__ x,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma, const double* __restrict__ sums,
                                                             const f32x4* __restrict__ dx_add, f32x4* __restrict__ dx,
                                                             int64_t n4, int C, float invR, int train) {
    const int g = blockIdx.y;
    const int C4 = C >> 2;
    const int64_t i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int c = (int)(i0 % C4) * 4;
    f32x4 mn, rs, k0, k1, gr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const size_t gc = (size_t)g * C + c + j;
        mn[j] = mean[gc];
        rs[j] = rstd[gc];
        k0[j] = (float)sums[((size_t)g * 2 + 0) * C + c + j];
        k1[j] = (float)sums[((size_t)g * 2 + 1) * C + c + j];
        gr[j] = (gamma ? gamma[c + j] : 1.f) * rs[j];
    }
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const size_t base = (size_t)g * n4;
    for (int64_t i = i0; i < n4; i += stride) {
        f32x4 v = __builtin_nontemporal_load(gy + base + i);
        if (train) {
            const f32x4 xv = __builtin_nontemporal_load(x + base + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (xv[j] - mn[j]) * rs[j];
                v[j] = v[j] - k0[j] * invR - xh * k1[j] * invR;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= gr[j];
        if (dx_add) v += __builtin_nontemporal_load(dx_add + base + i);
        dx[base + i] = v;
    }
}

__global__ void bn_param_grad_kernel(const double* __restrict__ sums, float* __restrict__ dgamma, float* __restrict__ dbeta, int G,
                                     int C) {
    GRID_STRIDE(c, C) {
        double s0 = 0.0, s1 = 0.0;
        for (int g = 0; g < G; ++g) {
            s0 += sums[((size_t)g * 2 + 0) * C + c];
            s1 += sums[((size_t)g * 2 + 1) * C + c];
        }
        if (dbeta) dbeta[c] += (float)s0;
        if (dgamma) dgamma[c] += (float)s1;
    }
}

__global__ void affine_kernel(const float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ b,
                              float* __restrict__ y, int64_t R, int C, int relu) {
    const int g = blockIdx.y;
    const int64_t n = R * C;
    GRID_STRIDE(i, n) {
        const int c = i % C;
        float v = fmaf(x[(size_t)g * n + i], a[(size_t)g * C + c], b[(size_t)g * C + c]);
        if (relu) v = fmaxf(v, 0.f);
        y[(size_t)g * n + i] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// pooling and layout
// ------------------------------------------------------------------------------------------------
__global__ void pool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ addend, float* __restrict__ y, int N, int D,
                                int H, int W, int C, int pd, float scale, const float* __restrict__ mask_src,
                                const float* __restrict__ mask_a, const float* __restrict__ mask_b, int mask_group) {
    const int Do = D / pd, Ho = H / 2, Wo = W / 2, C4 = C / 4;
    const int64_t total = (int64_t)N * Do * Ho * Wo * C4;
    GRID_STRIDE(i, total) {
        const int c4 = i % C4;
        int64_t t = i / C4;
        const int wo = t % Wo;
        t /= Wo;
        const int ho = t % Ho;
        t /= Ho;
        const int d_o = t % Do;
        const int n = t / Do;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int dz = 0; dz < pd; ++dz)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const size_t off = ((((size_t)n * D + d_o * pd + dz) * H + ho * 2 + dy) * W + wo * 2 + dx) * C + c4 * 4;
                    s += *reinterpret_cast<const f32x4*>(x + off);
                }
        s *= scale;
        const size_t o = (size_t)i * 4;
        if (addend) s += *reinterpret_cast<const f32x4*>(addend + o);
        if (mask_src) {
            f32x4 m = *reinterpret_cast<const f32x4*>(mask_src + o);
            if (mask_a) {
                const size_t gc = (size_t)(n / mask_group) * C + c4 * 4;
                const f32x4 a = *reinterpret_cast<const f32x4*>(mask_a + gc);
                const f32x4 b = *reinterpret_cast<const f32x4*>(mask_b + gc);
                m = m * a + b;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] = m[j] > 0.f ? s[j] : 0.f;
        }
        *reinterpret_cast<f32x4*>(y + o) = s;
    }
}

// y[n][j] = (x[n][2j] + x[n][2j + 1]) / 2 (+ addend[n][j]): the depth half of AvgPool3d(2) over planes of `plane4` float4 (the spatial
// half already rode in the conv that produced x: dgmr_conv_args.pool2 on a 3x3x3 conv).  blockIdx.y walks the output planes.
__global__ __launch_bounds__(256) void pool_depth2_kernel(const f32x4* __restrict__ x, const f32x4* __restrict__ addend,
                                                           f32x4* __restrict__ y, int N, int D, int64_t plane4) {
    const int Do = D / 2;
    for (int op = blockIdx.y; op < N * Do; op += gridDim.y) {
        const int n = op / Do, j = op - n * Do;
        const f32x4* x0 = x + ((size_t)n * D + 2 * j) * plane4;
        const size_t ob = (size_t)op * plane4;
        GRID_STRIDE(i, plane4) {
            f32x4 v = (x0[i] + x0[plane4 + i]) * 0.5f;
            if (addend) v += addend[ob + i];
            y[ob + i] = v;
        }
    }
}

__global__ void pool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int D, int H, int W, int C, int pd,
                                float scale) {
    const int Do = D / pd, Ho = H / 2, Wo = W / 2, C4 = C / 4;
    const int64_t total = (int64_t)N * D * H * W * C4;
    GRID_STRIDE(i, total) {
        const int c4 = i % C4;
        int64_t t = i / C4;
        const int w = t % W;
        t /= W;
        const int h = t % H;
        t /= H;
        const int d = t % D;
        const int n = t / D;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int d_o = d / pd, ho = h >> 1, wo = w >> 1;
        if (d_o < Do && ho < Ho && wo < Wo) {
            const size_t off = ((((size_t)n * Do + d_o) * Ho + ho) * Wo + wo) * C + c4 * 4;
            v = *reinterpret_cast<const f32x4*>(dy + off) * scale;
        }
        *reinterpret_cast<f32x4*>(dx + (size_t)i * 4) = v;
    }
}

__global__ void frames_s2d_kernel(const float* __restrict__ fr, const int32_t* __restrict__ idx, float* __restrict__ out, int B,
                                  int T, int C, int H, int W, int F, int p, int frame_major, int idx_group) {
    const int Ho = H / (2 * p), Wo = W / (2 * p), Co = 4 * C;
    const int64_t total = (int64_t)B * F * Ho * Wo * Co;
    const float inv = 1.f / (float)(p * p);
    GRID_STRIDE(i, total) {
        const int co = i % Co;
        int64_t t = i / Co;
        const int wo = t % Wo;
        t /= Wo;
        const int ho = t % Ho;
        t /= Ho;
        int b, f;
        if (frame_major) {
            b = t % B;
            f = t / B;
        } else {
            f = t % F;
            b = t / F;
        }
        const int c = co >> 2, dy = (co >> 1) & 1, dx = co & 1;
        const int tf = idx ? idx[(b / idx_group) * F + f] : f;
        const float* src = fr + (((size_t)b * T + tf) * C + c) * H * W;
        float s = 0.f;
        for (int py = 0; py < p; ++py)
            for (int px = 0; px < p; ++px) s += src[(size_t)((2 * ho + dy) * p + py) * W + (2 * wo + dx) * p + px];
        out[i] = s * inv;
    }
}

// Gradient of frames_s2d as a GATHER over the pixels of dframes: every element sums the (at most F) selected frames that read it, in
// frame order - no atomics, no zero-fill, bit-identical from run to run (the scatter form added duplicates of a randomly drawn
// frame index in whatever order the workgroups arrived: three draws of one frame among the spatial discriminator's eight happen in
// ~10 % of the steps).  dframes is WRITTEN, every element.
__global__ void frames_s2d_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ idx, float* __restrict__ dfr,
                                      int B, int T, int C, int H, int W, int F, int p, int frame_major, int idx_group) {
    const int Ho = H / (2 * p), Wo = W / (2 * p), Co = 4 * C;
    const int64_t total = (int64_t)B * T * C * H * W;
    const float inv = 1.f / (float)(p * p);
    GRID_STRIDE(i, total) {
        const int x = i % W;
        int64_t t = i / W;
        const int y = t % H;
        t /= H;
        const int c = t % C;
        t /= C;
        const int tf = t % T;
        const int b = t / T;
        const int xo = x / p, yo = y / p;  // pixel of the (pooled) frame
        const int wo = xo >> 1, dx = xo & 1, ho = yo >> 1, dy = yo & 1;
        float g = 0.f;
        if (ho < Ho && wo < Wo) {
            const int co = (c << 2) | (dy << 1) | dx;
            const int f_lo = idx ? 0 : tf, f_hi = idx ? F : min(tf + 1, F);  // no index list: frame f reads frame f
            for (int f = f_lo; f < f_hi; ++f) {
                const int src = idx ? idx[(b / idx_group) * F + f] : f;
                if (src != tf) continue;
                const int64_t n = frame_major ? (int64_t)f * B + b : (int64_t)b * F + f;
                g += dout[((n * Ho + ho) * Wo + wo) * Co + co] * inv;
            }
        }
        dfr[i] = g;
    }
}

template <bool BWD>
__global__ void d2s_frames_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int T, int t, int C, int h, int w) {
    // forward: frames[b][t][c][2y+dy][2x+dx] = x[b][y][x][c*4+dy*2+dx]; one thread per frame pixel (coalesced frame side)
    const int H = 2 * h, W = 2 * w;
    const int64_t total = (int64_t)B * C * H * W;
    GRID_STRIDE(i, total) {
        const int X = i % W;
        int64_t r = i / W;
        const int Y = r % H;
        r /= H;
        const int c = r % C;
        const int b = r / C;
        const size_t fi = ((((size_t)b * T + t) * C + c) * H + Y) * W + X;
        const size_t xi = (((size_t)b * h + (Y >> 1)) * w + (X >> 1)) * (4 * C) + c * 4 + (Y & 1) * 2 + (X & 1);
        if (BWD) dst[xi] = src[fi];
        else dst[fi] = src[xi];
    }
}

__global__ void copy_channels_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t R, int C, int src_C, int src_off,
                                     int src_cs, int dst_C, int dst_off, int dst_cs, int accumulate) {
    const int64_t total = R * C;
    GRID_STRIDE(i, total) {
        const int c = i % C;
        const int64_t r = i / C;
        const float v = src[(size_t)r * src_C + src_off + (size_t)c * src_cs];
        float* d = dst + (size_t)r * dst_C + dst_off + (size_t)c * dst_cs;
        *d = accumulate ? *d + v : v;
    }
}

// ------------------------------------------------------------------------------------------------
// ConvGRU gating, elementwise
// ------------------------------------------------------------------------------------------------
__global__ void gru_gate_fwd_kernel(const f32x4* __restrict__ pr, const f32x4* __restrict__ h, f32x4* __restrict__ rh, int64_t n4) {
    GRID_STRIDE(i, n4) {
        const f32x4 p = pr[i], hv = h[i];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = sigmoidf_(p[j]) * hv[j];
        rh[i] = o;
    }
}

__global__ void gru_gate_bwd_kernel(const f32x4* __restrict__ d_rh, const f32x4* __restrict__ pr, const f32x4* __restrict__ h,
                                    f32x4* __restrict__ dpr, f32x4* __restrict__ dh, int64_t n4) {
    GRID_STRIDE(i, n4) {
        const f32x4 g = d_rh[i], p = pr[i], hv = h[i];
        f32x4 a, b;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s = sigmoidf_(p[j]);
            a[j] = g[j] * hv[j] * s * (1.f - s);
            b[j] = g[j] * s;
        }
        dpr[i] = a;
        dh[i] = b;
    }
}

__global__ void gru_blend_fwd_kernel(const f32x4* __restrict__ pu, const f32x4* __restrict__ h, const f32x4* __restrict__ pc,
                                     f32x4* __restrict__ out, int64_t n4) {
    GRID_STRIDE(i, n4) {
        const f32x4 u = pu[i], hv = h[i], c = pc[i];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s = sigmoidf_(u[j]);
            o[j] = s * hv[j] + (1.f - s) * fmaxf(c[j], 0.f);
        }
        out[i] = o;
    }
}

__global__ void gru_blend_bwd_kernel(const f32x4* __restrict__ dout, const f32x4* __restrict__ pu, const f32x4* __restrict__ h,
                                     const f32x4* __restrict__ pc, f32x4* __restrict__ dpu, f32x4* __restrict__ dh,
                                     f32x4* __restrict__ dpc, int64_t n4) {
    GRID_STRIDE(i, n4) {
        const f32x4 g = dout[i], u = pu[i], hv = h[i], c = pc[i];
        f32x4 a, b, d;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s = sigmoidf_(u[j]);
            const float rc = fmaxf(c[j], 0.f);
            a[j] = g[j] * (hv[j] - rc) * s * (1.f - s);
            b[j] = g[j] * s;
            d[j] = c[j] > 0.f ? g[j] * (1.f - s) : 0.f;
        }
        dpu[i] = a;
        dh[i] = b;
        dpc[i] = d;
    }
}

// count += number of NaN / Inf among x (exponent field all ones); integer atomics: order-independent
__global__ void nonfinite_count_kernel(const float* __restrict__ x, int64_t n, int32_t* __restrict__ count) {
    int bad = 0;
    GRID_STRIDE(i, n) bad += ((__float_as_uint(x[i]) & 0x7f800000u) == 0x7f800000u) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(count, bad);
}

__global__ void axpby_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, float alpha,
                             float beta, int64_t n) {
    GRID_STRIDE(i, n) y[i] = b ? alpha * a[i] + beta * b[i] : alpha * a[i];
}

__global__ void scale_by_dev_kernel(const float* __restrict__ x, const float* __restrict__ s, float host_scale,
                                    float* __restrict__ y, int64_t n) {
    const float f = s[0] * host_scale;
    GRID_STRIDE(i, n) y[i] = x[i] * f;
}

__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, int64_t n) {
    GRID_STRIDE(i, n) dx[i] = x[i] > 0.f ? dy[i] : 0.f;
}

__global__ void fill_kernel(float* __restrict__ p, float v, int64_t n) { GRID_STRIDE(i, n) p[i] = v; }

// ------------------------------------------------------------------------------------------------
// latent attention (dgmr/layers/Attention.py:9-20).  Tensors are one sample, channels-last [H][W][Cq]; the
// reference's einsum treats the NCHW view [Cq][H][W] as "[h w c]": position p = cq*H + y, feature = x.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t att_addr(int p, int f, int Cq, int H, int W) {
    const int cq = p / H, y = p - cq * H;
    return ((size_t)y * W + f) * Cq + cq;
}

__global__ void attention_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                     float* __restrict__ beta, float* __restrict__ out, int Cq, int H, int W) {
    extern __shared__ float sh[];  // logits [L] + red[32]
    const int L = Cq * H, p = blockIdx.x;
    float* logit = sh;
    float* red = sh + L;
    float mx = -INFINITY;
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
        float s = 0.f;
        for (int f = 0; f < W; ++f) s = fmaf(q[att_addr(p, f, Cq, H, W)], k[att_addr(l, f, Cq, H, W)], s);
        logit[l] = s;
        mx = fmaxf(mx, s);
    }
    // block max
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = red[0];
    for (int i = 1; i < (int)((blockDim.x + 63) >> 6); ++i) mx = fmaxf(mx, red[i]);
    float sum = 0.f;
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
        const float e = __expf(logit[l] - mx);
        logit[l] = e;
        sum += e;
    }
    sum = block_sum(sum, red);
    const float inv = 1.f / sum;
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
        const float bv = logit[l] * inv;
        logit[l] = bv;
        beta[(size_t)p * L + l] = bv;
    }
    __syncthreads();
    for (int f = threadIdx.x; f < W; f += blockDim.x) {
        float s = 0.f;
        for (int l = 0; l < L; ++l) s = fmaf(logit[l], v[att_addr(l, f, Cq, H, W)], s);
        out[att_addr(p, f, Cq, H, W)] = s;
    }
}

// per query position p: dlogit[p][l] = beta*(dbeta - sum_l beta*dbeta) -> tmp ; dq[p][f] = sum_l dlogit*k[l][f]
__global__ void attention_bwd_q_kernel(const float* __restrict__ dout, const float* __restrict__ k, const float* __restrict__ v,
                                       const float* __restrict__ beta, float* __restrict__ dq, float* __restrict__ tmp, int Cq,
                                       int H, int W) {
    extern __shared__ float sh[];
    const int L = Cq * H, p = blockIdx.x;
    float* dl = sh;
    float* red = sh + L;
    float dot = 0.f;
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
        float s = 0.f;
        for (int f = 0; f < W; ++f) s = fmaf(dout[att_addr(p, f, Cq, H, W)], v[att_addr(l, f, Cq, H, W)], s);
        dl[l] = s;
        dot = fmaf(beta[(size_t)p * L + l], s, dot);
    }
    dot = block_sum(dot, red);
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
        const float g = beta[(size_t)p * L + l] * (dl[l] - dot);
        dl[l] = g;
        tmp[(size_t)p * L + l] = g;
    }
    __syncthreads();
    for (int f = threadIdx.x; f < W; f += blockDim.x) {
        float s = 0.f;
        for (int l = 0; l < L; ++l) s = fmaf(dl[l], k[att_addr(l, f, Cq, H, W)], s);
        dq[att_addr(p, f, Cq, H, W)] = s;
    }
}

// per key position l: dk[l][f] = sum_p dlogit[p][l]*q[p][f]; dv[l][f] = sum_p beta[p][l]*dout[p][f]
__global__ void attention_bwd_kv_kernel(const float* __restrict__ dout, const float* __restrict__ q, const float* __restrict__ beta,
                                        const float* __restrict__ tmp, float* __restrict__ dk, float* __restrict__ dv, int Cq, int H,
                                        int W) {
    const int L = Cq * H, l = blockIdx.x;
    for (int f = threadIdx.x; f < W; f += blockDim.x) {
        float a = 0.f, b = 0.f;
        for (int p = 0; p < L; ++p) {
            a = fmaf(tmp[(size_t)p * L + l], q[att_addr(p, f, Cq, H, W)], a);
            b = fmaf(beta[(size_t)p * L + l], dout[att_addr(p, f, Cq, H, W)], b);
        }
        dk[att_addr(l, f, Cq, H, W)] = a;
        dv[att_addr(l, f, Cq, H, W)] = b;
    }
}

// ------------------------------------------------------------------------------------------------
// discriminator heads
// ------------------------------------------------------------------------------------------------
__global__ void relu_sum_hw_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int HW, int C) {
    GRID_STRIDE(i, (int64_t)N * C) {
        const int c = i % C;
        const int n = i / C;
        float s = 0.f;
        for (int p = 0; p < HW; ++p) s += fmaxf(x[((size_t)n * HW + p) * C + c], 0.f);
        y[i] = s;
    }
}

__global__ void relu_sum_hw_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, int N,
                                       int HW, int C) {
    GRID_STRIDE(i, (int64_t)N * HW * C) {
        const int c = i % C;
        const int n = i / ((int64_t)HW * C);
        dx[i] = x[i] > 0.f ? dy[(size_t)n * C + c] : 0.f;
    }
}

__global__ void linear1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                   const float* __restrict__ scale, float* __restrict__ y, int C, int scale_group) {
    __shared__ float red[32];
    const int n = blockIdx.x;
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) s = fmaf(x[(size_t)n * C + c], w[c], s);
    s = block_sum(s, red);
    if (threadIdx.x == 0) y[n] = s * (scale ? scale[n / scale_group] : 1.f) + (bias ? bias[0] : 0.f);
}

// one thread per (group, channel): dx rows of the group, the group's raw weight gradient; thread (0,0) also the bias gradient
__global__ void linear1_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ w,
                                   const float* __restrict__ scale, float* __restrict__ dx, float* __restrict__ gw_raw,
                                   float* __restrict__ gb, int N, int C, int scale_group) {
    const int groups = N / scale_group;
    GRID_STRIDE(i, (int64_t)groups * C) {
        const int c = i % C, q = i / C;
        float g = 0.f;
        const float wc = w[c] * (scale ? scale[q] : 1.f);
        for (int n = q * scale_group; n < (q + 1) * scale_group; ++n) {
            const float d = dy[n];
            dx[(size_t)n * C + c] = d * wc;
            g = fmaf(d, x[(size_t)n * C + c], g);
        }
        gw_raw[(size_t)q * C + c] = g;
        if (i == 0 && gb) {
            float s = 0.f;
            for (int n = 0; n < N; ++n) s += dy[n];
            gb[0] = s;
        }
    }
}

// dst[t][n][i] = src[n][t][i]  (16-byte items)
__global__ void permute_nt_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int N, int T, int64_t inner4) {
    const int64_t total = (int64_t)N * T * inner4;
    GRID_STRIDE(j, total) {
        const int64_t i = j % inner4;
        const int64_t r = j / inner4;
        const int n = r % N, t = r / N;
        dst[j] = src[((int64_t)n * T + t) * inner4 + i];
    }
}

// ------------------------------------------------------------------------------------------------
// losses, Adam
// ------------------------------------------------------------------------------------------------
__global__ void hinge_disc_kernel(const float* __restrict__ s_real, const float* __restrict__ s_gen, float* __restrict__ loss,
                                  float* __restrict__ d_real, float* __restrict__ d_gen, int n_real, int n_gen) {
    __shared__ float red[32];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < n_real; i += blockDim.x) {
        const float v = 1.f - s_real[i];
        a += fmaxf(v, 0.f);
        if (d_real) d_real[i] = v > 0.f ? -1.f / n_real : 0.f;
    }
    for (int i = threadIdx.x; i < n_gen; i += blockDim.x) {
        const float v = 1.f + s_gen[i];
        b += fmaxf(v, 0.f);
        if (d_gen) d_gen[i] = v > 0.f ? 1.f / n_gen : 0.f;
    }
    a = block_sum(a, red);
    b = block_sum(b, red);
    if (threadIdx.x == 0) loss[0] = a / n_real + b / n_gen;
}

__global__ void grid_cell_kernel(const float* __restrict__ preds, int K, int64_t stride, const float* __restrict__ target,
                                 const float* __restrict__ weights, float cap, double* __restrict__ acc, float* __restrict__ dweight,
                                 int64_t n, int det) {
    __shared__ float red[32];
    float s = 0.f;
    const float invK = 1.f / (float)K;
    GRID_STRIDE(i, n) {
        float m = 0.f;
        for (int k = 0; k < K; ++k) m += preds[(size_t)k * stride + i];
        m *= invK;
        const float y = target[i];
        const float w = weights ? weights[i] : fmaxf(y + 1.f, cap);
        const float d = (m - y) * w;
        s += fabsf(d);
        if (dweight) dweight[i] = (d > 0.f ? w : (d < 0.f ? -w : 0.f)) * invK;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        if (det) acc[1 + blockIdx.x] = (double)s;  // deterministic mode: grid_cell_finish_kernel adds the workgroups' sums in order
        else atomicAdd(acc, (double)s);
    }
}

__global__ __launch_bounds__(256) void grid_cell_finish_kernel(double* __restrict__ acc, float* __restrict__ loss, float mult, int nb) {
    // deterministic mode: the workgroups' sums acc[1 .. nb] in a fixed order - thread t takes rows b = t (mod 256) in increasing order,
    // thread 0 adds the 256 partial sums in thread order
    __shared__ double part[256];
    if (nb > 0) {
        double a = 0.0;
        for (int b = threadIdx.x; b < nb; b += 256) a += acc[1 + b];
        part[threadIdx.x] = a;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int k = 0; k < 256; ++k) t += part[k];
            acc[0] += t;
        }
    }
    if (threadIdx.x != 0) return;
    loss[0] = (float)(acc[0] * (double)mult);
    acc[0] = 0.0;
}

// torch.optim.Adam's update, scalar for scalar (torch/optim/adam.py, _multi_tensor_adam): exp_avg.lerp_(grad, 1 - beta1);
// exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2); denom = sqrt(exp_avg_sq) / sqrt(bias_correction2) + eps;
// param.addcdiv_(exp_avg, denom, value = -lr / bias_correction1).  The scalars are formed in double on the host and rounded to
// float once, as torch does (1 - 0.999 in float arithmetic is 4.7e-5 off).
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                            float w1 /* 1 - beta1 */, float beta2, float w2 /* 1 - beta2 */, float eps, float step_size,
                            float bc2_sqrt) {
    GRID_STRIDE(i, n) {
        const float gi = g[i], m0 = m[i];
        const float diff = gi - m0;
        const float mi = w1 < 0.5f ? fmaf(w1, diff, m0) : gi - diff * (1.f - w1);  // at::lerp
        const float vi = fmaf(w2 * gi, gi, beta2 * v[i]);
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    }
}

// Every tensor of an optimiser in ONE launch: block b works on chunk (b - d.block0) of tensor d, found by bisection over the
// descriptors' first blocks.  Same per-element arithmetic as adam_kernel (bit-identical updates).
constexpr int ADAM_CHUNK = 4096;  // elements per block
__global__ __launch_bounds__(256) void adam_multi_kernel(const dgmr_adam_desc* __restrict__ descs, int n_tensors, float w1, float beta2, float w2,
                                                         float eps) {
    int lo = 0, hi = n_tensors - 1;
    const int b = blockIdx.x;
    while (lo < hi) {  // last descriptor whose block0 <= b
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].block0 <= b) lo = mid;
        else hi = mid - 1;
    }
    const dgmr_adam_desc d = descs[lo];
    const int64_t i0 = (int64_t)(b - d.block0) * ADAM_CHUNK, i1 = i0 + ADAM_CHUNK < d.n ? i0 + ADAM_CHUNK : d.n;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        const float gi = d.g[i], m0 = d.m[i];
        const float diff = gi - m0;
        const float mi = w1 < 0.5f ? fmaf(w1, diff, m0) : gi - diff * (1.f - w1);
        const float vi = fmaf(w2 * gi, gi, beta2 * d.v[i]);
        d.m[i] = mi;
        d.v[i] = vi;
        d.p[i] -= d.step_size * (mi / (sqrtf(vi) / d.bc2_sqrt + eps));
    }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int dgmr_spectral_sigma(const float* w, float* u, float* v, float* u_save, float* v_save, float* inv_sigma,
                                   float* scratch, float* tmp, int Cout, int Cin, int taps, float eps, int train, void* stream) {
    DGMR_CHECK_ARG(w && u && v && inv_sigma && scratch && tmp, "dgmr_spectral_sigma: null pointer");
    DGMR_CHECK_ARG(Cin % 4 == 0, "dgmr_spectral_sigma: Cin=%d must be a multiple of 4", Cin);
    const int K = Cin * taps;
    float* t = tmp;
    float* s = tmp + Cout;
    if (train) {
        hipLaunchKernelGGL(sn_rows_kernel, dim3(Cout), dim3(256), 0, ST, w, v, t, K, Cin, taps);
        hipLaunchKernelGGL(sn_cols_kernel, dim3((K + 63) / 64), dim3(256), 0, ST, w, t, s, u, u_save, scratch, Cout, K, eps);
        hipLaunchKernelGGL(sn_finish_train_kernel, dim3(std::min((K + 255) / 256, 64)), dim3(256), 0, ST, s, v, v_save, inv_sigma,
                           scratch, K, Cin, taps, eps);
    } else {
        hipLaunchKernelGGL(sn_rows_kernel, dim3(Cout), dim3(256), 0, ST, w, v, t, K, Cin, taps);
        hipLaunchKernelGGL(sn_finish_eval_kernel, dim3(std::min((Cout + K + 255) / 256, 64)), dim3(256), 0, ST, u, v, u_save, v_save,
                           inv_sigma, t, Cout, K);
    }
    hipLaunchKernelGGL(zero_kernel, dim3(1), dim3(64), 0, ST, scratch, 4);
    DGMR_CHECK_LAUNCH();
    return 0;
}


extern "C" int dgmr_spectral_sigma_seq(const float* w, const float* gram, float* u, float* v, float* u_hist, float* v_hist,
                                       float* inv_sigma, float* scratch, float* tmp, int Cout, int Cin, int taps, float eps,
                                       int T, void* stream) {
    DGMR_CHECK_ARG(w && gram && u && v && u_hist && v_hist && inv_sigma && scratch && tmp, "dgmr_spectral_sigma_seq: null pointer");
    DGMR_CHECK_ARG(Cin % 4 == 0, "dgmr_spectral_sigma_seq: Cin=%d must be a multiple of 4", Cin);
    DGMR_CHECK_ARG(T >= 1 && T <= SN_TMAX, "dgmr_spectral_sigma_seq: T=%d out of range [1, %d]", T, SN_TMAX);
    DGMR_CHECK_ARG(Cout <= 8192, "dgmr_spectral_sigma_seq: Cout=%d too large", Cout);
    const int K = Cin * taps;
    float* t0 = tmp;           // [Cout]
    float* dnorm = tmp + Cout; // [T]
    hipLaunchKernelGGL(sn_rows_kernel, dim3(Cout), dim3(256), 0, ST, w, v, t0, K, Cin, taps);
    const int threads = Cout >= 512 ? 1024 : (Cout >= 128 ? 512 : 256);
    hipLaunchKernelGGL(sn_gram_chain_kernel, dim3(1), dim3(threads), (2 * Cout + 32) * sizeof(float), ST, gram, t0, scratch, u,
                       u_hist, dnorm, inv_sigma, Cout, T, eps);
    hipLaunchKernelGGL(sn_cols_seq_kernel, dim3((K + 63) / 64), dim3(256), 0, ST, w, u_hist, dnorm, v, v_hist, Cout, K, Cin, taps, T);
    hipLaunchKernelGGL(zero_kernel, dim3(1), dim3(64), 0, ST, scratch, 4);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_spectral_sigma_seq_multi(const dgmr_sn_desc* descs_dev, int n, int total_row_blocks, int total_col_blocks,
                                             int total_iter_blocks, int max_cout, int max_T, float* arena, void* stream) {
    DGMR_CHECK_ARG(descs_dev && arena && n > 0 && total_row_blocks > 0 && total_col_blocks > 0 && total_iter_blocks > 0,
                   "dgmr_spectral_sigma_seq_multi: bad args");
    DGMR_CHECK_ARG(max_cout > 0 && max_cout <= 8192, "dgmr_spectral_sigma_seq_multi: max_cout=%d", max_cout);
    DGMR_CHECK_ARG(max_T >= 1 && max_T <= 4096, "dgmr_spectral_sigma_seq_multi: max_T=%d", max_T);
    hipLaunchKernelGGL(sn_rows_multi_kernel, dim3(total_row_blocks), dim3(256), 0, ST, descs_dev, n, arena);
    for (int t = 0; t <= max_T; ++t)  // launch t: iteration t of every module that has one (t == T: its bookkeeping only)
        hipLaunchKernelGGL(sn_iter_multi_kernel, dim3(total_iter_blocks), dim3(256), (max_cout + 64) * sizeof(float), ST, descs_dev, n,
                           arena, t);
    hipLaunchKernelGGL(sn_cols_multi_kernel, dim3(total_col_blocks), dim3(256), 0, ST, descs_dev, n, arena);
    DGMR_CHECK_LAUNCH();
    return 0;
}

static inline dim3 reduce_grid(int G, int64_t R) {
    int64_t bx = (R + 63) / 64;
    if (bx > 512) bx = 512;
    if (bx < 1) bx = 1;
    return dim3((unsigned)bx, (unsigned)G);
}

// deterministic mode: row blocks of a [G][R][C] reduction = reduce_grid's, capped so that the (1 + nb) G 2 C doubles of the caller's buffer
// stay below 32 MB
static inline int det_blocks(int G, int64_t R, int C) {
    const int64_t n = (int64_t)G * 2 * C;
    const int64_t cap = std::max<int64_t>(1, (32ll << 20) / (n * 8));
    return (int)std::min<int64_t>(reduce_grid(G, R).x, cap);
}
extern "C" int64_t dgmr_reduce_doubles(int G, int64_t R, int C) {
    const int64_t n = (int64_t)std::max(G, 1) * 2 * C;
    return g_deterministic ? (1 + (int64_t)det_blocks(std::max(G, 1), R, C)) * n : n;
}
static inline dim3 reduce_grid_mode(int G, int64_t R, int C) {
    return g_deterministic ? dim3((unsigned)det_blocks(G, R, C), (unsigned)G) : reduce_grid(G, R);
}
static inline void reduce_finish(double* sums, int G, int64_t R, int C, hipStream_t s) {
    if (!g_deterministic) return;
    const int64_t n = (int64_t)G * 2 * C;
    hipLaunchKernelGGL(reduce_rows_finish_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, s, sums, det_blocks(G, R, C), n);
}

extern "C" int dgmr_bn_stats(const float* x, double* sums, int G, int64_t R, int C, void* stream) {
    DGMR_CHECK_ARG(x && sums && G > 0 && R > 0 && C > 0, "dgmr_bn_stats: bad args");
    hipLaunchKernelGGL(bn_stats_kernel, reduce_grid_mode(G, R, C), dim3(256), 0, ST, x, sums, R, C, g_deterministic);
    reduce_finish(sums, G, R, C, ST);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_bn_partial_reduce(const float* partials, double* sums, int G, int64_t rows_per_group, int C, void* stream) {
    DGMR_CHECK_ARG(partials && sums && G > 0 && rows_per_group > 0 && C > 0, "dgmr_bn_partial_reduce: bad args");
    const int C2 = 2 * C;
    if (g_deterministic) {
        hipLaunchKernelGGL(bn_partial_reduce_det_kernel, dim3((C2 + 31) / 32, G), dim3(256), 0, ST, partials, sums, rows_per_group, C2);
        DGMR_CHECK_LAUNCH();
        return 0;
    }
    int zs = (int)std::min<int64_t>(64, (rows_per_group + 31) / 32);  // >= 32 rows per slice
    hipLaunchKernelGGL(bn_partial_reduce_kernel, dim3((C2 + 255) / 256, G, zs), dim3(256), 0, ST, partials, sums, rows_per_group, C2);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_bn_bwd_center(double* sums, const float* mean, const float* rstd, int G, int C, void* stream) {
    DGMR_CHECK_ARG(sums && mean && rstd && G > 0 && C > 0, "dgmr_bn_bwd_center: bad args");
    hipLaunchKernelGGL(bn_bwd_center_kernel, dim3(ew_blocks((int64_t)G * C)), dim3(EW_THREADS), 0, ST, sums, mean, rstd, G * C, C);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_bn_finalize(const double* sums, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                int64_t* num_batches_tracked, float* a, float* b, float* save_mean, float* save_rstd, int G,
                                int64_t R, int C, float eps, float momentum, const int32_t* order, void* stream) {
    DGMR_CHECK_ARG(running_mean && running_var && a && b, "dgmr_bn_finalize: null pointer");
    DGMR_CHECK_ARG(sums || G == 1, "dgmr_bn_finalize: eval mode needs G == 1");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + BNF_C - 1) / BNF_C), dim3(BNF_C * BNF_L), 0, ST, sums, gamma, beta, running_mean, running_var,
                       num_batches_tracked, a, b, save_mean, save_rstd, G, R, C, eps, momentum, order);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_bn_bwd_reduce(const float* gy, const float* x, const float* mean, const float* rstd, double* sums, int G,
                                  int64_t R, int C, void* stream) {
    DGMR_CHECK_ARG(gy && x && mean && rstd && sums, "dgmr_bn_bwd_reduce: null pointer");
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, reduce_grid_mode(G, R, C), dim3(256), 0, ST, gy, x, mean, rstd, sums, R, C, g_deterministic);
    reduce_finish(sums, G, R, C, ST);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_bn_bwd_apply(const float* gy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                 const double* sums, const float* dx_add, float* dx, float* dgamma, float* dbeta, int G,
                                 int64_t R, int C, int train, void* stream) {
    DGMR_CHECK_ARG(gy && x && mean && rstd && sums && dx, "dgmr_bn_bwd_apply: null pointer");
    const int64_t n4 = R * C / 4;
    if (C % 4 == 0 && C / 4 <= 4096 && al16(gy) && al16(x) && al16(dx) && al16(dx_add)) {
        // blocks: a multiple of m = (C/4) / gcd(C/4, 256) so that blocks * 256 is a multiple of C/4; ~2 float4 per thread per operand
        const int C4 = C / 4;
        int a = C4, b = 256;
        while (b) { const int t = a % b; a = b; b = t; }
        const int m = C4 / a;
        int64_t blocks = (n4 + 2 * 256 - 1) / (2 * 256);
        blocks = std::min<int64_t>(std::max<int64_t>((blocks + m - 1) / m * m, m), (int64_t)(65536 / m) * m);
        hipLaunchKernelGGL(bn_bwd_apply4_kernel, dim3((unsigned)blocks, G), dim3(256), 0, ST, (const f32x4*)gy, (const f32x4*)x, mean, rstd,
                           gamma, sums, (const f32x4*)dx_add, (f32x4*)dx, n4, C, 1.f / (float)R, train);
    } else {
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_blocks(R * C), G), dim3(EW_THREADS), 0, ST, gy, x, mean, rstd, gamma, sums,
                           dx_add, dx, R, C, train);
    }
    if (dgamma || dbeta)
        hipLaunchKernelGGL(bn_param_grad_kernel, dim3((C + 255) / 256), dim3(256), 0, ST, sums, dgamma, dbeta, G, C);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_colsum(const float* x, float* out, double* tmp, int64_t R, int C, int accumulate, void* stream) {
    DGMR_CHECK_ARG(x && out && tmp, "dgmr_colsum: null pointer");
    if (R <= 4096 && C >= 4096 && C % 4 == 0) {  // few rows, many columns: parallel over columns, no atomics
        hipLaunchKernelGGL(sum_rows_kernel, dim3(ew_blocks(C / 4)), dim3(EW_THREADS), 0, ST, (const f32x4*)x, (f32x4*)out, (int)R,
                           (int64_t)(C / 4), accumulate);
        DGMR_CHECK_LAUNCH();
        return 0;
    }
    (void)hipMemsetAsync(tmp, 0, sizeof(double) * 2 * C, ST);  // (deterministic mode: tmp holds dgmr_reduce_doubles(1, R, C) doubles)
    hipLaunchKernelGGL(colsum_kernel, reduce_grid_mode(1, R, C), dim3(256), 0, ST, x, tmp, R, C, g_deterministic);
    reduce_finish(tmp, 1, R, C, ST);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, ST, tmp, out, C, accumulate);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_group_rowsum(const float* x, const float* w, float* out, int groups, int rows, int64_t n, int rows_per_w,
                                 int w_stride, void* stream) {
    if (w_stride < 1) w_stride = 1;
    DGMR_CHECK_ARG(x && out && groups >= 1 && rows >= 1 && n > 0 && n % 4 == 0 && rows_per_w >= 1 && (n / 4) % w_stride == 0,
                   "dgmr_group_rowsum: groups=%d rows=%d n=%lld (multiple of 4) rows_per_w=%d w_stride=%d", groups, rows, (long long)n,
                   rows_per_w, w_stride);
    hipLaunchKernelGGL(group_rowsum_kernel, dim3(ew_blocks(n / 4), groups), dim3(EW_THREADS), 0, ST, (const f32x4*)x, w, (f32x4*)out,
                       rows, (int64_t)(n / 4), rows_per_w, w_stride);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_repeat_interleave(const float* src, float* dst, int64_t nblocks, int64_t block, int repeat, void* stream) {
    DGMR_CHECK_ARG(src && dst && nblocks > 0 && block > 0 && block % 4 == 0 && repeat >= 1,
                   "dgmr_repeat_interleave: nblocks=%lld block=%lld (multiple of 4) repeat=%d", (long long)nblocks, (long long)block, repeat);
    const int64_t total4 = nblocks * repeat * (block / 4);
    hipLaunchKernelGGL(repeat_interleave_kernel, dim3(ew_blocks(total4)), dim3(EW_THREADS), 0, ST, (const f32x4*)src, (f32x4*)dst,
                       (int64_t)(block / 4), repeat, total4);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_repeat_rows(const float* src, float* dst, int64_t n, int repeat, void* stream) {
    DGMR_CHECK_ARG(src && dst && n > 0 && n % 4 == 0 && repeat >= 1, "dgmr_repeat_rows: n=%lld (multiple of 4) repeat=%d", (long long)n,
                   repeat);
    hipLaunchKernelGGL(repeat_rows_kernel, dim3(ew_blocks(n / 4 * repeat)), dim3(EW_THREADS), 0, ST, (const f32x4*)src, (f32x4*)dst,
                       (int64_t)(n / 4), (int64_t)(n / 4) * repeat);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_affine(const float* x, const float* a, const float* b, float* y, int G, int64_t R, int C, int relu,
                           void* stream) {
    DGMR_CHECK_ARG(x && a && b && y, "dgmr_affine: null pointer");
    hipLaunchKernelGGL(affine_kernel, dim3(ew_blocks(R * C), G), dim3(EW_THREADS), 0, ST, x, a, b, y, R, C, relu);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_pool_fwd(const float* x, const float* addend, float* y, int N, int D, int H, int W, int C, int pd, float scale,
                             const float* mask_src, const float* mask_a, const float* mask_b, int mask_group, void* stream) {
    DGMR_CHECK_ARG(x && y, "dgmr_pool_fwd: null pointer");
    DGMR_CHECK_ARG(C % 4 == 0 && (pd == 1 || pd == 2), "dgmr_pool_fwd: C=%d pd=%d unsupported", C, pd);
    if (scale == 0.f) scale = 1.f / (4.f * pd);
    if (mask_group < 1) mask_group = 1;
    const int64_t total = (int64_t)N * (D / pd) * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, ST, x, addend, y, N, D, H, W, C, pd, scale,
                       mask_src, mask_a, mask_b, mask_group);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_pool_depth2(const float* x, const float* addend, float* y, int N, int D, int64_t plane, void* stream) {
    DGMR_CHECK_ARG(x && y, "dgmr_pool_depth2: null pointer");
    DGMR_CHECK_ARG(N > 0 && D >= 2 && plane > 0 && plane % 4 == 0 && al16(x) && al16(y) && al16(addend),
                   "dgmr_pool_depth2: N=%d D=%d plane=%lld (planes of whole, aligned float4)", N, D, (long long)plane);
    const int64_t plane4 = plane / 4;
    const int planes = N * (D / 2);
    hipLaunchKernelGGL(pool_depth2_kernel, dim3((unsigned)std::min<int64_t>((plane4 + 255) / 256, 1024), (unsigned)std::min(planes, 65535)),
                       dim3(256), 0, ST, (const f32x4*)x, (const f32x4*)addend, (f32x4*)y, N, D, plane4);
    DGMR_CHECK_LAUNCH();
    return 0;
}

// z[n][r][c][co*9 + ky*3 + kx] = sum_{i,j in {0,1}} dy[n][2r + 1 - ky + i][2c + 1 - kx + j][co]  (zero outside the map): the window
// sums of the output gradient that meet input pixel (r, c) under tap (ky, kx) of a 3x3 conv on the nearest-2x upsampled map.  One
// block = P consecutive pixels of an input row: the 4 x (2P + 2) x C window goes through LDS once, every output is 4 LDS reads and
// one coalesced store.
__global__ void upsample_wgrad_sums_kernel(const float* __restrict__ dy, float* __restrict__ z, int H, int W, int C, int P) {
    extern __shared__ float win[];  // [4][2P + 2][C]
    const int wblocks = W / P;
    const int64_t b = blockIdx.x;
    const int c0 = (int)(b % wblocks) * P;
    const int r = (int)((b / wblocks) % H);
    const int n = (int)(b / ((int64_t)wblocks * H));
    const int H2 = 2 * H, W2 = 2 * W, cols = 2 * P + 2, C4 = C >> 2;
    for (int i = threadIdx.x; i < 4 * cols * C4; i += blockDim.x) {
        const int c4 = i % C4, col = (i / C4) % cols, row = i / (C4 * cols);
        const int R = 2 * r - 1 + row, Cc = 2 * c0 - 1 + col;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)R < (unsigned)H2 && (unsigned)Cc < (unsigned)W2)
            v = *reinterpret_cast<const f32x4*>(dy + (((size_t)n * H2 + R) * W2 + Cc) * C + c4 * 4);
        *reinterpret_cast<f32x4*>(win + ((size_t)row * cols + col) * C + c4 * 4) = v;
    }
    __syncthreads();
    const int C9 = 9 * C;
    float* zrow = z + (((size_t)n * H + r) * W + c0) * C9;
    for (int rem = threadIdx.x; rem < C9; rem += blockDim.x) {  // (divisions by constants only: the loop over pixels is inside)
        const int co = rem / 9, tap = rem - co * 9, ky = tap / 3, kx = tap - ky * 3;
        const float* w0 = win + ((size_t)(2 - ky) * cols + 2 - kx) * C + co;
        const int rs = cols * C;
        for (int j = 0; j < P; ++j, w0 += 2 * C) zrow[(size_t)j * C9 + rem] = (w0[0] + w0[C]) + (w0[rs] + w0[rs + C]);
    }
}

extern "C" int dgmr_upsample_wgrad_sums(const float* dy, float* z, int N, int H, int W, int C, void* stream) {
    DGMR_CHECK_ARG(dy && z && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "dgmr_upsample_wgrad_sums: bad args (C=%d)", C);
    int P = 8;
    while (P > 1 && (W % P != 0 || (size_t)4 * (2 * P + 2) * C * sizeof(float) > 48 * 1024)) P >>= 1;
    const size_t lds = (size_t)4 * (2 * P + 2) * C * sizeof(float);
    DGMR_CHECK_ARG(lds <= 64 * 1024, "dgmr_upsample_wgrad_sums: C=%d too wide", C);
    const int64_t blocks = (int64_t)N * H * (W / P);
    DGMR_CHECK_ARG(blocks < (1ll << 31), "dgmr_upsample_wgrad_sums: grid too large");
    hipLaunchKernelGGL(upsample_wgrad_sums_kernel, dim3((unsigned)blocks), dim3(256), lds, ST, dy, z, H, W, C, P);
    DGMR_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The sampler's output layer (generators.py:159-166): relu(BatchNorm(x)) -> spectrally-normalised 1x1 conv to 4 channels, on
// T x draws x B maps of 128 x 128 x 48: 28 M pixels, 5.4 GB per pass over x and 48 multiply-adds per output - pure HBM streaming,
// which the MFMA conv kernels did at 0.4 ... 1.4 TB/s.  Three fp32 VALU kernels, 16 lanes per pixel (one float4 of channels each,
// C <= 64): forward; backward pass 1 (data gradient formed in registers, NOT written: BatchNorm's backward sums, the raw weight
// gradient per call group and the bias gradient as per-block partials, deterministic); backward pass 2 (gradient recomputed,
// BatchNorm backward applied, dx written).  x is read three times and written once in total; exact fp32 in every precision mode.
// A block walks `ppb` consecutive pixels of ONE call group (same BatchNorm statistics, same 1/sigma).
struct HeadGeom {
    int64_t ppg;  // pixels per call group
    int ppb;      // pixels per block (multiple of 16, divides ppg)
    int bpg;      // blocks per group
    int C, C4;
};

__device__ __forceinline__ float sum16(float v) {
    v += __shfl_xor(v, 1, 16);
    v += __shfl_xor(v, 2, 16);
    v += __shfl_xor(v, 4, 16);
    v += __shfl_xor(v, 8, 16);
    return v;
}

__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                       const float* __restrict__ scale, float* __restrict__ y, HeadGeom gm) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane & 15, j = lane >> 4;
    const bool act = q < gm.C4;
    const int g = blockIdx.x / gm.bpg;
    const int64_t p0 = (int64_t)g * gm.ppg + (int64_t)(blockIdx.x - g * gm.bpg) * gm.ppb;
    const int cq = act ? q * 4 : 0;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 av = act ? *reinterpret_cast<const f32x4*>(a + (size_t)g * gm.C + cq) : zero;
    const f32x4 bv = act ? *reinterpret_cast<const f32x4*>(b + (size_t)g * gm.C + cq) : zero;
    f32x4 wv[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) wv[o] = act ? *reinterpret_cast<const f32x4*>(w + (size_t)o * gm.C + cq) : zero;
    const float sc = scale ? scale[g] : 1.f;
    const f32x4 bs = bias ? *reinterpret_cast<const f32x4*>(bias) : zero;
    for (int it = wave * 4 + j; it < gm.ppb; it += 16) {
        const int64_t p = p0 + it;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + p * gm.C + cq);
        f32x4 s;
        float xh[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) xh[c] = fmaxf(fmaf(av[c], xv[c], bv[c]), 0.f);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) d = fmaf(wv[o][c], xh[c], d);
            s[o] = sum16(act ? d : 0.f);
        }
        if (q == 0) {
            f32x4 o4;
#pragma unroll
            for (int o = 0; o < 4; ++o) o4[o] = fmaf(s[o], sc, bs[o]);
            *reinterpret_cast<f32x4*>(y + p * 4) = o4;
        }
    }
}

// the data gradient of pixel p, channels 4q .. 4q+3, behind the relu(BatchNorm) mask: g = [a x + b > 0] * (1/sigma) * W^T dy
__device__ __forceinline__ void head_grad(const f32x4& xv, const f32x4& dyv, const f32x4& av, const f32x4& bv, const f32x4 (&wv)[4],
                                          float sc, float (&gr)[4], float (&xh)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float pre = fmaf(av[c], xv[c], bv[c]);
        xh[c] = fmaxf(pre, 0.f);
        float t = wv[0][c] * dyv[0];
        t = fmaf(wv[1][c], dyv[1], t);
        t = fmaf(wv[2][c], dyv[2], t);
        t = fmaf(wv[3][c], dyv[3], t);
        gr[c] = pre > 0.f ? t * sc : 0.f;
    }
}

__global__ __launch_bounds__(256) void head_bwd_sums_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                            const float* __restrict__ b, const float* __restrict__ w,
                                                            const float* __restrict__ scale, const float* __restrict__ dy,
                                                            float* __restrict__ bn_part, float* __restrict__ w_part,
                                                            float* __restrict__ bias_part, HeadGeom gm) {
    __shared__ float red[4][16][28];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane & 15, j = lane >> 4;
    const bool act = q < gm.C4;
    const int g = blockIdx.x / gm.bpg;
    const int64_t p0 = (int64_t)g * gm.ppg + (int64_t)(blockIdx.x - g * gm.bpg) * gm.ppb;
    const int cq = act ? q * 4 : 0;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 av = act ? *reinterpret_cast<const f32x4*>(a + (size_t)g * gm.C + cq) : zero;
    const f32x4 bv = act ? *reinterpret_cast<const f32x4*>(b + (size_t)g * gm.C + cq) : zero;
    f32x4 wv[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) wv[o] = act ? *reinterpret_cast<const f32x4*>(w + (size_t)o * gm.C + cq) : zero;
    const float sc = scale ? scale[g] : 1.f;
    float acc[28];  // S0[4] | S1[4] | dW[4 o][4 c] | db[4]
#pragma unroll
    for (int i = 0; i < 28; ++i) acc[i] = 0.f;
    for (int it = wave * 4 + j; it < gm.ppb; it += 16) {
        const int64_t p = p0 + it;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + p * gm.C + cq);
        const f32x4 dyv = *reinterpret_cast<const f32x4*>(dy + p * 4);
        float gr[4], xh[4];
        head_grad(xv, dyv, av, bv, wv, sc, gr, xh);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acc[c] += gr[c];
            acc[4 + c] = fmaf(gr[c], xv[c], acc[4 + c]);
#pragma unroll
            for (int o = 0; o < 4; ++o) acc[8 + o * 4 + c] = fmaf(dyv[o], xh[c], acc[8 + o * 4 + c]);
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[24 + o] += dyv[o];
    }
    // fold the four pixel slots of the wave, then the four waves; one partial row per block
#pragma unroll
    for (int i = 0; i < 28; ++i) {
        float v = acc[i];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (j == 0) red[wave][q][i] = v;
    }
    __syncthreads();
    if (wave == 0 && j == 0 && act) {
        float tot[28];
#pragma unroll
        for (int i = 0; i < 28; ++i) tot[i] = (red[0][q][i] + red[1][q][i]) + (red[2][q][i] + red[3][q][i]);
        float* bp = bn_part + (size_t)blockIdx.x * 2 * gm.C;
        float* wp = w_part + (size_t)blockIdx.x * 4 * gm.C;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bp[cq + c] = tot[c];
            bp[gm.C + cq + c] = tot[4 + c];
#pragma unroll
            for (int o = 0; o < 4; ++o) wp[(size_t)o * gm.C + cq + c] = tot[8 + o * 4 + c];
        }
        if (q == 0) {
#pragma unroll
            for (int o = 0; o < 4; ++o) bias_part[(size_t)blockIdx.x * 4 + o] = tot[24 + o];
        }
    }
}

__global__ __launch_bounds__(256) void head_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                             const float* __restrict__ b, const float* __restrict__ w,
                                                             const float* __restrict__ scale, const float* __restrict__ dy,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma, const double* __restrict__ sums,
                                                             float* __restrict__ dx, HeadGeom gm, int train) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane & 15, j = lane >> 4;
    const bool act = q < gm.C4;
    const int g = blockIdx.x / gm.bpg;
    const int64_t p0 = (int64_t)g * gm.ppg + (int64_t)(blockIdx.x - g * gm.bpg) * gm.ppb;
    const int cq = act ? q * 4 : 0;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 av = act ? *reinterpret_cast<const f32x4*>(a + (size_t)g * gm.C + cq) : zero;
    const f32x4 bv = act ? *reinterpret_cast<const f32x4*>(b + (size_t)g * gm.C + cq) : zero;
    f32x4 wv[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) wv[o] = act ? *reinterpret_cast<const f32x4*>(w + (size_t)o * gm.C + cq) : zero;
    const float sc = scale ? scale[g] : 1.f;
    // the same arithmetic as bn_bwd_apply_kernel: v = (g - S0/R - xhat * S1/R) * gamma * rstd
    const float invR = 1.f / (float)gm.ppg;
    float mu[4], rs[4], k0[4], k1[4], gs[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const size_t gc = (size_t)g * gm.C + cq + c;
        mu[c] = mean[gc];
        rs[c] = rstd[gc];
        k0[c] = (float)sums[((size_t)g * 2 + 0) * gm.C + cq + c] * invR;
        k1[c] = (float)sums[((size_t)g * 2 + 1) * gm.C + cq + c] * invR;
        gs[c] = (gamma ? gamma[cq + c] : 1.f) * rs[c];
    }
    for (int it = wave * 4 + j; it < gm.ppb; it += 16) {
        const int64_t p = p0 + it;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + p * gm.C + cq);
        const f32x4 dyv = *reinterpret_cast<const f32x4*>(dy + p * 4);
        float gr[4], xh[4];
        head_grad(xv, dyv, av, bv, wv, sc, gr, xh);
        f32x4 o4;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = gr[c];
            if (train) {
                const float xn = (xv[c] - mu[c]) * rs[c];
                v = v - k0[c] - xn * k1[c];
            }
            o4[c] = v * gs[c];
        }
        if (act) *reinterpret_cast<f32x4*>(dx + p * gm.C + cq) = o4;
    }
}

static bool head_geom(int64_t M, int64_t ppg, int C, HeadGeom* gm) {
    if (!(M > 0 && ppg > 0 && M % ppg == 0 && C % 4 == 0 && C >= 4 && C <= 64 && ppg % 16 == 0)) return false;
    int ppb = 4096;
    while (ppb > 16 && ppg % ppb != 0) ppb >>= 1;
    if (ppg % ppb != 0) return false;
    gm->ppg = ppg;
    gm->ppb = ppb;
    gm->bpg = (int)(ppg / ppb);
    gm->C = C;
    gm->C4 = C / 4;
    return (M / ppg) * gm->bpg < (1ll << 31);
}

extern "C" int dgmr_head_blocks(int64_t M, int64_t pixels_per_group, int C) {
    HeadGeom gm;
    return head_geom(M, pixels_per_group, C, &gm) ? (int)((M / pixels_per_group) * gm.bpg) : 0;
}

extern "C" int dgmr_head_fwd(const float* x, const float* a, const float* b, const float* w, const float* bias, const float* scale,
                             float* y, int64_t M, int64_t pixels_per_group, int C, void* stream) {
    HeadGeom gm;
    DGMR_CHECK_ARG(x && a && b && w && y, "dgmr_head_fwd: null pointer");
    DGMR_CHECK_ARG(head_geom(M, pixels_per_group, C, &gm), "dgmr_head_fwd: M=%lld pixels_per_group=%lld C=%d unsupported", (long long)M,
                   (long long)pixels_per_group, C);
    hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)((M / pixels_per_group) * gm.bpg)), dim3(256), 0, ST, x, a, b, w, bias, scale, y, gm);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_head_bwd_sums(const float* x, const float* a, const float* b, const float* w, const float* scale, const float* dy,
                                  float* bn_partials, float* w_partials, float* bias_partials, int64_t M, int64_t pixels_per_group,
                                  int C, void* stream) {
    HeadGeom gm;
    DGMR_CHECK_ARG(x && a && b && w && dy && bn_partials && w_partials && bias_partials, "dgmr_head_bwd_sums: null pointer");
    DGMR_CHECK_ARG(head_geom(M, pixels_per_group, C, &gm), "dgmr_head_bwd_sums: M=%lld pixels_per_group=%lld C=%d unsupported",
                   (long long)M, (long long)pixels_per_group, C);
    hipLaunchKernelGGL(head_bwd_sums_kernel, dim3((unsigned)((M / pixels_per_group) * gm.bpg)), dim3(256), 0, ST, x, a, b, w, scale, dy,
                       bn_partials, w_partials, bias_partials, gm);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_head_bwd_apply(const float* x, const float* a, const float* b, const float* w, const float* scale, const float* dy,
                                   const float* mean, const float* rstd, const float* gamma, const double* sums, float* dx,
                                   float* dgamma, float* dbeta, int64_t M, int64_t pixels_per_group, int C, int train, void* stream) {
    HeadGeom gm;
    DGMR_CHECK_ARG(x && a && b && w && dy && mean && rstd && sums && dx, "dgmr_head_bwd_apply: null pointer");
    DGMR_CHECK_ARG(head_geom(M, pixels_per_group, C, &gm), "dgmr_head_bwd_apply: M=%lld pixels_per_group=%lld C=%d unsupported",
                   (long long)M, (long long)pixels_per_group, C);
    const int G = (int)(M / pixels_per_group);
    hipLaunchKernelGGL(head_bwd_apply_kernel, dim3((unsigned)(G * gm.bpg)), dim3(256), 0, ST, x, a, b, w, scale, dy, mean, rstd, gamma, sums,
                       dx, gm, train);
    if (dgamma || dbeta)
        hipLaunchKernelGGL(bn_param_grad_kernel, dim3((C + 255) / 256), dim3(256), 0, ST, sums, dgamma, dbeta, G, C);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_pool_bwd(const float* dy, float* dx, int N, int D, int H, int W, int C, int pd, float scale, void* stream) {
    DGMR_CHECK_ARG(dy && dx, "dgmr_pool_bwd: null pointer");
    DGMR_CHECK_ARG(C % 4 == 0 && (pd == 1 || pd == 2), "dgmr_pool_bwd: C=%d pd=%d unsupported", C, pd);
    if (scale == 0.f) scale = 1.f / (4.f * pd);
    const int64_t total = (int64_t)N * D * H * W * (C / 4);
    hipLaunchKernelGGL(pool_bwd_kernel, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, ST, dy, dx, N, D, H, W, C, pd, scale);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_frames_s2d(const float* frames, const int32_t* idx, float* out, int B, int T, int C, int H, int W, int F,
                               int pool, int frame_major, int idx_group, void* stream) {
    if (idx_group < 1) idx_group = B;
    DGMR_CHECK_ARG(frames && out, "dgmr_frames_s2d: null pointer");
    const int p = pool ? 2 : 1;
    DGMR_CHECK_ARG(H % (2 * p) == 0 && W % (2 * p) == 0, "dgmr_frames_s2d: H=%d W=%d not divisible by %d", H, W, 2 * p);
    const int64_t total = (int64_t)B * F * (H / (2 * p)) * (W / (2 * p)) * 4 * C;
    hipLaunchKernelGGL(frames_s2d_kernel, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, ST, frames, idx, out, B, T, C, H, W, F, p,
                       frame_major, idx_group);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_frames_s2d_bwd(const float* dout, const int32_t* idx, float* dframes, int B, int T, int C, int H, int W, int F,
                                   int pool, int frame_major, int idx_group, void* stream) {
    if (idx_group < 1) idx_group = B;
    DGMR_CHECK_ARG(dout && dframes, "dgmr_frames_s2d_bwd: null pointer");
    const int p = pool ? 2 : 1;
    const int64_t total = (int64_t)B * T * C * H * W;
    hipLaunchKernelGGL(frames_s2d_bwd_kernel, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, ST, dout, idx, dframes, B, T, C, H, W, F,
                       p, frame_major, idx_group);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_d2s_frames(const float* x, float* frames, int B, int T, int t, int C, int h, int w, void* stream) {
    DGMR_CHECK_ARG(x && frames && t >= 0 && t < T, "dgmr_d2s_frames: bad args");
    const int64_t total = (int64_t)B * C * 4 * h * w;
    hipLaunchKernelGGL(d2s_frames_kernel<false>, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, ST, x, frames, B, T, t, C, h, w);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_d2s_frames_bwd(const float* dframes, float* dx, int B, int T, int t, int C, int h, int w, void* stream) {
    DGMR_CHECK_ARG(dframes && dx && t >= 0 && t < T, "dgmr_d2s_frames_bwd: bad args");
    const int64_t total = (int64_t)B * C * 4 * h * w;
    hipLaunchKernelGGL(d2s_frames_kernel<true>, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, ST, dframes, dx, B, T, t, C, h, w);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_copy_channels(const float* src, float* dst, int64_t R, int C, int src_C, int src_off, int src_cstride,
                                  int dst_C, int dst_off, int dst_cstride, int accumulate, void* stream) {
    DGMR_CHECK_ARG(src && dst && R > 0 && C > 0, "dgmr_copy_channels: bad args");
    hipLaunchKernelGGL(copy_channels_kernel, dim3(ew_blocks(R * C)), dim3(EW_THREADS), 0, ST, src, dst, R, C, src_C, src_off,
                       src_cstride, dst_C, dst_off, dst_cstride, accumulate);
    DGMR_CHECK_LAUNCH();
    return 0;
}

#define CHECK_N4(n) DGMR_CHECK_ARG((n) > 0 && (n) % 4 == 0, "%s: n=%lld must be a positive multiple of 4", __func__, (long long)(n))

extern "C" int dgmr_gru_gate_fwd(const float* pr, const float* h, float* rh, int64_t n, void* stream) {
    CHECK_N4(n);
    hipLaunchKernelGGL(gru_gate_fwd_kernel, dim3(ew_blocks(n / 4)), dim3(EW_THREADS), 0, ST, (const f32x4*)pr, (const f32x4*)h,
                       (f32x4*)rh, n / 4);
    DGMR_CHECK_LAUNCH();
    return 0;
}
extern "C" int dgmr_gru_gate_bwd(const float* d_rh, const float* pr, const float* h, float* dpr, float* dh, int64_t n, void* stream) {
    CHECK_N4(n);
    hipLaunchKernelGGL(gru_gate_bwd_kernel, dim3(ew_blocks(n / 4)), dim3(EW_THREADS), 0, ST, (const f32x4*)d_rh, (const f32x4*)pr,
                       (const f32x4*)h, (f32x4*)dpr, (f32x4*)dh, n / 4);
    DGMR_CHECK_LAUNCH();
    return 0;
}
extern "C" int dgmr_gru_blend_fwd(const float* pu, const float* h, const float* pc, float* out, int64_t n, void* stream) {
    CHECK_N4(n);
    hipLaunchKernelGGL(gru_blend_fwd_kernel, dim3(ew_blocks(n / 4)), dim3(EW_THREADS), 0, ST, (const f32x4*)pu, (const f32x4*)h,
                       (const f32x4*)pc, (f32x4*)out, n / 4);
    DGMR_CHECK_LAUNCH();
    return 0;
}
extern "C" int dgmr_gru_blend_bwd(const float* dout, const float* pu, const float* h, const float* pc, float* dpu, float* dh,
                                  float* dpc, int64_t n, void* stream) {
    CHECK_N4(n);
    hipLaunchKernelGGL(gru_blend_bwd_kernel, dim3(ew_blocks(n / 4)), dim3(EW_THREADS), 0, ST, (const f32x4*)dout, (const f32x4*)pu,
                       (const f32x4*)h, (const f32x4*)pc, (f32x4*)dpu, (f32x4*)dh, (f32x4*)dpc, n / 4);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_nonfinite_count(const float* x, int64_t n, int32_t* count, void* stream) {
    DGMR_CHECK_ARG(x && count && n > 0, "dgmr_nonfinite_count: bad args");
    hipLaunchKernelGGL(nonfinite_count_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, ST, x, n, count);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_axpby(const float* a, const float* b, float* y, float alpha, float beta, int64_t n, void* stream) {
    DGMR_CHECK_ARG(a && y && n > 0, "dgmr_axpby: bad args");
    hipLaunchKernelGGL(axpby_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, ST, a, b, y, alpha, beta, n);
    DGMR_CHECK_LAUNCH();
    return 0;
}
extern "C" int dgmr_scale_by_dev(const float* x, const float* s, float host_scale, float* y, int64_t n, void* stream) {
    DGMR_CHECK_ARG(x && s && y && n > 0, "dgmr_scale_by_dev: bad args");
    hipLaunchKernelGGL(scale_by_dev_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, ST, x, s, host_scale, y, n);
    DGMR_CHECK_LAUNCH();
    return 0;
}
extern "C" int dgmr_relu_bwd(const float* dy, const float* x, float* dx, int64_t n, void* stream) {
    DGMR_CHECK_ARG(dy && x && dx && n > 0, "dgmr_relu_bwd: bad args");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, ST, dy, x, dx, n);
    DGMR_CHECK_LAUNCH();
    return 0;
}
extern "C" int dgmr_fill(float* p, float value, int64_t n, void* stream) {
    DGMR_CHECK_ARG(p && n > 0, "dgmr_fill: bad args");
    hipLaunchKernelGGL(fill_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, ST, p, value, n);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_attention_fwd(const float* q, const float* k, const float* v, float* beta, float* out, int Cq, int H, int W,
                                  void* stream) {
    DGMR_CHECK_ARG(q && k && v && beta && out, "dgmr_attention_fwd: null pointer");
    const int L = Cq * H;
    hipLaunchKernelGGL(attention_fwd_kernel, dim3(L), dim3(256), (L + 32) * sizeof(float), ST, q, k, v, beta, out, Cq, H, W);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_attention_bwd(const float* dout, const float* q, const float* k, const float* v, const float* beta, float* dq,
                                  float* dk, float* dv, float* tmp, int Cq, int H, int W, void* stream) {
    DGMR_CHECK_ARG(dout && q && k && v && beta && dq && dk && dv && tmp, "dgmr_attention_bwd: null pointer");
    const int L = Cq * H;
    hipLaunchKernelGGL(attention_bwd_q_kernel, dim3(L), dim3(256), (L + 32) * sizeof(float), ST, dout, k, v, beta, dq, tmp, Cq, H, W);
    hipLaunchKernelGGL(attention_bwd_kv_kernel, dim3(L), dim3(64), 0, ST, dout, q, beta, tmp, dk, dv, Cq, H, W);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_relu_sum_hw_fwd(const float* x, float* y, int N, int HW, int C, void* stream) {
    DGMR_CHECK_ARG(x && y, "dgmr_relu_sum_hw_fwd: null pointer");
    hipLaunchKernelGGL(relu_sum_hw_fwd_kernel, dim3(ew_blocks((int64_t)N * C)), dim3(EW_THREADS), 0, ST, x, y, N, HW, C);
    DGMR_CHECK_LAUNCH();
    return 0;
}
extern "C" int dgmr_relu_sum_hw_bwd(const float* dy, const float* x, float* dx, int N, int HW, int C, void* stream) {
    DGMR_CHECK_ARG(dy && x && dx, "dgmr_relu_sum_hw_bwd: null pointer");
    hipLaunchKernelGGL(relu_sum_hw_bwd_kernel, dim3(ew_blocks((int64_t)N * HW * C)), dim3(EW_THREADS), 0, ST, dy, x, dx, N, HW, C);
    DGMR_CHECK_LAUNCH();
    return 0;
}
extern "C" int dgmr_linear1_fwd(const float* x, const float* w, const float* bias, const float* scale, float* y, int N, int C,
                                int scale_group, void* stream) {
    DGMR_CHECK_ARG(x && w && y, "dgmr_linear1_fwd: null pointer");
    if (scale_group < 1) scale_group = N;
    DGMR_CHECK_ARG(N % scale_group == 0, "dgmr_linear1_fwd: N=%d not divisible by scale_group=%d", N, scale_group);
    hipLaunchKernelGGL(linear1_fwd_kernel, dim3(N), dim3(256), 0, ST, x, w, bias, scale, y, C, scale_group);
    DGMR_CHECK_LAUNCH();
    return 0;
}
extern "C" int dgmr_linear1_bwd(const float* dy, const float* x, const float* w, const float* scale, float* dx, float* gw_raw,
                                float* gb, int N, int C, int scale_group, void* stream) {
    DGMR_CHECK_ARG(dy && x && w && dx && gw_raw, "dgmr_linear1_bwd: null pointer");
    if (scale_group < 1) scale_group = N;
    DGMR_CHECK_ARG(N % scale_group == 0, "dgmr_linear1_bwd: N=%d not divisible by scale_group=%d", N, scale_group);
    const int64_t total = (int64_t)(N / scale_group) * C;
    hipLaunchKernelGGL(linear1_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, ST, dy, x, w, scale, dx, gw_raw, gb, N, C,
                       scale_group);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_permute_nt(const float* src, float* dst, int N, int T, int64_t inner, void* stream) {
    DGMR_CHECK_ARG(src && dst && N > 0 && T > 0 && inner > 0 && inner % 4 == 0, "dgmr_permute_nt: bad args");
    const int64_t total = (int64_t)N * T * (inner / 4);
    hipLaunchKernelGGL(permute_nt_kernel, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, ST, (const f32x4*)src, (f32x4*)dst, N, T,
                       inner / 4);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_hinge_disc(const float* s_real, const float* s_gen, float* loss, float* d_real, float* d_gen, int n_real,
                               int n_gen, void* stream) {
    DGMR_CHECK_ARG(s_real && s_gen && loss && n_real > 0 && n_gen > 0, "dgmr_hinge_disc: bad args");
    hipLaunchKernelGGL(hinge_disc_kernel, dim3(1), dim3(256), 0, ST, s_real, s_gen, loss, d_real, d_gen, n_real, n_gen);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t dgmr_grid_cell_acc_doubles(int64_t n) { return g_deterministic ? 1 + (int64_t)ew_blocks(n) : 1; }

extern "C" int dgmr_grid_cell_loss(const float* preds, int K, int64_t pred_stride, const float* target, const float* weights, float cap,
                                   double* acc, float* loss, float mult, float* dweight, int64_t n, void* stream) {
    DGMR_CHECK_ARG(preds && target && acc && loss && K > 0 && n > 0, "dgmr_grid_cell_loss: bad args");
    hipLaunchKernelGGL(grid_cell_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, ST, preds, K, pred_stride, target, weights, cap,
                       acc, dweight, n, g_deterministic);
    hipLaunchKernelGGL(grid_cell_finish_kernel, dim3(1), dim3(256), 0, ST, acc, loss, mult, g_deterministic ? ew_blocks(n) : 0);
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_adam(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                         int step, void* stream) {
    DGMR_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "dgmr_adam: bad args");
    DGMR_CHECK_ARG(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0, "dgmr_adam: betas (%g, %g) out of [0, 1)", beta1, beta2);
    const double bc1 = 1.0 - std::pow(beta1, (double)step);
    const double bc2 = 1.0 - std::pow(beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, ST, p, g, m, v, n, (float)(1.0 - beta1), (float)beta2,
                       (float)(1.0 - beta2), (float)eps, (float)(lr / bc1), (float)std::sqrt(bc2));
    DGMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int dgmr_adam_chunk(void) { return ADAM_CHUNK; }

extern "C" int dgmr_adam_multi(const dgmr_adam_desc* descs, int n_tensors, int total_blocks, double beta1, double beta2, double eps,
                               void* stream) {
    DGMR_CHECK_ARG(descs && n_tensors > 0 && total_blocks > 0, "dgmr_adam_multi: bad args");
    hipLaunchKernelGGL(adam_multi_kernel, dim3(total_blocks), dim3(256), 0, ST, descs, n_tensors, (float)(1.0 - beta1), (float)beta2,
                       (float)(1.0 - beta2), (float)eps);
    DGMR_CHECK_LAUNCH();
    return 0;
}
