/*
 * dgmr_hip.h — C ABI of libdgmr_hip.so: the MI355X (gfx950) kernels behind the DGMR training step.
 *
 * The reference (openclimatefix/skillful_nowcasting) has NO native code and no FFI: every FLOP of
 * `DGMR.training_step` (dgmr/dgmr.py:137-218) runs inside stock torch ops.  The boundary this library
 * replaces is therefore "the torch op call sites on the hot path" (SURVEY.md §8a); each entry point
 * below names the reference call sites it stands in for.  The Python host side
 * (skillful_nowcasting_amd/_lib.py) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - Activations are fp32, channels-last: NHWC for 2-D, NDHWC for 3-D ("N D H W C"; 2-D == D 1).
 *   - Conv weights are fp32 [Cout][KD][KH][KW][Cin] (a torch OIHW / OIDHW parameter held in
 *     channels_last memory format has exactly this physical layout).
 *   - Cin % 4 == 0 and Cout % 4 == 0 for every conv (true for every conv on the DGMR path; the single
 *     Linear(768 -> 1) heads use dgmr_linear1_*).
 *   - Every buffer (incl. workspaces) is allocated by the caller.  Kernels never allocate, free or
 *     synchronise; every launch goes to the `stream` argument (a hipStream_t passed as void*), so the
 *     calls are graph-capture safe.
 *   - Return value: 0 on success, negative on error; dgmr_last_error() returns a thread-local message.
 */
#ifndef DGMR_HIP_H
#define DGMR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGMR_ABI_VERSION 11

int dgmr_abi_version(void);
const char* dgmr_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Convolution (implicit GEMM on v_mfma_f32_32x32x2_f32).
 * Replaces torch.nn.Conv2d / Conv3d forward+backward at every `conv2d(` / `get_conv_layer` call site:
 * dgmr/common.py:43-66,113-137,192-215,266-286,350-384,451-455; dgmr/generators.py:52-56,67-73,84-90,
 * 101-107,115-121; dgmr/layers/ConvGRU.py:29-55; dgmr/layers/Attention.py:38-66 — together with the
 * ops the reference applies around them, fused into the operand load / epilogue:
 *   relu on load            F.relu / nn.ReLU before a conv (common.py:77,80,146,151,230,232,296,298)
 *   affine+relu on load     BatchNorm2d(train) + ReLU before a conv (common.py:76-80,145-151; generators.py:176)
 *   nearest 2x on load      nn.Upsample(scale_factor=2) before a conv (common.py:142,148)
 *   1/sigma epilogue scale  spectral_norm's W/sigma (torch/nn/utils/parametrizations.py:506-521)
 *   residual add            `x2 + sc` (common.py:83,154,237,300)
 * ---------------------------------------------------------------------------------------------- */
typedef struct dgmr_conv_args {
    const float* x;        /* input [N][D][Hin][Win][Cin]; Hin,Win = H,W  (or H/2,W/2 when upsample) */
    const float* w;        /* [Cout][KD][KH][KW][Cin] */
    const float* bias;     /* [Cout] or NULL */
    const float* scale;    /* per-group multiplier (1/sigma): scale[n / scale_group]; NULL = 1 */
    const float* pre_a;    /* NULL, or per-(group,channel) a: x' = relu(x*a+b), [N/pre_group][Cin] */
    const float* pre_b;
    const float* addend;   /* NULL or [M][Cout]: added to the accumulator BEFORE scale */
    const float* residual; /* NULL or [M][Cout]: added after scale+bias */
    const float* mask_src; /* NULL or [M][Cout]: y = (mask_src*mask_a+mask_b > 0) ? y : 0  (relu backward) */
    const float* mask_a;   /* NULL (plain mask_src > 0) or [N/mask_group][Cout] */
    const float* mask_b;
    float* y;              /* output [N][D][H][W][Cout] */
    int32_t N, D, H, W;    /* output extent (== input extent after the optional upsample) */
    int32_t Cin, Cout;
    int32_t KD, KH, KW;    /* each 1 or 3; 'same' zero padding K/2, stride 1 */
    int32_t upsample;      /* 1: x is [N][D][H/2][W/2][Cin], nearest-upsampled on the fly */
    int32_t pre_relu;      /* 1: relu on the input (ignored when pre_a != NULL, which implies relu) */
    int32_t scale_group;   /* samples per scale group (>=1) */
    int32_t pre_group;     /* samples per pre_a/pre_b group */
    int32_t mask_group;
    int32_t act_relu;      /* 1: relu after scale+bias, before the residual add (F.relu(conv(..)), common.py:424) */
    /* -- ABI 3 -- */
    int32_t w_cin;         /* weights are the input-channel slice [w_coff, w_coff+Cin) of a [Cout][KD][KH][KW][w_cin] tensor
                              (the x / h halves of a ConvGRU conv applied to torch.cat([x, h], 1): ConvGRU.py:69,79); 0: dense */
    int32_t w_coff;
    int32_t epi_mode;      /* DGMR_EPI_*: fused tail after scale+bias (replaces act_relu / residual / mask when != PLAIN) */
    int32_t ksplit;        /* out: ignored on input; the library splits K itself when splitk_ws is given and the grid is small */
    const float* gru_h;    /* [M][Cout] previous hidden state (GRU modes) */
    const float* gru_pu;   /* [M][Cout] update-gate pre-activation (DGMR_EPI_GRU_BLEND) */
    float* pre_out;        /* [M][Cout] receives the pre-activation (scale+bias applied) in the GRU modes (needed by the backward);
                              NULL: not stored (forwards without a graph: a fifth of the recurrent step's HBM traffic) */
    float* splitk_ws;      /* NULL, or scratch for split-K partial sums */
    int64_t splitk_ws_bytes;
    const uint16_t* w_split; /* NULL, or the SAME weights (of the slice, if w_cin/w_coff select one) pre-split into dense bf16
                                planes [P][Cout][KH*KW][Cin] by dgmr_split_weights (P = 2 in the modes bf16x3 / bf16, 3 in bf16x6 - the
                                library reads as many planes as the mode set when the launch is issued): lets the bf16 modes run 3x3
                                convs of the big feature maps through the LDS-window kernel */
    int32_t residual_up;     /* 1: residual is [N][H/2][W/2][Cout], added with nearest-2x upsampling (the 1x1 shortcut of an upsampling
                                G-block evaluated before the upsample: conv1x1(up(x)) == up(conv1x1(x)), common.py:142-143,154) */
    int32_t reserved0;       /* must be 0 (library-internal) */
    /* -- ABI 6 -- */
    float* stats_out;        /* NULL, or [dgmr_conv_stats_rows(args)][2][Cout]: per-workgroup-tile partial sums (sum y, sum y^2) of the
                                OUTPUT, one row per pixel tile - the BatchNorm batch statistics of the next layer taken in this conv's
                                epilogue instead of by a second pass over y (GBlock: bn2(first_conv(..)), common.py:76-80,145-151).
                                Rows are ordered like the samples; dgmr_bn_partial_reduce folds them per statistics group.
                                With mask_src (data gradient through relu(BatchNorm(x))) the second sum is sum y * mask_src instead:
                                with dgmr_bn_bwd_center the two sums BatchNorm's backward needs (common.py:76,145 backwards). */
    /* -- ABI 7 -- */
    const uint16_t* w_phase; /* NULL, or (with `upsample`, 3x3, bf16 modes) the tap sums of the four output-pixel parities as bf16 planes
                                [2][4*Cout][2*2][Cin]: dgmr_upsample_phase_weights then dgmr_split_weights.  The upsampling conv of
                                UpsampleGBlock (common.py:142,148) then runs as four 2x2 convs on the low-resolution input - 16 instead of
                                36 multiply steps per input pixel, same sums up to the fp32 rounding of the tap sums. */
    int32_t pool2;           /* 1: y = 2x2 sum pool of the conv, [N][H/2][W/2][Cout] (H, W stay the conv's map; mask_src / stats_out refer to
                                the pooled output) - the data gradient of an upsampling conv (common.py:142,148 backwards) without
                                writing the full-resolution gradient.  Needs w_phase = dgmr_pool2_phase_weights + dgmr_split_weights
                                ([2][Cout][16][Cin]) and a conv the window kernel takes: ask dgmr_conv_pool2_supported.
                                ABI 10: also the FORWARD of a DBlock's last conv + AvgPool (common.py:233-237; tap sums x 0.25): `residual`
                                is then [N][H/2][W/2][Cout]; and 3x3x3 convs (KD = 3, D > 1), plane by plane, w_phase =
                                [2][Cout][3 * 16][Cin] (dgmr_pool2_phase_weights per depth tap): y = [N][D][H/2][W/2][Cout], the 2 x 2
                                spatial half of AvgPool3d - dgmr_pool_depth2 finishes it. */
    int32_t reserved1;
    /* -- ABI 8 -- DGMR_EPI_GRU_GATES2: the ConvGRU's read AND update gate convs in one launch (ConvGRU.py:69-76: both convolve the same
       cat[x, h]); see the define below */
    const float* scale2;     /* 1/sigma of the second (update) gate conv, indexed like `scale` */
    const float* bias2;      /* [gru_split] */
    const float* addend2;    /* [M][gru_split], or NULL */
    float* y2;               /* [M][gru_split]: the update gate's pre-activation */
    int32_t gru_split;       /* C: output columns [0, C) are the read gate's, [C, 2C) the update gate's; Cout == 2 C, C % 4 == 0 */
    int32_t reserved2;
} dgmr_conv_args;

/* ---- the sampler's output layer: relu(BatchNorm(x)) -> 1x1 conv to 4 channels (generators.py:159-166), streaming fp32 kernels ----
 * x: [M][C] (C <= 64, % 4); a, b: BatchNorm affine per call group [G][C]; w: [4][C]; scale: 1/sigma per call group [G] or NULL;
 * a call group = pixels_per_group consecutive pixels (M = G * pixels_per_group).  Exact fp32 in every precision mode.
 * dgmr_head_blocks: number of per-block partial rows the backward writes (0: shape unsupported, use the conv entry points). */
int dgmr_head_blocks(int64_t M, int64_t pixels_per_group, int C);
int dgmr_head_fwd(const float* x, const float* a, const float* b, const float* w, const float* bias, const float* scale, float* y,
                  int64_t M, int64_t pixels_per_group, int C, void* stream);
/* Backward pass 1 - the data gradient g = [a x + b > 0] (1/sigma) W^T dy is formed in registers and NOT written: per block
 * bn_partials[blk][2][C] = (sum g, sum g x)  (-> dgmr_bn_partial_reduce, dgmr_bn_bwd_center), w_partials[blk][4][C] = sum dy (x) relu(a x + b)
 * (the raw weight gradient; blocks of a call group are consecutive: dgmr_wgrad_reduce with nsplit = blocks), bias_partials[blk][4]. */
int dgmr_head_bwd_sums(const float* x, const float* a, const float* b, const float* w, const float* scale, const float* dy,
                       float* bn_partials, float* w_partials, float* bias_partials, int64_t M, int64_t pixels_per_group, int C,
                       void* stream);
/* Backward pass 2 - g recomputed, BatchNorm's backward applied (dgmr_bn_bwd_apply's arithmetic), dx written; dgamma / dbeta
 * accumulated from `sums` when given. */
int dgmr_head_bwd_apply(const float* x, const float* a, const float* b, const float* w, const float* scale, const float* dy,
                        const float* mean, const float* rstd, const float* gamma, const double* sums, float* dx, float* dgamma,
                        float* dbeta, int64_t M, int64_t pixels_per_group, int C, int train, void* stream);

/* Weight gradient of an upsampling conv (common.py:142,148 backwards) as a 1x1 problem: z[n][r][c][co*9 + ky*3 + kx] = the sum of the
 * 2x2 pixels of dy ([N][2H][2W][C]) that meet input pixel (r, c) under tap (ky, kx), so that dW[co][ky][kx][ci] = sum over the INPUT
 * pixels of z * pre(x) - dgmr_conv_wgrad with KH = KW = 1, Cout = 9 C, dy = z: a quarter of the multiply steps of the gradient taken
 * on the upsampled map, and its partial[..][co*9 + tap][ci] is the layout dgmr_wgrad_reduce expects.  z: [N][H][W][9 C] floats. */
int dgmr_upsample_wgrad_sums(const float* dy, float* z, int N, int H, int W, int C, void* stream);
/* 1 when dgmr_conv_fwd accepts these arguments with pool2 = 1 (host arithmetic, no launch). */
int dgmr_conv_pool2_supported(const dgmr_conv_args* a);
/* out[co][(p*2+q)*4 + a*2+b][ci]: the 4x4 stride-2 kernel of "3x3 conv then 2x2 sum pool" (row u = 2a + 1 - p sums the taps ky with
 * ky + i = u, i in {0,1}), grouped by the parity (p, q) of the input pixel.  w: [Cout][3][3][Cin] fp32; out: [Cout][16][Cin] fp32. */
int dgmr_pool2_phase_weights(const float* w, float* out, int Cout, int Cin, void* stream);

/* out[(py*2+px)*Cout + co][a][b][ci] = sum of w[co][ky][kx][ci] over the taps (ky, kx) that read input pixel (h + py - 1 + a,
 * w + px - 1 + b) when the conv runs on the nearest-2x upsampled map at output pixel (2h + py, 2w + px): ky in {0} | {1,2} for py = 0,
 * a = 0 | 1; {0,1} | {2} for py = 1; kx likewise.  w: [Cout][3][3][Cin] fp32 (channels-last OIHW); out: [4*Cout][2][2][Cin] fp32. */
int dgmr_upsample_phase_weights(const float* w, float* out, int Cout, int Cin, void* stream);
/* 1 when dgmr_conv_fwd accepts these arguments with epi_mode = DGMR_EPI_GRU_GATES2 in the current arithmetic mode (host arithmetic only) */
int dgmr_conv_gates2_supported(const dgmr_conv_args* a);

/* Number of partial-sum rows dgmr_conv_fwd writes to stats_out for these arguments (host arithmetic, no launch): 0 when the kernel
 * the library would dispatch has no fused statistics (only the LDS-window 3x3 kernels of the bf16 modes do), then stats_out must
 * be NULL and the caller takes the statistics with dgmr_bn_stats. */
int dgmr_conv_stats_rows(const dgmr_conv_args* a);

#define DGMR_EPI_PLAIN 0
#define DGMR_EPI_GRU_GATE 1  /* pre_out = v ; y = sigmoid(v) * gru_h                              (ConvGRU.py:69-71,78) */
#define DGMR_EPI_GRU_BLEND 2 /* pre_out = v ; y = s*gru_h + (1-s)*relu(v), s = sigmoid(gru_pu)     (ConvGRU.py:80-84) */
/* Read and update gate of one ConvGRU step fused (ConvGRU.py:69-76): ONE conv with Cout = 2 C output columns whose weights (w_split
 * rows) are the read gate's [0, C) followed by the update gate's [C, 2C) - the input halo (h, or cat[x, h]) is staged once for both.
 * Every tensor of the epilogue has C channels per pixel:
 *   column c <  C:  v = (acc + addend[m][c]) * scale[g] + bias[c];        pre_out[m][c] = v (if given);  y[m][c] = sigmoid(v) * gru_h[m][c]
 *   column c >= C:  v = (acc + addend2[m][c-C]) * scale2[g] + bias2[c-C];  y2[m][c-C] = v
 * Only for convs the LDS-DMA window kernel takes, with 16-byte aligned tensors: ask dgmr_conv_gates2_supported first. */
#define DGMR_EPI_GRU_GATES2 3

/* out[k][i] = bf16(w_i - out[0][i] - ... - out[k-1][i]), k < planes, for the [rows][Cin] slice [w_coff, w_coff+Cin) of a
 * [rows][w_cin] weight matrix (rows = Cout * taps; w_cin == 0: dense).  Cin must be even.  planes: 2 for DGMR_PREC_BF16X3 /
 * DGMR_PREC_BF16, 3 for DGMR_PREC_BF16X6 (three bf16 hold all 24 mantissa bits: the planes sum to w exactly).  Valid until the
 * weights change.  plane_stride: elements between two planes of `out` (0: dense, rows * Cin) - lets several weight tensors share one
 * plane set (the fused ConvGRU gates: the read gate's rows followed by the update gate's).  (ABI 8: `planes`, `plane_stride` added.) */
int dgmr_split_weights(const float* w, uint16_t* out, int64_t rows, int Cin, int w_cin, int w_coff, int planes, int64_t plane_stride,
                       void* stream);

/* Arithmetic of the forward / data-gradient contraction (process-wide; tensors in HBM stay fp32, accumulation is fp32):
 *   DGMR_PREC_F32     exact fp32 on v_mfma_f32_32x32x2_f32 (157 TF peak) -- the parity mode
 *   DGMR_PREC_BF16X3  each operand split into two bf16 terms, 3 x v_mfma_f32_32x32x16_bf16 per product: products carry 16
 *                     significant bits (~2^-16 relative error per product - NOT fp32 arithmetic), 833 TF effective peak
 *   DGMR_PREC_BF16    operands rounded to bf16 (BASELINE.json configs[1]): 2.5 PF peak, ~3 significant digits
 *   DGMR_PREC_BF16X6  (ABI 8) three bf16 terms per operand (exact: 3 x 8 bits = the fp32 mantissa), the six products a_i * b_j with
 *                     i + j <= 2 on six MFMAs: the dropped terms are <= 2^-25 |ab|, below the rounding of an fp32 product -
 *                     fp32-faithful contractions at 417 TF effective peak (2.6 x the exact-f32 MFMA's)
 * A caller may switch the mode between launches (it is read when a launch is issued): the module layer runs the discriminator
 * forward in BF16X6 inside a BF16X3 step (skillful_nowcasting_amd.ops.set_precision). */
#define DGMR_PREC_F32 0
#define DGMR_PREC_BF16X3 1
#define DGMR_PREC_BF16 2
#define DGMR_PREC_BF16X6 3
int dgmr_set_precision(int mode);
int dgmr_get_precision(void);

/* y = act(conv(pre(x), w) (+addend) *scale + bias) (+residual), masked.  Forward AND data-gradient (the
 * latter with dgmr_conv_flip_weights()'ed weights and dy as x). */
int dgmr_conv_fwd(const dgmr_conv_args* a, void* stream);

/* w_t[Cin][KD][KH][KW][Cout] = w[Cout][KD-1-kd][KH-1-kh][KW-1-kw][w_coff + ci]: weights of the transposed
 * (data-gradient) convolution, for the input-channel slice [w_coff, w_coff+Cin) of a tensor with w_cin input channels
 * (w_cin == 0: dense, w_cin = Cin). */
int dgmr_conv_flip_weights(const float* w, float* w_t, int Cout, int Cin, int KD, int KH, int KW, int w_cin, int w_coff,
                           void* stream);

typedef struct dgmr_wgrad_args {
    const float* x;      /* forward input, as in dgmr_conv_args (pre_* / upsample applied on load) */
    const float* dy;     /* [M][Cout] gradient of the conv output (before scale: caller folds scale later) */
    const float* pre_a;
    const float* pre_b;
    float* partial;      /* workspace [nsplit][Cout][K] (K = KD*KH*KW*Cin), written, not accumulated */
    int32_t N, D, H, W, Cin, Cout, KD, KH, KW;
    int32_t upsample, pre_relu, pre_group;
    int32_t nsplit;      /* >= 1: the M = N*D*H*W reduction is cut into nsplit contiguous slabs */
    int32_t groups;      /* >= 1, divides N and nsplit: consecutive N/groups samples form a group (one spectral-norm call:
                            forecast step / frame); a slab never straddles two groups */
    float* bias_grad;    /* NULL, or [Cout]: += column sums of dy (the conv's bias gradient) in the same pass */
    /* -- ABI 11 -- deterministic bias gradient: a workspace of `bias_rows` rows of Cout floats (bias_rows as filled in by
       dgmr_conv_wgrad_plan).  With it every slab's column sums land in a row of their own (or, for the kernels whose workgroups meet
       in a channel, a fixed-order column-sum pass over dy fills the rows) and ONE thread per channel adds the rows up in order:
       bias_grad is bit-identical from run to run.  NULL: the slabs' sums meet in float atomics (any order), whatever
       dgmr_set_deterministic says - the rows are the caller's to provide. */
    float* bias_partial;
    int32_t bias_rows;
    int32_t bias_stride; /* set by the library (0 / Cout); ignored on input */
} dgmr_wgrad_args;

/* Weight-gradient partial sums: partial[s] = dy[slab s]^T * im2col(pre(x))[slab s]. */
int dgmr_conv_wgrad(const dgmr_wgrad_args* a, void* stream);
/* Suggested nsplit (a multiple of groups) for a problem (pure host arithmetic). */
int dgmr_conv_wgrad_nsplit(int M, int Cout, int K, int groups);
/* Same with the whole geometry in view (the kernel choice depends on it): fills a->nsplit from the other fields; the caller then
 * sizes `partial` and calls dgmr_conv_wgrad with the same struct. */
int dgmr_conv_wgrad_plan(dgmr_wgrad_args* a);

/* With P_q = sum of group q's slabs:  g[Cout*K] = sum_q scale[q] * P_q  (scale == NULL: 1);  dot[q] += <P_q, w>  (dot == NULL:
 * skipped; otherwise [groups], zeroed by the caller / the previous finalize).  groups <= 128.
 * ABI 11: `dot` holds dgmr_wgrad_dot_floats(groups) floats - `groups` normally; in deterministic mode groups * (1 + 1024): the
 * workgroups leave their partial dots in rows of their own behind the first `groups` floats and one thread per group adds them up in
 * order (otherwise they meet in float atomics). */
int dgmr_wgrad_dot_floats(int groups);
int dgmr_wgrad_reduce(const float* partial, int nsplit, int groups, int64_t numel, const float* w, const float* scale, float* g,
                      float* dot, void* stream);
/* Same for a weight gradient computed on an input-channel slice: partial is [nsplit][Cout][taps][cin_slice]; element
 * (co, tap, ci) is written to / dotted with index (co*taps + tap)*cin_total + coff + ci of g / w.  Two calls (x and h halves)
 * fill one g and accumulate one set of dots. */
int dgmr_wgrad_reduce_slice(const float* partial, int nsplit, int groups, int Cout, int taps, int cin_slice, int cin_total, int coff,
                            const float* w, const float* scale, float* g, float* dot, void* stream);
/* Spectral-norm chain rule (torch/nn/utils/parametrizations.py:515-521), summed over the `groups` calls of the module that
 * one batched launch covered (g already carries the 1/sigma_q factors, see dgmr_wgrad_reduce):
 *   gw[i][k] (+)= g[i][k] - sum_q dot[q]*inv_sigma[q]^2 * u[q][i]*v[q][perm(k)],  then dot[0..groups) = 0.
 * u: [groups][Cout], v: [groups][K] in torch's logical (ci, kd, kh, kw) order; w/g in physical (kd,kh,kw,ci) order.
 * accumulate != 0 adds into gw instead of overwriting.  u == v == NULL: no rank-1 term (plain or scalar-gain conv). */
int dgmr_sn_wgrad_finalize(const float* g, float* gw, float* dot, const float* inv_sigma, const float* u, const float* v,
                           int Cout, int Cin, int taps, int groups, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Spectral-norm power iteration — torch/nn/utils/parametrizations.py:454-521, called on every
 * `spectral_norm(` module forward (list in SURVEY.md §8a row a14).
 * train != 0: u <- normalize(W v), v <- normalize(W^T u) in place (eps-clamped), inv_sigma = 1/(u^T W v).
 * train == 0: inv_sigma = 1/(u^T W v) with the stored u, v.
 * u_save/v_save (may be NULL) receive copies of the u, v used for sigma (needed by the backward).
 * scratch: 4 floats, zero on entry, left zero on exit.
 * ---------------------------------------------------------------------------------------------- */
int dgmr_spectral_sigma(const float* w, float* u, float* v, float* u_save, float* v_save, float* inv_sigma,
                        float* scratch, float* tmp /* [Cout + K] */, int Cout, int Cin, int taps, float eps, int train,
                        void* stream);
/* T consecutive train-mode calls of ONE module (a sampler conv is called once per forecast step, a discriminator conv once
 * per frame: generators.py:153-178, discriminators.py:119-133,201-226) in one go.  gram = W W^T [Cout][Cout] (computed by the
 * caller with dgmr_conv_fwd on the [Cout][K] weight matrix, valid until W changes): the T dependent power iterations then
 * run on gram alone and W is read twice instead of 2T times.  Outputs per call t: inv_sigma[t], u_hist[t][Cout],
 * v_hist[t][K] (the vectors sigma_t was computed with); u, v are left at their values after call T.  T <= 32.
 * tmp: [Cout + T] floats; scratch as above. */
int dgmr_spectral_sigma_seq(const float* w, const float* gram, float* u, float* v, float* u_hist, float* v_hist,
                            float* inv_sigma, float* scratch, float* tmp, int Cout, int Cin, int taps, float eps, int T,
                            void* stream);

/* The same for many modules at once: every spectral-norm call sequence of one generator / discriminator forward in three
 * launches (the iterations do not depend on activations; the chains of all modules run concurrently, one workgroup each).
 * descs: DEVICE array of n descriptors.  Outputs live in one caller-allocated float arena (offsets in floats):
 * inv_sigma[T], u_hist[T][Cout], v_hist[T][K], tmp[3*Cout + T].  row_block0 / col_block0 / iter_block0: exclusive prefix sums of
 * Cout, of ceil(K/64) and of ceil(Cout/32) over the descriptors; the totals are passed alongside.  max_cout: largest Cout (sizes
 * the iteration kernel's LDS), max_T: largest T (the chain runs as max_T + 1 launches, one power iteration of every module each). */
typedef struct dgmr_sn_desc {
    const float* w;
    const float* gram;
    float* u;
    float* v;
    int64_t inv_sigma_off;
    int64_t u_hist_off;
    int64_t v_hist_off;
    int64_t tmp_off;
    int32_t Cout, Cin, taps, T;
    float eps;
    int32_t row_block0, col_block0, iter_block0;
    /* perm[t] (DEVICE array of T ints, or NULL = identity): the group ("slot") of the batched launch that the t-th call of the
     * sequence belongs to.  inv_sigma / u_hist / v_hist are written at slot perm[t]; the module's u, v end at the LAST call's.
     * Batched generator draws run the calls (draw d, step t) as groups [t][d] of one batch while the reference's call order is
     * draw-major (and draw-reversed in the activation-checkpoint recompute, dgmr/dgmr.py:176). */
    const int32_t* perm;
} dgmr_sn_desc;
int dgmr_spectral_sigma_seq_multi(const dgmr_sn_desc* descs_dev, int n, int total_row_blocks, int total_col_blocks,
                                  int total_iter_blocks, int max_cout, int max_T, float* arena, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Per-channel reductions / BatchNorm — torch.nn.BatchNorm2d (common.py:38-39,108-109; generators.py:113)
 * and BatchNorm1d (discriminators.py:102,194) in train and eval mode.
 * x is [G][R][C] (G groups of R rows); everything is per (group, channel).
 * ---------------------------------------------------------------------------------------------- */
/* sums[g][0][c] += sum_r x ; sums[g][1][c] += sum_r x^2   (double accumulators, zeroed by the caller). */
int dgmr_bn_stats(const float* x, double* sums, int G, int64_t R, int C, void* stream);
/* train: mean/var from sums -> a = gamma*rstd, b = beta - mean*a ; updates running stats once per group (momentum, unbiased
 * var) and num_batches_tracked += G.  The updates are applied in the order order[0], order[1], ... (DEVICE array of G group
 * indices; NULL = 0..G-1): the order in which the reference called the module on those groups.
 * eval (sums == NULL): a,b from running stats (G == 1).  save_mean/save_rstd: [G][C] for the backward. */
int dgmr_bn_finalize(const double* sums, const float* gamma, const float* beta, float* running_mean, float* running_var,
                     int64_t* num_batches_tracked, float* a, float* b, float* save_mean, float* save_rstd, int G,
                     int64_t R, int C, float eps, float momentum, const int32_t* order, void* stream);
/* sums[g][0][c] += sum over the group's rows of partials[row][0][c], sums[g][1][c] likewise: `partials` is what dgmr_conv_fwd wrote to
 * stats_out ([G * rows_per_group][2][C] floats, fp32 sums over one pixel tile each); double accumulation from here on. */
int dgmr_bn_partial_reduce(const float* partials, double* sums, int G, int64_t rows_per_group, int C, void* stream);
/* In place: sums[g][1][c] = rstd[g][c] * (sums[g][1][c] - mean[g][c] * sums[g][0][c]): turns (sum g, sum g * x) - folded by
 * dgmr_bn_partial_reduce from the stats_out rows of a data-gradient conv - into dgmr_bn_bwd_reduce's (sum g, sum g * xhat). */
int dgmr_bn_bwd_center(double* sums, const float* mean, const float* rstd, int G, int C, void* stream);
/* sums[g][0][c] = sum_r g ; sums[g][1][c] = sum_r g * xhat   with xhat = (x-mean)*rstd  (sums zeroed by caller). */
int dgmr_bn_bwd_reduce(const float* gy, const float* x, const float* mean, const float* rstd, double* sums, int G,
                       int64_t R, int C, void* stream);
/* dx = a*(gy - sums0/R - xhat*sums1/R) (+ dx_add); dgamma[c] += sum_g sums1, dbeta[c] += sum_g sums0 (if non-NULL). */
int dgmr_bn_bwd_apply(const float* gy, const float* x, const float* mean, const float* rstd, const float* gamma,
                      const double* sums, const float* dx_add, float* dx, float* dgamma, float* dbeta, int G, int64_t R,
                      int C, int train, void* stream);
/* out[c] (+)= sum_r x[r][c]  (conv / linear bias gradient).  tmp: 2*C doubles of scratch (deterministic mode:
 * dgmr_reduce_doubles(1, R, C)), cleared by the call itself; not touched at all when R <= 4096 && C >= 4096 && C % 4 == 0
 * (few rows, many columns: one thread per column quad) - any non-NULL pointer will do there. */
int dgmr_colsum(const float* x, float* out, double* tmp, int64_t R, int C, int accumulate, void* stream);
/* y = x*a[g][c]+b[g][c]  (BatchNorm1d apply; no relu) */
int dgmr_affine(const float* x, const float* a, const float* b, float* y, int G, int64_t R, int C, int relu, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pooling / resampling / layout — nn.AvgPool2d/3d (common.py:189-191; discriminators.py:68,165),
 * PixelUnshuffle / PixelShuffle (common.py:326; discriminators.py:69,166; generators.py:123),
 * nn.Upsample backward (common.py:121), einops rearrange (common.py:423).
 * ---------------------------------------------------------------------------------------------- */
/* y[N][D/pd][H/2][W/2][C] = mean over pd x 2 x 2 window (pd in {1,2}) (+ addend).  scale overrides 1/window when != 0
 * (scale = 1 gives the sum-pool that is the backward of a nearest upsample).  Optional relu-backward mask on the
 * output: y = (mask_src*mask_a+mask_b > 0) ? y : 0. */
int dgmr_pool_fwd(const float* x, const float* addend, float* y, int N, int D, int H, int W, int C, int pd, float scale,
                  const float* mask_src, const float* mask_a, const float* mask_b, int mask_group, void* stream);
/* dx[N][D][H][W][C] = dy[n][d/pd][h/2][w/2][c] * scale  (scale = 1/window for avg-pool backward). */
int dgmr_pool_bwd(const float* dy, float* dx, int N, int D, int H, int W, int C, int pd, float scale, void* stream);
/* y[N][D/2][plane] = (x[n][2j] + x[n][2j+1]) / 2 (+ addend): the depth half of nn.AvgPool3d(2) (common.py:189) over planes of `plane`
 * floats (H/2 * W/2 * C), when the 2 x 2 spatial half already rode in the 3x3x3 conv that produced x (dgmr_conv_args.pool2).  A last
 * odd plane of x is dropped, like AvgPool3d does. */
int dgmr_pool_depth2(const float* x, const float* addend, float* y, int N, int D, int64_t plane, void* stream);
/* frames [B][T][C][H][W] (reference layout) -> channels-last space-to-depth tiles.
 * out[(b*F+f)][H/(2p)][W/(2p)][4C] with channel (c*4 + dy*2 + dx) (PixelUnshuffle(2) order), frame = idx[f],
 * p = pool ? 2 : 1 (AvgPool2d(2) first).  idx == NULL -> frames 0..F-1.  out_frame_major: 1 -> row (f*B+b).
 * idx is [B / idx_group][F]: samples b .. b + idx_group - 1 share one row of frame indices (idx_group <= 0: one row for all) -
 * several discriminator calls, each with its own random frame draw (discriminators.py:199), batched into one. */
int dgmr_frames_s2d(const float* frames, const int32_t* idx, float* out, int B, int T, int C, int H, int W, int F, int pool,
                    int frame_major, int idx_group, void* stream);
int dgmr_frames_s2d_bwd(const float* dout, const int32_t* idx, float* dframes /* written, every element (ABI 11: gather form, no atomics) */, int B, int T, int C, int H,
                        int W, int F, int pool, int frame_major, int idx_group, void* stream);
/* channels-last [B][h][w][4C] -> frames[b][t][c][2h][2w] (PixelShuffle(2)) and its backward. */
int dgmr_d2s_frames(const float* x, float* frames, int B, int T, int t, int C, int h, int w, void* stream);
int dgmr_d2s_frames_bwd(const float* dframes, float* dx, int B, int T, int t, int C, int h, int w, void* stream);
/* dst[t][n][i] = src[n][t][i], i < inner (inner % 4 == 0): frames of a [N][T][H][W][C] tensor to a frame-major batch
 * (discriminators.py:119-120 `x[:, :, idx]` for every idx at once) and back. */
int dgmr_permute_nt(const float* src, float* dst, int N, int T, int64_t inner, void* stream);
/* Strided channel copy: dst[r][dst_off + c*dst_cstride] (+)= src[r][src_off + c*src_cstride], c < C.
 * Implements torch.cat(dim=1) / channel slicing / "b t c h w -> b (c t) h w" on channels-last tensors. */
int dgmr_copy_channels(const float* src, float* dst, int64_t R, int C, int src_C, int src_off, int src_cstride, int dst_C,
                       int dst_off, int dst_cstride, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ConvGRU gating — dgmr/layers/ConvGRU.py:69-85.
 * ---------------------------------------------------------------------------------------------- */
/* rh = sigmoid(pr) * h */
int dgmr_gru_gate_fwd(const float* pr, const float* h, float* rh, int64_t n, void* stream);
/* dpr = d_rh * h * s*(1-s), dh = d_rh * s  (s = sigmoid(pr)) */
int dgmr_gru_gate_bwd(const float* d_rh, const float* pr, const float* h, float* dpr, float* dh, int64_t n, void* stream);
/* out = s(pu)*h + (1-s(pu))*relu(pc) */
int dgmr_gru_blend_fwd(const float* pu, const float* h, const float* pc, float* out, int64_t n, void* stream);
int dgmr_gru_blend_bwd(const float* dout, const float* pu, const float* h, const float* pc, float* dpu, float* dh, float* dpc,
                       int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Elementwise helpers.
 * ---------------------------------------------------------------------------------------------- */
/* y = x * s[0] * host_scale   (s is a device scalar: an upstream gradient) */
int dgmr_scale_by_dev(const float* x, const float* s, float host_scale, float* y, int64_t n, void* stream);
/* y = alpha*a + beta*b (b may be NULL) */
/* ------------------------------------------------------------------------------------------------
 * (ABI 11) Deterministic mode.  on != 0: every cross-workgroup sum of the step is formed in a fixed order - bias gradients through
 * dgmr_wgrad_args.bias_partial, <P_q, W> through the rows behind `dot`, per-channel statistics (dgmr_bn_stats, dgmr_bn_bwd_reduce,
 * dgmr_colsum, dgmr_bn_partial_reduce, dgmr_grid_cell_loss) through per-workgroup partial rows in the caller's `sums` / `tmp` / `acc`
 * buffer, which then holds dgmr_reduce_doubles(G, R, C) doubles instead of G*2*C.  Two identical runs then give bit-identical
 * parameters and buffers (tests/test_gpu_determinism.py).  Off: those sums meet in float / double atomics. */
int dgmr_set_deterministic(int on);
int dgmr_get_deterministic(void);
/* doubles the caller provides (zeroed) as `sums` / `tmp` / `acc` of a [G][R][C] per-channel reduction: G*2*C, or, in deterministic
 * mode, (1 + nb) * G*2*C with nb = the number of row blocks the library will launch (<= 512, capped so that the buffer stays below 32 MB) */
int64_t dgmr_reduce_doubles(int G, int64_t R, int C);
/* (ABI 11) count += number of non-finite values (NaN, +-Inf) among x[0..n): the opt-in stand-in for the reference's
 * torch.autograd.set_detect_anomaly(True) (dgmr/dgmr.py:130) - parameter gradients are written by kernels here, so autograd's anomaly
 * mode cannot see them.  count: one int32 on the device, accumulated (zero it first). */
int dgmr_nonfinite_count(const float* x, int64_t n, int32_t* count, void* stream);

int dgmr_axpby(const float* a, const float* b, float* y, float alpha, float beta, int64_t n, void* stream);
/* dst[i][:] = src[:] for i < repeat, rows of n floats (n % 4 == 0): einops 'b c h w -> (repeat b) c h w' at b == 1
 * (generators.py:146-148) */
int dgmr_repeat_rows(const float* src, float* dst, int64_t n, int repeat, void* stream);
/* x is [groups][rows][n] (n % 4 == 0): out[g][:] = sum_r w[((g*rows + r) / rows_per_w) * w_stride + col_block] * x[g][r][:] where
 * col_block = column / (n / w_stride) (w_stride 1: one weight per row); w == NULL: plain sums.
 * Sums the per-sample gradients of a ConvGRU whose input is the same latent for every sample and step (generators.py:146-149). */
int dgmr_group_rowsum(const float* x, const float* w, float* out, int groups, int rows, int64_t n, int rows_per_w, int w_stride,
                      void* stream);
/* dst[(k*repeat + r)][:] = src[k][:] for k < nblocks, r < repeat, rows of `block` floats (block % 4 == 0):
 * einops 'k c h w -> (k repeat) c h w' - one latent map per generator draw handed to the `repeat` samples of that draw. */
int dgmr_repeat_interleave(const float* src, float* dst, int64_t nblocks, int64_t block, int repeat, void* stream);
/* dx = (x > 0) ? dy : 0 */
int dgmr_relu_bwd(const float* dy, const float* x, float* dx, int64_t n, void* stream);
int dgmr_fill(float* p, float value, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Latent attention — dgmr/layers/Attention.py:9-20 (attention_einsum).  q, k, v, out are ONE sample,
 * channels-last [H][W][Cq].  The reference hands the NCHW view [Cq][H][W] to an einsum written for
 * "[h w c]": position p = cq*H + y (L = Cq*H of them), feature index = x (length W).  beta: [L][L] saved
 * softmax.  tmp: [L][L] scratch.
 * ---------------------------------------------------------------------------------------------- */
int dgmr_attention_fwd(const float* q, const float* k, const float* v, float* beta, float* out, int Cq, int H, int W,
                       void* stream);
int dgmr_attention_bwd(const float* dout, const float* q, const float* k, const float* v, const float* beta, float* dq,
                       float* dk, float* dv, float* tmp, int Cq, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Discriminator heads — discriminators.py:127-131,217-219: sum_hw(relu(x)), Linear(C -> 1).
 * ---------------------------------------------------------------------------------------------- */
int dgmr_relu_sum_hw_fwd(const float* x, float* y, int N, int HW, int C, void* stream);
int dgmr_relu_sum_hw_bwd(const float* dy, const float* x, float* dx, int N, int HW, int C, void* stream);
/* y[n] = scale[n / scale_group] * <x[n], w> + bias   (scale_group rows per spectral-norm call; <= 0: all N rows) */
int dgmr_linear1_fwd(const float* x, const float* w, const float* bias, const float* scale, float* y, int N, int C,
                     int scale_group, void* stream);
/* dx[n][c] = dy[n]*scale[q]*w[c]; gw_raw[q][c] = sum_{n in group q} dy[n]*x[n][c]; gb = sum_n dy[n]   (q = n / scale_group) */
int dgmr_linear1_bwd(const float* dy, const float* x, const float* w, const float* scale, float* dx, float* gw_raw, float* gb,
                     int N, int C, int scale_group, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Losses — dgmr/losses.py:172-192,307-319; dgmr/dgmr.py:20-33.
 * ---------------------------------------------------------------------------------------------- */
/* loss = mean(relu(1 - real)) + mean(relu(1 + gen)); d_real/d_gen = gradient * gscale */
int dgmr_hinge_disc(const float* s_real, const float* s_gen, float* loss, float* d_real, float* d_gen, int n_real, int n_gen,
                    void* stream);
/* loss = mult * sum_i |mean_k pred_k[i] - y[i]| * w[i];  pred_k = preds + k*pred_stride;  w = weights (explicit, [n]) or, when
 * weights == NULL, the reference's default weight_fn max(y[i]+1, cap) (dgmr/dgmr.py:20-33) evaluated in the kernel.
 * acc: one double, zero on entry, left zero.  dweight[i] (optional) = sign(.)*w/K, the per-prediction gradient / mult. */
/* (ABI 11) doubles behind `acc` (zeroed by the caller): 1, or in deterministic mode 1 + the number of workgroups of the launch */
int64_t dgmr_grid_cell_acc_doubles(int64_t n);
int dgmr_grid_cell_loss(const float* preds, int K, int64_t pred_stride, const float* target, const float* weights, float cap,
                        double* acc, float* loss, float mult, float* dweight, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Adam — torch.optim.Adam as constructed at dgmr/dgmr.py:292-300 (eps 1e-8, no weight decay, no amsgrad).
 * ---------------------------------------------------------------------------------------------- */
/* Hyper-parameters travel as doubles: torch forms 1 - beta, lr / (1 - beta1^step) and sqrt(1 - beta2^step) in double and rounds
 * each to float once; the kernel does the same (lerp for the first moment, as torch's foreach implementation). */
int dgmr_adam(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
              int step, void* stream);
/* (ABI 9) All tensors of one optimiser (dgmr/dgmr.py:292-300: one Adam per network) in ONE launch.  descs: device array, one entry per
 * tensor in ascending block0 order; block0 = first workgroup of the tensor when every tensor gets ceil(n / dgmr_adam_chunk())
 * workgroups; step_size = lr / (1 - beta1^step) and bc2_sqrt = sqrt(1 - beta2^step) rounded from double by the caller, as dgmr_adam
 * forms them (torch keeps one step counter per parameter: they travel per tensor).  Same arithmetic per element as dgmr_adam. */
typedef struct dgmr_adam_desc {
    float* p;
    const float* g;
    float* m;
    float* v;
    int64_t n;
    int32_t block0;
    float step_size;
    float bc2_sqrt;
    int32_t reserved;
} dgmr_adam_desc;
int dgmr_adam_chunk(void);
int dgmr_adam_multi(const dgmr_adam_desc* descs, int n_tensors, int total_blocks, double beta1, double beta2, double eps, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Measurement hooks (bench.py's roofline leg; not part of the reference's surface).  When enabled, every conv /
 * wgrad launch is bracketed by HIP events on its launch stream; collect() returns, per tile variant, the summed
 * kernel time, the summed algorithmic FLOPs (2*M*K*Cout) and the launch count, then clears the records.
 * ---------------------------------------------------------------------------------------------- */
int dgmr_profile_enable(int on);
int dgmr_profile_variants(void);
const char* dgmr_profile_variant_name(int variant);
int dgmr_profile_collect(double* total_ms, double* total_flops, int64_t* launches, int n);
/* The same plus executed_flops[v]: the flops the matrix pipe really issued - 16/36 of the algorithmic figure for upsampling convs run
 * as phase convs / pooled data gradients (w_phase), times the MFMAs per product of the mode the launch was issued in (3 in
 * DGMR_PREC_BF16X3, 6 in DGMR_PREC_BF16X6; ABI 8: this factor used to be the caller's to apply). */
int dgmr_profile_collect2(double* total_ms, double* total_flops, double* executed_flops, int64_t* launches, int n);
/* (ABI 9) The same records grouped by INSTANTIATED kernel instead of by class: tile (output-channel block, 128- / 256-pixel), mode
 * (plain / phase / pooled 3x3, 3-D, ConvGRU epilogue, split-K, 1x1), launch size (fewer than 1024 workgroups = "small": such a launch
 * cannot fill 256 CUs four times over) and arithmetic.  Writes "name\tlaunches\ttotal_ms\talgorithmic_flops\texecuted_flops\n" lines
 * (NUL-terminated, truncated to cap) and returns the bytes needed.  Does not clear the records: call it BEFORE dgmr_profile_collect*. */
int dgmr_profile_collect_detail(char* buf, int cap);
/* Dispatch override for tools/conv_bench.py's tile / split-K sweeps (process-wide; -1 = the library's own choice, which is
 * also the state at load): variant = index of a conv_fwd_dgrad<..> tile as listed by dgmr_profile_variant_name, ksplit = number
 * of K slabs (needs a workspace in the args), window = 0 never / 1 the register-staged LDS-window 3x3 kernel whenever the geometry allows / 2 likewise, with the experimental
 * 256-pixel tiles / 3 the LDS-DMA window kernel where eligible (what -1 picks, but also below the automatic size threshold) /
 * 4 its private-weight-slice variant / 5 its 16-column-block variant for <= 48 output channels / 6 (ABI 9) the LDS-DMA kernel with
 * the block shapes of the wave-specialised kernel (the bit-for-bit reference of 7) / 7 (ABI 9) the wave-specialised persistent
 * kernel (conv_win_ws.h: matrix waves + loader / epilogue waves in one workgroup per CU) wherever it applies, at any size;
 * wgrad_window = 0 never an LDS-window weight-gradient kernel / 1 the one-role kernel of round 2 (wgrad_win.h) wherever the geometry
 * allows / 2 the wave-specialised one (wgrad_ws.h: loader waves + matrix waves, ds_read_b64_tr_b16 fragments) with three matrix
 * waves (32x32x16 MFMAs, one filter row each) / 3 (= automatic) with four (16x16x32 MFMAs, one per SIMD; bf16x6: three) / 4 like 3,
 * and upsampling convs by output-pixel parity on the low-resolution map (wgrad_ws.h PHASE) even under DGMR_WGRAD_PHASES=0 (by default
 * that is the library's own choice where the four-wave kernel applies) / 5 (round 6) like 3, but layers with <= 48 output channels on
 * the 64-column tile of rounds 3 - 5 instead of the pixel-split 48-column one (wgrad_ws.h PSPLIT; the A/B reference) and no phase launches. */
int dgmr_conv_tune(int variant, int ksplit, int window, int wgrad_window);
/* Kernel-phase timing switches for tools/conv_bench.py (process-wide, 0 at load and in every product launch): bit 0 = the LDS-window
 * conv kernels return before their epilogue, bit 1 = they stage only their first input halo.  Outputs are then garbage by design;
 * only the launch duration is meaningful (how much of a launch is operand staging / matrix work / epilogue).  Bit 3 (8) = the
 * window kernels use their lane-per-channel epilogue instead of the 16-byte one (A/B: results are bit-identical); 64 / 128 = every
 * wave sleeps ~3.4 / ~6.8 us after issuing its last store (how long does a finished wave wait for its stores anyway?); 16 = the
 * wave-specialised weight-gradient kernel without its matrix work (the loaders' time alone); 256 (round 5) = phase launches of the
 * upsampling convs with one workgroup per ROW parity that computes both column parities from one staged halo, instead of one
 * workgroup per output-pixel parity (conv_win_glds.h PAIR; same results bit for bit - tests/test_gpu_kernels.py; measured no faster,
 * so it is not the default: DGMR_PHASE_PAIR=1 switches it on for a whole process). */
int dgmr_debug_flags(int flags);

#ifdef __cplusplus
}
#endif
#endif /* DGMR_HIP_H */
