"""two fp16 planes per operand ("fp16x3": three v_mfma_f32_32x32x16_f16 per product) against the bf16 split modes
the library runs.  Host-side NUMERICAL emulation (no GPU): every plane product is exact in fp32 (8 x 8 / 11 x 11 mantissa bits), so the
modes are emulated with float32 matmuls of the planes; errors against float64.  What it answers: (1) product accuracy per mode on a
conv-shaped contraction, (2) what fp16's 5-bit exponent does to it at the magnitudes the step really has (the reference's Q1 quirk puts
g_loss at ~1e11, so dY reaches 1e5 ... 1e8), (3) whether a power-of-two per-tensor scale repairs it.     python tools/fp16x3_probe.py"""
import torch

torch.manual_seed(0)


def planes(x, dtype, n):
    out, r = [], x.clone()
    for _ in range(n):
        p = r.to(dtype).to(torch.float32)
        out.append(p)
        r = r - p
    return out


def contract(a, b, dtype, na, nb, terms):
    pa, pb = planes(a, dtype, na), planes(b, dtype, nb)
    y = torch.zeros(a.shape[0], b.shape[1], dtype=torch.float32)
    for i, j in terms:
        y += pa[i] @ pb[j]  # exact products, fp32 accumulation (as the MFMA does)
    return y


MODES = {
    "bf16   (1 MFMA)": (torch.bfloat16, 1, 1, [(0, 0)]),
    "bf16x3 (3 MFMA)": (torch.bfloat16, 2, 2, [(0, 0), (0, 1), (1, 0)]),
    "bf16x6 (6 MFMA)": (torch.bfloat16, 3, 3, [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]),
    "fp16   (1 MFMA)": (torch.float16, 1, 1, [(0, 0)]),
    "fp16x3 (3 MFMA)": (torch.float16, 2, 2, [(0, 0), (0, 1), (1, 0)]),
}


def run(tag, a, b, scale_a=1.0, scale_b=1.0):
    ref = a.double() @ b.double()
    f32 = (a @ b).double()
    print(f"{tag}: |a| max {a.abs().max():.2e}, |b| max {b.abs().max():.2e}; plain fp32 matmul error {((f32 - ref).abs().max() / ref.abs().max()).item():.2e}")
    for name, (dt, na, nb, terms) in MODES.items():
        y = contract(a * scale_a, b * scale_b, dt, na, nb, terms).double() / (scale_a * scale_b)
        e = ((y - ref).abs().max() / ref.abs().max()).item()
        print(f"    {name:18s} max error / max |y| = {e:.2e}" + ("   <- overflow / underflow" if not (e < 1e-1) else ""))


M, K, N = 2048, 864, 96  # a 3x3 conv of 96 channels: K = 9 x 96
act = torch.relu(torch.randn(M, K))            # activations behind BatchNorm + ReLU
w = torch.randn(K, N) * 0.05                    # weights / sigma
run("forward  (activations x weights)", act, w)
dy_big = torch.randn(M, N) * 3e6                # dY at the magnitudes of the reference's Q1 quirk (g_loss ~ 1e11)
run("weight gradient, dY ~ 3e6 (no scaling)", act.t().contiguous(), dy_big)
s = 2.0 ** -torch.ceil(torch.log2(dy_big.abs().max())).item() * 2.0 ** 14  # power of two that puts max |dY| at ~2^14
run(f"weight gradient, dY ~ 3e6, dY scaled by 2^{int(torch.log2(torch.tensor(s)).item())}", act.t().contiguous(), dy_big, 1.0, s)
dy_small = torch.randn(M, N) * 1e-7
run("weight gradient, dY ~ 1e-7 (no scaling)", act.t().contiguous(), dy_small)
# a heavy-tailed operand: most of a gradient map sits 1e-5 below its few largest entries (what a max-abs scale cannot help)
dy_tail = torch.randn(M, N) * 1e-3
dy_tail[::97] *= 1e6
s = 2.0 ** -torch.ceil(torch.log2(dy_tail.abs().max())).item() * 2.0 ** 14
run("weight gradient, heavy-tailed dY (max / typical = 1e6), max-abs scaled", act.t().contiguous(), dy_tail, 1.0, s)
