"""CPU study (no GPU, no library): where the error of e4m3 attention comes from, and what mixed forms would buy.
Attention of one head is computed in fp32 with the MFMA operands rounded to e4m3 (per-head scale 448 / amax, like the fp8 kernels)
or to bf16, operand group by operand group; relative L2 error against the fp32 result on exact (bf16-valued) inputs.

    python tools/fp8_precision_study.py [S]

Data: `randn` (the test inputs) and `clustered` (the SVG2 bench inputs: 64-mode Gaussian mixture for q and k, bench_svg2.clustered)."""
import sys

import torch

torch.manual_seed(0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D, H = 128, 4


def clustered(H, N, D, modes=64, spread=0.35):
    centers = torch.randn(H, modes, D) * 1.5
    lab = torch.randint(0, modes, (H, N))
    x = torch.gather(centers, 1, lab[..., None].expand(-1, -1, D)) + spread * torch.randn(H, N, D)
    return x.to(torch.bfloat16).float()


def e4m3(x, per_head=True):
    """round to e4m3 after scaling the head to 448 (what f8_quantize_kernel does); returns the dequantised values"""
    amax = x.abs().amax(dim=(-1, -2), keepdim=True).clamp_min(1e-20)
    s = 448.0 / amax
    return (x * s).to(torch.float8_e4m3fn).float() / s


def bf16(x):
    return x.to(torch.bfloat16).float()


def attention(q, k, v, rq, rk, rp, rv):
    s = (rq(q) @ rk(k).transpose(-1, -2)) / D ** 0.5
    m = s.amax(-1, keepdim=True)
    p = torch.exp(s - m)
    l = p.sum(-1, keepdim=True)          # row sums in fp32 from the unrounded probabilities, like the kernels
    # e4m3 probabilities: the kernels scale p by 2^8 (it is <= 1 against its own reference) — a power of two, no effect on rounding
    pr = rp(p * 256.0) / 256.0 if rp is e4m3_p else rp(p)
    return (pr @ rv(v)) / l


def e4m3_p(x):
    return x.to(torch.float8_e4m3fn).float()


ident = lambda x: x  # noqa: E731
forms = [
    ("all operands bf16 (the 16-bit kernels)", bf16, bf16, bf16, bf16),
    ("all operands e4m3 (the fp8 kernels)", e4m3, e4m3, e4m3_p, e4m3),
    ("QK^T e4m3, PV bf16", e4m3, e4m3, bf16, bf16),
    ("QK^T bf16, PV e4m3 (P and V)", bf16, bf16, e4m3_p, e4m3),
    ("only Q e4m3", e4m3, bf16, bf16, bf16),
    ("only K e4m3", bf16, e4m3, bf16, bf16),
    ("only P e4m3", bf16, bf16, e4m3_p, bf16),
    ("only V e4m3", bf16, bf16, bf16, e4m3),
]
print(f"S = {S}, D = {D}, {H} heads; relative L2 of the output against fp32 attention on the same (bf16-valued) inputs")
for name, gen in (("randn", lambda: torch.randn(H, S, D).to(torch.bfloat16).float()), ("clustered (SVG2 bench data)", lambda: clustered(H, S, D))):
    q, k = gen(), gen()
    v = torch.randn(H, S, D).to(torch.bfloat16).float()
    ref = attention(q, k, v, ident, ident, ident, ident)
    print(f"--- q, k: {name}")
    for label, rq, rk, rp, rv in forms:
        o = attention(q, k, v, rq, rk, rp, rv)
        err = ((o - ref).norm() / ref.norm()).item()
        print(f"  {label:42s} {100 * err:7.3f} %")
