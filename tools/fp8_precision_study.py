"""CPU study (no GPU, no library): where the error of 8-bit attention comes from, and whether any 8-bit form of the QK^T / PV operands is
usable (BASELINE.json configs[4]).  Attention of a few heads is computed in fp32 with the MFMA operands rounded
operand group by operand group; the figure is the relative L2 error of the output against fp32 attention on the same (bf16-valued) inputs.

    python tools/fp8_precision_study.py [S]

Operand roundings:
  bf16            the 16-bit kernels
  e4m3 / head     OCP e4m3, one scale 448 / amax per head and tensor (what csrc/attention_f8.hip does)
  e4m3 / token    one scale per token row (q, k: per row; v: per row — folded into P's column on the device)
  int8 / token    symmetric int8, one scale 127 / amax per token row (gfx950 I8 MFMA: 2x the bf16 rate)
  K - mean        K with its per-head mean over tokens subtracted first (softmax-invariant: adds a per-query constant to every score)
  P e4m3          probabilities p <= 1 (against the row's reference) x 2^8, e4m3
  P int8          probabilities x 127, rounded to uint8-like (0 .. 127)
Data: `randn`; `clustered` (the SVG2 bench inputs: 64-mode Gaussian mixture for q and k, spread 0.35); `heavy` = clustered with 4 outlier
channels of q and k scaled x10 (real DiT q / k carry a few large channels); `peaked` = clustered x 2 (sharper softmax)."""
import sys

import torch

torch.manual_seed(0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D, H = 128, 4


def clustered(H, N, D, modes=64, spread=0.35, gain=1.0):
    centers = torch.randn(H, modes, D) * 1.5
    lab = torch.randint(0, modes, (H, N))
    x = torch.gather(centers, 1, lab[..., None].expand(-1, -1, D)) + spread * torch.randn(H, N, D)
    return (gain * x).to(torch.bfloat16).float()


def heavy(H, N, D):
    x = clustered(H, N, D)
    x[..., :4] *= 10.0
    return x.to(torch.bfloat16).float()


def bf16(x):
    return x.to(torch.bfloat16).float()


def e4m3_head(x):
    s = 448.0 / x.abs().amax(dim=(-1, -2), keepdim=True).clamp_min(1e-20)
    return (x * s).to(torch.float8_e4m3fn).float() / s


def e4m3_token(x):
    s = 448.0 / x.abs().amax(dim=-1, keepdim=True).clamp_min(1e-20)
    return (x * s).to(torch.float8_e4m3fn).float() / s


def int8_token(x):
    s = 127.0 / x.abs().amax(dim=-1, keepdim=True).clamp_min(1e-20)
    return torch.round(x * s).clamp(-127, 127) / s


def k_smooth(r):
    return lambda k: r(k - k.mean(dim=-2, keepdim=True))


def p_e4m3(p):
    return (p * 256.0).to(torch.float8_e4m3fn).float() / 256.0


def p_int8(p):
    return torch.round(p * 127.0) / 127.0


def attention(q, k, v, rq, rk, rp, rv):
    s = (rq(q) @ rk(k).transpose(-1, -2)) / D ** 0.5
    m = s.amax(-1, keepdim=True)
    p = torch.exp(s - m)
    l = p.sum(-1, keepdim=True)          # row sums in fp32 from the unrounded probabilities, like the kernels
    return (rp(p) @ rv(v)) / l


forms = [
    ("bf16 everywhere (the 16-bit kernels)", bf16, bf16, bf16, bf16),
    ("e4m3 / head everywhere (svg_*_attention_fp8 today)", e4m3_head, e4m3_head, p_e4m3, e4m3_head),
    ("QK^T e4m3 / head, PV bf16", e4m3_head, e4m3_head, bf16, bf16),
    ("QK^T e4m3 / token, PV bf16", e4m3_token, e4m3_token, bf16, bf16),
    ("QK^T e4m3 / token + K - mean, PV bf16", e4m3_token, k_smooth(e4m3_token), bf16, bf16),
    ("QK^T int8 / token, PV bf16", int8_token, int8_token, bf16, bf16),
    ("QK^T int8 / token + K - mean, PV bf16", int8_token, k_smooth(int8_token), bf16, bf16),
    ("QK^T bf16, PV e4m3 (P x 2^8, V / head)", bf16, bf16, p_e4m3, e4m3_head),
    ("QK^T bf16, PV e4m3 (P x 2^8, V / token)", bf16, bf16, p_e4m3, e4m3_token),
    ("QK^T bf16, P int8, V int8 / token", bf16, bf16, p_int8, int8_token),
    ("QK^T int8 / token + K - mean, PV e4m3 (V / token)", int8_token, k_smooth(int8_token), p_e4m3, e4m3_token),
    ("QK^T e4m3 / token + K - mean, PV e4m3 (V / token)", e4m3_token, k_smooth(e4m3_token), p_e4m3, e4m3_token),
    ("only P e4m3", bf16, bf16, p_e4m3, bf16),
    ("only V e4m3 / head", bf16, bf16, bf16, e4m3_head),
]
datasets = {
    "randn": lambda: tuple(torch.randn(H, S, D).to(torch.bfloat16).float() for _ in range(3)),
    "clustered": lambda: (clustered(H, S, D), clustered(H, S, D), torch.randn(H, S, D).to(torch.bfloat16).float()),
    "heavy": lambda: (heavy(H, S, D), heavy(H, S, D), torch.randn(H, S, D).to(torch.bfloat16).float()),
    "peaked": lambda: (clustered(H, S, D, gain=2.0), clustered(H, S, D, gain=2.0), torch.randn(H, S, D).to(torch.bfloat16).float()),
}
print(f"S = {S}, D = {D}, {H} heads; relative L2 of the output against fp32 attention on the same (bf16-valued) inputs")
print(f"| {'operand rounding':58s} | " + " | ".join(f"{n:>9s}" for n in datasets) + " |")
print("|" + "-" * 60 + "|" + "|".join("-" * 11 for _ in datasets) + "|")
data = {n: f() for n, f in datasets.items()}
ident = lambda x: x  # noqa: E731
ref = {n: attention(*d, ident, ident, ident, ident) for n, d in data.items()}
for name, rq, rk, rp, rv in forms:
    cells = []
    for n, d in data.items():
        o = attention(*d, rq, rk, rp, rv)
        cells.append(f"{((o - ref[n]).norm() / ref[n].norm()).item():9.2e}")
    print(f"| {name:58s} | " + " | ".join(cells) + " |")
