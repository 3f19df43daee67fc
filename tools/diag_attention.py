#!/usr/bin/env python3
"""Structured diagnostics for the attention kernel (run on the GPU box when a parity test fails): each case isolates
one stage (softmax normalisation, V path, key<->d mapping, masks) and prints where the error sits."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
import torch
from oracle import svg_oracle as O
from svg import _native as nat


def report(name, o, ref):
    o = o.float().cpu(); ref = ref.float()
    err = (o - ref).abs()
    rl2 = ((o - ref).norm() / ref.norm().clamp(min=1e-9)).item()
    print(f"[{name}] rel_l2={rl2:.3e} max_abs={err.max().item():.3e} nan={int(torch.isnan(o).sum())}")
    if rl2 > 5e-3:
        S, D = o.shape[-2], o.shape[-1]
        e2 = err.reshape(-1, S, D)[0]
        qb = e2.reshape(-1, min(32, S), D).amax(dim=(1, 2)) if S % 32 == 0 else None
        db = e2.reshape(S, D // 32, 32).amax(dim=(0, 2))
        print("   max err per 32-row q block:", [round(x, 3) for x in (qb.tolist() if qb is not None else [])][:16])
        print("   max err per 32-col d block:", [round(x, 3) for x in db.tolist()])
        print("   o[0,:8]  :", [round(x, 3) for x in o.reshape(-1, S, D)[0, 0, :8].tolist()])
        print("   ref[0,:8]:", [round(x, 3) for x in ref.reshape(-1, S, D)[0, 0, :8].tolist()])


def main():
    nat.load()
    torch.manual_seed(0)
    for variant in (0, 1):
        for D in (128, 64):
            for S in (64, 128, 200, 1000):
                dense = nat.BandMask(**O.dense_band_params(S))
                q = torch.randn(1, 1, S, D); k = torch.randn(1, 1, S, D); v = torch.randn(1, 1, S, D)
                cases = {
                    "v=ones": (q, k, torch.ones_like(v)),
                    "q=0 (uniform softmax)": (torch.zeros_like(q), k, v),
                    "v=onehot(d==key%D)": (q, k, torch.nn.functional.one_hot(torch.arange(S) % D, D).float()[None, None]),
                    "random": (q, k, v),
                }
                for name, (a, b, c) in cases.items():
                    a, b, c = (x.to(torch.bfloat16) for x in (a, b, c))
                    o = nat.band_attention(a.cuda(), b.cuda(), c.cuda(), dense, variant=variant)
                    report(f"var{variant} D{D} S{S} {name}", o, O.masked_attention(a, b, c, None))
    # masks
    F_, P_, ctx, L, mul = 5, 150, 40, 11, 2.3
    V = F_ * P_; S = V + ctx
    q, k, v = (torch.randn(1, 2, S, 128).to(torch.bfloat16) for _ in range(3))
    for name, prm, m in (("hy", O.hy_band_params(S, ctx, L, F_, P_, mul), O.hy_mask(S, ctx, L, F_, P_, mul)),
                         ("cog", O.cog_band_params(S, ctx, F_, P_, mul), O.cog_mask(S, ctx, F_, P_, mul))):
        o = nat.band_attention(q.cuda(), k.cuda(), v.cuda(), nat.BandMask(**prm))
        report(f"mask {name}", o, O.masked_attention(q, k, v, m))


if __name__ == "__main__":
    main()
