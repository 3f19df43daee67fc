#!/usr/bin/env python3
"""SVG1 band attention across the reference's model geometries (production masks of svg/models/*/utils.py), alternating
spatial / temporal heads: ms, algorithmic PFLOP/s (4 D H #unmasked pairs / time), speed-up over the same kernel in dense mode."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
import torch  # noqa: E402
from svg import _native as nat  # noqa: E402
from svg.models.hyvideo.utils import sparsity_to_width  # noqa: E402
from svg.models.hyvideo import utils as hy  # noqa: E402
from svg.models.wan import utils as wan  # noqa: E402
from svg.models.cog import utils as cog  # noqa: E402


def pairs(m, S):
    """#allowed (q, k) pairs of a BandMask (the kernels' interval form, BandPolicy::row_intervals)"""
    q = np.arange(S, dtype=np.int64)
    real = m.real_len
    rq = q < real
    rowf = (q >= m.rowfull_lo) & (q < m.rowfull_hi)
    lo = np.where(rq, np.where(rowf, 0, np.maximum(q - m.band + 1, 0)), real)
    hi = np.where(rq, np.where(rowf, real, np.minimum(q + m.band, real)), S)
    alen = np.maximum(hi - lo, 0)
    ch = min(m.colfull_hi, real)
    b0, b1 = m.colfull_lo, max(ch, m.colfull_lo)
    use_b = rq & ~rowf
    inter = np.maximum(np.minimum(hi, b1) - np.maximum(lo, b0), 0)
    blen = np.where(use_b, (b1 - b0) - inter, 0)
    return int((alen + blen).sum())


CASES = [
    # name, cfg*H, D, F, P, ctx, text_first, mask builder
    ("HunyuanVideo 720p 129f (s=0.25)", 24, 128, 33, 3600, 256, False,
     lambda F, P, ctx: hy.generate_temporal_head_mask_mod(ctx, 64, F, P, mul=sparsity_to_width(0.25, ctx, F, P))),
    ("Wan 2.1 720p 81f (s=0.30)", 40, 128, 21, 3600, 0, False,
     lambda F, P, ctx: wan.generate_temporal_head_mask_mod(ctx, ctx, F, P, mul=sparsity_to_width(0.30, ctx, F, P))),
    ("CogVideoX-v1 480p 49f (s=0.25)", 96, 64, 13, 1350, 226, True,
     lambda F, P, ctx: cog.generate_temporal_head_mask_mod(ctx, F, P, mul=sparsity_to_width(0.25, ctx, F, P))),
    ("CogVideoX-v1.5 768p 81f (s=0.25)", 96, 64, 11, 4080, 226, True,
     lambda F, P, ctx: cog.generate_temporal_head_mask_mod(ctx, F, P, mul=sparsity_to_width(0.25, ctx, F, P))),
]


def main():
    fp8 = len(sys.argv) > 1 and sys.argv[1] == "fp8"         # e4m3 kernels (head_dim 128 geometries only), pre-pass included
    pre = len(sys.argv) > 1 and sys.argv[1] == "pre"         # svg_band_attention_prescaled on a q that carries the softmax scale
    variant = int(sys.argv[1]) if len(sys.argv) > 1 and not fp8 and not pre else 0   # schedule of svg_band_attention (include/svg_attn.h)
    print("fp8 (e4m3) kernels, quantise pre-pass included" if fp8 else
          ("pre-scaled q (svg_band_attention_prescaled: two-phase body at every head_dim)" if pre else f"schedule variant {variant}"))
    dev = torch.device("cuda", 0)
    print("| model geometry | S | density | sparse ms | PFLOP/s (algorithmic) | dense ms | dense PFLOP/s | speed-up |")
    print("|---|---|---|---|---|---|---|---|")
    for name, BH, D, F_, P_, ctx, text_first, mk in CASES:
        if fp8 and D != 128:
            continue
        S = F_ * P_ + ctx
        mask = mk(F_, P_, ctx)
        q, k, v = (torch.randn(1, BH, S, D, device=dev, dtype=torch.bfloat16) for _ in range(3))
        o = torch.empty_like(q)
        best = torch.tensor([[h % 2 for h in range(BH)]], device=dev, dtype=torch.int64)
        vid0 = ctx if text_first else 0
        dmask = nat.BandMask(real_len=mask.real_len, band=S + 1, colfull_lo=0, colfull_hi=0, rowfull_lo=0, rowfull_hi=0)

        def t(fn):
            fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            return min(ts)

        if fp8:
            ms = t(lambda: nat.band_attention_fp8(q, k, v, mask, head_perm_flag=best, vid0=vid0, num_frame=F_, frame_size=P_, out=o))
            dms = t(lambda: nat.band_attention_fp8(q, k, v, dmask, out=o))
        elif pre:
            qs = (q.float() * nat.softmax_q_scale(D)).to(q.dtype)
            ms = t(lambda: nat.band_attention(qs, k, v, mask, head_perm_flag=best, vid0=vid0, num_frame=F_, frame_size=P_, out=o, q_prescaled=True))
            dms = t(lambda: nat.band_attention(qs, k, v, dmask, out=o, q_prescaled=True))
            del qs
        else:
            ms = t(lambda: nat.band_attention(q, k, v, mask, head_perm_flag=best, vid0=vid0, num_frame=F_, frame_size=P_, out=o, variant=variant))
            dms = t(lambda: nat.band_attention(q, k, v, dmask, out=o, variant=variant))
        np_, dp = pairs(mask, S), pairs(dmask, S)
        fl, dfl = 4.0 * D * BH * np_, 4.0 * D * BH * dp
        print(f"| {name} | {S} | {np_ / S / S:.4f} | {ms:.3f} | {fl / ms / 1e12:.3f} | {dms:.3f} | {dfl / dms / 1e12:.3f} | {dms / ms:.2f}x |", flush=True)
        del q, k, v, o


if __name__ == "__main__":
    main()
