#!/usr/bin/env python3
"""Print the per-phase cycle trace of the two-phase ping-pong attention schedule (svg_band_attention variant 2) on the
HunyuanVideo 720p layer-call shape.  Needs the diagnostics library: `python sparse-videogen_amd/build.py --ablations`, then
SVG_ATTN_LIB=sparse-videogen_amd/lib/libsvgattn_abl.so python tools/pp_trace.py [band] [abl,abl,...]
(variant 64 | abl << 8 of svg_band_attention; svg_debug_pp_trace in include/svg_attn.h)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
import torch  # noqa: E402
from svg import _native as nat  # noqa: E402

H, D, F_, P_, ctx = 4, 128, 33, 3600, 256
V = F_ * P_
S = V + ctx
dev = torch.device("cuda", 0)
q, k, v = (torch.randn(1, H, S, D, device=dev, dtype=torch.bfloat16) for _ in range(3))
band = int(sys.argv[1]) if len(sys.argv) > 1 else 15616
mask = nat.BandMask(real_len=V + 64, band=band, colfull_lo=V, colfull_hi=V + 64, rowfull_lo=V, rowfull_hi=V + 64)
abls = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0]
mode = sys.argv[3] if len(sys.argv) > 3 else "pp2"   # pp2: two-phase ping-pong (variant 64 | abl << 8); w4: one wave per SIMD (variant 32)
for variant in [(32 if mode == "w4" else 64) | (a << 8) for a in abls]:
    # (trace code 3 = the pre-scaled-q body: hand it a q that carries the softmax scale, as svg_band_attention_prescaled gets it)
    qq = (q.float() * nat.softmax_q_scale(D)).to(q.dtype) if (mode == "pp2" and (variant >> 8) == 3) else q
    o = nat.band_attention(qq, k, v, mask, variant=variant)
    tr = nat.debug_pp_trace()
    nT = max(tr["tiles"], 1)
    print(f"--- {mode} schedule, ablation {variant >> 8}")
    print(f"tiles {tr['tiles']} loop ticks {tr['loop_ticks']} = {tr['loop_ticks'] / nT:.0f} per tile")
    names = ["A<bar", "bar", "A>bar", "A->B", "B", "B->A"] if mode == "w4" else ["M", "bar", "N", "bar"]
    for w, acc in enumerate(tr["waves"]):
        if w in ((0, 1, 2, 3) if mode == "w4" else (0, 4)):
            print(f"wave {w}: " + "  ".join(f"{n} {a / nT:7.1f}" for n, a in zip(names, acc)) + f"   sum {sum(acc[:6]) / nT:7.1f}")
