#!/usr/bin/env python3
"""Print the per-cluster cycle trace of the ping-pong attention schedule on the HunyuanVideo 720p layer-call shape
(variant bits 5 + 6 of svg_band_attention; see svg_debug_pp_trace in include/svg_attn.h)."""
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
base = int(sys.argv[3]) if len(sys.argv) > 3 else 32   # 32: four-cluster schedule, 128: two-phase schedule
for variant in [base | 64 | (a << 8) for a in abls]:
    o = nat.band_attention(q, k, v, mask, variant=variant)
    tr = nat.debug_pp_trace()
    nT = max(tr["tiles"], 1)
    print(f"--- variant {base} ablation {variant >> 8}")
    print(f"tiles {tr['tiles']} loop ticks {tr['loop_ticks']} = {tr['loop_ticks'] / nT:.0f} per tile")
    names = ["LK", "bar", "QK", "bar", "SV", "bar", "PV", "bar"] if base == 32 else ["M", "bar", "N", "bar"]
    for w, acc in enumerate(tr["waves"]):
        if w in (0, 4):
            print(f"wave {w}: " + "  ".join(f"{n} {a / nT:7.1f}" for n, a in zip(names, acc)) + f"   sum {sum(acc) / nT:7.1f}")
