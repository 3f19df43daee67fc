"""Error pattern of the one-wave-per-SIMD attention body on a small dense problem: per query row (mod 64) and per column (d).
SVG_ATTN_LIB=.../libsvgattn_abl.so adds variant 32 (the fast launch alone, without the exact launch behind it)."""
import sys, torch
sys.path.insert(0, "sparse-videogen_amd"); sys.path.insert(0, ".")
from svg import _native as nat
from oracle import svg_oracle as O
torch.manual_seed(0)
dev = torch.device("cuda", 0)
variants = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [3]
for (S, D, dt) in [(256, 128, torch.bfloat16), (790, 128, torch.bfloat16), (790, 64, torch.bfloat16)]:
    q, k, v = (torch.randn(1, 1, S, D).to(dt) for _ in range(3))
    m = nat.BandMask(**O.dense_band_params(S))
    ref = O.masked_attention(q, k, v, None).float()[0, 0]
    for var in variants:
        o = nat.band_attention(q.to(dev), k.to(dev), v.to(dev), m, variant=var).float().cpu()[0, 0]
        err = (o - ref).abs()
        err[torch.isnan(err)] = 1e9
        rows = (err.amax(1) > 0.02).nonzero().flatten().tolist()
        cols = (err.amax(0) > 0.02).nonzero().flatten().tolist()
        print(f"S={S} D={D} variant {var}: max err {float(err.max()):.3g} nan {int(torch.isnan(o).sum())} bad rows {len(rows)} bad cols {len(cols)}")
        if rows:
            print("   rows mod 64:", sorted(set(r % 64 for r in rows)))
            print("   rows:", rows[:40])
            print("   cols:", cols[:64])
            r0 = rows[0]
            print("   o[r0,:8]  ", [round(float(x), 4) for x in o[r0, :8]])
            print("   ref[r0,:8]", [round(float(x), 4) for x in ref[r0, :8]])
            print("   ratio     ", [round(float(x), 4) for x in (o[r0, :8] / ref[r0, :8])])
