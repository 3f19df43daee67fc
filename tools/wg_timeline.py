#!/usr/bin/env python3
"""Where a launch of the band-attention kernel spends its time outside the tile loops: per-workgroup s_memtime stamps
(entry, loop start, loop end, exit) and hardware ids of the traced two-phase kernel (variant 64, diagnostics library: build.py --ablations + SVG_ATTN_LIB), folded into
per-CU occupancy, prologue / epilogue durations and the gap between consecutive workgroups on the same CU.
python tools/wg_timeline.py [band] [heads: spatial|temporal|alt] [H]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
import torch  # noqa: E402
from svg import _native as nat  # noqa: E402

band = int(sys.argv[1]) if len(sys.argv) > 1 else 15616
heads = sys.argv[2] if len(sys.argv) > 2 else "spatial"
H = int(sys.argv[3]) if len(sys.argv) > 3 else 24
D, F_, P_, ctx, L = 128, 33, 3600, 256, 64
V = F_ * P_
S = V + ctx
dev = torch.device("cuda", 0)
q, k, v = (torch.randn(1, H, S, D, device=dev, dtype=torch.bfloat16) for _ in range(3))
mask = nat.BandMask(real_len=V + L, band=band, colfull_lo=V, colfull_hi=V + L, rowfull_lo=V, rowfull_hi=V + L)
pat = {"alt": lambda h: h % 2, "spatial": lambda h: 0, "temporal": lambda h: 1}[heads]
best = torch.tensor([[pat(h) for h in range(H)]], device=dev, dtype=torch.int64)
kw = dict(head_perm_flag=best, vid0=0, num_frame=F_, frame_size=P_)
for _ in range(2):
    nat.band_attention(q, k, v, mask, variant=64, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
nat.band_attention(q, k, v, mask, variant=64, **kw)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
nwg = H * (-(-V // 256) + -(-L // 256) + -(-(ctx - L) // 256))   # q-tiles never straddle the video | prompt | pad boundaries
tr = nat.debug_wg_trace(min(nwg, 16384)).astype(np.int64)
t0, t1, t2, t3, hw, xcc = (tr[:, i] for i in range(6))
ok = t3 > 0
cu = ((xcc & 0xF) << 12) | (hw & 0xFF00) >> 4     # (xcc, se, sh, cu): HW_ID cu_id [11:8], sh_id [12], se_id [15:13]
keys = np.unique(cu[ok])
# s_memtime counters of different CUs are not aligned: every span is taken inside one CU; the CU with the longest span is
# taken to be busy for the whole launch, which calibrates the tick
spans = np.array([t3[ok & (cu == c)].max() - t0[ok & (cu == c)].min() for c in keys], dtype=np.float64)
span = float(spans.max())
tick_ns = ms * 1e6 / span
print(f"band {band} heads {heads}: {ms:.3f} ms, {ok.sum()} / {nwg} workgroups traced, median per-CU span {span:.0f} ticks -> {tick_ns:.3f} ns/tick")
pro, loop, epi = (t1 - t0)[ok], (t2 - t1)[ok], (t3 - t2)[ok]
for name, a in (("prologue", pro), ("tile loop", loop), ("epilogue", epi)):
    print(f"  {name:9s}: mean {a.mean() * tick_ns / 1e3:8.2f} us  median {np.median(a) * tick_ns / 1e3:8.2f}  p95 {np.percentile(a, 95) * tick_ns / 1e3:8.2f}"
          f"  max {a.max() * tick_ns / 1e3:8.2f}   sum/CU {a.sum() * tick_ns / 1e6 / 256:7.3f} ms")
gaps = []
for c in keys:
    m = ok & (cu == c)
    o = np.argsort(t0[m])
    a0, a3 = t0[m][o], t3[m][o]
    if len(a0) > 1:
        gaps.append(a0[1:] - a3[:-1])
gaps = np.concatenate(gaps)
print(f"  CUs seen {len(keys)}; workgroups per CU {ok.sum() / len(keys):.1f}")
print(f"  gap between consecutive workgroups on a CU: mean {gaps.mean() * tick_ns / 1e3:.2f} us  median {np.median(gaps) * tick_ns / 1e3:.2f}"
      f"  p95 {np.percentile(gaps, 95) * tick_ns / 1e3:.2f}   sum/CU {gaps.sum() * tick_ns / 1e6 / len(keys):.3f} ms")
idle = ms - spans * tick_ns / 1e6      # per CU: launch time not covered by [first entry, last exit]
print(f"  per-CU time outside [first entry, last exit]: mean {idle.mean():.3f} ms  median {np.median(idle):.3f}  max {idle.max():.3f}")
for x in range(8):
    sel = (keys >> 12) == x
    if sel.any():
        print(f"    XCD {x}: {int((ok & ((cu >> 12) == x)).sum()):5d} workgroups, CU span mean {spans[sel].mean() * tick_ns / 1e6:7.3f} ms  "
              f"min {spans[sel].min() * tick_ns / 1e6:7.3f}  max {spans[sel].max() * tick_ns / 1e6:7.3f}")
if len(sys.argv) > 4:   # detail of one XCD
    xsel = int(sys.argv[4])
    print(f"  XCD {xsel} (clocks of different CUs are not aligned: 'start' is relative to the CU's own first entry)")
    for c in keys[(keys >> 12) == xsel][:40]:
        m = ok & (cu == c)
        o = np.argsort(t0[m])
        a0, a3, ids = t0[m][o], t3[m][o], np.nonzero(m)[0][o]
        print(f"    cu {c & 0xFFF:03x}: n {len(a0):3d} span {(a3[-1] - a0[0]) * tick_ns / 1e3:9.1f} us"
              f"  first ids {ids[:4].tolist()}  longest {((a3 - a0).max()) * tick_ns / 1e3:8.1f} us")
