#!/usr/bin/env python3
"""Launch timeline of the variable-block attention kernel on the SVG2 bench workload (Wan 720p): per-workgroup prologue (block-row
lookup, run-list compaction, Q load, pipeline fill), tile loop and epilogue, from the s_memtime stamps of variant 5."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from svg import _native as nat  # noqa: E402
from svg import kmeans_utils as KU  # noqa: E402
from svg.models import _core  # noqa: E402
import bench_svg2 as B  # noqa: E402

dev = torch.device("cuda", 0)
H, D, F_, P_, ctx, L, QC, KC = B.WORKLOADS["wan720p"]
S = F_ * P_
gen = torch.Generator(device=dev).manual_seed(0)
q = B.clustered(H, S, D, 64, dev, gen)[None]
k = B.clustered(H, S, D, 64, dev, gen)[None]
v = torch.randn(1, H, S, D, device=dev, dtype=torch.bfloat16)
store = _core.CentroidStore()
(ql, qc, qs, _, qidx), (kl, kc, ks, _, kidx) = _core.kmeans_clustering(store, 0, q, k, QC, KC, 50, 2)
dmap = KU.identify_dynamic_map(qc.view(1, H, QC, D), kc.view(1, H, KC, D), qs.view(1, H, QC), ks.view(1, H, KC), 0.9, 0.1)
args = (q.view(H, S, D), k.view(H, S, D), v.view(H, S, D), dmap.view(H, QC, KC).contiguous(), qs.view(H, QC).contiguous(),
        ks.view(H, KC).contiguous())
kw = dict(q_row_idx=qidx.contiguous(), kv_row_idx=kidx.contiguous())
for _ in range(2):
    nat.varblock_attention(*args, variant=5, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
nat.varblock_attention(*args, variant=5, **kw)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
nwg = int(torch.ceil(qs.float() / 256).sum().item())
tr = nat.debug_wg_trace(min(nwg, 16384)).astype(np.int64)
t0, t1, t2, t3, hw, xcc = (tr[:, i] for i in range(6))
ok = t3 > 0
cu = ((xcc & 0xF) << 12) | (hw & 0xFF00) >> 4
keys = np.unique(cu[ok])
spans = np.array([t3[ok & (cu == c)].max() - t0[ok & (cu == c)].min() for c in keys], dtype=np.float64)
tick_ns = ms * 1e6 / spans.max()
print(f"variable-block attention (Wan 720p SVG2 workload): {ms:.3f} ms, {ok.sum()} / {nwg} workgroups traced, {tick_ns:.3f} ns/tick")
for name, a in (("prologue", (t1 - t0)[ok]), ("tile loop", (t2 - t1)[ok]), ("epilogue", (t3 - t2)[ok])):
    print(f"  {name:9s}: mean {a.mean() * tick_ns / 1e3:8.2f} us  median {np.median(a) * tick_ns / 1e3:8.2f}  p95 {np.percentile(a, 95) * tick_ns / 1e3:8.2f}"
          f"  max {a.max() * tick_ns / 1e3:8.2f}   sum/CU {a.sum() * tick_ns / 1e6 / len(keys):7.3f} ms")
idle = ms - spans * tick_ns / 1e6
print(f"  per-CU time outside [first entry, last exit]: mean {idle.mean():.3f} ms  max {idle.max():.3f}")
