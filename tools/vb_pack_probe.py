#!/usr/bin/env python3
"""How many ragged last tiles the device-side matching of svg_varblock_attention (variant 3) packs on the SVG2 bench data, and what that
does to the number of (q-tile, key-tile) iterations — counted from the launch order the call leaves in its workspace."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
import torch  # noqa: E402
import bench_svg2 as B  # noqa: E402
from svg import _native as nat  # noqa: E402
from svg.kmeans_utils import identify_dynamic_map  # noqa: E402
from svg.models import _core  # noqa: E402

nat.load()
dev = torch.device("cuda", 0)
H, D, F_, P_, ctx, L, QC, KC = B.WORKLOADS["wan720p"]
S = F_ * P_
gen = torch.Generator(device=dev).manual_seed(0)
q = B.clustered(H, S, D, 64, dev, gen)[None]; k = B.clustered(H, S, D, 64, dev, gen)[None]
v = torch.randn(1, H, S, D, device=dev, dtype=torch.bfloat16, generator=gen)
store = _core.CentroidStore()
_core.kmeans_clustering(store, 0, q, k, QC, KC, 50, 2)
(ql, qc, qs, _, qidx), (kl, kc, ks, _, kidx) = _core.kmeans_clustering(store, 0, q, k, QC, KC, 50, 2)
dmap = identify_dynamic_map(qc.view(1, H, QC, D), kc.view(1, H, KC, D), qs.view(1, H, QC), ks.view(1, H, KC), 0.9, 0.1)[0].contiguous()
qs, ks = qs.view(H, QC).contiguous(), ks.view(H, KC).contiguous()
for variant in tuple(int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("3", "6"))):   # 3: packing (16x16x32 body at head_dim 128), 9: the same on the 32x32x16 body, 6: no packing
    ws = nat.varblock_workspace(H, H, QC, KC, S, dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    nat.varblock_attention(q[0], k[0], v[0], dmap, qs, ks, q_row_idx=qidx.contiguous(), kv_row_idx=kidx.contiguous(), variant=variant, workspace=ws)
    clk = nat.ClockProbe(dev)
    clk.start(max_ms=5000)
    ev[0].record()
    for _ in range(6):
        nat.varblock_attention(q[0], k[0], v[0], dmap, qs, ks, q_row_idx=qidx.contiguous(), kv_row_idx=kidx.contiguous(), variant=variant, workspace=ws)
    ev[1].record()
    clk.arm_stop()
    torch.cuda.synchronize()
    mhz = clk.result()
    order = nat.varblock_launch_order(ws, H, QC, KC).cpu()
    keys = (dmap.float() * ks[:, None, :].float()).sum(-1).cpu()          # [H, QC]
    kt = torch.ceil(keys / 64)
    pairs = int((order[:, 2] >= 0).sum())
    iters = 0.0
    dm, ksc = dmap.cpu(), ks.cpu().float()
    for head, e, pj in order.tolist():
        i = e >> 16
        if pj >= 0:
            iters += float(torch.ceil(((dm[head, i] | dm[head, pj]).float() @ ksc[head]) / 64))
        else:
            iters += float(kt[head, i])
    print(f"variant {variant}: workgroups {order.shape[0]}, packed pairs {pairs} ({pairs / H:.1f} per head), key-tile iterations {iters:.0f}, "
          f"call {ev[0].elapsed_time(ev[1]) / 6:.3f} ms, sustained shader clock {mhz} MHz")
