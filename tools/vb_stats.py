import sys, torch
sys.path.insert(0, "sparse-videogen_amd"); sys.path.insert(0, ".")
from svg import _native as nat
from svg.models import _core
import bench_svg2 as B
nat.load()
dev = torch.device("cuda", 0)
H, D, F_, P_, ctx, L, QC, KC = B.WORKLOADS["wan720p"]
S = F_ * P_
gen = torch.Generator(device=dev).manual_seed(0)
q = B.clustered(H, S, D, 64, dev, gen)[None]; k = B.clustered(H, S, D, 64, dev, gen)[None]
store = _core.CentroidStore()
(ql, qc, qs, _, qidx), (kl, kc, ks, _, kidx) = _core.kmeans_clustering(store, 0, q, k, QC, KC, 50, 2)
qs = qs.view(H, QC).float()
print("q sizes: mean %.1f std %.1f min %d max %d" % (qs.mean(), qs.std(), qs.min(), qs.max()))
for BM in (64, 128, 256):
    tiles = torch.ceil(qs / BM).sum().item()
    print(f"BM={BM}: tiles {tiles:.0f}, padded rows {tiles*BM:.0f} vs real {H*S} -> row efficiency {H*S/(tiles*BM):.3f}")
# mixed: full 256 tiles + tail in 128 granularity
full = torch.floor(qs / 256); rem = qs - full * 256
rows = (full * 256 + torch.ceil(rem / 128) * 128).sum().item()
print("mixed 256 + 128-granular tail: efficiency %.3f" % (H * S / rows))
rows = (full * 256 + torch.ceil(rem / 64) * 64).sum().item()
print("mixed 256 + 64-granular tail: efficiency %.3f" % (H * S / rows))

# ---- launch model of the 256-row two-phase kernel: dispatch id b = head * max_tiles + w goes to XCD b % 8, each XCD hands its
#      workgroups in order to the first free of its 32 CUs; cost of a workgroup = its KV tiles (64 keys each) ----
import heapq
import numpy as np
ks = ks.view(H, KC).float()

from svg import kmeans_utils as KU
dmap = KU.identify_dynamic_map(qc.view(1, H, QC, D), kc.view(1, H, KC, D), qs.view(1, H, QC).int(), ks.view(1, H, KC).int(), 0.9, 0.1)[0]
keys_per_row = (dmap.float() * ks[:, None, :]).sum(-1)          # [H, QC] active keys of block-row i
nT = torch.ceil(keys_per_row / 64).cpu().numpy()
nsub = torch.ceil(qs / 256).cpu().numpy().astype(int)
max_tiles = S // 256 + QC
cost = []
for h in range(H):
    w = [nT[h, i] for i in range(QC) for _ in range(nsub[h, i])]
    w += [0.0] * (max_tiles - len(w))
    cost += w
cost = np.array(cost)
def makespan(order_cost):
    ends = []
    for x in range(8):
        cus = [0.0] * 32
        heapq.heapify(cus)
        for c in order_cost[x::8]:
            t0 = heapq.heappop(cus)
            heapq.heappush(cus, t0 + c + 3.0)      # ~3 tile-times of prologue/epilogue per workgroup
        ends.append(max(cus))
    return ends
ideal = (cost.sum() + 3.0 * (cost > 0).sum()) / 256
e = makespan(cost)
print(f"dispatch order as shipped: per-XCD finish {np.round(np.array(e) / ideal, 3)}  makespan / ideal = {max(e) / ideal:.3f}")
lpt = np.sort(cost)[::-1]
e = makespan(lpt)
print(f"longest-first order:       per-XCD finish {np.round(np.array(e) / ideal, 3)}  makespan / ideal = {max(e) / ideal:.3f}")
print(f"workgroups {int((cost > 0).sum())}, tiles per workgroup mean {cost[cost > 0].mean():.1f} max {cost.max():.0f}; row occupancy of the 256-row tiles "
      f"{H * S / (256.0 * (cost > 0).sum()):.3f}")
