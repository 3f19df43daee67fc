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
