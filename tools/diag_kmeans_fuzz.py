#!/usr/bin/env python3
"""Diagnose a label mismatch of tests/test_gpu_fuzz.py::test_fuzz_kmeans_iter[trial]: for every point whose label differs from the
oracle's argmin print both indices, both distances, and whether the two centroids are bit-identical copies.  usage: diag_kmeans_fuzz.py <trial>"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "sparse-videogen_amd"), str(ROOT / "tests")]
import torch  # noqa: E402
from oracle import svg_oracle as O  # noqa: E402
from svg import _native as nat  # noqa: E402
from test_gpu_fuzz import Rng  # noqa: E402

trial = int(sys.argv[1]) if len(sys.argv) > 1 else 6
r = Rng(4000 + trial)
B, N, K, D = r.ri(1, 3), r.ri(64, 6000), r.ri(1, 400), r.pick((64, 128))
dtype = r.pick((torch.bfloat16, torch.float16))
modes = r.ri(1, 40)
centers = r.randn(B, modes, D) * 2
x = torch.gather(centers, 1, torch.randint(0, modes, (B, N), generator=r.g)[..., None].expand(-1, -1, D)) + 0.5 * r.randn(B, N, D)
x = x.to(dtype)
pick = torch.randint(0, N, (K,), generator=r.g)
c0 = x[:, pick].clone()
if K > 1:
    c0[:, -1] = 100.0
xd = x.cuda()
xsq = nat.kmeans_xsq(xd)
buf = nat.KmeansBuffers(B, N, K, D, xd.device)
c_out = torch.empty_like(c0.cuda())
nat.kmeans_iter(xd, xsq, c0.cuda().contiguous(), c_out, buf)
dist = O.kmeans_distances(x, xsq.cpu(), c0)
raw = xsq.cpu()[:, :, None] + O.kmeans_csq(c0)[:, None, :] - 2.0 * torch.einsum("bnd,bkd->bnk", x.float(), c0.float())
lab = buf.labels.cpu().long()
ref = dist.argmin(-1)
print(f"trial {trial}: B {B} N {N} K {K} D {D} {dtype} modes {modes}; mismatches {(lab != ref).sum().item()} of {B * N}")
for b, n in (lab != ref).nonzero().tolist():
    g, w = lab[b, n].item(), ref[b, n].item()
    same = torch.equal(c0[b, g], c0[b, w])
    print(f"  b {b} n {n}: kernel {g} (d {dist[b, n, g].item():.6g}, unclamped {raw[b, n, g].item():.6g}) oracle {w} (d {dist[b, n, w].item():.6g}, "
          f"unclamped {raw[b, n, w].item():.6g}) centroids identical: {same}; point is centroid kernel/oracle: "
          f"{torch.equal(x[b, n], c0[b, g])}/{torch.equal(x[b, n], c0[b, w])}; zero-distance centroids: {(dist[b, n] == 0).nonzero().flatten().tolist()}")
