#!/usr/bin/env python3
"""Loop-level pin of flash-kmeans: runs the REFERENCE's own `batch_kmeans_Euclid`, `_euclid_iter` and the host half of
`triton_centroid_update_sorted_euclid` (svg/kmeans_utils.py:375-421,629-733) on CPU in this container.  Only the two Triton
kernel LAUNCHES are replaced — they cannot run without a GPU:
  * `euclid_assign_triton`        -> the torch form the reference keeps commented next to the call (:631-635) with the kernel's
                                     rounding points (:512-538): x^2 summed in the input dtype, c*c rounded, fp32 cross term
  * `_centroid_update_chunk_kernel[grid](...)` -> fp32 index_add of the gathered rows + counts (what the kernel accumulates)
Everything the reference does around them — the sort, clamp(count, 1), empty clusters keep the old centroid, the cast back to
the input dtype, the centre shift, the `< tol` break and the 'centroids one update ahead of the labels' return convention — is
the reference's code.  Writes tests/golden/kmeans_loop_golden.npz;  python tests/golden/make_golden_kmeans.py"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_golden as MG  # noqa: E402  (stubs for the third-party modules the reference imports)


class _FakeChunkKernel:
    def __getitem__(self, grid):
        def launch(x, sorted_idx, sorted_ids, sums, cnts, B, N, D, K, BLOCK_N=256):
            for b in range(B):
                rows = x[b][sorted_idx[b].long()].float()
                sums[b].index_add_(0, sorted_ids[b].long(), rows)
                cnts[b] += torch.bincount(sorted_ids[b].long(), minlength=K).to(cnts.dtype)
        return launch


def torch_assign(x, centroids, x_sq, out=None, **kw):
    cent_sq = (centroids * centroids).float().sum(dim=-1)
    cross = torch.einsum("bnd,bkd->bnk", x.float(), centroids.float())
    dist = (x_sq.float()[:, :, None] + cent_sq[:, None, :] - 2.0 * cross).clamp_min(0.0)
    return dist.argmin(dim=-1)


def main():
    MG.install_stubs()
    sys.path.insert(0, MG.REF)
    torch.Tensor.cuda = lambda self, *a, **k: self
    import svg.kmeans_utils as KU

    KU.euclid_assign_triton = torch_assign
    KU._centroid_update_chunk_kernel = _FakeChunkKernel()
    KU._euclid_iter_compiled = KU._euclid_iter
    is_cuda = torch.Tensor.is_cuda
    torch.Tensor.is_cuda = property(lambda self: True)      # the host wrappers assert .is_cuda
    out = {}
    try:
        for tag, (B, N, D, K, iters, dt, seed) in {
            "a": (3, 500, 64, 12, 6, torch.bfloat16, 0),
            "b": (2, 700, 128, 40, 3, torch.bfloat16, 1),     # more clusters than well-separated modes: empty clusters appear
            "c": (1, 300, 64, 8, 50, torch.float32, 2),        # converges before max_iters (tol break)
        }.items():
            g = torch.Generator().manual_seed(seed)
            centers = torch.randn(B, 6, D, generator=g) * 2
            lab = torch.randint(0, 6, (B, N), generator=g)
            x = (torch.gather(centers, 1, lab[..., None].expand(-1, -1, D)) + 0.3 * torch.randn(B, N, D, generator=g)).to(dt)
            init = x[:, :K].clone()
            if tag == "b":
                init[:, -3:] = 50.0                             # far-away seeds: they stay empty
            ids, cent, sizes, n_it = KU.batch_kmeans_Euclid(x, K, max_iters=iters, tol=1e-4, init_centroids=init.clone())
            out[f"{tag}_x"] = x.float().numpy()
            out[f"{tag}_init"] = init.float().numpy()
            out[f"{tag}_meta"] = np.array([B, N, D, K, iters, {torch.bfloat16: 0, torch.float32: 2}[dt]], dtype=np.int64)
            out[f"{tag}_ids"] = ids.numpy().astype(np.int32)
            out[f"{tag}_centroids"] = cent.float().numpy()
            out[f"{tag}_sizes"] = sizes.numpy().astype(np.int32)
            out[f"{tag}_iters"] = np.int64(n_it)
    finally:
        torch.Tensor.is_cuda = is_cuda
    p = HERE / "kmeans_loop_golden.npz"
    np.savez_compressed(p, **out)
    print(f"wrote {p} ({p.stat().st_size / 1024:.0f} KB), iterations: " + ", ".join(f"{t}={int(out[t + '_iters'])}" for t in "abc"))


if __name__ == "__main__":
    main()
