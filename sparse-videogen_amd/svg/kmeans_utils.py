"""SVG2 operators — same module path and function names as the reference `svg/kmeans_utils.py`, backed by the HIP
kernels of libsvgattn (flash-kmeans: csrc/kmeans.hip, top-p block selection: csrc/dynmap.hip, variable-block attention:
csrc/attention.hip).  No Triton, no flashinfer, no cuVS.

Differences that are deliberate and documented:
  * batch_kmeans_Euclid(check_every=0) runs without any host synchronisation: the reference's stopping rule
    (`center_shift < tol`, :723, a read-back per iteration) is evaluated on the device and freezes the result;
    the default (check_every=1) reads the shift back after every iteration exactly like the reference;
  * centroid sums are reduced in a fixed order (bit-reproducible), the reference uses fp32 atomics;
  * `sorted_indices` is the *stable* argsort of the labels (the reference's order inside a cluster is unspecified);
  * the variable-block attention needs no planning pass, no 4 GiB index buffer and no 512 MB workspace per call.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _native
from .timer import time_logging_decorator


@time_logging_decorator("Level 4- density calculation")
def density_calculation(dynamic_map, q_cluster_sizes, k_cluster_sizes):
    """ref: svg/kmeans_utils.py:13-31.  dynamic_map [cfg, H, QC, KC] bool, sizes [cfg, H, QC] / [cfg, H, KC] -> [cfg, H]"""
    cfg, H, QC, KC = dynamic_map.shape
    if dynamic_map.is_cuda:
        d = _native.map_density(dynamic_map.reshape(cfg * H, QC, KC).contiguous(),
                                q_cluster_sizes.reshape(cfg * H, QC).to(torch.int32).contiguous(),
                                k_cluster_sizes.reshape(cfg * H, KC).to(torch.int32).contiguous())
        return d.reshape(cfg, H)
    blk = q_cluster_sizes[:, :, :, None] * k_cluster_sizes[:, :, None, :]
    return torch.sum(blk * dynamic_map, dim=(2, 3)) / torch.sum(blk, dim=(2, 3))


class KMeansState:
    """Reusable device buffers of batch_kmeans_Euclid for one (B, N, K, D) shape."""

    def __init__(self, x: torch.Tensor, n_clusters: int):
        B, N, D = x.shape
        self.shape = (B, N, n_clusters, D, x.dtype, x.device)
        self.buf = _native.KmeansBuffers(B, N, n_clusters, D, x.device)
        self.c = [torch.empty((B, n_clusters, D), dtype=x.dtype, device=x.device) for _ in range(2)]

    def matches(self, x, n_clusters):
        B, N, D = x.shape
        return self.shape == (B, N, n_clusters, D, x.dtype, x.device)


# per-iteration path (check_every >= 1, sharded stopping rule): label / count / centroid buffers per (shape, device, stream), bounded like
# the other workspaces (svg._native.WorkspaceCache) — two streams on one shape never share a KMeansState
_STATE_CACHE = _native.WorkspaceCache()
_LOOP_WORK = _native._KMEANS_WS   # scratch of svg_kmeans_loop per (B, N, K, D, dtype, device, stream), bounded (svg._native.WorkspaceCache)


@time_logging_decorator("Level 4 - batch kmeans euclid")
def batch_kmeans_Euclid(x, n_clusters, max_iters=100, tol=1e-4, init_centroids=None, verbose=False, check_every=1,
                        return_sorted_indices=False, shift_reduce=None):
    """ref: batch_kmeans_Euclid, svg/kmeans_utils.py:684-733.

    x: [B, N, D] bf16/fp16 GPU tensor.  Returns (cluster_ids int64 [B, N], centroids [B, K, D], cluster_sizes int32 [B, K],
    n_iters) — and the stable sorted indices int32 [B, N] when `return_sorted_indices` (they come for free from the
    centroid update and save the argsort of permute_tensor_by_labels_triton).

    Semantics of the reference's loop (:716-733), in both modes: iteration `it` assigns with the current centroids and computes
    new ones; if the largest centre shift (over all batches) is below `tol` the loop stops and returns the labels / sizes of
    THAT iteration, the OLD centroids and n_iters = it + 1; otherwise the new centroids become current — so without
    convergence the returned centroids are one update ahead of the returned labels.

    check_every = 1 (default): like the reference, the shift is read back after every iteration (one host sync each; n > 1
        reads it every n-th iteration only, i.e. may overshoot by up to n - 1 iterations).
    check_every = 0: no host synchronisation at all — every iteration is launched, and a device-side flag freezes the result at
        the iteration where the reference would have stopped (a few torch.where over the result tensors per iteration).  Same
        labels, centroids and sizes as check_every = 1; n_iters is then a 0-dim int64 GPU tensor instead of an int.
    shift_reduce: optional callable applied in place to the 0-dim maximum centre shift before it is compared with `tol` — the
        head-sharded layer-call (svg.distributed) passes an all-reduce(MAX) so that the stopping rule stays the reference's
        maximum over ALL heads."""
    _red = shift_reduce if shift_reduce is not None else (lambda t: t)
    assert x.is_cuda, "batch_kmeans_Euclid requires GPU tensors"
    assert max_iters >= 1, "max_iters must be >= 1 (the reference raises NameError for 0)"
    B, N, D = x.shape
    loop_in_library = not check_every and shift_reduce is None and not verbose
    # (the library loop reads batches that lie further apart than N * D in place — svg_kmeans_loop_strided: the video tokens of a
    #  [H, S, D] tensor with text rows behind them; every other path takes the contiguous copy the reference makes)
    if not (loop_in_library and x.dim() == 3 and x.stride(2) == 1 and x.stride(1) == D and x.stride(0) >= N * D):
        x = x.contiguous()
    st = None
    if not loop_in_library:     # (the svg_kmeans_loop path keeps its own scratch: no KMeansState for it)
        key = _native.WorkspaceCache.key(B, N, n_clusters, D, x.dtype, device=x.device)
        st = _STATE_CACHE.get(key)
        if st is None:
            st = _STATE_CACHE[key] = KMeansState(x, n_clusters)
    xsq = None   # the reference's x_sq (:704) is not needed: the assignment kernel takes argmax_k (<x, c_k> - |c_k|^2 / 2)
    if init_centroids is None:
        # ref :706-709 — random points of x as initial centres (device RNG, not reproducible across platforms)
        indices = torch.randint(0, N, (B, n_clusters), device=x.device)
        c_in = torch.gather(x, dim=1, index=indices[..., None].expand(-1, -1, D)).contiguous()
    else:
        c_in = init_centroids.reshape(B, n_clusters, D).contiguous()
    cur = c_in
    if check_every:
        n_done = 0
        for it in range(max_iters):
            c_out = st.c[it & 1]
            _native.kmeans_iter(x, xsq, cur, c_out, st.buf)
            n_done = it + 1
            if verbose:
                print(f"Iter {it}, center shift: {st.buf.shift.max().item():.6f}")
            if (it + 1) % check_every == 0 and _red(st.buf.shift.max()).item() < tol:
                break  # converged: like the reference, keep the OLD centroids (`cur`)
            cur = c_out
        out = (st.buf.labels.to(torch.int64), cur.clone(), st.buf.counts.clone(), n_done)
        if return_sorted_indices:
            return out + (st.buf.sorted_idx.clone(),)
        return out
    # ---- device-side convergence: nothing below reads a value back to the host ----
    if loop_in_library:
        # the whole loop inside the library (svg_kmeans_loop): the iterations and a commit kernel that applies the stopping rule on the
        # device — the same result as the torch statement below (kept for the sharded path, whose stopping rule needs an all-reduce
        # between the iterations), without its ~10 framework launches per iteration
        labels_r, cent_r, counts_r, n_r, sorted_r = _native.kmeans_loop(x, xsq, c_in, max_iters, tol, work=_LOOP_WORK)
        out = (labels_r.to(torch.int64), cent_r, counts_r, n_r.to(torch.int64))
        return out + (sorted_r,) if return_sorted_indices else out
    labels_r = cent_r = counts_r = sorted_r = None
    stopped = torch.zeros((), dtype=torch.bool, device=x.device)   # the reference's loop has left at an earlier iteration
    n_r = torch.zeros((), dtype=torch.int64, device=x.device)
    for it in range(max_iters):
        c_out = st.c[it & 1]
        _native.kmeans_iter(x, xsq, cur, c_out, st.buf)
        conv_now = _red(st.buf.shift.max()) < tol
        if it == 0:
            labels_r, counts_r, sorted_r = st.buf.labels.clone(), st.buf.counts.clone(), st.buf.sorted_idx.clone()
            cent_r = torch.where(conv_now, c_in, c_out)
        else:
            run = ~stopped
            labels_r = torch.where(run, st.buf.labels, labels_r)
            counts_r = torch.where(run, st.buf.counts, counts_r)
            sorted_r = torch.where(run, st.buf.sorted_idx, sorted_r)
            cent_r = torch.where(run & ~conv_now, c_out, cent_r)
        n_r = n_r + (~stopped).to(torch.int64)
        stopped = stopped | conv_now
        cur = c_out
    out = (labels_r.to(torch.int64), cent_r, counts_r, n_r)
    if return_sorted_indices:
        return out + (sorted_r,)
    return out


@time_logging_decorator("Level 4 - weighted softmax")
def weighted_softmax(scores, weights):
    """ref: svg/kmeans_utils.py:852-861 (torch; kept for API parity — the fused HIP op is identify_dynamic_map)"""
    input_dtype = scores.dtype
    scores = scores.float()
    weights = weights.float()
    max_score = torch.max(scores, dim=-1, keepdim=True)[0]
    weighted_exp = weights * torch.exp(scores - max_score)
    return (weighted_exp / torch.sum(weighted_exp, dim=-1, keepdim=True).clamp(min=1e-12)).to(input_dtype)


@time_logging_decorator("Level 4 - identify dynamic map")
def identify_dynamic_map(query_centroids, key_centroids, q_cluster_sizes, k_cluster_sizes, p, min_kc_ratio=0):
    """ref: svg/kmeans_utils.py:864-896.  centroids [B, H, QC|KC, D], k_cluster_sizes [B, H, KC] -> bool [B, H, QC, KC]"""
    B, H, QC, D = query_centroids.shape
    KC = key_centroids.shape[2]
    assert query_centroids.is_cuda, "identify_dynamic_map requires GPU tensors"
    preserve = int(min_kc_ratio * KC) if min_kc_ratio > 0 else 0
    m = _native.identify_dynamic_map(query_centroids.reshape(B * H, QC, D).contiguous(),
                                     key_centroids.reshape(B * H, KC, D).contiguous(),
                                     k_cluster_sizes.reshape(B * H, KC).to(torch.int32).contiguous(), float(p), preserve)
    return m.reshape(B, H, QC, KC)


@time_logging_decorator("Level 3 - dynamic block sparse fwd flashinfer on GPU")
def dynamic_block_sparse_fwd_flashinfer(q, k, v, block_mask_map, block_row_sz, block_col_sz, is_cpu: bool = True, *,
                                        q_sorted_indices: Optional[torch.Tensor] = None,
                                        kv_sorted_indices: Optional[torch.Tensor] = None):
    """ref: svg/kmeans_utils.py:1319-1392 (name kept for drop-in; there is no flashinfer underneath).

    q, k, v: [B, H, S, D]; block_mask_map bool [B, H, QB, KB]; block_row_sz / block_col_sz int [B, H, QB] / [B, H, KB].
    `is_cpu` is accepted and ignored (map and sizes may live on either device; they are moved to the GPU).
    Optional q_sorted_indices / kv_sorted_indices ([B*H, S] int32): q/k/v are then the ORIGINAL (un-permuted) tensors and
    the permutation / inverse permutation is fused into the kernel."""
    B, H, S, D = q.shape
    QB, KB = block_row_sz.shape[-1], block_col_sz.shape[-1]
    assert block_mask_map.shape == (B, H, QB, KB)
    dev = q.device
    bm = block_mask_map.to(dev).reshape(B * H, QB, KB).contiguous()
    rs = block_row_sz.to(dev).reshape(B * H, QB).to(torch.int32).contiguous()
    cs = block_col_sz.to(dev).reshape(B * H, KB).to(torch.int32).contiguous()
    qi = None if q_sorted_indices is None else q_sorted_indices.reshape(B * H, S).to(torch.int32).contiguous()
    ki = None if kv_sorted_indices is None else kv_sorted_indices.reshape(B * H, k.shape[2]).to(torch.int32).contiguous()
    with time_logging_decorator("Level 4 - Running"):
        o = _native.varblock_attention(q.reshape(B * H, S, D).contiguous(), k.reshape(B * H, k.shape[2], D).contiguous(),
                                       v.reshape(B * H, v.shape[2], D).contiguous(), bm, rs, cs, q_row_idx=qi, kv_row_idx=ki)
    return o.reshape(B, H, S, D)


def dynamic_block_sparse_fwd_torch(q, k, v, dynamic_map, qc_size, kc_size):
    """ref: svg/kmeans_utils.py:902-995 — plain torch statement of the same attention (any device, slow).  Kept because
    the reference exposes it; it is dense attention under the block mask expanded to elements."""
    B, H, S, D = q.shape
    out = torch.zeros_like(q)
    for b in range(B):
        for h in range(H):
            em = torch.repeat_interleave(torch.repeat_interleave(dynamic_map[b, h], qc_size[b, h].long(), dim=0),
                                         kc_size[b, h].long(), dim=1)
            s = (q[b, h].float() @ k[b, h].float().T) * D ** -0.5
            s = s.masked_fill(~em, float("-inf"))
            p = torch.softmax(s, dim=-1)
            p = torch.where(em.any(dim=-1, keepdim=True), p, torch.zeros_like(p))
            out[b, h] = (p @ v[b, h].float()).to(q.dtype)
    return out


@time_logging_decorator("Level 4 - permute tensor by labels")
def permute_tensor_by_labels(tensor, labels, dim):
    """ref: svg/kmeans_utils.py:828-838 (torch version, any device; stable order)"""
    sorted_indices = torch.argsort(labels.to(tensor.device), dim=-1, stable=True)
    gi = sorted_indices
    for _ in range(dim + 1, tensor.dim()):
        gi = gi.unsqueeze(-1)
    return torch.gather(tensor, dim, gi.expand(list(tensor.shape))), sorted_indices


@time_logging_decorator("Level 4 - inverse permutation")
def apply_inverse_permutation(permuted_tensor, sorted_indices, dim):
    """ref: svg/kmeans_utils.py:841-849"""
    inv = torch.argsort(sorted_indices, dim=-1)
    gi = inv
    for _ in range(dim + 1, permuted_tensor.dim()):
        gi = gi.unsqueeze(-1)
    return torch.gather(permuted_tensor, dim, gi.expand(permuted_tensor.shape))


def dynamic_block_sparse_fwd_triton(q, k, v, dynamic_map, qc_size, kc_size):
    """ref: svg/kmeans_utils.py:1205-1317 — the reference's Triton implementation of the variable-block attention, same arguments
    (q, k, v [B, H, S, D] in cluster order, dynamic_map bool [B, H, QC, KC], sizes [B, H, QC] / [B, H, KC]).  Here it is the same HIP
    kernel as `dynamic_block_sparse_fwd_flashinfer` (tests/test_gpu_triton_golden.py compares that kernel with the OUTPUT of the
    reference's Triton kernel)."""
    return dynamic_block_sparse_fwd_flashinfer(q, k, v, dynamic_map, qc_size, kc_size, is_cpu=False)


# ---- the two halves of a Euclidean iteration under the reference's names (ref svg/kmeans_utils.py:562-627, 375-421, 208-255) ----
# (host plumbing added at the end of round 3 on the kernels batch_kmeans_Euclid runs; GPU test: tests/test_gpu_reference_calls.py)
def euclid_assign_triton(x, centroids, x_sq=None, out=None, *, BLOCK_N: int = 128, BLOCK_K: int = 128):
    """-> cluster ids int64 [B, N] (ref :562-627).  `x_sq` and the tile sizes are accepted and not needed: the HIP kernel takes
    argmax_k (<x, c_k> - |c_k|^2 / 2), with fp32 centroid norms (the Triton kernel reduces them in the input type)."""
    assert x.is_cuda and centroids.is_cuda, "All tensors must be on CUDA"
    assert centroids.dtype == x.dtype, "centroids dtype mismatch"
    ids = _native.kmeans_assign(x.contiguous(), centroids.contiguous()).to(torch.int64)
    if out is not None:
        out.copy_(ids)
        return out
    return ids


def triton_centroid_update_sorted_euclid(x, cluster_ids, old_centroids, *, BLOCK_N: int = 256):
    """-> (centroids [B, K, D] of x.dtype, counts int32 [B, K]) (ref :375-421): fp32 means in a fixed order (bit-reproducible, unlike
    the reference's atomics), empty clusters keep the old centroid."""
    assert x.is_cuda and cluster_ids.is_cuda, "Inputs must be on CUDA device"
    cent, counts, _, _ = _native.kmeans_update(x.contiguous(), cluster_ids.to(torch.int32).contiguous(), old_centroids.to(x.dtype).contiguous())
    return cent, counts


def triton_centroid_update_euclid(x, cluster_ids, old_centroids):
    """ref :208-255 (the unsorted, one-atomic-per-token kernel): the same result as the sorted form -> centroids only"""
    return triton_centroid_update_sorted_euclid(x, cluster_ids, old_centroids)[0]
