"""CPU oracle for the sparse-attention hot path of svg-project/Sparse-VideoGen.

TEST INFRASTRUCTURE ONLY.  Nothing under sparse-videogen_amd/ imports this module; only tests/, bench.py's
`cpu_baseline` leg and __graft_entry__.smoke() may, and only as the checker.  Every function restates (in plain
torch on CPU, fp32 unless the reference itself computes in the input dtype) what the cited reference code computes.
Citations are `path:line` under the reference tree (/root/reference).

Pinning status (see tests/golden/make_golden.py, which imports the reference itself to produce the fixtures):
  pinned by running the reference's own code here : mask_mod masks (hy/wan/cog), flex_attention output on CPU,
      placement (hy/cog ref fns), torch permute / inverse, weighted_softmax, identify_dynamic_map,
      dynamic_block_sparse_fwd_torch, density_calculation, sparsity_to_width, get_attention_mask (hy/wan/cog) and
      sample_mse (Hunyuan processor method).
      Beyond the committed geometries: tools/fuzz_oracle_vs_reference.py draws random ones (frame counts, ragged frame sizes, text lengths,
      multipliers, cluster counts with empty clusters, top-p / min_kc_ratio) and compares these functions with the reference's on each —
      all equal (profiles/r03zt_fuzz_oracle_vs_reference.txt); tools/fuzz_oracle_vs_triton.py does the same against the interpreted Triton
      kernels in float32 (profiles/r03zu_fuzz_oracle_vs_triton.txt: equal; k-means labels differ only on fp32 near-ties, counted there).
  pinned at loop level: flash-kmeans — tests/golden/make_golden_kmeans.py runs the reference's own batch_kmeans_Euclid /
      _euclid_iter / host half of triton_centroid_update_sorted_euclid on CPU with only the two Triton kernel LAUNCHES replaced
      (svg/kmeans_utils.py:258-554; their bodies are restated here from the Triton source and the commented torch form :631-635).
  pinned by the reference's own torch references / generators: QK-norm + RoPE prologue (make_golden_prologue.py), BSR masks of
      the uniform-block ops (make_golden_bsr.py).
  pinned at KERNEL level by executing the reference's own `@triton.jit` sources with Triton's interpreter (TRITON_INTERPRET=1, fp32 and
      fp16; tests/golden/make_golden_triton.py, tests/test_triton_golden.py): the flash-kmeans assignment and update kernels and the
      loop on both of them, the reference's Triton statement of the variable-block attention (_dynamic_block_sparse_fwd_kernel), the
      head-placement and permutation kernels of the three models, the LayerNorm / RMSNorm / modulate kernels of the Wan block.
      Found that way: the LayerNorm kernels count the zero padding up to the next power of two in the VARIANCE (a reference quirk for
      hidden sizes such as 1536 / 5120; `fp32_layernorm` below states FP32LayerNorm, the specification — see its docstring).
  pinned at PROCESSOR level by executing the reference's processors (same generator, sections 8-16; section 14 is the Wan block forward on both of its branches, held against the glue functions below): `attention_core_logic` of the SVG1
      processors (Hunyuan, Wan, CogVideoX) and of the SAP processors (Hunyuan, Wan), and the whole `__call__` of the Wan, Hunyuan (double-
      and single-stream) and CogVideoX SVG processors on a duck-typed attention module — the composition of the functions below that the
      tests use as the checker for the product's processors is itself held to the reference's output (1e-3, fp32).
  PARITY UNPINNED: (a) the bf16 rounding points of the two k-means norms (`kmeans_xsq`, `kmeans_csq`): the interpreter of this
      image's Triton cannot compute in bfloat16, and on the GPU `tl.sum` of a 16-bit tensor reduces in that type in an
      implementation-chosen order, so they are not bit-defined by the reference either — labels are compared up to that noise;
      (b) the flashinfer variable-block kernel (third-party, GPU-only), anchored on the reference's own test
      (svg/kernels/test/test_sparse_attn_dyn_blk_wan.py:38-133: dense attention under the repeat-interleaved block mask) and now also
      on the reference's Triton kernel for the same operator.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------------------------
# sparsity -> band width, mask_mod predicates
# ----------------------------------------------------------------------------------------------------------


def sparsity_to_width(sparsity: float, context_length: int, num_frame: int, frame_size: int) -> float:
    """ref: svg/models/hyvideo/utils.py:142-151 (identical in wan/utils.py:51-60, cog/utils.py:49-58)"""
    seq_len = context_length + num_frame * frame_size
    total = seq_len ** 2
    s = (sparsity * total - 2 * seq_len * context_length) / total
    width = seq_len * (1 - math.sqrt(1 - s))
    return width / frame_size


def _q_index(S: int, rows) -> torch.Tensor:
    """query indices as a column: all of them, or the selected `rows` (full-size spot checks: the whole [S, S] mask does not fit)"""
    return torch.arange(S)[:, None] if rows is None else torch.as_tensor(rows, dtype=torch.long)[:, None]


def hy_mask(S: int, context_length: int, prompt_length: int, num_frames: int, P: int, mul: float, rows=None) -> torch.Tensor:
    """Dense bool mask of generate_temporal_head_mask_mod, ref: svg/models/hyvideo/utils.py:20-44"""
    q = _q_index(S, rows)
    k = torch.arange(S)[None, :]
    real_length = num_frames * P + prompt_length
    real = (k < real_length) & (q < real_length)
    fake = (k >= real_length) & (q >= real_length)
    two_frame = math.floor(mul * P / 128) * 128
    band = (q - k).abs() < two_frame
    text_col = (num_frames * P <= k) & (k < real_length)
    text_row = (num_frames * P <= q) & (q < real_length)
    return (real & (band | text_col | text_row)) | fake


def wan_mask(S: int, num_frames: int, P: int, mul: float, rows=None) -> torch.Tensor:
    """ref: svg/models/wan/utils.py:25-41 (ceil, <=, first-frame sink columns)"""
    q = _q_index(S, rows)
    k = torch.arange(S)[None, :]
    two_frame = math.ceil(mul * P / 128) * 128
    return ((q - k).abs() <= two_frame) | (k < P)


def cog_mask(S: int, prompt_length: int, num_frames: int, P: int, mul: float, attn_sink: bool = False, rows=None) -> torch.Tensor:
    """ref: svg/models/cog/utils.py:30-46 (text first)"""
    q = _q_index(S, rows)
    k = torch.arange(S)[None, :]
    first_row = q < prompt_length
    first_col = k < (prompt_length + P) if attn_sink else k < prompt_length
    two_frame = math.floor(mul * P / 128) * 128
    return first_col | first_row | ((q - k).abs() < two_frame)


# the six integers of svg_band_mask_t (include/svg_attn.h) for each model
def hy_band_params(S, context_length, prompt_length, num_frames, P, mul):
    V = num_frames * P
    real = V + prompt_length
    return dict(real_len=real, band=math.floor(mul * P / 128) * 128, colfull_lo=V, colfull_hi=real, rowfull_lo=V,
                rowfull_hi=real)


def wan_band_params(S, num_frames, P, mul):
    return dict(real_len=S, band=math.ceil(mul * P / 128) * 128 + 1, colfull_lo=0, colfull_hi=P, rowfull_lo=0, rowfull_hi=0)


def cog_band_params(S, prompt_length, num_frames, P, mul, attn_sink=False):
    return dict(real_len=S, band=math.floor(mul * P / 128) * 128, colfull_lo=0,
                colfull_hi=prompt_length + (P if attn_sink else 0), rowfull_lo=0, rowfull_hi=prompt_length)


def dense_band_params(S, valid_len=None):
    return dict(real_len=S if valid_len is None else valid_len, band=S + 1, colfull_lo=0, colfull_hi=0, rowfull_lo=0,
                rowfull_hi=0)


def band_mask(S: int, real_len: int, band: int, colfull_lo: int, colfull_hi: int, rowfull_lo: int, rowfull_hi: int):
    """Dense restatement of the predicate documented in include/svg_attn.h (svg_band_mask_t)."""
    q = torch.arange(S)[:, None]
    k = torch.arange(S)[None, :]
    rq, rk = q < real_len, k < real_len
    in_band = (q - k).abs() < band
    colf = (k >= colfull_lo) & (k < colfull_hi)
    rowf = (q >= rowfull_lo) & (q < rowfull_hi)
    return (rq & rk & (in_band | colf | rowf)) | (~rq & ~rk)


# ----------------------------------------------------------------------------------------------------------
# attention under an element mask
# ----------------------------------------------------------------------------------------------------------


def masked_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, mask: Optional[torch.Tensor],
                     scale: Optional[float] = None) -> torch.Tensor:
    """softmax(q k^T * scale + mask) v in fp32; q,k,v [..., S, D]; mask bool [Sq, Skv] or broadcastable.
    Rows without any allowed key give zeros (ref: svg/kmeans_utils.py:993 clamp, flex_attention convention).
    This is what flex_attention(q,k,v,block_mask) (hyvideo/attention.py:401-403) and the dense-masked reference of
    svg/kernels/test/test_sparse_attn.py:66-87 compute."""
    qf, kf, vf = q.float(), k.float(), v.float()
    scale = 1.0 / math.sqrt(q.shape[-1]) if scale is None else scale
    s = torch.matmul(qf, kf.transpose(-1, -2)) * scale
    if mask is not None:
        s = s.masked_fill(~mask, float("-inf"))
    m = s.max(dim=-1, keepdim=True).values
    m = torch.where(torch.isinf(m), torch.zeros_like(m), m)
    p = torch.exp(s - m)
    l = p.sum(dim=-1, keepdim=True)
    o = torch.matmul(p, vf) / l.clamp(min=1e-30)
    return torch.where(l > 0, o, torch.zeros_like(o))


def chunked_masked_attention(q, k, v, mask_fn, scale=None, chunk=2048):
    """Same as masked_attention but builds the mask chunk-by-chunk: mask_fn(q0, q1) -> bool [q1-q0, Skv]."""
    outs = []
    for q0 in range(0, q.shape[-2], chunk):
        q1 = min(q.shape[-2], q0 + chunk)
        outs.append(masked_attention(q[..., q0:q1, :], k, v, mask_fn(q0, q1), scale))
    return torch.cat(outs, dim=-2)


# ----------------------------------------------------------------------------------------------------------
# layout transformation (placement)
# ----------------------------------------------------------------------------------------------------------


def _video_slice(S, context_length, text_first):
    return (context_length, S) if text_first else (0, S - context_length)


def head_placement(x: torch.Tensor, best_mask_idx: torch.Tensor, context_length: int, num_frame: int, frame_size: int,
                   text_first: bool = False, inverse: bool = False) -> torch.Tensor:
    """ref: ref_hunyuan_sparse_head_placement svg/models/hyvideo/placement.py:156-184 with
    hunyuan_token_reorder_to_token_major :6-17 (forward) / ref_hunyuan_hidden_states_placement :390-401 with
    _to_frame_major :20-31 (inverse); text-first: svg/models/cog/placement.py:6-31.  Correct for context_length == 0
    (the reference torch helper slices `[:-0]` and breaks there; its Triton kernel does not — SURVEY hazard 6)."""
    cfg, H, S, D = x.shape
    lo, hi = _video_slice(S, context_length, text_first)
    out = x.clone()
    vid = x[:, :, lo:hi, :]
    if not inverse:
        perm = vid.reshape(cfg, H, num_frame, frame_size, D).transpose(2, 3).reshape(cfg, H, hi - lo, D)
    else:
        perm = vid.reshape(cfg, H, frame_size, num_frame, D).transpose(2, 3).reshape(cfg, H, hi - lo, D)
    sel = best_mask_idx.to(torch.bool)
    out[:, :, lo:hi, :] = torch.where(sel[:, :, None, None], perm, vid)
    return out


# ----------------------------------------------------------------------------------------------------------
# token permutation
# ----------------------------------------------------------------------------------------------------------


def stable_argsort(labels: torch.Tensor) -> torch.Tensor:
    """argsort(labels, stable=True): the tie order the reference leaves open (permute.py:113) fixed to 'by index'."""
    return torch.argsort(labels, dim=-1, stable=True)


def permute_by_labels(t: torch.Tensor, labels: torch.Tensor):
    """ref: permute_tensor_by_labels svg/kmeans_utils.py:828-838 / permute.py:82-128.  t [B,H,S,D], labels [B*H,S]"""
    B, H, S, D = t.shape
    idx = stable_argsort(labels.reshape(B * H, S))
    out = torch.gather(t.reshape(B * H, S, D), 1, idx[..., None].expand(-1, -1, D)).reshape(B, H, S, D)
    return out, idx.to(torch.int32)


def inverse_permutation(t_perm: torch.Tensor, sorted_indices: torch.Tensor) -> torch.Tensor:
    """ref: apply_inverse_permutation svg/kmeans_utils.py:841-849 / permute.py:131-170: out[idx[s]] = in[s]"""
    B, H, S, D = t_perm.shape
    out = torch.empty_like(t_perm).reshape(B * H, S, D)
    out.scatter_(1, sorted_indices.long().reshape(B * H, S)[..., None].expand(-1, -1, D), t_perm.reshape(B * H, S, D))
    return out.reshape(B, H, S, D)


# ----------------------------------------------------------------------------------------------------------
# flash-kmeans (kernel bodies restated from the Triton source and its commented torch form; the LOOP — finalisation, empty
# clusters, shift / tol break, return convention — is pinned against the reference's own batch_kmeans_Euclid run on CPU with only
# the two Triton launches replaced: tests/golden/make_golden_kmeans.py, tests/test_oracle_golden.py)
# ----------------------------------------------------------------------------------------------------------


def kmeans_xsq(x: torch.Tensor) -> torch.Tensor:
    """ref: svg/kmeans_utils.py:704 `(x**2).sum(dim=-1)` in the input dtype, cast to fp32 inside the kernel (:512)"""
    return (x ** 2).sum(dim=-1).float()


def kmeans_csq(c: torch.Tensor) -> torch.Tensor:
    """ref: svg/kmeans_utils.py:531 `tl.sum(c_tile * c_tile)`: product rounded to the input dtype, fp32 sum"""
    return (c * c).float().sum(dim=-1)


def kmeans_distances(x, xsq, c):
    """ref: svg/kmeans_utils.py:533-538 dist = x_sq + cent_sq - 2 x.c (fp32 accumulate), clamped at 0"""
    cross = torch.einsum("bnd,bkd->bnk", x.float(), c.float())
    return (xsq[:, :, None] + kmeans_csq(c)[:, None, :] - 2.0 * cross).clamp_min(0.0)


def kmeans_assign(x, xsq, c):
    """argmin with lowest-index tie-break (ref :543-548 strict '<' across chunks, first-index tl.argmin inside)"""
    return kmeans_distances(x, xsq, c).argmin(dim=-1)


def kmeans_update(x, labels, c_old):
    """ref: triton_centroid_update_sorted_euclid svg/kmeans_utils.py:375-421 (fp32 sums, clamp(count,1), empty
    cluster keeps the old centroid, cast to x.dtype)"""
    B, N, D = x.shape
    K = c_old.shape[1]
    sums = torch.zeros(B, K, D, dtype=torch.float32)
    sums.scatter_add_(1, labels[..., None].expand(-1, -1, D), x.float())
    counts = torch.zeros(B, K, dtype=torch.int64)
    counts.scatter_add_(1, labels, torch.ones_like(labels))
    cent = sums / counts.float().clamp(min=1.0)[..., None]
    cent = torch.where((counts == 0)[..., None], c_old.float(), cent)
    return cent.to(x.dtype), counts.to(torch.int32)


def kmeans_iter(x, xsq, c):
    """one Lloyd iteration, ref: _euclid_iter svg/kmeans_utils.py:629-643"""
    labels = kmeans_assign(x, xsq, c)
    c_new, counts = kmeans_update(x, labels, c)
    shift = (c_new - c).norm(dim=-1).max()
    return c_new, shift, labels, counts


def batch_kmeans_euclid(x, n_clusters, max_iters=100, tol=1e-4, init_centroids=None):
    """ref: batch_kmeans_Euclid svg/kmeans_utils.py:684-733 incl. its quirk: returned centroids are one update ahead of
    the returned labels unless it converged (SURVEY hazard 5).  init_centroids is required (device RNG otherwise)."""
    assert init_centroids is not None, "pass init_centroids: the reference's random init is device-RNG dependent"
    B, N, D = x.shape
    xsq = kmeans_xsq(x)
    c = init_centroids.view(B, n_clusters, D)
    for it in range(max_iters):
        c_new, shift, labels, counts = kmeans_iter(x, xsq, c)
        if shift < tol:
            break
        c = c_new
    return labels, c, counts, it + 1


# ----------------------------------------------------------------------------------------------------------
# SVG2 block selection and variable-block attention
# ----------------------------------------------------------------------------------------------------------


def weighted_softmax(scores: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
    """ref: svg/kmeans_utils.py:852-861"""
    dt = scores.dtype
    s = scores.float()
    w = weights.float()
    e = torch.exp(s - s.max(dim=-1, keepdim=True)[0])
    we = w * e
    return (we / we.sum(dim=-1, keepdim=True).clamp(min=1e-12)).to(dt)


def identify_dynamic_map(qc, kc, q_sizes, k_sizes, p, min_kc_ratio=0.0, exact=False):
    """ref: svg/kmeans_utils.py:864-896, with the sort made stable (ties -> lower cluster index; SURVEY hazard 2).

    exact=False: the reference's own arithmetic — torch's 16-bit matmul (fp32 accumulation in torch's order) and the fp32
    softmax; this is what the golden vectors generated by the reference pin.
    exact=True: the same rounding points with the rounded quantities computed exactly enough to be order-independent (dot
    products and softmax in fp64, then the same conversions to the input dtype).  It differs from exact=False only where the
    reference's fp32 accumulation ORDER decides a rounding (tests/test_oracle_golden.py shows every such entry is a <= 1 ulp
    near-tie) and it is what the HIP kernel reproduces bit for bit (csrc/dynmap.hip)."""
    B, H, QC, D = qc.shape
    KC = kc.shape[2]
    if exact:
        s0 = torch.matmul(qc.double(), kc.double().transpose(-2, -1)).to(qc.dtype)   # double -> float -> 16 bit, like Elt::from_double
        scores = s0 / (D ** 0.5)
        w = k_sizes.unsqueeze(-2).double()
        e = torch.exp(scores.double() - scores.double().max(dim=-1, keepdim=True)[0])
        we = w * e
        probs = (we / we.sum(dim=-1, keepdim=True).clamp(min=1e-12)).to(qc.dtype)
    else:
        scores = torch.matmul(qc, kc.transpose(-2, -1)) / (D ** 0.5)
        probs = weighted_softmax(scores, k_sizes.unsqueeze(-2).float())
    sp, si = torch.sort(probs, dim=-1, descending=True, stable=True)
    cum = torch.cumsum(sp, dim=-1)
    rm = cum > p
    rm[..., 1:] = rm[..., :-1].clone()
    rm[..., 0] = False
    if min_kc_ratio > 0:
        rm[..., : int(min_kc_ratio * KC)] = False
    out = torch.zeros(B, H, QC, KC, dtype=torch.bool)
    out.scatter_(-1, si, ~rm)
    return out


def block_mask_to_element_mask(block_map, row_sz, col_sz):
    """ref: svg/kernels/test/test_sparse_attn_dyn_blk_wan.py:47-58"""
    r = torch.repeat_interleave(block_map, row_sz.long(), dim=0)
    return torch.repeat_interleave(r, col_sz.long(), dim=1)


def dynamic_block_sparse_fwd(q, k, v, dynamic_map, qc_size, kc_size, scale=None):
    """ref: dynamic_block_sparse_fwd_torch svg/kmeans_utils.py:902-995 (== dense attention under the expanded block
    mask, which is how the reference's GPU test checks flashinfer).  q [B,H,Sq,D], k/v [B,H,Skv,D]."""
    B, H, S, D = q.shape
    out = torch.zeros(B, H, S, D, dtype=torch.float32)
    for b in range(B):
        for h in range(H):
            em = block_mask_to_element_mask(dynamic_map[b, h], qc_size[b, h], kc_size[b, h])
            out[b, h] = masked_attention(q[b, h], k[b, h], v[b, h], em, scale)
    return out


def density_calculation(dynamic_map, q_sizes, k_sizes):
    """ref: svg/kmeans_utils.py:13-31"""
    blk = q_sizes[:, :, :, None] * k_sizes[:, :, None, :]
    return (blk * dynamic_map).sum(dim=(2, 3)) / blk.sum(dim=(2, 3))


def dynamic_map_post_processing(dyn_map, qc_sz, kc_sz, q_sorted_indices, video_length, context_length, prompt_length):
    """ref: Hunyuan_SAPAttn_Processor2_0.dynamic_map_post_processing svg/models/hyvideo/attention.py:657-702
    (the map / sizes / indices part; the q,k,v write-back is a plain slice copy)."""
    dyn_map = F.pad(dyn_map, (0, 2, 0, 2), value=0)
    dyn_map[:, :, -2, :-1] = True
    dyn_map[:, :, :-1, -2] = True
    dyn_map[:, :, -1, -1] = True
    unprompt = context_length - prompt_length
    qc_sz = F.pad(qc_sz, (0, 2), value=0)
    qc_sz[:, :, -2] = prompt_length
    qc_sz[:, :, -1] = unprompt
    kc_sz = F.pad(kc_sz, (0, 2), value=0)
    kc_sz[:, :, -2] = prompt_length
    kc_sz[:, :, -1] = unprompt
    qsi = F.pad(q_sorted_indices, (0, context_length), value=0)
    qsi[:, video_length:] = torch.arange(video_length, video_length + context_length)
    return dyn_map, qc_sz, kc_sz, qsi


# ----------------------------------------------------------------------------------------------------------
# online profiler
# ----------------------------------------------------------------------------------------------------------


def _blocked_band(n: int, thres_blocks: float) -> torch.Tensor:
    nb = math.ceil(n / 128)
    i = torch.arange(nb)
    blk = (i[:, None] - i[None, :]).abs() < thres_blocks
    return blk.repeat_interleave(128, 0).repeat_interleave(128, 1)[:n, :n]


def profile_masks(model: str, context_length: int, num_frame: int, frame_size: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Full [S,S] float profiling masks (spatial, temporal).
    ref: get_attention_mask hyvideo/utils.py:47-93 (1.5 frames, text rows/cols ones), wan/utils.py:63-110 (2 frames,
    first-frame sink set BEFORE the token-major permutation, ctx == 0), cog/utils.py:61-88 (text first; the spatial
    band is applied to the un-offset index range [0, ceil(V/128)*128), the temporal mask has no text rows/cols)."""
    V = num_frame * frame_size
    S = V + context_length

    def temporal(pix):
        return pix.reshape(frame_size, num_frame, frame_size, num_frame).permute(1, 0, 3, 2).reshape(V, V)

    if model == "hy":
        band = _blocked_band(V, (frame_size * 1.5) // 128)
        out = []
        for pix in (band, temporal(band)):
            m = torch.zeros(S, S)
            m[:V, :V] = pix.float()
            m[V:, :] = 1
            m[:, V:] = 1
            out.append(m)
        return out[0], out[1]
    if model == "wan":
        assert context_length == 0
        band = _blocked_band(V, (frame_size * 2) // 128)
        pix = band.clone()
        pix[:, :frame_size] = True
        return pix.float(), temporal(pix).float()
    if model == "cog":
        c = context_length
        sp = torch.zeros(S, S)
        sp[:c, :] = 1
        sp[:, :c] = 1
        band = _blocked_band(V, (frame_size * 1.5) // 128)
        nb128 = min(S, math.ceil(V / 128) * 128)
        big = _blocked_band(nb128, (frame_size * 1.5) // 128).float()
        sp[:nb128, :nb128] = torch.maximum(sp[:nb128, :nb128], big)
        tp = torch.zeros(S, S)
        tp[c:, c:] = temporal(band).float()
        return sp, tp
    raise ValueError(model)


def profile_mask_rows(model: str, context_length: int, num_frame: int, frame_size: int, rows) -> Tuple[torch.Tensor, torch.Tensor]:
    """Rows `rows` of the two masks of `profile_masks`, each [len(rows), S] float, without the [S, S] tensors (at HunyuanVideo 720p one
    of them is 56 GB).  Same references: every entry is the blocked band `|i // 128 - j // 128| < thres` evaluated at the coordinates the
    permutation `reshape(P, F, P, F).permute(1, 0, 3, 2)` assigns to (row, column) — temporal[f P + p, f' P + p'] = band[p F + f, p' F + f']
    — plus the models' all-ones text rows / columns and sink columns.  tests/test_oracle_golden.py holds it to `profile_masks` entry for
    entry at small sizes (and through it to the reference's own get_attention_mask outputs in tests/golden)."""
    F_, P_ = num_frame, frame_size
    V = F_ * P_
    S = V + context_length
    r = torch.as_tensor(rows, dtype=torch.long)[:, None]
    c = torch.arange(S)[None, :]

    def tm(x):      # token-major coordinate of a frame-major video index
        return (x % P_) * F_ + x // P_

    def blk_band(i, j, thres):
        return ((i // 128) - (j // 128)).abs() < thres

    if model == "hy":
        th = (P_ * 1.5) // 128
        vid = (r < V) & (c < V)
        sp = torch.where(vid, blk_band(r, c, th), torch.ones(1, dtype=torch.bool))
        tp = torch.where(vid, blk_band(tm(r), tm(c), th), torch.ones(1, dtype=torch.bool))
        return sp.float(), tp.float()
    if model == "wan":
        assert context_length == 0
        th = (P_ * 2) // 128
        sp = blk_band(r, c, th) | (c < P_)
        tp = blk_band(tm(r), tm(c), th) | (tm(c) < P_)
        return sp.float(), tp.float()
    if model == "cog":
        ctx = context_length
        th = (P_ * 1.5) // 128
        nb128 = min(S, math.ceil(V / 128) * 128)
        sp = (r < ctx) | (c < ctx) | ((r < nb128) & (c < nb128) & blk_band(r, c, th))
        rv, cv = (r - ctx).clamp(min=0), (c - ctx).clamp(min=0)
        tp = (r >= ctx) & (c >= ctx) & blk_band(tm(rv), tm(cv), th)
        return sp.float(), tp.float()
    raise ValueError(model)


def sample_mse(q, k, v, sampled_rows, masks, rows_gathered: bool = False):
    """ref: sample_mse svg/models/hyvideo/attention.py:376-399, computed in the input dtype exactly like the reference
    (bf16 matmul outputs, bf16 softmax outputs).  masks: two [rows_available, S] float masks.  -> [2, cfg, H]"""
    cfg, H, S, D = q.shape
    sq = q[:, :, sampled_rows, :]
    scores = torch.matmul(sq, k.transpose(-2, -1)) / (D ** 0.5)
    golden = torch.matmul(F.softmax(scores, dim=-1), v)
    out = torch.zeros(len(masks), cfg, H, dtype=q.dtype)
    for i, m in enumerate(masks):
        sm = m if rows_gathered else m[sampled_rows, :]      # rows_gathered: masks are already [len(sampled_rows), S] (profile_mask_rows)
        w = F.softmax(scores.masked_fill(sm == 0, float("-inf")), dim=-1)
        hs = torch.matmul(w, v)
        out[i] = torch.mean((hs - golden) ** 2, dim=(2, 3))
    return out


def sample_mse_fp32(q, k, v, sampled_rows, masks, rows_gathered: bool = False):
    """Same quantity in fp32 throughout (what the HIP profiler computes with emulate_bf16 = 0)."""
    return sample_mse(q.float(), k.float(), v.float(), sampled_rows, masks, rows_gathered)


# =====================================================================================================================
# Pre-attention prologue: QK normalisation + rotary embedding.
# The reference's CUDA sources (svg/kernels/csrc/include/{norm,rope}/*.cuh) are not part of the checkout; the semantics are
# the torch references its own tests compare the kernels with.  Pinned: tests/golden/prologue_golden.npz is produced by
# executing those reference functions themselves (tests/golden/make_golden_prologue.py).
# =====================================================================================================================
def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """ref: replica_host_rms_norm, svg/kernels/test/test_rms_norm.py:31-36 (== diffusers RMSNorm.forward): fp32 variance,
    the normalised value is rounded to the input dtype before it is multiplied by the weight."""
    dt = x.dtype
    xf = x.to(torch.float32)
    var = xf.pow(2).mean(-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return weight * xf.to(dt)


def layer_norm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """ref: ref_host_layer_norm, svg/kernels/test/test_layer_norm.py:24-29 — fp32 statistics, one rounding at the end
    (written out instead of F.layer_norm so that the rounding points are explicit)."""
    xf = x.to(torch.float32)
    mean = xf.mean(-1, keepdim=True)
    var = (xf - mean).pow(2).mean(-1, keepdim=True)
    y = (xf - mean) * torch.rsqrt(var + eps) * weight.to(torch.float32) + bias.to(torch.float32)
    return y.to(x.dtype)


def rope_cossin(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """ref: ref_host_apply_rope, svg/kernels/test/test_apply_rope.py:24-37 (diffusers apply_rotary_emb, use_real_unbind_dim=-1)"""
    xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    x_rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return (x.float() * cos + x_rot.float() * sin).to(x.dtype)


def rope_complex(x: torch.Tensor, freqs_real: torch.Tensor, freqs_imag: torch.Tensor) -> torch.Tensor:
    """ref: ref_host_apply_rope_complex, svg/kernels/test/test_apply_rope_complex.py:25-36 — complex128 x complex64"""
    freqs = torch.complex(freqs_real, freqs_imag)
    xc = torch.view_as_complex(x.to(torch.float64).unflatten(3, (-1, 2)))
    return torch.view_as_real(xc * freqs.unsqueeze(0).unsqueeze(0)).flatten(3, 4).type_as(x)


def apply_qk_rope(q, k, a, b, len_text_prompt: int, kind: str):
    """kind: 'cossin' (first len_text_prompt positions skipped, ops.h:80-136), 'txtlast' (last ones skipped, :138-196),
    'complex' (first ones skipped, fp64 complex multiply, :198-260).  Returns new tensors."""
    S = q.shape[2]
    L = int(len_text_prompt)
    lo, hi = (0, S - L) if kind == "txtlast" else (L, S)
    out = []
    for x in (q, k):
        y = x.clone()
        seg = x[:, :, lo:hi]
        y[:, :, lo:hi] = rope_complex(seg, a, b) if kind == "complex" else rope_cossin(seg, a, b)
        out.append(y)
    return out


# =====================================================================================================================
# Transformer-block glue of the Wan blocks.  The semantics are the torch fall-back branches of the
# reference's block forward (svg/models/wan/custom_models.py:44-108), which its Triton kernels replace.
# =====================================================================================================================
def fp32_layernorm(x: torch.Tensor, weight=None, bias=None, eps: float = 1e-5) -> torch.Tensor:
    """ref: `self.norm1(hidden_states.float())` with diffusers FP32LayerNorm, custom_models.py:44-47 — fp32 in, fp32 out.
    NOT what the reference's Triton kernels compute when the hidden size is not a power of two: they load the row zero-padded to
    N2 = next_power_of_2(N) and the padding enters the variance, var' = var + (N2 - N) / N * mean^2 (svg/kernels/triton/layernorm.py:35-41,
    :134-140; shown by executing them, tests/test_triton_golden.py::test_glue_kernels).  The two agree when the row mean is 0 — the
    reference's own test feeds randn — and differ by (N2 - N) / (2 N) * mean^2 / var relative otherwise (Wan 14B, N = 5120: 0.3 mean^2 / var)."""
    w = weight.float() if weight is not None else None
    b = bias.float() if bias is not None else None
    return F.layer_norm(x.float(), (x.shape[-1],), w, b, eps)


def modulate_shift(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, out_dtype) -> torch.Tensor:
    """ref: `(norm_hidden_states * (1 + scale_msa) + shift_msa).type_as(hidden_states)`, custom_models.py:58"""
    return (x.float() * (1 + scale.float()) + shift.float()).to(out_dtype)


def modulate_gate_residual(residual: torch.Tensor, x: torch.Tensor, gate: torch.Tensor, out_dtype) -> torch.Tensor:
    """ref: `(hidden_states.float() + attn_output * gate_msa).type_as(hidden_states)`, custom_models.py:69"""
    return (residual.float() + x.float() * gate.float()).to(out_dtype)


# ----------------------------------------------------------------------------------------------------------
# remainder packing of the variable-block kernel (no reference counterpart: an execution detail of this engine whose RESULT —
# index work — has to be reproducible; restated here so that the device-side matching can be checked for equality)
# ----------------------------------------------------------------------------------------------------------


def varblock_pair_partners(block_map: torch.Tensor, q_sizes: torch.Tensor, k_sizes: torch.Tensor, tile_rows: int = 256,
                           rounds: int = 3, min_common: int = 8) -> torch.Tensor:
    """block_map bool [H, QB, KB], q_sizes [H, QB], k_sizes [H, KB] -> int32 [H, QB]: the partner array of
    csrc/attention.hip varblock_pair_score_kernel / varblock_pair_match_kernel.  Per head and round every block-row with an unmatched
    ragged last tile (q_size % tile_rows > 0) chooses, among the other unmatched ones whose remainder fits beside its own, the one
    with the most active non-empty key blocks in common (at least `min_common`; ties: the lowest index); mutual choices become pairs
    (partner[i] = j for the lower index i, -2 for j); -1: alone."""
    H, QB, KB = block_map.shape
    out = torch.full((H, QB), -1, dtype=torch.int32)
    for h in range(H):
        bits = (block_map[h].bool() & (k_sizes[h] > 0)[None]).to(torch.float32)
        common = (bits @ bits.T).to(torch.int64)                 # [QB, QB] exact counts
        rem = (q_sizes[h].to(torch.int64) % tile_rows).clone()
        for _ in range(rounds):
            best = torch.full((QB,), -1, dtype=torch.int64)
            for i in range(QB):
                if rem[i] <= 0:
                    continue
                ok = (rem > 0) & (rem + rem[i] <= tile_rows)
                ok[i] = False
                sc = torch.where(ok, common[i], torch.full_like(common[i], -1))
                j = int(torch.argmax(sc))                        # first maximum = lowest index
                if sc[j] >= min_common:
                    best[i] = j
            for i in range(QB):
                j = int(best[i])
                if j >= 0 and int(best[j]) == i:
                    rem[i] = 0
                    out[h, i] = j if i < j else -2
    return out
