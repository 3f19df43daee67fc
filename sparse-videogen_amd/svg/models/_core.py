"""Model-independent attention core of the SVG1 / SVG2 processors.

The three model packages (hyvideo, wan, cog) keep the reference's processor classes and call into these functions from
their `attention_core_logic`.  On GPU tensors everything below is libsvgattn (HIP); on CPU tensors only the DENSE branch
exists (torch SDPA — the reference's own CPU-capable path, e.g. svg/models/wan/attention.py:279-281) and the sparse
branch raises: there is no CPU fallback for the sparse hot path.
"""
from __future__ import annotations

import json
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn.functional as F

from .. import _native
from .. import distributed as _dist
from ..kmeans_utils import batch_kmeans_Euclid, density_calculation, identify_dynamic_map
from ..timer import time_logging_decorator


# ---------------------------------------------------------------------------------------------------------------
# precision of the sparse attention kernels: "bf16" (the input dtype: bf16 / fp16 MFMA, default) or "fp8" (e4m3 QK^T / PV,
# svg_band_attention_fp8 / svg_varblock_attention_fp8; head_dim 128 only — other head sizes stay on the 16-bit kernels).
# No reference counterpart (the reference lists fp8 attention as future work, README.md:117): an opt-in of this package.
# ---------------------------------------------------------------------------------------------------------------
_ATTENTION_DTYPE = {"value": "bf16"}


def set_attention_dtype(name: str) -> None:
    assert name in ("bf16", "fp8"), name
    _ATTENTION_DTYPE["value"] = name


def attention_dtype() -> str:
    return _ATTENTION_DTYPE["value"]


def _use_fp8(q: torch.Tensor) -> bool:
    return _ATTENTION_DTYPE["value"] == "fp8" and q.shape[-1] == 128


@dataclass
class Geometry:
    """Token layout of one model: [text?][video F*P][text?]"""

    context_length: int
    num_frame: int
    frame_size: int
    text_first: bool = False

    @property
    def video_length(self) -> int:
        return self.num_frame * self.frame_size

    @property
    def vid0(self) -> int:
        return self.context_length if self.text_first else 0

    @property
    def seq_len(self) -> int:
        return self.context_length + self.video_length


def is_full_attention(layer_idx: int, timestep, first_layers_fp, first_times_fp) -> bool:
    """ref: svg/models/hyvideo/attention.py:491-496 — dense for the first layers and the first (large) timesteps."""
    if layer_idx < first_layers_fp:
        return True
    from .context import timestep_value

    return timestep_value(timestep) > first_times_fp   # read back once per transformer forward, not once per layer



LN2 = 0.6931471805599453   # sm_scale of a kernel that applies sm_scale * log2(e) itself to a q that already carries the softmax scale

# Attention straight out of / into the projection layout (include/svg_attn.h, svg_attn_layout_t): q / k / v views that are not
# contiguous (a `proj(x).unflatten(2, (H, -1)).transpose(1, 2)` the caller did not copy) are read in place, and the result comes back as a
# [cfg, H, S, D] tensor STORED token-major, so that the processors' `hidden_states.transpose(1, 2).flatten(2, 3)` (ref:
# wan/attention.py:168-170, hyvideo/attention.py:202) is a view instead of a 2 S H D-byte copy.  The default schedules, one GPU (the head-sharded
# path gathers contiguous head slices); anything else falls back to copies inside svg/_native.py.  False: every output is contiguous.
TOKEN_MAJOR_IO = True


@time_logging_decorator("Level 3 - Dense Flash Attention")
def dense_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, valid_len: Optional[int] = None,
                    q_prescaled: bool = False) -> torch.Tensor:
    """Dense attention [cfg, H, S, D].  valid_len < S: two independent segments [0, valid) and [valid, S) — what the
    reference gets from flash_attn_varlen_func with cu_seqlens [0, valid, S] (hyvideo/attention.py:452-470).
    q_prescaled: q carries sm_scale * log2(e) (see qkv_from_projections(q_scale=...)); GPU only."""
    S = q.shape[2]
    if q.is_cuda:
        real = S if valid_len is None else int(valid_len)
        mask = _native.BandMask(real_len=real, band=S + 1, colfull_lo=0, colfull_hi=0, rowfull_lo=0, rowfull_hi=0)
        return _native.band_attention(q, k, v, mask, q_prescaled=q_prescaled, token_major_out=TOKEN_MAJOR_IO and not _dist.active())
    assert not q_prescaled, "a pre-scaled q only exists on the GPU path"
    if valid_len is None or valid_len >= S:
        return F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, is_causal=False)
    vl = int(valid_len)
    o = torch.empty_like(q)
    o[:, :, :vl] = F.scaled_dot_product_attention(q[:, :, :vl], k[:, :, :vl], v[:, :, :vl])
    o[:, :, vl:] = F.scaled_dot_product_attention(q[:, :, vl:], k[:, :, vl:], v[:, :, vl:])
    return o


def _require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what}: the sparse path runs only on the GPU through libsvgattn (no CPU fallback); "
                           f"CPU tensors can only take the dense branch")


@time_logging_decorator("Level 3 - sample_mse")
def sample_mse(q, k, v, geo: Geometry, prof: "_native.ProfileDesc", num_sampled_rows: int, sample_max_row: int, skip_flag=None,
               generator=None, q_prescaled: bool = False):
    """ref: sample_mse svg/models/hyvideo/attention.py:376-399.  Rows are drawn with the CPU generator exactly like the
    reference (`torch.randint(low=0, high=sample_mse_max_row, size=(n,))` without device=) unless `generator` is given.
    -> [2, cfg, H] in q's dtype: the reference stores the MSEs in `query.dtype` (:388) before the argmin, so two candidates that
    round to the same bf16 value tie and resolve to index 0 (spatial) there — the fp32 results of the kernel are rounded the same way."""
    cfg, H, S, D = q.shape
    n = min(num_sampled_rows, S)
    rows = torch.randint(low=0, high=sample_max_row, size=(n,), generator=generator)
    mses = _native.sample_mse(q, k, v, rows.to(q.device, non_blocking=True), prof, skip_flag=skip_flag,   # (4-D views: no reshape copy)
                              sm_scale=LN2 if q_prescaled else None)   # (the profiler multiplies its scale by log2(e) itself)
    return mses.reshape(2, cfg, H).to(q.dtype)


_SWITCH_GEN = None
_SWITCH_SEED = None


def reseed_switch_generator(seed: Optional[int] = None) -> None:
    """(Re)build the CPU generator of the device-switched path from `seed` (default: the global seed, torch.initial_seed()).
    Called by svg.utils.seed.seed_everything and by the replace_*_attention install hooks, so that seeding a second video in the
    same process resets the sampled profiler rows exactly like a fresh process.  Under svg.distributed.enable() the seed of rank
    0 is broadcast: every rank draws the same rows, whatever it was seeded with (the sharded result equals the single-GPU one)."""
    global _SWITCH_GEN, _SWITCH_SEED
    base = torch.initial_seed() if seed is None else int(seed)
    use = base
    if _dist.active():
        grp = _dist.current_group()
        on_gpu = torch.distributed.get_backend(grp) == "nccl"
        t = torch.tensor([base % (2 ** 63)], dtype=torch.int64, device="cuda" if on_gpu else "cpu")
        torch.distributed.broadcast(t, src=torch.distributed.get_global_rank(grp, 0) if grp is not None else 0, group=grp)
        use = int(t.item())
    _SWITCH_GEN = torch.Generator().manual_seed(use % (2 ** 63))
    _SWITCH_SEED = (base, _dist.active())


def _switch_generator():
    """CPU generator of the device-switched path.  The reference (and the host-side branch) draw the profiler's rows from the global
    CPU generator on SPARSE steps only; the switched path does not know on the host whether the step is dense, so it must not
    touch the global stream at all: it draws from this generator, derived from the global seed — re-derived whenever
    torch.initial_seed() changes (a later torch.manual_seed) or sharding is switched on, and by `reseed_switch_generator()`, which
    the seeding / install hooks call (re-seeding with the SAME seed cannot be seen from the seed alone).  (Deviation: at equal seeds
    the sampled rows differ from the reference's; every other consumer of the global CPU RNG sees the reference's stream.)"""
    if _SWITCH_GEN is None or _SWITCH_SEED != (torch.initial_seed(), _dist.active()):
        reseed_switch_generator()
    return _SWITCH_GEN


def dense_flag_on_device(timestep, first_times_fp):
    """int32 [1] on the timestep's device: 1 = this denoise step is a dense warm-up step (`timestep[0] > first_times_fp`,
    ref hyvideo/attention.py:495) — the comparison stays on the GPU; None when the timestep is not a GPU tensor."""
    if not (torch.is_tensor(timestep) and timestep.is_cuda):
        return None
    return (timestep.reshape(-1)[:1] > first_times_fp).to(torch.int32)


def svg1_attention_device_switch(q, k, v, geo: Geometry, mask: "_native.BandMask", dense_mask: "_native.BandMask",
                                 prof: "_native.ProfileDesc", num_sampled_rows: int, sample_max_row: int, dense_flag,
                                 _local: bool = False, q_prescaled: bool = False):
    """Dense warm-up step or sparse step, decided on the device: the profiler and the attention kernel read
    `dense_flag`; on a dense step the profiler returns at once and the kernel runs `dense_mask` without the head placement.
    Same attention results as the host-side branch of attention_core_logic (ref: hyvideo/attention.py:491-524) for the same sampled
    rows; the rows come from a dedicated CPU generator (`_switch_generator`).  The returned best_mask_idx is -1 on a dense step."""
    _require_gpu(q, "SVG1 attention")
    if _dist.active() and not _local:   # svg.distributed.enable(): this rank's heads only, outputs all-gathered
        return _dist.run_sharded(lambda qh, kh, vh: svg1_attention_device_switch(
            qh, kh, vh, geo, mask, dense_mask, prof, num_sampled_rows, sample_max_row, dense_flag, _local=True,
            q_prescaled=q_prescaled), (q, k, v), _dist.current_group())
    mses = sample_mse(q, k, v, geo, prof, num_sampled_rows, sample_max_row, skip_flag=dense_flag, generator=_switch_generator(),
                      q_prescaled=q_prescaled)
    best_mask_idx = torch.argmin(mses, dim=0)
    out = _native.band_attention_switch(q, k, v, mask, dense_mask, dense_flag, head_perm_flag=best_mask_idx, vid0=geo.vid0,
                                        num_frame=geo.num_frame, frame_size=geo.frame_size, q_prescaled=q_prescaled,
                                        token_major_out=TOKEN_MAJOR_IO and not _local)
    return out, torch.where(dense_flag.reshape(()) != 0, torch.full_like(best_mask_idx, -1), best_mask_idx)


def svg1_sparse_attention(q, k, v, geo: Geometry, mask: "_native.BandMask", prof: "_native.ProfileDesc",
                          num_sampled_rows: int, sample_max_row: int, fused: bool = True, _local: bool = False,
                          q_prescaled: bool = False):
    """The sparse branch of attention_core_logic (ref: hyvideo/attention.py:507-524):
    online profiling -> best_mask_idx -> placement -> block-sparse attention -> inverse placement.
    fused=True folds both placements into the attention kernel (bit-identical result, ~5.9 GB less HBM traffic at
    Hunyuan 720p); fused=False runs the three kernels of the reference pipeline."""
    _require_gpu(q, "SVG1 sparse attention")
    if _dist.active() and not _local:   # svg.distributed.enable(): this rank's heads only, outputs all-gathered
        return _dist.run_sharded(lambda qh, kh, vh: svg1_sparse_attention(
            qh, kh, vh, geo, mask, prof, num_sampled_rows, sample_max_row, fused, _local=True, q_prescaled=q_prescaled), (q, k, v),
            _dist.current_group())
    mses = sample_mse(q, k, v, geo, prof, num_sampled_rows, sample_max_row, q_prescaled=q_prescaled)
    best_mask_idx = torch.argmin(mses, dim=0)  # [cfg, H] int64; NaN wins like torch.argmin in the reference
    pk = dict(head_perm_flag=best_mask_idx, vid0=geo.vid0, num_frame=geo.num_frame, frame_size=geo.frame_size)
    if fused:
        with time_logging_decorator("Level 3 - sparse_flex_attention"):
            if _use_fp8(q):   # (the fp8 pre-pass folds whatever scale it is given into its quantisation of q)
                out = _native.band_attention_fp8(q.contiguous(), k.contiguous(), v.contiguous(), mask, sm_scale=LN2 if q_prescaled else None, **pk)
            else:
                out = _native.band_attention(q, k, v, mask, q_prescaled=q_prescaled, token_major_out=TOKEN_MAJOR_IO and not _local, **pk)
        return out, best_mask_idx
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    qo, ko, vo = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    with time_logging_decorator("Level 3 - fast_sparse_head_placement"):
        _native.head_placement([q, k, v], [qo, ko, vo], best_mask_idx, geo.context_length, geo.num_frame, geo.frame_size,
                               geo.text_first, inverse=False)
    with time_logging_decorator("Level 3 - sparse_flex_attention"):
        hs = _native.band_attention(qo, ko, vo, mask, q_prescaled=q_prescaled)
    out = torch.empty_like(hs)
    with time_logging_decorator("Level 3 - fast_hidden_states_placement"):
        _native.head_placement([hs], [out], best_mask_idx, geo.context_length, geo.num_frame, geo.frame_size, geo.text_first,
                               inverse=True)
    return out, best_mask_idx


# ---------------------------------------------------------------------------------------------------------------
# SVG2
# ---------------------------------------------------------------------------------------------------------------
class CentroidStore:
    """Per-layer k-means state (ref: class-level dicts of Hunyuan_SAPAttn_Processor2_0, hyvideo/attention.py:566-568).
    Unlike the reference it can be reset between videos (`clear()`)."""

    def __init__(self):
        self.q = {}
        self.k = {}

    def has(self, layer_idx):
        return layer_idx in self.q

    def clear(self):
        self.q.clear()
        self.k.clear()


KMEANS_TWO_STREAMS = True    # q-side and k-side k-means loops on two streams (kmeans_clustering below); False: one after the other
_SIDE_STREAMS = {}


def _side_stream(device) -> "torch.cuda.Stream":
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[key]


@time_logging_decorator("Level 3.5 - kmeans clustering")
def kmeans_clustering(store: CentroidStore, layer_idx: int, q_video, k_video, num_q_centroids, num_k_centroids, iter_init,
                      iter_step, head_shard=None):
    """ref: kmeans_init / kmeans_step / kmeans_clustering, hyvideo/attention.py:576-626: random init from the data on the
    first call of a layer, warm start from the previous denoise step's centroids afterwards.
    head_shard = (h0, h1, H): q_video / k_video hold heads [h0, h1) of H (svg.distributed) — the random initial points are drawn
    for all H heads, in the order of the unsharded call, and this rank keeps its rows; the stopping rule's maximum centre shift is
    all-reduced — so that the result does not depend on the sharding."""
    cfg, H, N, D = q_video.shape
    first = not store.has(layer_idx)
    iters = iter_init if first else iter_step
    qi = None if first else store.q[layer_idx]
    ki = None if first else store.k[layer_idx]
    red = _dist.all_reduce_max_ if head_shard is not None else None
    if first and head_shard is not None:
        h0, h1, H_all = head_shard
        rows = (torch.arange(cfg, device=q_video.device)[:, None] * H_all + torch.arange(h0, h1, device=q_video.device)[None]).reshape(-1)

        def draw(x, n_clusters):   # ref svg/kmeans_utils.py:706-709 for all cfg * H heads, then this rank's rows
            idx = torch.randint(0, N, (cfg * H_all, n_clusters), device=x.device).index_select(0, rows)
            return torch.gather(x.reshape(cfg * H, N, D), 1, idx[..., None].expand(-1, -1, D)).contiguous()

        qi, ki = draw(q_video, num_q_centroids), draw(k_video, num_k_centroids)
    # (check_every=0: the reference's stopping rule evaluated on the device — same result, no read-back per iteration)
    def run_q():
        return batch_kmeans_Euclid(q_video.reshape(cfg * H, N, D), num_q_centroids, max_iters=iters, init_centroids=qi,
                                   return_sorted_indices=True, check_every=0, shift_reduce=red)

    def run_k():
        return batch_kmeans_Euclid(k_video.reshape(cfg * H, N, D), num_k_centroids, max_iters=iters, init_centroids=ki,
                                   return_sorted_indices=True, check_every=0, shift_reduce=red)

    if KMEANS_TWO_STREAMS and red is None and q_video.is_cuda:
        # The two Lloyd loops are independent until the block map: the q side runs on a side stream beside the k side, so that the
        # HBM-bound update / sort / commit launches of one side share the chip with the MFMA-bound assignment of the other
        # (Wan 2.1 720p: 3.5 instead of 3.9 - 4.1 ms per layer-call, 86 instead of 99 ms for the 50-iteration init; bit-identical results —
        # profiles/r05d_svg2_two_streams.txt).  A sharded call keeps one stream: its stopping rule all-reduces on the current stream.
        cur = torch.cuda.current_stream(q_video.device)
        side = _side_stream(q_video.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            ql, qc, qs, qit, qidx = run_q()
        kl, kc, ks, kit, kidx = run_k()
        cur.wait_stream(side)
        for t_ in (ql, qc, qs, qit, qidx):   # allocated on the side stream's pool, consumed on the current stream from here on
            if isinstance(t_, torch.Tensor) and t_.is_cuda:
                t_.record_stream(cur)
    else:
        ql, qc, qs, qit, qidx = run_q()
        kl, kc, ks, kit, kidx = run_k()
    store.q[layer_idx] = qc
    store.k[layer_idx] = kc
    if first:
        print(f"Centroids initialized at layer {layer_idx}. Init step: {iter_init}")
    return (ql, qc, qs, qit, qidx), (kl, kc, ks, kit, kidx)


def svg2_sparse_attention(q, k, v, geo: Geometry, store: CentroidStore, layer_idx: int, num_q_centroids: int,
                          num_k_centroids: int, top_p: float, min_kc_ratio: float, iter_init: int, iter_step: int,
                          prompt_length: int = 0, logging_file: Optional[str] = None, timestep=None, _head_shard=None):
    """The sparse branch of the SAP processors (ref: hyvideo/attention.py:747-804, wan/attention.py:529-559):
    k-means on the video tokens -> top-p block map -> (Hunyuan) two pseudo clusters for prompt / unused prompt ->
    variable-block attention with the token permutation fused in (the result is already in the original order)."""
    _require_gpu(q, "SVG2 sparse attention")
    if _dist.active() and _head_shard is None:   # svg.distributed.enable(): this rank's heads only, outputs all-gathered
        grp = _dist.current_group()
        mine = _dist.shard_heads(q.shape[1], torch.distributed.get_rank(grp), torch.distributed.get_world_size(grp))
        shard = (mine[0], mine[-1] + 1, q.shape[1]) if mine else (0, 0, q.shape[1])
        return _dist.run_sharded(lambda qh, kh, vh: svg2_sparse_attention(
            qh, kh, vh, geo, store, layer_idx, num_q_centroids, num_k_centroids, top_p, min_kc_ratio, iter_init, iter_step,
            prompt_length, logging_file, timestep, _head_shard=shard), (q, k, v), grp)
    cfg, H, S, D = q.shape
    assert cfg == 1, "Batch size must be 1 for kmeans block sparse attention"
    assert not geo.text_first, "SVG2 is defined for text-last models (Hunyuan, Wan)"
    V, ctx = geo.video_length, geo.context_length
    q, k = q.contiguous(), k.contiguous()   # (the k-means reads them as [H, N, D]; v may stay a strided view: only the attention kernel reads it)
    # the video tokens as VIEWS when the loop runs inside the library (svg_kmeans_loop_strided reads heads S * D apart in place); the
    # head-sharded path (torch statement of the loop, all-reduced stopping rule) takes the copies the reference makes
    in_place = ctx and cfg == 1 and _head_shard is None and not _dist.active()
    qv = (q[:, :, :V] if in_place else q[:, :, :V].contiguous()) if ctx else q
    kv = (k[:, :, :V] if in_place else k[:, :, :V].contiguous()) if ctx else k
    with time_logging_decorator("Level 3 - semantic aware permutation"):
        (ql, qc, qs, _, qidx), (kl, kc, ks, _, kidx) = kmeans_clustering(store, layer_idx, qv, kv, num_q_centroids,
                                                                         num_k_centroids, iter_init, iter_step,
                                                                         head_shard=_head_shard)
        q_sizes = qs.view(cfg, H, num_q_centroids)
        k_sizes = ks.view(cfg, H, num_k_centroids)
        dyn_map = identify_dynamic_map(qc.view(cfg, H, num_q_centroids, D), kc.view(cfg, H, num_k_centroids, D), q_sizes,
                                       k_sizes, top_p, min_kc_ratio)
    if ctx:
        with time_logging_decorator("Level 3 - dynamic map post processing"):
            dyn_map, q_sizes, k_sizes, qidx, kidx = dynamic_map_post_processing(dyn_map, q_sizes, k_sizes, qidx, kidx, V, ctx,
                                                                                prompt_length)
    QB, KB = q_sizes.shape[-1], k_sizes.shape[-1]
    f8 = _use_fp8(q)
    # (rows_covered: the cluster sizes of every head add up to S — k-means counts over the V video tokens plus the text pseudo-clusters of
    #  dynamic_map_post_processing — so the kernel writes every output row and the wrapper skips the zero fill)
    out = _native.varblock_attention(q, k, v, dyn_map.view(H, QB, KB).contiguous(),
                                     q_sizes.view(H, QB).contiguous(), k_sizes.view(H, KB).contiguous(),
                                     q_row_idx=qidx.contiguous(), kv_row_idx=kidx.contiguous(), fp8=f8,
                                     token_major_out=TOKEN_MAJOR_IO and _head_shard is None, rows_covered=True)
    if logging_file is not None:
        from .context import timestep_value

        densities = density_calculation(dyn_map, q_sizes, k_sizes)
        if _head_shard is not None:   # one log line per layer-call with all heads, written by rank 0
            densities = _dist.all_gather_heads(densities.reshape(cfg, H, 1, 1), _head_shard[2], _dist.current_group()).reshape(cfg, -1)
        if _head_shard is None or torch.distributed.get_rank(_dist.current_group()) == 0:
            DENSITY_LOG.push(logging_file, {"timestep": timestep_value(timestep) if timestep is not None else None,
                                            "layer": layer_idx}, densities)
    return out.reshape(cfg, H, S, D)   # (a view in either storage order: cfg == 1)


class _DensityLog:
    """Density logging off the critical path.  The reference reads `densities.mean().item()` / `.tolist()` back in
    every sparse layer-call (hyvideo/attention.py:786-802): one device synchronisation per layer.  Here the per-head densities are
    copied to pinned host memory asynchronously and the JSON lines (same fields, same order) are written once their copy has
    completed — opportunistically on later calls, and in any case by `flush()` (registered with atexit; call it before reading
    the file)."""

    def __init__(self):
        self.pending = []   # (path, meta, pinned host tensor, event)

    def push(self, path: str, meta: dict, densities: torch.Tensor) -> None:
        host = torch.empty(densities.shape, dtype=torch.float32, pin_memory=densities.is_cuda)
        host.copy_(densities.float(), non_blocking=True)
        ev = None
        if densities.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
        self.pending.append((path, meta, host, ev))
        self.flush(wait=False)

    def flush(self, wait: bool = True) -> None:
        while self.pending:
            path, meta, host, ev = self.pending[0]
            if ev is not None:
                if wait:
                    ev.synchronize()
                elif not ev.query():
                    return
            entry = dict(meta, avg_density=float(host.mean()), density=host.tolist())
            with open(path, "a") as f:
                f.write(json.dumps(entry) + "\n")
            self.pending.pop(0)


DENSITY_LOG = _DensityLog()


def flush_density_log() -> None:
    """Write every pending density record (see _DensityLog)."""
    DENSITY_LOG.flush(wait=True)


import atexit  # noqa: E402

atexit.register(flush_density_log)


def dynamic_map_post_processing(dyn_map, qc_sz, kc_sz, q_sorted_indices, k_sorted_indices, video_length, context_length,
                                prompt_length):
    """ref: hyvideo/attention.py:657-702 — append the prompt / unused-prompt pseudo clusters.  The q,k,v write-back of
    the reference (permuted video tokens copied in front of the text tokens) is not needed: the attention kernel
    gathers rows through the (identity-padded) sorted indices."""
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
    tail = torch.arange(video_length, video_length + context_length, device=q_sorted_indices.device, dtype=torch.int32)
    tail = tail.expand(q_sorted_indices.shape[0], -1)
    q_sorted_indices = torch.cat([q_sorted_indices, tail], dim=1)
    k_sorted_indices = torch.cat([k_sorted_indices, tail], dim=1)
    return dyn_map, qc_sz, kc_sz, q_sorted_indices, k_sorted_indices


# ---------------------------------------------------------------------------------------------------------------------
# Pre-attention prologue on libsvgattn (the reference's `_kernels` fast path: hyvideo/attention.py:157-188,
# wan/attention.py:42-48, cog/attention.py:19-34).  Each helper returns False when the HIP path does not apply (CPU tensors,
# unsupported head_dim / dtype, a norm module it does not recognise) so that the caller runs the model's own torch modules —
# exactly the reference's `except ImportError` branch.
# ---------------------------------------------------------------------------------------------------------------------
_FAST_DIMS = (32, 64, 128, 256)


def _fast_ok(*ts) -> bool:
    if not all(t.is_cuda for t in ts):
        return False                      # CPU tensors: the model's own torch modules (the reference's CPU-capable path)
    _native.load()                        # GPU tensors without the library: raise, never fall back silently
    return all(t.is_cuda and t.is_contiguous() and t.dtype in (torch.bfloat16, torch.float16) and t.dim() == 4
               and t.shape[-1] in _FAST_DIMS for t in ts)


def _norm_desc(norm, D: int, dtype, device):
    """(kind, weight, bias, eps) of an RMSNorm / LayerNorm over head_dim, or None if the module is something else."""
    if norm is None:
        return None
    w = getattr(norm, "weight", None)
    if w is not None and tuple(w.shape) != (D,):
        return None
    eps = getattr(norm, "eps", None)
    if eps is None:
        return None
    b = getattr(norm, "bias", None)
    is_layer = isinstance(norm, torch.nn.LayerNorm) or b is not None
    cast = lambda t: None if t is None else t.detach().to(device=device, dtype=dtype).contiguous()  # noqa: E731
    return (2 if is_layer else 1), cast(w), cast(b), float(eps)


def qk_norm_inplace(norm_q, norm_k, query, key) -> bool:
    """In-place QK normalisation over head_dim (ref fast path: `_kernels.rms_norm_forward` / `layer_norm_forward`)."""
    if norm_q is None or norm_k is None or not _fast_ok(query, key):
        return False
    D = query.shape[-1]
    dq, dk = _norm_desc(norm_q, D, query.dtype, query.device), _norm_desc(norm_k, D, key.dtype, key.device)
    if dq is None or dk is None or dq[0] != dk[0] or dq[3] != dk[3]:
        return False
    if query.shape[0] == key.shape[0] and query.shape[2] == key.shape[2]:
        _native.qk_norm_rope(query, key, dq[0], dq[1], dq[2], dk[1], dk[2], dq[3])
    else:   # cross attention: different sequence lengths, one call per tensor
        _native.qk_norm_rope(query, None, dq[0], dq[1], dq[2], None, None, dq[3])
        _native.qk_norm_rope(key, None, dk[0], dk[1], dk[2], None, None, dk[3])
    return True


def _tables(a, b, rows: int, cols: int, device):
    a = a.detach().to(device=device, dtype=torch.float32).reshape(-1, a.shape[-1]).contiguous()
    b = b.detach().to(device=device, dtype=torch.float32).reshape(-1, b.shape[-1]).contiguous()
    return (a, b) if a.shape == (rows, cols) and b.shape == (rows, cols) else None


def qk_rope_inplace(query, key, cos, sin, rope_lo: int, rope_hi: int, complex_pairs: bool = False, norm_q=None,
                    norm_k=None, q_scale: float = 1.0) -> bool:
    """In-place rotary embedding of positions [rope_lo, rope_hi) — optionally fused with the QK normalisation (one pass over
    q and k instead of three).  cos / sin: [rope_hi - rope_lo, D] fp32 (complex_pairs: real / imag [.., D / 2]).
    q_scale != 1: folded into the pass's last rounding of q (the attention core then runs its pre-scaled kernels)."""
    if not _fast_ok(query, key) or rope_hi <= rope_lo:
        return False
    D = query.shape[-1]
    tb = _tables(cos, sin, rope_hi - rope_lo, D // 2 if complex_pairs else D, query.device)
    if tb is None:
        return False
    kind, qw, qb, kw, kb, eps = 0, None, None, None, None, 0.0
    if norm_q is not None or norm_k is not None:
        dq, dk = _norm_desc(norm_q, D, query.dtype, query.device), _norm_desc(norm_k, D, key.dtype, key.device)
        if dq is None or dk is None or dq[0] != dk[0] or dq[3] != dk[3]:
            return False
        kind, qw, qb, kw, kb, eps = dq[0], dq[1], dq[2], dk[1], dk[2], dq[3]
    _native.qk_norm_rope(query, key, kind, qw, qb, kw, kb, eps, 2 if complex_pairs else 1, tb[0], tb[1], rope_lo, rope_hi,
                         q_scale=q_scale)
    return True


def prescale_supported(query) -> bool:
    """the pre-scaled attention kernels exist for 16-bit GPU tensors with head_dim 64 / 128"""
    return bool(query.is_cuda and query.dtype in (torch.bfloat16, torch.float16) and query.shape[-1] in (64, 128))


def value_in_place(value: torch.Tensor, heads: int):
    """The v projection's output [bsz, S, heads * D] as the [bsz, heads, S, D] VIEW the attention kernels read in place
    (svg_attn_layout_t; TOKEN_MAJOR_IO, one GPU, head_dim 64 / 128, 16-bit, on the GPU) — or None: the caller makes the head-major copy."""
    if not (TOKEN_MAJOR_IO and value.is_cuda and value.dim() == 3 and value.is_contiguous() and not _dist.active()):
        return None
    if value.dtype not in (torch.bfloat16, torch.float16) or value.shape[-1] not in (heads * 64, heads * 128):
        return None
    return value.unflatten(2, (heads, -1)).transpose(1, 2)


def qkv_from_projections(query, key, value, heads: int, norm_q, norm_k, cos, sin, rope_lo: int, rope_hi: int,
                         complex_pairs: bool = False, q_scale: float = 1.0):
    """Projection outputs [bsz, S, heads * D] -> head-major q, k, v [bsz, heads, S, D] with QK-norm + RoPE applied to q, k in
    the same pass (svg_qk_norm_rope_transpose): replaces three transpose copies + norm + norm + rope.  None: not applicable."""
    ts = (query, key, value)
    if not all(t.is_cuda for t in ts):
        return None
    _native.load()
    if not all(t.dim() == 3 and t.is_contiguous() and t.dtype in (torch.bfloat16, torch.float16) for t in ts):
        return None
    D = query.shape[-1] // heads
    if D not in _FAST_DIMS or query.shape != key.shape or value.shape != query.shape:
        return None
    kind, qw, qb, kw, kb, eps = 0, None, None, None, None, 0.0
    if norm_q is not None or norm_k is not None:
        dq, dk = _norm_desc(norm_q, D, query.dtype, query.device), _norm_desc(norm_k, D, key.dtype, key.device)
        if dq is None or dk is None or dq[0] != dk[0] or dq[3] != dk[3]:
            return None
        kind, qw, qb, kw, kb, eps = dq[0], dq[1], dq[2], dk[1], dk[2], dq[3]
    rk, tb = 0, (None, None)
    if cos is not None:
        tb = _tables(cos, sin, rope_hi - rope_lo, D // 2 if complex_pairs else D, query.device)
        if tb is None:
            return None
        rk = 2 if complex_pairs else 1
    q, k = _native.qk_norm_rope_transpose(query, key, heads, heads, kind, qw, qb, kw, kb, eps, rk, tb[0], tb[1], rope_lo, rope_hi,
                                          q_scale=q_scale)   # q_scale != 1: q leaves the prologue carrying the softmax scale
    v = value_in_place(value, heads) if q_scale == 1.0 else None   # (a pre-scaled q takes the entry points without strides)
    if v is None:
        v, _ = _native.qk_norm_rope_transpose(value, None, heads, 0)
    return q, k, v
