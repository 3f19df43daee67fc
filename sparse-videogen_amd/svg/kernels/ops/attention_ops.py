"""Uniform-block (BSR) sparse attention — same module path, names and call protocol as the reference's alternative backend
(svg/kernels/ops/attention_ops.py; the Wan variant attention_ops_wan.py has the same interface), on libsvgattn instead of
flashinfer: the BSR pattern is expanded to a block map with the text block in front and runs on the variable-block kernel, so
the reference's three flashinfer calls + merge_state (video x video BSR, video x text, text x all) are one launch.

q, k, v: [seq_len, num_heads, head_dim] with the text tokens FIRST (the reference's layout, attention_ops.py:139-150)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch

from ... import _native


def _to_bsr(attn_mask: np.ndarray):
    counts = (attn_mask != -1).sum(axis=1)
    indptr = np.concatenate(([0], np.cumsum(counts))).astype(np.int32)
    cols = attn_mask[attn_mask != -1]
    cols = np.concatenate((cols, [0] * 256))   # the reference pads the column indices (attention_ops.py:49-51)
    return indptr, cols.astype(np.int32)


def _device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


def _gen_temporal_mask(num_frames: int, num_tokens_per_frame: int, multiplier: float):
    """ref: attention_ops.py:9-56 — reordered sliding window on blocks of num_tokens_per_frame / 10 tokens:
    block (i, j) active iff |centre_i - centre_j| < multiplier * num_tokens_per_frame."""
    assert num_tokens_per_frame % 10 == 0
    bs = num_tokens_per_frame // 10
    nb = num_frames * num_tokens_per_frame // bs
    centre = np.arange(nb) * bs + bs // 2
    active = np.abs(centre[:, None] - centre[None, :]) < multiplier * num_tokens_per_frame
    attn_mask = np.where(active, np.arange(nb)[None, :], -1)
    indptr, cols = _to_bsr(attn_mask)
    dev = _device()
    return torch.from_numpy(indptr).to(dev), torch.from_numpy(cols).to(dev), (bs, bs)


def _gen_spatial_mask(num_frames: int, num_tokens_per_frame: int, multiplier: int):
    """ref: attention_ops.py:59-105 — frame-sized blocks, |frame_i - frame_j| <= multiplier or j == 0 (attention sink)"""
    assert multiplier >= 0 and num_frames > 0
    f = np.arange(num_frames)
    active = (np.abs(f[:, None] - f[None, :]) <= multiplier) | (f[None, :] == 0)
    attn_mask = np.where(active, f[None, :], -1)
    indptr, cols = _to_bsr(attn_mask)
    dev = _device()
    return torch.from_numpy(indptr).to(dev), torch.from_numpy(cols).to(dev), (num_tokens_per_frame, num_tokens_per_frame)


@dataclass
class FAMetadata:
    len_text_promt: int   # (sic — the reference's field name)
    num_frames: int
    num_tokens_per_frame: int
    temporal_mask_metadata: Optional[Tuple[torch.Tensor, torch.Tensor, Tuple[int, int]]]
    spatial_mask_metadata: Optional[Tuple[torch.Tensor, torch.Tensor, Tuple[int, int]]]
    workspace: Optional[torch.Tensor] = None   # unused here (flashinfer's 128 MiB workspace in the reference)


def init_sparse_attn(len_text_prompt: int, num_frames: int, num_tokens_per_frame: int, temporal_multiplier: float,
                     spatial_multiplier: int) -> FAMetadata:
    """ref: attention_ops.py:117-136"""
    return FAMetadata(len_text_prompt, num_frames, num_tokens_per_frame,
                      _gen_temporal_mask(num_frames, num_tokens_per_frame, temporal_multiplier),
                      _gen_spatial_mask(num_frames, num_tokens_per_frame, spatial_multiplier), None)


def sparse_attn_forward(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, metadata: FAMetadata,
                        sparse_pattern: str = "temporal") -> torch.Tensor:
    """ref: attention_ops.py:139-197"""
    assert sparse_pattern in ["temporal", "spatial"]
    indptr, indices, (R, Cb) = metadata.temporal_mask_metadata if sparse_pattern == "temporal" else metadata.spatial_mask_metadata
    L = metadata.len_text_promt
    S, Hq, D = q.shape
    Hkv = k.shape[1]
    video = S - L
    assert video % R == 0 and video % Cb == 0 and k.shape[0] == S
    MB, NB = video // R, video // Cb
    bm, qs, ks = _native.bsr_to_block_map(indptr, indices, MB, NB, R, Cb, L, Hkv)
    qh, kh, vh = (x.permute(1, 0, 2).contiguous() for x in (q, k, v))
    o = _native.varblock_attention(qh, kh, vh, bm, qs, ks)
    return o.permute(1, 0, 2).contiguous()
