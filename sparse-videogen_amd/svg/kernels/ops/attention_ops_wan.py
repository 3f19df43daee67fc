"""Uniform-block (BSR) sparse attention, Wan variant — same module path, names and call protocol as the reference
(svg/kernels/ops/attention_ops_wan.py: the video-only form, no text block; block size = the largest divisor of the frame size below
256), on libsvgattn instead of flashinfer's BlockSparseAttentionWrapper: the BSR pattern becomes a block map and runs on the
variable-block kernel.  q, k, v: [seq_len, num_heads, head_dim]."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch

from ... import _native
from .attention_ops import _device, _to_bsr


def get_factor(num_frames: int, num_tokens_per_frame: int) -> int:
    """ref: attention_ops_wan.py:12-25 — the largest divisor of num_tokens_per_frame that is below 256 (1 if there is none)"""
    n = int(num_tokens_per_frame)
    for f in range(min(n, 255), 0, -1):
        if n % f == 0:
            return f
    return 1


def ref_gen_temporal_mask(num_frames: int, num_tokens_per_frame: int, multiplier: float) -> np.ndarray:
    """ref: attention_ops_wan.py:96-130 — [row blocks, col blocks], entry = column block index where the block is active, -1 elsewhere:
    |centre of the row block - centre of the column block| < multiplier * num_tokens_per_frame (reordered sliding window)"""
    bs = get_factor(num_frames, num_tokens_per_frame)
    total = num_frames * num_tokens_per_frame
    assert total % bs == 0
    nb = total // bs
    centre = np.arange(nb) * bs + bs // 2
    active = np.abs(centre[:, None] - centre[None, :]) < multiplier * num_tokens_per_frame
    return np.where(active, np.arange(nb)[None, :], -1)


def gen_temporal_mask(num_frames: int, num_tokens_per_frame: int, multiplier: float) -> Tuple[torch.Tensor, torch.Tensor, Tuple[int, int]]:
    """ref: attention_ops_wan.py:48-93 — the same mask in BSR form (row pointer, padded column indices, block size); the reference also
    prints the sparsity and saves a figure, which is left out"""
    bs = get_factor(num_frames, num_tokens_per_frame)
    indptr, cols = _to_bsr(ref_gen_temporal_mask(num_frames, num_tokens_per_frame, multiplier))
    dev = _device()
    return torch.from_numpy(indptr).to(dev), torch.from_numpy(cols).to(dev), (bs, bs)


@dataclass
class WanFAMetadata:
    num_frames: int
    num_tokens_per_frame: int
    temporal_mask_metadata: Tuple[torch.Tensor, torch.Tensor, Tuple[int, int]]
    workspace: Optional[torch.Tensor] = None   # unused here (flashinfer's workspace in the reference)


def wan_sparse_attn_forward(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, metadata: WanFAMetadata) -> torch.Tensor:
    """ref: attention_ops_wan.py:141-184"""
    indptr, indices, (R, Cb) = metadata.temporal_mask_metadata
    assert q.shape[0] % R == 0, f"Query length {q.shape[0]} % block_size {R} != 0"
    assert k.shape[0] % Cb == 0, f"Key length {k.shape[0]} % block_size {Cb} != 0"
    assert k.shape[0] == v.shape[0], f"Key length {k.shape[0]} != Value length {v.shape[0]}"
    Hkv = k.shape[1]
    bm, qs, ks = _native.bsr_to_block_map(indptr, indices, q.shape[0] // R, k.shape[0] // Cb, R, Cb, 0, Hkv)
    qh, kh, vh = (x.permute(1, 0, 2).contiguous() for x in (q, k, v))
    return _native.varblock_attention(qh, kh, vh, bm, qs, ks).permute(1, 0, 2).contiguous()
