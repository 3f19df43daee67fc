"""Shared implementation of the layout-transformation operators (one HIP kernel, svg_head_placement) behind the three
model-specific modules svg/models/{hyvideo,wan,cog}/placement.py, which keep the reference's function names."""
from __future__ import annotations

import torch

from .. import _native


def _check(q, outs, best_mask_idx, context_length, num_frame, frame_size):
    cfg, num_heads, seq_len, head_dim = q.shape
    assert seq_len == context_length + num_frame * frame_size, (
        f"Query Shape: {seq_len} is not equivalent to {context_length} + {num_frame} * {frame_size}")
    assert best_mask_idx.shape == (cfg, num_heads)


def sparse_head_placement(query, key, value, query_out, key_out, value_out, best_mask_idx, context_length, num_frame,
                          frame_size, text_first):
    """Q/K/V rows of temporal heads (best_mask_idx == 1): frame-major f*P+p -> token-major p*F+f; text rows and spatial
    heads copied.  One launch for the three tensors."""
    _check(query, None, best_mask_idx, context_length, num_frame, frame_size)
    _native.head_placement([query, key, value], [query_out, key_out, value_out], best_mask_idx, context_length, num_frame,
                           frame_size, text_first, inverse=False)


def hidden_states_placement(hidden_states, hidden_states_out, best_mask_idx, context_length, num_frame, frame_size,
                            text_first):
    """Inverse transformation on the attention output."""
    _check(hidden_states, None, best_mask_idx, context_length, num_frame, frame_size)
    _native.head_placement([hidden_states], [hidden_states_out], best_mask_idx, context_length, num_frame, frame_size,
                           text_first, inverse=True)
    return hidden_states_out


def _video_range(seq_len, context_length, text_first):
    return (context_length, seq_len) if text_first else (0, seq_len - context_length)


def torch_placement(x: torch.Tensor, best_mask_idx, context_length, num_frame, frame_size, text_first, inverse):
    """Plain-torch statement of the same transformation (any device) — the `ref_*` functions of the reference modules.
    Unlike the reference helper it is also correct for context_length == 0."""
    cfg, H, S, D = x.shape
    lo, hi = _video_range(S, context_length, text_first)
    vid = x[:, :, lo:hi]
    a, b = (num_frame, frame_size) if not inverse else (frame_size, num_frame)
    perm = vid.reshape(cfg, H, a, b, D).transpose(2, 3).reshape(cfg, H, hi - lo, D)
    out = x.clone()
    out[:, :, lo:hi] = torch.where(best_mask_idx.to(torch.bool)[:, :, None, None], perm, vid)
    return out


def token_reorder(tensor: torch.Tensor, fix_len: int, reorder_len: int, reorder_num_frame: int, frame_size: int, text_first: bool,
                  to_token_major: bool) -> torch.Tensor:
    """The reference's `*_token_reorder_to_token_major` / `*_token_reorder_to_frame_major` helpers (hyvideo/placement.py:6-31,
    cog/placement.py:6-31): IN PLACE on the video part of EVERY head of `tensor` [B, H, fix_len + reorder_len, D] (text last, or first
    for CogVideoX), returns the tensor.  Correct for fix_len == 0 too (the reference's text-last helpers slice `[:-0]` there)."""
    assert reorder_len == reorder_num_frame * frame_size
    assert tensor.shape[2] == fix_len + reorder_len
    lo, hi = _video_range(tensor.shape[2], fix_len, text_first)
    a, b = (reorder_num_frame, frame_size) if to_token_major else (frame_size, reorder_num_frame)
    vid = tensor[:, :, lo:hi]
    tensor[:, :, lo:hi] = vid.reshape(tensor.shape[0], tensor.shape[1], a, b, tensor.shape[3]).transpose(2, 3).reshape(vid.shape)
    return tensor
