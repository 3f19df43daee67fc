"""Wan 2.1 helpers of the SVG1 path — same names as the reference module svg/models/wan/utils.py; masks are analytic
descriptors evaluated inside the HIP kernels (see hyvideo/utils.py)."""
from __future__ import annotations

import math
from math import ceil

from ... import _native
from ..hyvideo.utils import dense_mask, sparsity_to_width  # noqa: F401  (identical formula, ref wan/utils.py:51-60)


def generate_temporal_head_mask_mod(context_length: int = 226, prompt_length: int = 226, num_frames: int = 13,
                                    token_per_frame: int = 1350, mul: float = 2) -> _native.BandMask:
    """ref: svg/models/wan/utils.py:25-41: |q-k| <= ceil(mul*P/128)*128 or k in the first frame (attention sink)."""
    S = context_length + num_frames * token_per_frame
    band = ceil(mul * token_per_frame / 128) * 128 + 1  # '<=' of the reference == '<' band + 1
    return _native.BandMask(real_len=S, band=min(band, S + 1), colfull_lo=0, colfull_hi=token_per_frame, rowfull_lo=0,
                            rowfull_hi=0)


def generate_dense_mask_mod(seq_len: int) -> _native.BandMask:
    """ref: svg/models/wan/utils.py:44-48"""
    return dense_mask(seq_len)


def get_attention_mask(mask_name, sample_mse_max_row, context_length, num_frame, frame_size):
    """ref: svg/models/wan/utils.py:63-110: 2-frame 128-blocked band + first-frame sink (set in the mask's own
    coordinate order, i.e. before the token-major permutation for the temporal mask)."""
    assert context_length == 0, "Wan has no text tokens in the self-attention sequence"
    V = num_frame * frame_size
    bb = int((frame_size * 2) // 128)
    coord = 0 if mask_name == "spatial" else 1
    return _native.ProfileVariant(coord, 0, V, bb, frame_size, 0, 0)


def profile_desc(context_length, num_frame, frame_size, emulate_bf16=True) -> _native.ProfileDesc:
    d = _native.ProfileDesc(0, num_frame, frame_size, int(emulate_bf16))
    d.variant[0] = get_attention_mask("spatial", 0, context_length, num_frame, frame_size)
    d.variant[1] = get_attention_mask("temporal", 0, context_length, num_frame, frame_size)
    return d


# ---- the uniform-block (BSR) alternative backend of the reference's Wan processors (ref wan/utils.py:63-240; its generators live in
#      svg/kernels/ops/attention_ops_wan.py, which wan/utils.py duplicates) ----
from ...kernels.ops.attention_ops_wan import gen_temporal_mask, get_factor  # noqa: E402,F401


def flashinfer_sparse_attn_forward(q, k, v, temporal_mask_metadata):
    """ref: wan/utils.py:188-240 — q, k, v [cfg, H, S, D] -> [cfg, H, S, D] under the BSR temporal mask (name kept; the BSR pattern runs on
    the variable-block HIP kernel, see svg/kernels/ops/attention_ops_wan.py).  cfg and heads are folded into the head axis like the
    reference does."""
    from ...kernels.ops.attention_ops_wan import WanFAMetadata, wan_sparse_attn_forward

    cfg, num_heads, seq_len, head_dim = q.shape
    qs, ks, vs = (x.permute(2, 0, 1, 3).reshape(seq_len, cfg * num_heads, head_dim) for x in (q, k, v))
    o = wan_sparse_attn_forward(qs, ks, vs, WanFAMetadata(0, 0, temporal_mask_metadata, None))
    return o.reshape(seq_len, cfg, num_heads, head_dim).permute(1, 2, 0, 3).contiguous()
