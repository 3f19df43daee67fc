"""CogVideoX helpers of the SVG1 path — same names as the reference module svg/models/cog/utils.py (text tokens FIRST)."""
from __future__ import annotations

import math
from math import floor

from ... import _native
from ...utils.seed import seed_everything  # noqa: F401  (the reference module re-exports it)
from ..hyvideo.utils import dense_mask, sparsity_to_width  # noqa: F401  (identical formula, ref cog/utils.py:49-58)


def generate_temporal_head_mask_mod(prompt_length: int = 226, num_frames: int = 13, token_per_frame: int = 1350,
                                    mul: float = 2, attn_sink: bool = False) -> _native.BandMask:
    """ref: svg/models/cog/utils.py:30-46: text rows and text columns dense, |q-k| < floor(mul*P/128)*128 elsewhere."""
    S = prompt_length + num_frames * token_per_frame
    band = floor(mul * token_per_frame / 128) * 128
    col_hi = prompt_length + (token_per_frame if attn_sink else 0)
    return _native.BandMask(real_len=S, band=band, colfull_lo=0, colfull_hi=col_hi, rowfull_lo=0,
                            rowfull_hi=prompt_length)


def get_attention_mask(mask_name, context_length, num_frame, frame_size):
    """ref: svg/models/cog/utils.py:61-88, quirks included: the spatial band is laid over the UN-offset index range
    [0, ceil(V/128)*128) and the temporal mask has no text rows / columns (a sampled text row then has no visible key
    and its MSE is NaN, which `argmin` picks — exactly what the reference does)."""
    V = num_frame * frame_size
    S = V + context_length
    bb = int((frame_size * 1.5) // 128)
    if mask_name == "spatial":
        return _native.ProfileVariant(0, 0, min(S, math.ceil(V / 128) * 128), bb, 0, 0, context_length)
    return _native.ProfileVariant(1, context_length, V, bb, 0, 0, 0)


def profile_desc(context_length, num_frame, frame_size, emulate_bf16=True) -> _native.ProfileDesc:
    d = _native.ProfileDesc(context_length, num_frame, frame_size, int(emulate_bf16))
    d.variant[0] = get_attention_mask("spatial", context_length, num_frame, frame_size)
    d.variant[1] = get_attention_mask("temporal", context_length, num_frame, frame_size)
    return d
