"""HunyuanVideo helpers of the SVG1 path — same names as the reference module svg/models/hyvideo/utils.py.

The reference builds a torch flex_attention BlockMask from `generate_temporal_head_mask_mod` and two materialised
[10000, S] fp32 profiling masks (4.8 GB each) with `get_attention_mask`.  Here both are *descriptors*: six integers for
the attention mask (svg_band_mask_t) and an analytic predicate for the profiling masks (svg_profile_desc_t); the HIP
kernels evaluate them on the fly.
"""
from __future__ import annotations

import math
from math import floor

from ... import _native

try:   # the reference takes the template from diffusers (ref: svg/models/hyvideo/utils.py:8); optional here
    from diffusers.pipelines.hunyuan_video.pipeline_hunyuan_video import DEFAULT_PROMPT_TEMPLATE
except Exception:  # noqa: BLE001 — diffusers absent: the caller passes prompt_template (or a template with crop_start)
    DEFAULT_PROMPT_TEMPLATE = None


def sparsity_to_width(sparsity, context_length, num_frame, frame_size):
    """ref: svg/models/hyvideo/utils.py:142-151 — density target -> band half-width in frames."""
    seq_len = context_length + num_frame * frame_size
    total_elements = seq_len ** 2
    sparsity = (sparsity * total_elements - 2 * seq_len * context_length) / total_elements
    width = seq_len * (1 - math.sqrt(1 - sparsity))
    return width / frame_size


def generate_temporal_head_mask_mod(context_length: int = 226, prompt_length: int = 226, num_frames: int = 13,
                                    token_per_frame: int = 1350, mul: float = 2) -> _native.BandMask:
    """ref: svg/models/hyvideo/utils.py:20-44.  Returns the band-mask descriptor equal to the reference mask_mod:
    real = V + prompt_length; |q-k| < floor(mul*P/128)*128 inside the real part, prompt rows/cols dense, pad tokens
    attend only among themselves."""
    V = num_frames * token_per_frame
    real = V + prompt_length
    band = floor(mul * token_per_frame / 128) * 128
    return _native.BandMask(real_len=real, band=band, colfull_lo=V, colfull_hi=real, rowfull_lo=V, rowfull_hi=real)


def dense_mask(seq_len: int, valid_len: int | None = None) -> _native.BandMask:
    """Dense attention; valid_len < seq_len gives the two segments of cu_seqlens [0, valid, S]
    (ref: svg/models/hyvideo/attention.py:308-316,452-470)."""
    real = seq_len if valid_len is None else int(valid_len)
    return _native.BandMask(real_len=real, band=seq_len + 1, colfull_lo=0, colfull_hi=0, rowfull_lo=0, rowfull_hi=0)


def get_attention_mask(mask_name, sample_mse_max_row, context_length, num_frame, frame_size, device="cuda"):
    """ref: svg/models/hyvideo/utils.py:47-93.  Returns the analytic profiling-mask variant (spatial / temporal):
    128-blocked band of (1.5*P)//128 blocks over the video tokens, in frame-major or token-major order; text rows and
    columns all ones."""
    V = num_frame * frame_size
    bb = int((frame_size * 1.5) // 128)
    coord = 0 if mask_name == "spatial" else 1
    return _native.ProfileVariant(coord, 0, V, bb, 0, V, V + context_length)


def profile_desc(context_length, num_frame, frame_size, emulate_bf16=True) -> _native.ProfileDesc:
    d = _native.ProfileDesc(0, num_frame, frame_size, int(emulate_bf16))
    d.variant[0] = get_attention_mask("spatial", 0, context_length, num_frame, frame_size)
    d.variant[1] = get_attention_mask("temporal", 0, context_length, num_frame, frame_size)
    return d


def get_prompt_length(pipe, prompt, prompt_template=DEFAULT_PROMPT_TEMPLATE, max_sequence_length=256, device="cuda"):
    """ref: svg/models/hyvideo/utils.py:96-141 — number of real prompt tokens (prompt_length + pad = context_length = 256),
    needed before the masks are built.  Tokenises the templated prompt with the pipeline's tokenizer, drops the template
    prefix (`crop_start`) and sums the attention mask.  Host logic only (tokenizer call); returns a 0-dim tensor like the
    reference."""
    if prompt_template is None:
        raise RuntimeError("get_prompt_length: diffusers is not installed, pass prompt_template explicitly")
    prompt = [prompt] if isinstance(prompt, str) else prompt
    prompt = [prompt_template["template"].format(p) for p in prompt]
    crop_start = prompt_template.get("crop_start", None)
    if crop_start is None:
        tmpl = pipe.tokenizer(prompt_template["template"], padding="max_length", return_tensors="pt", return_length=False,
                              return_overflowing_tokens=False, return_attention_mask=False)
        crop_start = tmpl["input_ids"].shape[-1] - 2   # minus <|eot_id|> and the {} placeholder
    max_sequence_length += crop_start
    text_inputs = pipe.tokenizer(prompt, max_length=max_sequence_length, padding="max_length", truncation=True,
                                 return_tensors="pt", return_length=False, return_overflowing_tokens=False,
                                 return_attention_mask=True)
    prompt_attention_mask = text_inputs.attention_mask.to(device=device)
    if crop_start is not None and crop_start > 0:
        prompt_attention_mask = prompt_attention_mask[:, crop_start:]
    return prompt_attention_mask.sum()
