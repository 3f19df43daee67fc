"""Install hooks for HunyuanVideo — same function names and signatures as the reference module
svg/models/hyvideo/inference.py (`replace_hyvideo_flashattention`, `replace_hyvideo_attention`)."""
from __future__ import annotations

import os

import torch

from ...logger import logger
from .attention import (
    Hunyuan_SAPAttn_Processor2_0,
    Hunyuan_SVGAttn_Processor2_0,
    HunyuanVideoAttnProcessor2_0_FlashAttention,
    prepare_flexattention,
)
from .custom_models import register_transformer, replace_sparse_forward
from .utils import get_attention_mask, sparsity_to_width


def _self_attention_modules(pipe):
    """(layer_idx, attention module) of every transformer block, double-stream blocks first (ref :105-115)."""
    tr = pipe.transformer
    blocks = list(getattr(tr, "transformer_blocks", [])) + list(getattr(tr, "single_transformer_blocks", []))
    return [(i, b.attn) for i, b in enumerate(blocks)]


def replace_hyvideo_flashattention(pipe):
    """ref: hyvideo/inference.py:16-30 — dense processor on every block."""
    for layer_idx, attn in _self_attention_modules(pipe):
        attn.processor = HunyuanVideoAttnProcessor2_0_FlashAttention(layer_idx)


def replace_hyvideo_attention(
    pipe,
    height,
    width,
    num_frames,
    prompt_length,
    first_layers_fp,
    first_times_fp,
    pattern="SVG",
    # SVG
    num_sampled_rows=64,
    sample_mse_max_row=10000,
    sparsity=0.25,
    # SAP
    num_q_centroids=None,
    num_k_centroids=None,
    top_p_kmeans=None,
    min_kc_ratio=0,
    logging_file=None,
    kmeans_iter_init=0,
    kmeans_iter_step=0,
    zero_step_kmeans_init=False,
):
    """ref: hyvideo/inference.py:33-166.  Sets the class-level configuration of the processor class and installs one
    processor per block."""
    from .._core import reseed_switch_generator

    reseed_switch_generator()   # installing the processors (a new video) resets the switched path's profiler-row generator
    # geometry exactly as the reference derives it (:57-59)
    cfg_size, num_head, head_dim, dtype = 1, 24, 128, torch.bfloat16
    context_length, num_frame, frame_size = 256, 1 + num_frames // 4, height * width // 256
    prompt_length = int(prompt_length)

    if pattern == "SVG":
        AttnModule = Hunyuan_SVGAttn_Processor2_0
        multiplier = diag_width = sparsity_to_width(sparsity, context_length, num_frame, frame_size)
        AttnModule.num_sampled_rows = num_sampled_rows
        AttnModule.sample_mse_max_row = sample_mse_max_row
        AttnModule.attention_masks = [
            get_attention_mask(name, AttnModule.sample_mse_max_row, context_length, num_frame, frame_size)
            for name in ("spatial", "temporal")
        ]
        AttnModule.block_mask = prepare_flexattention(cfg_size, num_head, head_dim, dtype, "cuda", context_length,
                                                      prompt_length, num_frame, frame_size, diag_width=diag_width,
                                                      multiplier=multiplier)
        logger.info(f"SVG: sparsity {sparsity} -> width {multiplier:.4f} frames -> band {AttnModule.block_mask.band} tokens")
    elif pattern in ["SAP"]:
        logger.info(f"Configuring KMEANS_BLOCK attention with QC: {num_q_centroids}, KC: {num_k_centroids}, P: {top_p_kmeans}, "
                    f"min_kc_ratio: {min_kc_ratio}")
        if logging_file is not None:   # ref :126-130: make the directory and clear the density log
            os.makedirs(os.path.dirname(logging_file) or ".", exist_ok=True)
            with open(logging_file, "w") as f:
                f.write("")
        AttnModule = Hunyuan_SAPAttn_Processor2_0
        AttnModule.num_q_centroids = num_q_centroids
        AttnModule.num_k_centroids = num_k_centroids
        AttnModule.top_p_kmeans = top_p_kmeans
        AttnModule.min_kc_ratio = min_kc_ratio
        AttnModule.kmeans_iter_init = kmeans_iter_init
        AttnModule.kmeans_iter_step = kmeans_iter_step
        AttnModule.zero_step_kmeans_init = zero_step_kmeans_init
        AttnModule.logging_file = logging_file
        AttnModule.reset_state()
    else:
        assert pattern == "dense", f"Invalid pattern: {pattern}"   # ref :165-166: "dense" leaves the pipeline's own processors in place
        return None

    AttnModule.prompt_length = prompt_length
    AttnModule.context_length = context_length
    AttnModule.num_frame = num_frame
    AttnModule.frame_size = frame_size
    AttnModule.first_layers_fp = first_layers_fp
    AttnModule.first_times_fp = first_times_fp

    register_transformer(pipe.transformer)
    replace_sparse_forward()
    for layer_idx, attn in _self_attention_modules(pipe):
        attn.processor = AttnModule(layer_idx)
    return AttnModule
