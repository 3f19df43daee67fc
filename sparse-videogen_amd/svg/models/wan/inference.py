"""Install hook for Wan 2.1 — same name and signature as the reference `replace_wan_attention`
(svg/models/wan/inference.py:17-186)."""
from __future__ import annotations

import os

import torch

from ...logger import logger
from .attention import WanAttn_SAPAttn_Processor, WanAttn_SVGAttn_Processor2_0, prepare_flashinfer_attention, prepare_flexattention
from .custom_models import register_transformer, replace_sparse_forward
from .utils import get_attention_mask, sparsity_to_width


def replace_wan_attention(
    pipe,
    height,
    width,
    num_frames,
    first_layers_fp,
    first_times_fp,
    attention_backend="flexattn",
    pattern="SVG",
    num_sampled_rows=64,
    sample_mse_max_row=10000,
    sparsity=0.25,
    num_q_centroids=None,
    num_k_centroids=None,
    top_p_kmeans=None,
    min_kc_ratio=0,
    logging_file=None,
    kmeans_iter_init=0,
    kmeans_iter_step=0,
    zero_step_kmeans_init=False,
):
    from .._core import reseed_switch_generator

    reseed_switch_generator()   # installing the processors (a new video) resets the switched path's profiler-row generator
    context_length = 0
    num_frame_patches = 1 + num_frames // (pipe.vae_scale_factor_temporal * pipe.transformer.config.patch_size[0])
    mod_value = pipe.vae_scale_factor_spatial * pipe.transformer.config.patch_size[1]
    frame_patches_one_frame = int(height // mod_value) * int(width // mod_value)

    register_transformer(pipe.transformer)
    replace_sparse_forward()
    num_layers = len(pipe.transformer.blocks)

    if pattern == "SVG":
        AttnModule = WanAttn_SVGAttn_Processor2_0
        AttnModule.num_sampled_rows = num_sampled_rows
        AttnModule.sample_mse_max_row = sample_mse_max_row
        AttnModule.sparsity = sparsity
        AttnModule.attention_masks = [
            get_attention_mask(name, sample_mse_max_row, context_length, num_frame_patches, frame_patches_one_frame)
            for name in ("spatial", "temporal")
        ]
        multiplier = diag_width = sparsity_to_width(sparsity, context_length, num_frame_patches, frame_patches_one_frame)
        if attention_backend not in ("flexattn", "flashinfer"):
            raise ValueError(f"Attention backend {attention_backend} not supported")
        # both reference back-ends map to the same HIP kernel; the mask is the production (flex) mask
        AttnModule.block_mask = prepare_flexattention(1, None, None, torch.bfloat16, None, context_length, context_length,
                                                      num_frame_patches, frame_patches_one_frame, diag_width, multiplier)
        if attention_backend == "flashinfer":   # ref :92-117: the uniform-block (BSR) statement of the same mask, kept for callers that read it
            AttnModule.temporal_mask_metadata = prepare_flashinfer_attention(1, None, None, torch.bfloat16, None, context_length, context_length,
                                                                             num_frame_patches, frame_patches_one_frame, diag_width, multiplier)
        logger.info(f"SVG: sparsity {sparsity} -> width {multiplier:.4f} frames -> band {AttnModule.block_mask.band - 1} tokens")
    elif pattern == "SAP":
        logger.info(f"Configuring KMEANS_BLOCK attention with QC: {num_q_centroids}, KC: {num_k_centroids}, "
                    f"P: {top_p_kmeans}, min_kc_ratio: {min_kc_ratio}")
        if logging_file is not None:
            os.makedirs(os.path.dirname(logging_file) or ".", exist_ok=True)
            with open(logging_file, "w") as f:
                f.write("")
        AttnModule = WanAttn_SAPAttn_Processor
        AttnModule.logging_file = logging_file
        AttnModule.num_q_centroids = num_q_centroids
        AttnModule.num_k_centroids = num_k_centroids
        AttnModule.top_p_kmeans = top_p_kmeans
        AttnModule.min_kc_ratio = min_kc_ratio
        AttnModule.num_layers = num_layers
        AttnModule.kmeans_iter_init = kmeans_iter_init
        AttnModule.kmeans_iter_step = kmeans_iter_step
        AttnModule.zero_step_kmeans_init = zero_step_kmeans_init
    else:
        raise ValueError(f"Pattern '{pattern}' not supported")

    AttnModule.first_layers_fp = first_layers_fp
    AttnModule.first_times_fp = first_times_fp
    AttnModule.context_length = context_length
    AttnModule.num_frame = num_frame_patches
    AttnModule.frame_size = frame_patches_one_frame

    for layer_idx, m in enumerate(pipe.transformer.blocks):
        proc = AttnModule(layer_idx=layer_idx)
        proc.num_layers = num_layers
        if hasattr(m.attn1, "set_processor"):
            m.attn1.set_processor(proc)
        else:
            m.attn1.processor = proc
    print(f"Attention processors replaced with {pattern} pattern.")
    return AttnModule
