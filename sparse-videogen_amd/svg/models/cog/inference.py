"""Install hook for CogVideoX — same name and signature as the reference `replace_cog_attention`
(svg/models/cog/inference.py:26-73)."""
from __future__ import annotations

import torch

from .attention import CogVideoX_SparseAttn_Processor2_0, prepare_flexattention
from .custom_models import register_transformer, replace_sparse_forward
from .utils import get_attention_mask, sparsity_to_width

GEOMETRY = {"v1": (226, 13, 1350), "v1.5": (226, 11, 4080)}  # (context_length, num_frame, frame_size), ref :31-38


def replace_cog_attention(pipe, version, num_sampled_rows, sparsity, first_layers_fp, first_times_fp):
    if version not in GEOMETRY:
        raise ValueError(f"Unsupported version: {version}")
    context_length, num_frame, frame_size = GEOMETRY[version]
    AttnModule = CogVideoX_SparseAttn_Processor2_0
    AttnModule.num_sampled_rows = num_sampled_rows
    AttnModule.attention_masks = [get_attention_mask(n, context_length, num_frame, frame_size) for n in ("spatial", "temporal")]
    AttnModule.version = version
    AttnModule.first_layers_fp = first_layers_fp
    AttnModule.first_times_fp = first_times_fp
    multiplier = diag_width = sparsity_to_width(sparsity, context_length, num_frame, frame_size)
    AttnModule.context_length = context_length
    AttnModule.num_frame = num_frame
    AttnModule.frame_size = frame_size
    AttnModule.block_mask = prepare_flexattention(2, 48, 64, torch.bfloat16, "cuda", context_length, num_frame, frame_size,
                                                  diag_width, multiplier)
    register_transformer(pipe.transformer)
    replace_sparse_forward()
    num_layers = len(pipe.transformer.transformer_blocks)
    for layer_idx, m in enumerate(pipe.transformer.transformer_blocks):
        proc = AttnModule(layer_idx)
        proc.num_layers = num_layers
        if hasattr(m.attn1, "set_processor"):
            m.attn1.set_processor(proc)
        else:
            m.attn1.processor = proc
    return AttnModule
