"""Install hook for CogVideoX — same name and signature as the reference `replace_cog_attention`
(svg/models/cog/inference.py:26-73)."""
from __future__ import annotations

import torch

from .attention import CogVideoX_SparseAttn_Processor2_0, prepare_flexattention
from .custom_models import register_transformer, replace_sparse_forward
from .utils import get_attention_mask, sparsity_to_width

GEOMETRY = {"v1": (226, 13, 1350), "v1.5": (226, 11, 4080)}  # (context_length, num_frame, frame_size), ref :31-38


def sample_image(pipe, prompt, image_path, output_path, seed, version, num_step=50):
    """ref: svg/models/cog/inference.py:11-23 — one image-to-video sample with the pipeline's own sampler (diffusers
    `load_image` / `export_to_video`; this function is the entry script's glue, not part of the attention path)."""
    from diffusers.utils import export_to_video, load_image

    print("\n" * 5)
    print(f"Prompt: {prompt}")
    image = load_image(image_path)
    print(f"Image Is Ready. Seed is {seed}")
    if version == "v1":
        video = pipe(image=image, prompt=prompt, guidance_scale=6, use_dynamic_cfg=True, num_inference_steps=num_step).frames[0]
    elif version == "v1.5":
        video = pipe(image=image, prompt=prompt, num_videos_per_prompt=1, num_inference_steps=num_step, num_frames=81,
                     guidance_scale=6, height=768, width=1360).frames[0]
    else:
        raise ValueError(f"Unsupported version: {version}")
    export_to_video(video, output_path, fps=8)


def replace_cog_attention(pipe, version, num_sampled_rows, sparsity, first_layers_fp, first_times_fp):
    from .._core import reseed_switch_generator

    reseed_switch_generator()   # installing the processors (a new video) resets the switched path's profiler-row generator
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
