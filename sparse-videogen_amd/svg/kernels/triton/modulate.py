"""Same module path and function names as the reference (svg/kernels/triton/modulate.py), implemented with the HIP kernels of
csrc/glue.hip instead of Triton; `layernorm_modulate_forward` is the fused form (one pass over the hidden states)."""
from __future__ import annotations

import torch

from ... import _native


def triton_modulate_shift_forward(x, scale, shift, output_dtype=torch.float32):
    """y = x * (1 + scale) + shift  (ref: modulate.py:45-85)"""
    return _native.modulate_shift_forward(x.contiguous(), scale, shift, output_dtype)


def triton_modulate_gate_residual_forward(residual, x, gate, output_dtype=torch.float32):
    """y = residual + x * gate  (ref: modulate.py:125-164)"""
    return _native.modulate_gate_residual_forward(residual.contiguous(), x.contiguous(), gate, output_dtype)


def layernorm_modulate_forward(x, w, b, eps, scale=None, shift=None, output_dtype=None):
    """fp32 LayerNorm + modulate in one pass (replaces triton_layernorm_forward + triton_modulate_shift_forward,
    svg/models/wan/custom_models.py:37-60)"""
    from . import layernorm as _ln   # (its REFERENCE_PADDING switch applies to the fused form too)

    return _native.layernorm_modulate_forward(x.contiguous(), w, b, scale, shift, eps, output_dtype, reference_padding=_ln.REFERENCE_PADDING)
