"""Same module path and function names as the reference (svg/kernels/triton/layernorm.py), implemented with the HIP kernels of
csrc/glue.hip instead of Triton.  fp32 statistics, fp32 output like the reference's kernels."""
from __future__ import annotations

import torch

from ... import _native


def triton_layernorm_param_forward(x, w, b, eps):
    """ref: layernorm.py:64-106"""
    return _native.layernorm_forward(x.contiguous(), w.contiguous(), b.contiguous(), eps, torch.float32)


def triton_layernorm_noparam_forward(x, eps):
    """ref: layernorm.py:157-197"""
    return _native.layernorm_forward(x.contiguous(), None, None, eps, torch.float32)


def triton_layernorm_forward(x, w, b, eps, elementwise_affine=True):
    """ref: layernorm.py:204-216"""
    if elementwise_affine:
        assert w is not None and b is not None
        return triton_layernorm_param_forward(x, w, b, eps)
    assert w is None and b is None
    return triton_layernorm_noparam_forward(x, eps)
