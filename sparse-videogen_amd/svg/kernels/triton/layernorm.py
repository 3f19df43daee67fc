"""Same module path and function names as the reference (svg/kernels/triton/layernorm.py), implemented with the HIP kernels of
csrc/glue.hip instead of Triton.  fp32 statistics, fp32 output like the reference's kernels.

REFERENCE_PADDING (module switch, default False).  The reference's kernels load a row zero-padded to the next power of two and the
padding takes part in the variance: var' = var + (N2 - N) / N * mean^2 (ref layernorm.py:35-41, :134-140 — found by executing them
with Triton's interpreter, tests/test_triton_golden.py), so for hidden sizes such as 1536 / 5120 they do not compute the
FP32LayerNorm they replace (svg/models/wan/custom_models.py:44-47) unless the row mean is 0.  False: FP32LayerNorm (what this
package computes everywhere).  True: the reference's kernels as they are, for a bit-level comparison with a reference run."""
from __future__ import annotations

import torch

from ... import _native

REFERENCE_PADDING = False


def triton_layernorm_param_forward(x, w, b, eps):
    """ref: layernorm.py:64-106"""
    return _native.layernorm_forward(x.contiguous(), w.contiguous(), b.contiguous(), eps, torch.float32, reference_padding=REFERENCE_PADDING)


def triton_layernorm_noparam_forward(x, eps):
    """ref: layernorm.py:157-197"""
    return _native.layernorm_forward(x.contiguous(), None, None, eps, torch.float32, reference_padding=REFERENCE_PADDING)


def triton_layernorm_forward(x, w, b, eps, elementwise_affine=True):
    """ref: layernorm.py:204-216"""
    if elementwise_affine:
        assert w is not None and b is not None
        return triton_layernorm_param_forward(x, w, b, eps)
    assert w is None and b is None
    return triton_layernorm_noparam_forward(x, eps)
