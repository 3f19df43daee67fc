"""Same module path and function name as the reference (svg/kernels/triton/rmsnorm.py), on the HIP kernel of csrc/glue.hip
(svg_rmsnorm_forward) instead of Triton: fp32 statistics, y = (x * rstd * w) rounded ONCE to x.dtype — the reference kernel's
arithmetic (rmsnorm.py:8-48; checked against fixtures that kernel produced, tests/test_gpu_triton_golden.py)."""
from __future__ import annotations

from ... import _native
from .utils import flatten_if_batched  # noqa: F401  (re-exported like the reference module's import)


def triton_rmsnorm_forward(x, w, eps):
    """ref: rmsnorm.py:51-105 — x [M, N] or [B, S, N] contiguous -> y of the same shape and dtype"""
    assert x.is_contiguous(), "Input must be contiguous"
    return _native.rmsnorm_forward(x, None if w is None else w.contiguous(), eps, x.dtype)
