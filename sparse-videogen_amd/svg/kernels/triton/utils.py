"""ref: svg/kernels/triton/utils.py — the one helper the reference's kernel wrappers share."""
from __future__ import annotations


def flatten_if_batched(*tensors):
    """(B, N, D_i) -> (B * N, D_i) for every tensor if the first one is 3-D -> (list of tensors, batched, batch size or None)"""
    if not tensors:
        raise ValueError("At least one tensor must be provided.")
    first = tensors[0]
    assert first.dim() in (2, 3), "Input tensors must be batched (3D) or not batched (2D)"
    if first.dim() == 2:
        return list(tensors), False, None
    bs = first.shape[0]
    assert all(t.shape[0] == bs for t in tensors), "All input tensors must have the same batch size"
    assert all(t.shape[1] == first.shape[1] for t in tensors), "All input tensors must have the same sequence length"
    return [t.reshape(-1, t.shape[-1]) for t in tensors], True, bs
