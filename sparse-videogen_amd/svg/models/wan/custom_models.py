"""ref: svg/models/wan/custom_models.py — make `timestep` reach every self-attention processor.

`replace_sparse_forward()` keeps the reference's name and zero-argument call; it needs the pipeline's transformer, which
the install hook (`replace_wan_attention`) registers through `register_transformer`."""
from __future__ import annotations

from ..context import install_timestep_hook

_TRANSFORMERS = []


def register_transformer(transformer) -> None:
    if transformer not in _TRANSFORMERS:
        _TRANSFORMERS.append(transformer)


def replace_sparse_forward() -> None:
    for t in _TRANSFORMERS:
        install_timestep_hook(t)
