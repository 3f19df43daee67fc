"""ref: svg/models/cosmos/custom_models.py — make `timestep` reach every self-attention processor (see wan/custom_models.py)."""
from __future__ import annotations

from ..context import install_timestep_hook

_TRANSFORMERS = []


def register_transformer(transformer) -> None:
    if transformer not in _TRANSFORMERS:
        _TRANSFORMERS.append(transformer)


def replace_sparse_forward() -> None:
    for t in _TRANSFORMERS:
        install_timestep_hook(t)
