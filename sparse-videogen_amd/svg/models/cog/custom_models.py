"""ref: svg/models/cog/custom_models.py — make `timestep` reach every self-attention processor.

`replace_sparse_forward()` keeps the reference's name and zero-argument call; it needs the pipeline's transformer, which the install
hook (`replace_cog_attention`) registers through `register_transformer` (svg.models.context.TransformerRegistry)."""
from __future__ import annotations

from ..context import TransformerRegistry

_REGISTRY = TransformerRegistry()
register_transformer = _REGISTRY.register_transformer
replace_sparse_forward = _REGISTRY.replace_sparse_forward
