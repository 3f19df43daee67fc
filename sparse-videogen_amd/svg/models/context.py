"""How `timestep` reaches the attention processors.

The reference monkey-patches copies of diffusers' block / model `forward`s so that `timestep` is threaded through as a
keyword argument (svg/models/hyvideo/custom_models.py:134-256, pinned to diffusers 0.34 internals).  Here the
transformer's forward is wrapped once (`install_timestep_hook`): the wrapper publishes the current `timestep` in this
module and the processors read it when the keyword is not passed.  Both conventions work: an explicit
`timestep=` keyword (reference-style patched blocks) always wins."""
from __future__ import annotations

import functools
import inspect
from typing import Any, Optional

_CURRENT: dict = {"timestep": None, "host": None}


def current_timestep() -> Optional[Any]:
    return _CURRENT["timestep"]


def set_timestep(t) -> None:
    _CURRENT["timestep"] = t
    _CURRENT["host"] = None


def timestep_value(t) -> float:
    """`timestep[0]` as a Python float.  For the tensor published by the forward hook the device-to-host read happens once per
    transformer forward (the reference compares the GPU tensor in every layer: one sync per layer-call)."""
    import torch

    if t is _CURRENT["timestep"] and _CURRENT["host"] is not None:
        return _CURRENT["host"]
    t0 = t[0] if (torch.is_tensor(t) and t.dim() > 0) or isinstance(t, (list, tuple)) else t
    val = float(t0)
    if t is _CURRENT["timestep"]:
        _CURRENT["host"] = val
    return val


def install_timestep_hook(transformer) -> None:
    """Wrap `transformer.forward` so that its `timestep` argument is visible to the processors during the call."""
    if getattr(transformer, "_svg_timestep_hook", False):
        return
    orig = transformer.forward
    sig = inspect.signature(orig)

    @functools.wraps(orig)
    def forward(*args, **kwargs):
        try:
            bound = sig.bind_partial(*args, **kwargs)
            t = bound.arguments.get("timestep", None)
        except TypeError:
            t = kwargs.get("timestep", None)
        prev = (_CURRENT["timestep"], _CURRENT["host"])
        _CURRENT["timestep"], _CURRENT["host"] = t, None
        try:
            return orig(*args, **kwargs)
        finally:
            _CURRENT["timestep"], _CURRENT["host"] = prev

    transformer.forward = forward
    transformer._svg_timestep_hook = True


class TransformerRegistry:
    """What every model's `custom_models.py` needs: the install hook (`replace_*_attention`) registers the pipeline's transformer, and
    the reference-named, zero-argument `replace_sparse_forward()` makes `timestep` reach the processors of every registered
    transformer (plus, for Wan, rebinds the block forward: `also`).  One registry per model module."""

    def __init__(self, also=None):
        self._transformers, self._also = [], also

    def register_transformer(self, transformer) -> None:
        if transformer not in self._transformers:
            self._transformers.append(transformer)

    def replace_sparse_forward(self) -> None:
        for t in self._transformers:
            install_timestep_hook(t)
            if self._also is not None:
                self._also(t)
