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

_CURRENT: dict = {"timestep": None}


def current_timestep() -> Optional[Any]:
    return _CURRENT["timestep"]


def set_timestep(t) -> None:
    _CURRENT["timestep"] = t


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
        prev = _CURRENT["timestep"]
        _CURRENT["timestep"] = t
        try:
            return orig(*args, **kwargs)
        finally:
            _CURRENT["timestep"] = prev

    transformer.forward = forward
    transformer._svg_timestep_hook = True
