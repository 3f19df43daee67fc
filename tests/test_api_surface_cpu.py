"""The reference's entry scripts (cog_inference.py, hyvideo_t2v_inference.py, wan_*_inference.py, cosmos_t2v_inference.py)
must run against this package unchanged: every `from svg... import name` line resolves, and every imported function takes the
reference's parameters — same names, same order, same defaults.  The surface is recorded from the reference by AST
(tests/golden/make_golden_api.py -> api_surface.json)."""
import importlib
import inspect
import json
from pathlib import Path

import pytest

SURFACE = json.loads((Path(__file__).parent / "golden" / "api_surface.json").read_text())
IDS = [f"{o['script']}:{o['line']}:{o['name']}" for o in SURFACE]


@pytest.mark.parametrize("entry", SURFACE, ids=IDS)
def test_reference_import_line_resolves(entry):
    mod = importlib.import_module(entry["module"])
    assert hasattr(mod, entry["name"]), f"{entry['script']}:{entry['line']} from {entry['module']} import {entry['name']}"
    obj = getattr(mod, entry["name"])
    if entry["kind"] == "function":
        assert callable(obj)
        sig = inspect.signature(obj)
        ours = list(sig.parameters.values())
        ref = entry["params"]
        assert [p.name for p in ours[: len(ref)]] == [n for n, _ in ref], f"parameter order of {entry['name']}"
        for p, (name, default) in zip(ours, ref):
            if default is None:
                assert p.default is inspect.Parameter.empty, f"{entry['name']}({name}) has no default in the reference"
            elif default.isidentifier() and default not in ("None", "True", "False"):
                assert p.default is not inspect.Parameter.empty   # a module-level constant of the reference (e.g. a prompt template)
            else:
                assert p.default == eval(default), f"{entry['name']}({name}={default}) default differs: {p.default!r}"   # noqa: S307
        # extra parameters of ours must be optional
        assert all(p.default is not inspect.Parameter.empty for p in ours[len(ref):])


def test_dense_pattern_is_accepted_like_the_reference():
    """ref: hyvideo/inference.py:165-166 — `pattern == "dense"` passes the assert and installs nothing; anything else is an
    AssertionError (Wan raises ValueError for it, ref: wan/inference.py:176-177 — kept as is)."""
    from svg.models.hyvideo.inference import replace_hyvideo_attention

    class _T:
        transformer_blocks = []
        single_transformer_blocks = []

    class _Pipe:
        transformer = _T()

    assert replace_hyvideo_attention(_Pipe(), 720, 1280, 129, 64, 0.03, 0.1, pattern="dense") is None
    with pytest.raises(AssertionError, match="Invalid pattern"):
        replace_hyvideo_attention(_Pipe(), 720, 1280, 129, 64, 0.03, 0.1, pattern="nope")
