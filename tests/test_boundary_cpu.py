"""CPU checks of the drop-in boundary: the C-ABI library loads here (no GPU), exports every symbol include/svg_attn.h
declares, the ctypes structs match the C layout, argument validation returns error codes without touching the GPU, and
the Python package exposes the reference's module paths / names."""
import ctypes
import importlib
import re
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent


def declared_functions():
    src = (ROOT / "include" / "svg_attn.h").read_text()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(svg_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from svg import _native

    lib = _native.load()
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/svg_attn.h but not exported"
    assert set(names) == set(_native.SIGNATURES), set(names) ^ set(_native.SIGNATURES)
    assert "gfx950" in _native.build_info()


def test_ctypes_signatures_match_the_header_prototypes():
    """Every prototype of include/svg_attn.h against the ctypes signature the Python side binds it with (svg/_native.py SIGNATURES):
    same number of parameters, same class per parameter (pointer / int32 / int64 / size_t / float / double) and same return class —
    a mismatch here is a corrupted argument on the GPU box, found without one."""
    from svg import _native

    src = (ROOT / "include" / "svg_attn.h").read_text()
    src = re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", src, flags=re.S))
    protos = re.findall(r"\b([A-Za-z_][A-Za-z0-9_ ]*?[ \*]+)(svg_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S)
    assert {n for _, n, _ in protos} == set(_native.SIGNATURES)

    def c_class(t):
        t = t.strip()
        for pat, c in ((r"\*", "ptr"), (r"\bsize_t\b", "size"), (r"\b(int64_t|long long)\b", "i64"), (r"\b(int32_t|uint32_t|int|unsigned)\b", "i32"),
                       (r"\bfloat\b", "f32"), (r"\bdouble\b", "f64"), (r"^void$", "void")):
            if re.search(pat, t):
                return c
        return "?" + t

    def py_class(a):
        if a is None:
            return "void"
        if a in (ctypes.c_void_p, ctypes.c_char_p) or (isinstance(a, type) and issubclass(a, ctypes._Pointer)):
            return "ptr"
        return {ctypes.c_size_t: "size", ctypes.c_int64: "i64", ctypes.c_longlong: "i64", ctypes.c_int32: "i32", ctypes.c_int: "i32",
                ctypes.c_uint32: "i32", ctypes.c_float: "f32", ctypes.c_double: "f64"}.get(a, "?" + repr(a))

    for ret, name, params in protos:
        ps = [x.strip() for x in params.split(",") if x.strip() and x.strip() != "void"]
        want = [c_class(x if x.endswith("*") else re.sub(r"\b[A-Za-z_][A-Za-z0-9_]*\s*(\[[^\]]*\])?$", "", x)) for x in ps]
        res, args = _native.SIGNATURES[name]
        assert [py_class(a) for a in args] == want, (name, want, [py_class(a) for a in args])
        assert py_class(res) == c_class(ret), (name, ret, res)


def test_every_call_site_passes_as_many_arguments_as_the_signature_has():
    """Static check of svg/_native.py: each `lib.svg_*(...)` call passes exactly the number of positional arguments its ctypes
    signature declares (ctypes would raise only when the call runs — on the GPU box)."""
    import ast

    from svg import _native

    tree = ast.parse((ROOT / "sparse-videogen_amd" / "svg" / "_native.py").read_text())
    checked = 0
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr in _native.SIGNATURES:
            if any(isinstance(a, ast.Starred) for a in node.args):
                continue
            assert not node.keywords and len(node.args) == len(_native.SIGNATURES[node.func.attr][1]), (node.func.attr, node.lineno)
            checked += 1
    assert checked >= 45


def test_struct_layouts():
    from svg import _native

    assert ctypes.sizeof(_native.BandMask) == 24
    assert ctypes.sizeof(_native.PermDesc) == 24 and _native.PermDesc.vid0.offset == 8
    assert ctypes.sizeof(_native.ProfileVariant) == 28
    assert ctypes.sizeof(_native.ProfileDesc) == 16 + 2 * 28
    # svg_attn_layout_t: two int32, then four { batch, head, row } int64 triples (include/svg_attn.h)
    assert ctypes.sizeof(_native.TensorStrides) == 24 and ctypes.sizeof(_native.AttnLayout) == 8 + 4 * 24
    assert _native.AttnLayout.q.offset == 8 and _native.AttnLayout.o.offset == 8 + 3 * 24 and _native.TensorStrides.row.offset == 16


def test_argument_validation_returns_error_codes():
    from svg import _native

    lib = _native.load()
    assert lib.svg_strerror(0) == b"ok"
    m = _native.BandMask(10, 3, 0, 0, 0, 0)
    # null pointers / bad geometry are rejected before any launch
    assert lib.svg_band_attention(None, None, None, None, 1, 10, 128, 0, 1.0, ctypes.byref(m), None, 0, None) == -1
    assert lib.svg_permute_rows(None, None, None, 1, 1, 64, 0, None) == -1
    assert lib.svg_argsort_workspace_bytes(2, 5000, 100) == 2 * 5 * 100 * 4
    assert lib.svg_varblock_workspace_bytes(4, 4, 10, 20, 1000) > 0
    assert lib.svg_kmeans_workspace_bytes(2, 5000, 100, 128) >= lib.svg_argsort_workspace_bytes(2, 5000, 100)


def test_native_ops_refuse_cpu_tensors():
    from svg import _native

    x = torch.zeros(1, 4, 64, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _native.permute_rows(x, torch.zeros(1, 4, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _native.band_attention(x[None], x[None], x[None], _native.BandMask(4, 5, 0, 0, 0, 0))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from svg import _native

    monkeypatch.setenv("SVG_ATTN_LIB", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_native, "_lib", None)
    with pytest.raises(RuntimeError, match="could not be loaded"):
        _native.load()
    assert not _native.available()


REFERENCE_API = {
    "svg.models.hyvideo.attention": ["HunyuanVideoAttnProcessor2_0_FlashAttention", "Hunyuan_SVGAttn_Processor2_0",
                                     "Hunyuan_SAPAttn_Processor2_0", "prepare_flexattention"],
    "svg.models.hyvideo.inference": ["replace_hyvideo_flashattention", "replace_hyvideo_attention"],
    "svg.models.hyvideo.placement": ["hunyuan_sparse_head_placement", "hunyuan_hidden_states_placement",
                                     "ref_hunyuan_sparse_head_placement", "ref_hunyuan_hidden_states_placement"],
    "svg.models.hyvideo.utils": ["generate_temporal_head_mask_mod", "get_attention_mask", "sparsity_to_width"],
    "svg.models.hyvideo.custom_models": ["replace_sparse_forward"],
    "svg.models.wan.attention": ["WanAttn_SVGAttn_Processor2_0", "WanAttn_SAPAttn_Processor", "prepare_flexattention"],
    "svg.models.wan.inference": ["replace_wan_attention"],
    "svg.models.wan.placement": ["wan_sparse_head_placement", "wan_hidden_states_placement"],
    "svg.models.wan.utils": ["generate_temporal_head_mask_mod", "get_attention_mask", "sparsity_to_width"],
    "svg.models.cog.attention": ["CogVideoX_SparseAttn_Processor2_0", "prepare_flexattention"],
    "svg.models.cog.inference": ["replace_cog_attention"],
    "svg.models.cog.placement": ["sparse_head_placement", "hidden_states_placement", "ref_sparse_head_placement"],
    "svg.models.cog.utils": ["generate_temporal_head_mask_mod", "get_attention_mask", "sparsity_to_width"],
    "svg.kernels.triton.permute": ["permute_tensor_by_labels_triton", "apply_inverse_permutation_triton"],
    "svg.kmeans_utils": ["batch_kmeans_Euclid", "identify_dynamic_map", "dynamic_block_sparse_fwd_flashinfer",
                         "dynamic_block_sparse_fwd_torch", "density_calculation", "weighted_softmax"],
    "svg.timer": ["time_logging_decorator", "TimeLoggingContext", "operator_log_data", "print_operator_log_data"],
}


@pytest.mark.parametrize("module", sorted(REFERENCE_API))
def test_reference_module_paths_and_names(module):
    m = importlib.import_module(module)
    for name in REFERENCE_API[module]:
        assert hasattr(m, name), f"{module}.{name} missing (reference API)"


def test_mask_descriptors_equal_oracle_parameters():
    from oracle import svg_oracle as O
    from svg.models.cog.utils import generate_temporal_head_mask_mod as cog_mm
    from svg.models.hyvideo.utils import generate_temporal_head_mask_mod as hy_mm
    from svg.models.hyvideo.utils import sparsity_to_width
    from svg.models.wan.utils import generate_temporal_head_mask_mod as wan_mm

    w = sparsity_to_width(0.25, 256, 33, 3600)
    assert w == O.sparsity_to_width(0.25, 256, 33, 3600)
    S = 119056
    assert hy_mm(256, 64, 33, 3600, w).as_tuple() == tuple(O.hy_band_params(S, 256, 64, 33, 3600, w).values())
    ww = sparsity_to_width(0.3, 0, 21, 3600)
    assert wan_mm(0, 0, 21, 3600, ww).as_tuple() == tuple(O.wan_band_params(75600, 21, 3600, ww).values())
    wc = sparsity_to_width(0.25, 226, 13, 1350)
    assert cog_mm(226, 13, 1350, wc).as_tuple() == tuple(O.cog_band_params(17776, 226, 13, 1350, wc).values())
    assert hy_mm(256, 64, 33, 3600, w).band == 15616 and wan_mm(0, 0, 21, 3600, ww).band == 12417


def test_tools_and_benches_compile():
    """the measurement scripts (bench*.py, tools/*.py) are part of what the profiles/ numbers come from: keep them importable"""
    import py_compile
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    files = sorted(root.glob("bench*.py")) + sorted((root / "tools").glob("*.py")) + [root / "__graft_entry__.py"]
    assert len(files) >= 10
    for f in files:
        py_compile.compile(str(f), doraise=True)
