"""ctypes binding of libsvgattn.so (the C ABI declared in include/svg_attn.h).

PyTorch is only plumbing here: it owns the device memory and the HIP stream; every op below passes raw device
pointers + sizes + the current stream to the C entry point and raises on a non-zero status.  There is NO CPU
fallback: calling a native op without the library (or with CPU tensors) raises RuntimeError.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional, Sequence

import torch

_LIB_PATH = Path(__file__).resolve().parent.parent / "lib" / "libsvgattn.so"
_lib: Optional[C.CDLL] = None
_load_error: Optional[str] = None

SVG_DTYPE_BF16, SVG_DTYPE_F16 = 0, 1


class BandMask(C.Structure):
    """svg_band_mask_t"""

    _fields_ = [
        ("real_len", C.c_int32),
        ("band", C.c_int32),
        ("colfull_lo", C.c_int32),
        ("colfull_hi", C.c_int32),
        ("rowfull_lo", C.c_int32),
        ("rowfull_hi", C.c_int32),
    ]

    def as_tuple(self):
        return (self.real_len, self.band, self.colfull_lo, self.colfull_hi, self.rowfull_lo, self.rowfull_hi)


class PermDesc(C.Structure):
    """svg_perm_desc_t"""

    _fields_ = [
        ("head_perm_flag", C.c_void_p),
        ("vid0", C.c_int32),
        ("num_frame", C.c_int32),
        ("frame_size", C.c_int32),
    ]


class ProfileVariant(C.Structure):
    """svg_profile_variant_t"""

    _fields_ = [
        ("coord", C.c_int32),
        ("origin", C.c_int32),
        ("span", C.c_int32),
        ("band_blocks", C.c_int32),
        ("sink_cols", C.c_int32),
        ("text_lo", C.c_int32),
        ("text_hi", C.c_int32),
    ]


class ProfileDesc(C.Structure):
    """svg_profile_desc_t"""

    _fields_ = [
        ("vid0", C.c_int32),
        ("num_frame", C.c_int32),
        ("frame_size", C.c_int32),
        ("emulate_bf16", C.c_int32),
        ("variant", ProfileVariant * 2),
    ]


_I32, _I64, _F32, _VP, _SZ = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t


class TensorStrides(C.Structure):
    """svg_tensor_strides_t: element strides of a [B, H, S, D] view whose last dimension is contiguous."""
    _fields_ = [("batch", C.c_int64), ("head", C.c_int64), ("row", C.c_int64)]


class AttnLayout(C.Structure):
    """svg_attn_layout_t (include/svg_attn.h): where q, k, v, o of a *_strided attention call live."""
    _fields_ = [("heads_per_batch", C.c_int32), ("kv_heads_per_batch", C.c_int32), ("q", TensorStrides), ("k", TensorStrides),
                ("v", TensorStrides), ("o", TensorStrides)]


# name -> (restype, argtypes); must list every symbol include/svg_attn.h declares (tests check this)
SIGNATURES = {
    "svg_abi_version": (C.c_int, []),
    "svg_strerror": (C.c_char_p, [C.c_int]),
    "svg_last_hip_error": (C.c_int, []),
    "svg_build_info": (C.c_char_p, []),
    "svg_debug_pp_trace": (C.c_int, [_VP]),
    "svg_debug_wg_trace": (C.c_int, [_VP, _I32]),
    "svg_rms_norm_forward": (C.c_int, [_VP, _VP, C.c_int64, _I32, _I32, C.c_float, _VP]),
    "svg_layer_norm_forward": (C.c_int, [_VP, _VP, _VP, C.c_int64, _I32, _I32, _VP]),
    "svg_apply_qk_rope_inplace_cossin": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _VP]),
    "svg_apply_qk_rope_inplace_cossin_txtlast": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _VP]),
    "svg_apply_qk_rope_inplace_cossin_complex": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _VP]),
    "svg_bsr_to_block_map": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _VP, _VP, _VP, _VP]),
    "svg_layernorm_forward": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int64, _I32, _I32, _I32, _I32, C.c_float, _VP]),
    "svg_rmsnorm_forward": (C.c_int, [_VP, _VP, _VP, C.c_int64, _I32, _I32, _I32, _I32, C.c_float, _VP]),
    "svg_layernorm_forward_ex": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int64, _I32, _I32, _I32, _I32, C.c_float, _I32, _VP]),
    "svg_modulate_shift_forward": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int64, _I32, C.c_int64, _I32, _I32, _VP]),
    "svg_modulate_gate_residual_forward": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int64, _I32, C.c_int64, _I32, _I32, _I32, _VP]),
    "svg_layernorm_modulate_forward": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, C.c_int64, _I32, C.c_int64, _I32, _I32, _I32,
                                                 C.c_float, _VP]),
    "svg_layernorm_modulate_forward_ex": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, C.c_int64, _I32, C.c_int64, _I32, _I32, _I32,
                                                    C.c_float, _I32, _VP]),
    "svg_qk_norm_rope_transpose": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _VP, _VP, _VP, _VP,
                                             C.c_float, _I32, _VP, _VP, _I32, _I32, _VP]),
    "svg_qk_norm_rope": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _VP, _VP, _VP, _VP, C.c_float, _I32, _VP,
                                   _VP, _I32, _I32, _VP]),
    "svg_qk_norm_rope_qscale": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _VP, _VP, _VP, _VP, C.c_float, _I32, _VP,
                                          _VP, _I32, _I32, C.c_float, _VP]),
    "svg_qk_norm_rope_transpose_qscale": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _VP, _VP, _VP, _VP,
                                                    C.c_float, _I32, _VP, _VP, _I32, _I32, C.c_float, _VP]),
    "svg_rmsnorm_rope_transpose": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _VP, _VP, _I32, C.c_float, _I32, _VP,
                                             _VP, _I32, _I32, C.c_float, _VP]),
    "svg_head_placement": (C.c_int, [_VP, _VP, _I32, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _VP]),
    "svg_permute_rows": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32, _I32, _VP]),
    "svg_inverse_permute_rows": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32, _I32, _VP]),
    "svg_argsort_workspace_bytes": (_SZ, [_I32, _I32, _I32]),
    "svg_argsort_labels": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32, _VP, _SZ, _VP]),
    "svg_band_attention": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _F32, C.POINTER(BandMask),
                                     C.POINTER(PermDesc), _I32, _VP]),
    "svg_band_attention_strided": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _F32, C.POINTER(BandMask),
                                             C.POINTER(PermDesc), C.POINTER(AttnLayout), _VP]),
    "svg_band_attention_switch_strided": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _F32, C.POINTER(BandMask),
                                                    C.POINTER(PermDesc), C.POINTER(BandMask), _VP, C.POINTER(AttnLayout), _VP]),
    "svg_varblock_attention_strided": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _F32, _VP, _VP, _VP,
                                                 _I32, _I32, _VP, _VP, _VP, _SZ, C.POINTER(AttnLayout), _VP]),
    "svg_sample_mse_strided": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _F32, C.POINTER(ProfileDesc), _VP, _VP,
                                         _SZ, _VP, C.POINTER(AttnLayout), _VP]),
    "svg_band_attention_prescaled": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, C.POINTER(BandMask), C.POINTER(PermDesc), _VP]),
    "svg_band_attention_switch_prescaled": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, C.POINTER(BandMask),
                                                      C.POINTER(PermDesc), C.POINTER(BandMask), _VP, _VP]),
    "svg_debug_clock_probe": (C.c_int, [_VP, _VP, _I32, _VP]),
    "svg_band_attention_prescaled_notify_seg": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, C.POINTER(BandMask),
                                                          C.POINTER(PermDesc), _VP, _I32, _I32, _VP]),
    "svg_band_attention_notify_target": (_I32, [_I32, C.POINTER(BandMask)]),
    "svg_band_attention_notify": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _F32, C.POINTER(BandMask),
                                            C.POINTER(PermDesc), _VP, _I32, _VP]),
    "svg_varblock_attention_fp8_workspace_bytes": (_SZ, [_I32, _I32, _I32, _I32, _I32, _I32, _I32]),
    "svg_varblock_attention_fp8": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _F32, _VP, _VP, _VP, _I32, _I32,
                                             _VP, _VP, _VP, _SZ, _VP]),
    "svg_band_attention_fp8_workspace_bytes": (_SZ, [_I32, _I32, _I32]),
    "svg_band_attention_fp8": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _F32, _VP, _VP, _VP, _SZ, _VP]),
    "svg_band_attention_fp8_stage": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _F32, _VP, _VP, _VP, _SZ, _I32, _VP]),
    "svg_wait_counters": (C.c_int, [_VP, _I32, _I32, _VP]),
    "svg_wait_counters_deadline": (C.c_int, [_VP, _I32, _I32, _I32, _VP, _VP]),
    "svg_band_attention_notify_layout": (_I32, [_I32, C.POINTER(BandMask), _I32, _VP, _VP]),
    "svg_band_attention_notify_seg": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _F32, C.POINTER(BandMask),
                                                C.POINTER(PermDesc), _VP, _I32, _I32, _VP]),
    "svg_band_attention_switch": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _F32, C.POINTER(BandMask),
                                            C.POINTER(PermDesc), C.POINTER(BandMask), _VP, _VP]),
    "svg_sample_mse_flagged": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _F32, C.POINTER(ProfileDesc), _VP, _VP,
                                         _SZ, _VP, _VP]),
    "svg_varblock_workspace_bytes": (_SZ, [_I32, _I32, _I32, _I32, _I32]),
    "svg_varblock_attention": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _F32, _VP, _VP, _VP,
                                         _I32, _I32, _VP, _VP, _VP, _SZ, _I32, _VP]),
    "svg_sample_mse_workspace_bytes": (_SZ, [_I32, _I32, _I32, _I32]),
    "svg_sample_mse": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _F32, C.POINTER(ProfileDesc), _VP, _VP,
                                 _SZ, _VP]),
    "svg_kmeans_xsq": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _I32, _VP]),
    "svg_kmeans_workspace_bytes": (_SZ, [_I32, _I32, _I32, _I32]),
    "svg_kmeans_iter": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _VP, _SZ, _VP]),
    "svg_kmeans_assign": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _VP, _SZ, _VP]),
    "svg_kmeans_update": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _VP, _SZ, _VP]),
    "svg_kmeans_loop_workspace_bytes": (_SZ, [_I32, _I32, _I32, _I32]),
    "svg_kmeans_loop": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _F32, _VP, _SZ, _VP]),
    "svg_kmeans_loop_strided": (C.c_int, [_VP, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _F32, _VP, _SZ, _VP]),
    "svg_identify_dynamic_map": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _F32, _I32, _VP]),
    "svg_map_density": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _VP]),
}


SVG_ABI_VERSION = 4   # include/svg_attn.h


def lib_path() -> Path:
    return Path(os.environ.get("SVG_ATTN_LIB", str(_LIB_PATH)))


def load(strict: bool = True) -> Optional[C.CDLL]:
    """dlopen the library once and type every entry point.  strict=False returns None instead of raising."""
    global _lib, _load_error
    if _lib is not None:
        return _lib
    p = lib_path()
    try:
        lib = C.CDLL(str(p))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        got = int(lib.svg_abi_version())
        if got != SVG_ABI_VERSION:   # a library built from another revision of include/svg_attn.h: its entry points may take other arguments
            raise OSError(f"ABI version {got}, this binding was written against {SVG_ABI_VERSION} (rebuild: python sparse-videogen_amd/build.py)")
        _lib = lib
    except (OSError, AttributeError) as e:  # missing .so, missing symbol or ABI mismatch
        _load_error = f"{p}: {e}"
        if strict:
            raise RuntimeError(
                f"libsvgattn.so could not be loaded ({_load_error}). Build it with "
                f"`python sparse-videogen_amd/build.py` (hipcc, gfx950). There is no CPU fallback."
            ) from e
        return None
    return _lib


def available() -> bool:
    return load(strict=False) is not None


def build_info() -> str:
    return load().svg_build_info().decode()


def _check(rc: int, what: str) -> None:
    if rc != 0:
        lib = load()
        msg = lib.svg_strerror(rc).decode()
        raise RuntimeError(f"{what} failed: {msg} (code {rc}, hipError {lib.svg_last_hip_error()})")


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return SVG_DTYPE_BF16
    if t.dtype == torch.float16:
        return SVG_DTYPE_F16
    raise RuntimeError(f"libsvgattn supports bfloat16 / float16 tensors, got {t.dtype}")


def _dev(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("libsvgattn ops need GPU (HIP) tensors; there is no CPU fallback for the sparse path")
        if not t.is_contiguous():
            raise RuntimeError("libsvgattn ops need contiguous tensors")


def _gpu(*ts: Optional[torch.Tensor]) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("libsvgattn ops need GPU (HIP) tensors; there is no CPU fallback for the sparse path")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------------------
# strided attention tensors (svg_attn_layout_t): q / k / v read where a projection wrote them, o written where the output projection
# reads it — the reference's processors hand flex_attention / flash-attn `proj(x).unflatten(2, (H, -1)).transpose(1, 2)` VIEWS and
# copy what those cannot take (wan/attention.py:123-125,168-170)
# ------------------------------------------------------------------------------------------------------
def _view4(t: torch.Tensor) -> torch.Tensor:
    """[B, H, S, D] view of a [B, H, S, D] or [BH, S, D] tensor (no copy)."""
    return t if t.dim() == 4 else t.unsqueeze(0)


def _strided_ok(t4: torch.Tensor, dma: bool = False) -> bool:
    """Can a *_strided entry point read / write this [B, H, S, D] view in place?  (layout_from_abi, csrc/svg_common.h)"""
    B, H, S, D = t4.shape
    st = t4.stride()
    if st[3] != 1 or t4.data_ptr() % 16 != 0 or st[2] < D or st[2] >= (1 << 23):
        return False
    if any(x % 8 != 0 or x < 0 for x in st[:3]):
        return False
    return (not dma) or S * st[2] * 2 < (1 << 32)   # k / v: 32-bit byte offsets per head in the LDS-DMA requests


def _is_dense4(t4: torch.Tensor) -> bool:
    """contiguous [B, H, S, D]: what every entry point without `_strided` takes"""
    return t4.is_contiguous()


def attn_layout(q4: torch.Tensor, k4: torch.Tensor, v4: torch.Tensor, o4: torch.Tensor) -> AttnLayout:
    ts = [TensorStrides(t.stride(0), t.stride(1), t.stride(2)) for t in (q4, k4, v4, o4)]
    return AttnLayout(q4.shape[1], k4.shape[1], *ts)


def token_major_empty(like4: torch.Tensor) -> torch.Tensor:
    """An uninitialised [B, H, S, D] tensor stored token-major ([B, S, H, D] in memory): `out.transpose(1, 2).flatten(2, 3)` — what
    every processor of the reference does with the attention output before the output projection — is then a view, not a copy."""
    B, H, S, D = like4.shape
    return torch.empty((B, S, H, D), dtype=like4.dtype, device=like4.device).permute(0, 2, 1, 3)


def strided_attention_supported(q: torch.Tensor, variant: int = 0) -> bool:
    """The *_strided entry points exist for the default schedules (the two-phase bodies: 16x16x32 at head_dim 128, 32x32x16 at 64)."""
    return q.shape[-1] in (64, 128) and variant == 0 and q.dtype in (torch.bfloat16, torch.float16)


# ------------------------------------------------------------------------------------------------------
# typed wrappers (torch tensors in, torch tensors out)
# ------------------------------------------------------------------------------------------------------
def head_placement(srcs: Sequence[torch.Tensor], dsts: Sequence[torch.Tensor], best_mask_idx: torch.Tensor,
                   context_length: int, num_frame: int, frame_size: int, text_first: bool, inverse: bool) -> None:
    lib = load()
    n = len(srcs)
    assert 1 <= n <= 3 and len(dsts) == n
    _dev(*srcs, *dsts, best_mask_idx)
    x = srcs[0]
    BH = x.shape[0] * x.shape[1]
    S, D = x.shape[2], x.shape[3]
    for t in list(srcs) + list(dsts):
        assert t.shape == x.shape and t.dtype == x.dtype
    best = best_mask_idx.to(torch.int64).contiguous()
    assert best.numel() == BH
    sp = (C.c_void_p * 3)(*[s.data_ptr() for s in srcs], *([None] * (3 - n)))
    dp = (C.c_void_p * 3)(*[d.data_ptr() for d in dsts], *([None] * (3 - n)))
    rc = lib.svg_head_placement(C.cast(sp, _VP), C.cast(dp, _VP), n, best.data_ptr(), BH, S, D, _dtype_code(x),
                                context_length, num_frame, frame_size, int(text_first), int(inverse), _stream())
    _check(rc, "svg_head_placement")


def permute_rows(x: torch.Tensor, idx: torch.Tensor, inverse: bool = False) -> torch.Tensor:
    lib = load()
    _dev(x, idx)
    BH, S, D = x.shape
    assert idx.dtype == torch.int32 and idx.shape == (BH, S)
    y = torch.empty_like(x)
    fn = lib.svg_inverse_permute_rows if inverse else lib.svg_permute_rows
    _check(fn(x.data_ptr(), idx.data_ptr(), y.data_ptr(), BH, S, D, _dtype_code(x), _stream()), "svg_permute_rows")
    return y


def argsort_labels(labels: torch.Tensor, K: int):
    """labels int32 [B, N] in [0, K) -> (sorted_idx int32 [B, N] (stable), counts int32 [B, K])"""
    lib = load()
    _dev(labels)
    assert labels.dtype == torch.int32 and labels.dim() == 2
    B, N = labels.shape
    ws = torch.empty(lib.svg_argsort_workspace_bytes(B, N, K), dtype=torch.uint8, device=labels.device)
    sidx = torch.empty((B, N), dtype=torch.int32, device=labels.device)
    counts = torch.empty((B, K), dtype=torch.int32, device=labels.device)
    rc = lib.svg_argsort_labels(labels.data_ptr(), sidx.data_ptr(), counts.data_ptr(), B, N, K, ws.data_ptr(),
                                ws.numel(), _stream())
    _check(rc, "svg_argsort_labels")
    return sidx, counts


def band_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, mask: BandMask, sm_scale: Optional[float] = None,
                   head_perm_flag: Optional[torch.Tensor] = None, vid0: int = 0, num_frame: int = 1,
                   frame_size: int = 1, variant: int = 0, out: Optional[torch.Tensor] = None,
                   done: Optional[torch.Tensor] = None, done_nseg: int = 1, q_prescaled: bool = False,
                   token_major_out: bool = False) -> torch.Tensor:
    """q, k, v: [B, H, S, D] (or [BH, S, D]) bf16/fp16 GPU tensors -> o of the same shape.
    Strided views (last dimension contiguous — e.g. `proj(x).unflatten(2, (H, -1)).transpose(1, 2)`, or a slice of a fused QKV
    projection) are read in place by svg_band_attention_strided where it exists (head_dim 128, default schedule, plain q, no completion
    counters) and copied otherwise, as the reference does.  token_major_out: the result is a [B, H, S, D] tensor stored [B, S, H, D]
    (token_major_empty), so the processors' `.transpose(1, 2).flatten(2, 3)` is a view.
    q_prescaled: q already carries sm_scale * log2(e) (SOFTMAX_Q_SCALE(D) for the default scale; what qk_norm_rope*(q_scale=...)
    writes): svg_band_attention_prescaled, default schedule only.
    done: int32 [BH * (done_nseg + 1)] zeroed completion counters (svg_band_attention_notify[_seg]; see band_notify_target /
    band_notify_layout / wait_counters / notify_counters): counter (h, s) at done[h * done_nseg + s], the last BH words are scratch
    of the library (hidden per-head counters of heads that run with the fused layout permutation)."""
    lib = load()
    _dev(head_perm_flag)
    _gpu(q, k, v, out)
    assert q.shape == k.shape == v.shape and q.dtype == k.dtype == v.dtype
    S, D = q.shape[-2], q.shape[-1]
    BH = q.numel() // (S * D)
    scale = float(sm_scale) if sm_scale is not None else 1.0 / (D ** 0.5)
    perm = None
    if head_perm_flag is not None:
        flag = head_perm_flag.to(torch.int64).contiguous()
        assert flag.numel() == BH
        perm = PermDesc(flag.data_ptr(), vid0, num_frame, frame_size)
    dense_in = q.is_contiguous() and k.is_contiguous() and v.is_contiguous() and (out is None or out.is_contiguous())
    if (not dense_in or (token_major_out and out is None)) and done is None and not q_prescaled and strided_attention_supported(q, variant):
        q4, k4, v4 = _view4(q), _view4(k), _view4(v)
        o4 = _view4(out) if out is not None else (token_major_empty(q4) if token_major_out else torch.empty(q4.shape, dtype=q.dtype, device=q.device))
        if _strided_ok(q4) and _strided_ok(k4, True) and _strided_ok(v4, True) and _strided_ok(o4) and o4.shape == q4.shape:
            lay = attn_layout(q4, k4, v4, o4)
            rc = lib.svg_band_attention_strided(q.data_ptr(), k.data_ptr(), v.data_ptr(), o4.data_ptr(), BH, S, D, _dtype_code(q), scale,
                                                C.byref(mask), C.byref(perm) if perm is not None else None, C.byref(lay), _stream())
            _check(rc, "svg_band_attention_strided")
            return out if out is not None else (o4 if q.dim() == 4 else o4.squeeze(0))
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()   # (what no strided entry point takes is copied, as the reference does)
    if out is not None and not out.is_contiguous():
        out.copy_(band_attention(q, k, v, mask, sm_scale, head_perm_flag, vid0, num_frame, frame_size, variant, None, done, done_nseg,
                                 q_prescaled))
        return out
    o = torch.empty_like(q) if out is None else out
    if q_prescaled and done is not None:
        _dev(done)
        assert done.dtype == torch.int32 and done.is_contiguous() and variant == 0 and sm_scale is None
        rc = lib.svg_band_attention_prescaled_notify_seg(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), BH, S, D,
                                                         _dtype_code(q), C.byref(mask), C.byref(perm) if perm is not None else None,
                                                         done.data_ptr(), int(done.numel()), int(done_nseg), _stream())
        _check(rc, "svg_band_attention_prescaled_notify_seg")
        return o
    if q_prescaled:
        assert variant == 0 and sm_scale is None, "q_prescaled: default-schedule call, the scale lives in q"
        rc = lib.svg_band_attention_prescaled(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), BH, S, D, _dtype_code(q),
                                              C.byref(mask), C.byref(perm) if perm is not None else None, _stream())
        _check(rc, "svg_band_attention_prescaled")
        return o
    if done is not None:
        _dev(done)
        assert done.dtype == torch.int32 and done.is_contiguous() and variant == 0
        rc = lib.svg_band_attention_notify_seg(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), BH, S, D, _dtype_code(q),
                                               scale, C.byref(mask), C.byref(perm) if perm is not None else None, done.data_ptr(),
                                               int(done.numel()), int(done_nseg), _stream())   # too small: SVG_ERR_WORKSPACE
        _check(rc, "svg_band_attention_notify_seg")
        return o
    rc = lib.svg_band_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), BH, S, D, _dtype_code(q), scale,
                                C.byref(mask), C.byref(perm) if perm is not None else None, variant, _stream())
    _check(rc, "svg_band_attention")
    return o


def softmax_q_scale(D: int, sm_scale: Optional[float] = None) -> float:
    """The factor a pre-scaled q carries: sm_scale * log2(e) (default sm_scale = 1 / sqrt(D))."""
    return (float(sm_scale) if sm_scale is not None else 1.0 / (D ** 0.5)) * 1.4426950408889634


class WorkspaceCache:
    """Scratch tensors the wrappers keep between calls, keyed by (shape..., device, stream): two calls of one shape on different
    streams never share a buffer, and at most `capacity` entries stay alive (least recently used goes first), so transient streams
    cannot pile up multi-GB workspaces.  One policy for the fp8 workspaces and the k-means loop scratch."""

    def __init__(self, capacity: int = 8):
        from collections import OrderedDict

        self.capacity, self._d = capacity, OrderedDict()

    @staticmethod
    def key(*shape, device):
        return tuple(shape) + (device, _stream())

    def get(self, key):
        v = self._d.get(key)
        if v is not None:
            self._d.move_to_end(key)
        return v

    def __setitem__(self, key, value):
        self._d[key] = value
        self._d.move_to_end(key)
        while len(self._d) > self.capacity:
            self._d.popitem(last=False)

    def __len__(self):
        return len(self._d)

    def clear(self):
        self._d.clear()


_F8_WS = WorkspaceCache()
_KMEANS_WS = WorkspaceCache()


def band_attention_fp8(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, mask: BandMask, sm_scale: Optional[float] = None,
                       head_perm_flag: Optional[torch.Tensor] = None, vid0: int = 0, num_frame: int = 1, frame_size: int = 1,
                       out: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None, stage: int = 0) -> torch.Tensor:
    """band_attention with e4m3 QK^T / PV (svg_band_attention_fp8; D = 128).  Same arguments and output as band_attention; the
    quantise + placement pre-pass runs inside the call.  workspace: uint8 GPU tensor of band_attention_fp8_workspace_bytes(...)
    bytes, cached per (BH, S, device) when omitted.  stage 1 / 2: only the pre-pass / only the attention kernel on the workspace
    a stage-1 call has filled (svg_band_attention_fp8_stage)."""
    lib = load()
    _dev(q, k, v, head_perm_flag)
    assert q.shape == k.shape == v.shape and q.dtype == k.dtype == v.dtype
    S, D = q.shape[-2], q.shape[-1]
    BH = q.numel() // (S * D)
    o = torch.empty_like(q) if out is None else out
    _dev(o)
    scale = float(sm_scale) if sm_scale is not None else 1.0 / (D ** 0.5)
    perm = None
    if head_perm_flag is not None:
        flag = head_perm_flag.to(torch.int64).contiguous()
        assert flag.numel() == BH
        perm = PermDesc(flag.data_ptr(), vid0, num_frame, frame_size)
    need = int(lib.svg_band_attention_fp8_workspace_bytes(BH, S, D))
    if need == 0:
        raise RuntimeError(f"svg_band_attention_fp8: unsupported shape (D = {D}; only 128)")
    if workspace is None:
        key = WorkspaceCache.key("band", BH, S, device=q.device)
        workspace = _F8_WS.get(key)
        if workspace is None or workspace.numel() < need:
            workspace = _F8_WS[key] = torch.empty(need, dtype=torch.uint8, device=q.device)
    _dev(workspace)
    assert workspace.dtype == torch.uint8 and workspace.numel() >= need
    args = (q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), BH, S, D, _dtype_code(q), scale, C.byref(mask),
            C.byref(perm) if perm is not None else None, workspace.data_ptr(), workspace.numel())
    if stage:
        _check(lib.svg_band_attention_fp8_stage(*args, int(stage), _stream()), "svg_band_attention_fp8_stage")
    else:
        _check(lib.svg_band_attention_fp8(*args, _stream()), "svg_band_attention_fp8")
    return o


def notify_counters(BH: int, nseg: int, device) -> torch.Tensor:
    """Zeroed counter buffer for band_attention(done=...): BH * nseg segment counters + BH words of library scratch."""
    return torch.zeros(BH * (nseg + 1), device=device, dtype=torch.int32)


def band_notify_target(S: int, mask: BandMask) -> int:
    """Value a head's completion counter reaches when the head is done (svg_band_attention_notify_target)."""
    t = load().svg_band_attention_notify_target(int(S), C.byref(mask))
    assert t > 0
    return int(t)


def band_notify_layout(S: int, mask: BandMask, nseg: int):
    """(n, row_bounds [n + 1], targets [n]) of the per-segment completion counters (svg_band_attention_notify_layout):
    counter (h, s) >= targets[s]  =>  the physical rows [row_bounds[s], row_bounds[s + 1]) of head h are complete and visible
    (heads with the fused layout permutation release all their segments together, see include/svg_attn.h); n <= nseg."""
    rb = (C.c_int32 * (nseg + 1))()
    tg = (C.c_int32 * nseg)()
    n = load().svg_band_attention_notify_layout(int(S), C.byref(mask), int(nseg), C.cast(rb, C.c_void_p), C.cast(tg, C.c_void_p))
    assert n > 0
    return n, list(rb)[: n + 1], list(tg)[:n]


def wait_counters(counters: torch.Tensor, target: int, timeout_ms: int = 0, timed_out: Optional[torch.Tensor] = None) -> None:
    """Enqueue, on the current stream, a one-wave kernel that returns once every element of `counters` (int32, GPU) >= target.
    timeout_ms > 0: the kernel gives up after that time and stores 1 to `timed_out` (int32 [1], GPU) — it cannot hang the stream;
    the caller reads the flag before trusting what was queued behind the waiter."""
    _dev(counters)
    assert counters.dtype == torch.int32 and counters.is_contiguous()
    if timeout_ms > 0:
        assert timed_out is not None, "wait_counters(timeout_ms > 0) needs a `timed_out` flag tensor (int32 [1] on the GPU)"
        _dev(timed_out)
        assert timed_out.dtype == torch.int32 and timed_out.numel() >= 1
        _check(load().svg_wait_counters_deadline(counters.data_ptr(), counters.numel(), int(target), int(timeout_ms),
                                                 timed_out.data_ptr(), _stream()), "svg_wait_counters_deadline")
        return
    _check(load().svg_wait_counters(counters.data_ptr(), counters.numel(), int(target), _stream()), "svg_wait_counters")


def band_attention_switch(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, mask: BandMask, alt_mask: BandMask,
                          use_alt_flag: torch.Tensor, sm_scale: Optional[float] = None,
                          head_perm_flag: Optional[torch.Tensor] = None, vid0: int = 0, num_frame: int = 1, frame_size: int = 1,
                          out: Optional[torch.Tensor] = None, q_prescaled: bool = False, token_major_out: bool = False) -> torch.Tensor:
    """band_attention with a device-side switch: `use_alt_flag` (int32 [1] on the GPU) != 0 selects `alt_mask` without the head
    placement, otherwise `mask` with it (svg_band_attention_switch) — no host read of the flag.
    q_prescaled: q carries sm_scale * log2(e) (svg_band_attention_switch_prescaled, D = 128).
    Strided views / token_major_out: as band_attention (svg_band_attention_switch_strided: head_dim 128, plain q)."""
    lib = load()
    _dev(head_perm_flag, use_alt_flag)
    _gpu(q, k, v, out)
    assert q.shape == k.shape == v.shape and q.dtype == k.dtype == v.dtype
    assert use_alt_flag.dtype == torch.int32 and use_alt_flag.numel() >= 1
    S, D = q.shape[-2], q.shape[-1]
    BH = q.numel() // (S * D)
    scale = float(sm_scale) if sm_scale is not None else 1.0 / (D ** 0.5)
    perm = None
    if head_perm_flag is not None:
        flag = head_perm_flag.to(torch.int64).contiguous()
        assert flag.numel() == BH
        perm = PermDesc(flag.data_ptr(), vid0, num_frame, frame_size)
    dense_in = q.is_contiguous() and k.is_contiguous() and v.is_contiguous() and (out is None or out.is_contiguous())
    if (not dense_in or (token_major_out and out is None)) and not q_prescaled and strided_attention_supported(q):
        q4, k4, v4 = _view4(q), _view4(k), _view4(v)
        o4 = _view4(out) if out is not None else (token_major_empty(q4) if token_major_out else torch.empty(q4.shape, dtype=q.dtype, device=q.device))
        if _strided_ok(q4) and _strided_ok(k4, True) and _strided_ok(v4, True) and _strided_ok(o4) and o4.shape == q4.shape:
            lay = attn_layout(q4, k4, v4, o4)
            rc = lib.svg_band_attention_switch_strided(q.data_ptr(), k.data_ptr(), v.data_ptr(), o4.data_ptr(), BH, S, D, _dtype_code(q),
                                                       scale, C.byref(mask), C.byref(perm) if perm is not None else None,
                                                       C.byref(alt_mask), use_alt_flag.data_ptr(), C.byref(lay), _stream())
            _check(rc, "svg_band_attention_switch_strided")
            return out if out is not None else (o4 if q.dim() == 4 else o4.squeeze(0))
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    if out is not None and not out.is_contiguous():
        out.copy_(band_attention_switch(q, k, v, mask, alt_mask, use_alt_flag, sm_scale, head_perm_flag, vid0, num_frame, frame_size, None,
                                        q_prescaled))
        return out
    o = torch.empty_like(q) if out is None else out
    if q_prescaled:
        assert sm_scale is None, "q_prescaled: the scale lives in q"
        rc = lib.svg_band_attention_switch_prescaled(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), BH, S, D, _dtype_code(q),
                                                     C.byref(mask), C.byref(perm) if perm is not None else None, C.byref(alt_mask),
                                                     use_alt_flag.data_ptr(), _stream())
        _check(rc, "svg_band_attention_switch_prescaled")
        return o
    rc = lib.svg_band_attention_switch(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), BH, S, D, _dtype_code(q), scale,
                                       C.byref(mask), C.byref(perm) if perm is not None else None, C.byref(alt_mask),
                                       use_alt_flag.data_ptr(), _stream())
    _check(rc, "svg_band_attention_switch")
    return o


def varblock_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, block_map: torch.Tensor, q_sizes: torch.Tensor,
                       k_sizes: torch.Tensor, sm_scale: Optional[float] = None, q_row_idx: Optional[torch.Tensor] = None,
                       kv_row_idx: Optional[torch.Tensor] = None, variant: int = -1, fp8: bool = False,
                       workspace: Optional[torch.Tensor] = None, token_major_out: bool = False,
                       rows_covered: bool = False) -> torch.Tensor:
    """q: [Hq, Sq, D], k/v: [Hkv, Skv, D] (or [B, H, S, D]: heads = B * H); block_map bool [Hkv, QB, KB]; sizes int32 [Hkv, QB] /
    [Hkv, KB].  -> o of q's shape.
    fp8=True: e4m3 QK^T / PV (svg_varblock_attention_fp8, D = 128, default schedule only).
    workspace: optional uint8 GPU tensor of svg_varblock_workspace_bytes(...) bytes for the 16-bit call (tests read the launch
    order back from it; see varblock_launch_order).
    Strided views / token_major_out: as band_attention (svg_varblock_attention_strided: head_dim 128, variant -1 on block-rows large
    enough for the default body, 16-bit).
    rows_covered: the caller guarantees sum(q_sizes[h]) == Sq for every head (k-means cluster sizes do), so every output row is
    written by the kernel and the output is not zero-filled first (a 2 S H D-byte memset per call otherwise)."""
    lib = load()
    _dev(block_map, q_sizes, k_sizes, q_row_idx, kv_row_idx)
    _gpu(q, k, v)
    Sq, D = q.shape[-2], q.shape[-1]
    Skv = k.shape[-2]
    Hq, Hkv = q.numel() // (Sq * D), k.numel() // (Skv * D)
    QB, KB = q_sizes.shape[-1], k_sizes.shape[-1]
    assert block_map.shape == (Hkv, QB, KB) and block_map.dtype in (torch.bool, torch.uint8)
    assert q_sizes.dtype == torch.int32 and k_sizes.dtype == torch.int32
    if q_row_idx is not None:
        assert q_row_idx.dtype == torch.int32 and q_row_idx.shape == (Hq, Sq)
    if kv_row_idx is not None:
        assert kv_row_idx.dtype == torch.int32 and kv_row_idx.shape == (Hkv, Skv)
    scale = float(sm_scale) if sm_scale is not None else 1.0 / (D ** 0.5)
    dense_in = q.is_contiguous() and k.is_contiguous() and v.is_contiguous()
    if (not dense_in or token_major_out) and not fp8 and variant == -1 and strided_attention_supported(q) and Sq >= 160 * QB:
        q4, k4, v4 = _view4(q), _view4(k), _view4(v)
        o4 = token_major_empty(q4) if token_major_out else torch.empty(q4.shape, dtype=q.dtype, device=q.device)
        if _strided_ok(q4) and _strided_ok(k4, True) and _strided_ok(v4, True) and _strided_ok(o4):
            if not rows_covered:
                o4.zero_()
            need = int(lib.svg_varblock_workspace_bytes(Hq, Hkv, QB, KB, Sq))
            ws = torch.empty(need, dtype=torch.uint8, device=q.device) if workspace is None else workspace
            _dev(ws)
            assert ws.dtype == torch.uint8 and ws.numel() >= need
            lay = attn_layout(q4, k4, v4, o4)
            rc = lib.svg_varblock_attention_strided(q.data_ptr(), k.data_ptr(), v.data_ptr(), o4.data_ptr(), Hq, Hkv, Sq, Skv, D,
                                                    _dtype_code(q), scale, block_map.data_ptr(), q_sizes.data_ptr(), k_sizes.data_ptr(),
                                                    QB, KB, _ptr(q_row_idx), _ptr(kv_row_idx), ws.data_ptr(), ws.numel(), C.byref(lay),
                                                    _stream())
            _check(rc, "svg_varblock_attention_strided")
            return o4 if q.dim() == 4 else o4.squeeze(0)
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()   # (what the strided entry point does not take is copied)
    o = torch.empty_like(q) if rows_covered else torch.zeros_like(q)
    if fp8:
        need = int(lib.svg_varblock_attention_fp8_workspace_bytes(Hq, Hkv, QB, KB, Sq, Skv, D))
        if need == 0:
            raise RuntimeError(f"svg_varblock_attention_fp8: unsupported shape (D = {D}; only 128)")
        key = WorkspaceCache.key("vb", Hq, Hkv, Sq, Skv, QB, KB, device=q.device)
        ws = _F8_WS.get(key)
        if ws is None or ws.numel() < need:
            ws = _F8_WS[key] = torch.empty(need, dtype=torch.uint8, device=q.device)
        rc = lib.svg_varblock_attention_fp8(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), Hq, Hkv, Sq, Skv, D, _dtype_code(q),
                                            scale, block_map.data_ptr(), q_sizes.data_ptr(), k_sizes.data_ptr(), QB, KB,
                                            _ptr(q_row_idx), _ptr(kv_row_idx), ws.data_ptr(), ws.numel(), _stream())
        _check(rc, "svg_varblock_attention_fp8")
        return o
    need = int(lib.svg_varblock_workspace_bytes(Hq, Hkv, QB, KB, Sq))
    ws = torch.empty(need, dtype=torch.uint8, device=q.device) if workspace is None else workspace
    _dev(ws)
    assert ws.dtype == torch.uint8 and ws.numel() >= need
    rc = lib.svg_varblock_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), Hq, Hkv, Sq, Skv, D,
                                    _dtype_code(q), scale, block_map.data_ptr(), q_sizes.data_ptr(), k_sizes.data_ptr(),
                                    QB, KB, _ptr(q_row_idx), _ptr(kv_row_idx), ws.data_ptr(), ws.numel(), variant,
                                    _stream())
    _check(rc, "svg_varblock_attention")
    return o


def varblock_workspace(Hq: int, Hkv: int, QB: int, KB: int, Sq: int, device) -> torch.Tensor:
    return torch.zeros(int(load().svg_varblock_workspace_bytes(Hq, Hkv, QB, KB, Sq)), dtype=torch.uint8, device=device)


def varblock_launch_order(workspace: torch.Tensor, Hkv: int, QB: int, KB: int):
    """The launch order a 256-row variable-block call (variants 3 / 6 / 7) left in its workspace: int32 [n, 3] rows of
    (q head, block-row << 16 | sub-tile, partner) in dispatch order — partner >= 0: the tile also carries the ragged last tile of
    that block-row (remainder packing) — (layout: csrc/attention.hip run_varblock: plan prefix sums, two buckets + the partner per
    block-row, histogram, then [count, pad, entries])."""
    w = workspace.view(torch.int32)
    off = Hkv * (3 * (QB + 1) + (KB + 1)) + 3 * Hkv * QB + Hkv * 64
    n = int(w[off].item())
    return w[off + 2: off + 2 + 3 * n].view(n, 3).clone()


def varblock_partners(workspace: torch.Tensor, Hkv: int, QB: int, KB: int) -> torch.Tensor:
    """The remainder-packing partners a variant-3 call left in its workspace: int32 [Hkv, QB]; j >= 0: block-row i's ragged last tile
    also carries block-row j's, -2: carried by its partner, -1: alone (csrc/attention.hip varblock_pair_*_kernel)."""
    w = workspace.view(torch.int32)
    off = Hkv * (3 * (QB + 1) + (KB + 1)) + 2 * Hkv * QB
    return w[off: off + Hkv * QB].view(Hkv, QB).clone()


class ClockProbe:
    """Sustained shader clock over a span of GPU work (svg_debug_clock_probe): `start()` launches the one-wave probe on its own
    stream, `arm_stop()` raises its flag from a third stream behind the work enqueued so far, `result()` waits for the probe and returns
    MHz (None if the counters did not move)."""

    def __init__(self, device):
        self.dev = torch.device(device)
        self.flag = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.out = torch.zeros(2, dtype=torch.int64, device=self.dev)
        self.s_probe = torch.cuda.Stream(device=self.dev)
        self.s_flag = torch.cuda.Stream(device=self.dev)

    def start(self, max_ms: int = 20000) -> None:
        self.flag.zero_()
        self.out.zero_()
        torch.cuda.synchronize(self.dev)
        with torch.cuda.stream(self.s_probe):
            _check(load().svg_debug_clock_probe(self.flag.data_ptr(), self.out.data_ptr(), int(max_ms), _stream()),
                   "svg_debug_clock_probe")

    def arm_stop(self) -> None:
        """Raise the flag once everything enqueued so far on the CURRENT stream has finished (stream-ordered, no host wait) — call it
        right after the last launch of the span, BEFORE any device-wide synchronisation (which would wait for the probe itself)."""
        self.s_flag.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.s_flag):
            self.flag.fill_(1)

    def result(self):
        self.s_probe.synchronize()
        c, w = (int(x) for x in self.out.tolist())
        return round(100.0 * c / w, 1) if w > 0 and c > 0 else None


def clear_workspace_cache() -> None:
    """Drop the cached workspaces (band_attention_fp8 / varblock_attention(fp8=True) / kmeans_loop keep one per shape, device and stream,
    at most WorkspaceCache.capacity of each kind)."""
    _F8_WS.clear()
    _KMEANS_WS.clear()
    from . import kmeans_utils as _ku   # (its KMeansState cache of the per-iteration path follows the same policy)

    _ku._STATE_CACHE.clear()


def sample_mse(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, rows: torch.Tensor, prof: ProfileDesc,
               sm_scale: Optional[float] = None, skip_flag: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q,k,v [BH, S, D] (or [B, H, S, D]: BH = B * H); rows int64 [R] (device) -> mse float32 [2, BH].  skip_flag (int32 [1] on the
    GPU) != 0: the kernels return at once and the result is undefined (a dense step does not use it).
    Strided views are read in place by svg_sample_mse_strided (bf16: the second form of the kernel) and copied otherwise."""
    lib = load()
    _dev(rows, skip_flag)
    _gpu(q, k, v)
    S, D = q.shape[-2], q.shape[-1]
    BH = q.numel() // (S * D)
    R = rows.numel()
    out = torch.empty((2, BH), dtype=torch.float32, device=q.device)
    ws = torch.empty(lib.svg_sample_mse_workspace_bytes(BH, R, D, S), dtype=torch.uint8, device=q.device)
    scale = float(sm_scale) if sm_scale is not None else 1.0 / (D ** 0.5)
    if not (q.is_contiguous() and k.is_contiguous() and v.is_contiguous()) and q.dtype == torch.bfloat16:
        q4, k4, v4 = _view4(q), _view4(k), _view4(v)
        if _strided_ok(q4) and _strided_ok(k4, True) and _strided_ok(v4, True):
            if skip_flag is not None:
                assert skip_flag.dtype == torch.int32
                out.zero_()
            lay = attn_layout(q4, k4, v4, q4)
            rc = lib.svg_sample_mse_strided(q.data_ptr(), k.data_ptr(), v.data_ptr(), rows.data_ptr(), R, BH, S, D, _dtype_code(q), scale,
                                            C.byref(prof), out.data_ptr(), ws.data_ptr(), ws.numel(), _ptr(skip_flag), C.byref(lay),
                                            _stream())
            _check(rc, "svg_sample_mse_strided")
            return out
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    if skip_flag is not None:
        assert skip_flag.dtype == torch.int32
        out.zero_()
        rc = lib.svg_sample_mse_flagged(q.data_ptr(), k.data_ptr(), v.data_ptr(), rows.data_ptr(), R, BH, S, D, _dtype_code(q),
                                        scale, C.byref(prof), out.data_ptr(), ws.data_ptr(), ws.numel(), skip_flag.data_ptr(),
                                        _stream())
        _check(rc, "svg_sample_mse_flagged")
        return out
    rc = lib.svg_sample_mse(q.data_ptr(), k.data_ptr(), v.data_ptr(), rows.data_ptr(), R, BH, S, D, _dtype_code(q), scale,
                            C.byref(prof), out.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "svg_sample_mse")
    return out


def kmeans_xsq(x: torch.Tensor) -> torch.Tensor:
    lib = load()
    _dev(x)
    B, N, D = x.shape
    xsq = torch.empty((B, N), dtype=torch.float32, device=x.device)
    _check(lib.svg_kmeans_xsq(x.data_ptr(), xsq.data_ptr(), B, N, D, _dtype_code(x), _stream()), "svg_kmeans_xsq")
    return xsq


class KmeansBuffers:
    """Device buffers reused across Lloyd iterations (labels, counts, sorted indices, shift, workspace)."""

    def __init__(self, B: int, N: int, K: int, D: int, device):
        lib = load()
        self.labels = torch.empty((B, N), dtype=torch.int32, device=device)
        self.counts = torch.empty((B, K), dtype=torch.int32, device=device)
        self.sorted_idx = torch.empty((B, N), dtype=torch.int32, device=device)
        self.shift = torch.empty((B,), dtype=torch.float32, device=device)
        self.ws = torch.empty(lib.svg_kmeans_workspace_bytes(B, N, K, D), dtype=torch.uint8, device=device)


def kmeans_iter(x: torch.Tensor, xsq: Optional[torch.Tensor], c_in: torch.Tensor, c_out: torch.Tensor, buf: KmeansBuffers) -> None:
    """xsq: the reference's x_sq or None (the assignment kernel does not read it)"""
    lib = load()
    _dev(x, xsq, c_in, c_out)
    B, N, D = x.shape
    K = c_in.shape[1]
    assert c_in.shape == (B, K, D) and c_out.shape == (B, K, D) and c_in.dtype == x.dtype == c_out.dtype
    rc = lib.svg_kmeans_iter(x.data_ptr(), _ptr(xsq), c_in.data_ptr(), c_out.data_ptr(), buf.labels.data_ptr(),
                             buf.counts.data_ptr(), buf.sorted_idx.data_ptr(), buf.shift.data_ptr(), B, N, K, D,
                             _dtype_code(x), buf.ws.data_ptr(), buf.ws.numel(), _stream())
    _check(rc, "svg_kmeans_iter")


def kmeans_assign(x: torch.Tensor, centroids: torch.Tensor) -> torch.Tensor:
    """labels int32 [B, N] of the nearest centroid (lowest index on ties): the assignment half of kmeans_iter (svg_kmeans_assign)"""
    lib = load()
    _dev(x, centroids)
    B, N, D = x.shape
    K = centroids.shape[1]
    assert centroids.shape == (B, K, D) and centroids.dtype == x.dtype and x.is_contiguous() and centroids.is_contiguous()
    labels = torch.empty((B, N), dtype=torch.int32, device=x.device)
    ws = torch.empty(lib.svg_kmeans_workspace_bytes(B, N, K, D), dtype=torch.uint8, device=x.device)
    _check(lib.svg_kmeans_assign(x.data_ptr(), centroids.data_ptr(), labels.data_ptr(), B, N, K, D, _dtype_code(x), ws.data_ptr(),
                                 ws.numel(), _stream()), "svg_kmeans_assign")
    return labels


def kmeans_update(x: torch.Tensor, labels: torch.Tensor, centroids_in: torch.Tensor):
    """The update half on given labels (svg_kmeans_update) -> (centroids [B, K, D] of x.dtype, counts int32 [B, K], sorted_idx int32
    [B, N], shift float32 [B]); empty clusters keep centroids_in."""
    lib = load()
    _dev(x, labels, centroids_in)
    B, N, D = x.shape
    K = centroids_in.shape[1]
    assert labels.dtype == torch.int32 and labels.shape == (B, N) and labels.is_contiguous()
    assert centroids_in.shape == (B, K, D) and centroids_in.dtype == x.dtype and x.is_contiguous() and centroids_in.is_contiguous()
    cent = torch.empty_like(centroids_in)
    counts = torch.empty((B, K), dtype=torch.int32, device=x.device)
    sorted_idx = torch.empty((B, N), dtype=torch.int32, device=x.device)
    shift = torch.empty((B,), dtype=torch.float32, device=x.device)
    ws = torch.empty(lib.svg_kmeans_workspace_bytes(B, N, K, D), dtype=torch.uint8, device=x.device)
    _check(lib.svg_kmeans_update(x.data_ptr(), labels.data_ptr(), centroids_in.data_ptr(), cent.data_ptr(), counts.data_ptr(),
                                 sorted_idx.data_ptr(), shift.data_ptr(), B, N, K, D, _dtype_code(x), ws.data_ptr(), ws.numel(), _stream()),
           "svg_kmeans_update")
    return cent, counts, sorted_idx, shift


def kmeans_loop(x: torch.Tensor, xsq: Optional[torch.Tensor], c_init: torch.Tensor, max_iters: int, tol: float, work=None):
    """The whole Lloyd loop as one library call without host synchronisation (svg_kmeans_loop) -> (labels int32 [B, N], centroids
    [B, K, D], counts int32 [B, K], n_iters int32 [] on the device, sorted_idx int32 [B, N]).  `work`: a WorkspaceCache (or None: fresh
    scratch per call) that keeps the scratch tensors of a (B, N, K, D) shape per device AND stream between calls."""
    lib = load()
    _dev(xsq, c_init)
    _gpu(x)
    B, N, D = x.shape
    K = c_init.shape[1]
    assert c_init.shape == (B, K, D) and c_init.dtype == x.dtype and c_init.is_contiguous()
    # x: contiguous, or batches further apart than N * D with contiguous rows inside (the video tokens `q[:, :V]` of a [H, S, D] tensor):
    # svg_kmeans_loop_strided reads them in place
    batch_strided = (not x.is_contiguous()) and x.stride(2) == 1 and x.stride(1) == D and x.stride(0) >= N * D and x.stride(0) % 8 == 0 \
        and x.data_ptr() % 16 == 0
    if not x.is_contiguous() and not batch_strided:
        x = x.contiguous()
    key = WorkspaceCache.key("kmeans", B, N, K, D, x.dtype, device=x.device)
    w = None if work is None else work.get(key)
    if w is None:
        w = dict(ca=torch.empty_like(c_init), cb=torch.empty_like(c_init),
                 ws=torch.empty(lib.svg_kmeans_loop_workspace_bytes(B, N, K, D), dtype=torch.uint8, device=x.device))
        if work is not None:
            work[key] = w
    labels = torch.empty((B, N), dtype=torch.int32, device=x.device)
    sorted_idx = torch.empty((B, N), dtype=torch.int32, device=x.device)
    counts = torch.empty((B, K), dtype=torch.int32, device=x.device)
    cent = torch.empty_like(c_init)
    n_it = torch.zeros((), dtype=torch.int32, device=x.device)
    if batch_strided:
        rc = lib.svg_kmeans_loop_strided(x.data_ptr(), int(x.stride(0)), c_init.data_ptr(), w["ca"].data_ptr(), w["cb"].data_ptr(),
                                         labels.data_ptr(), counts.data_ptr(), sorted_idx.data_ptr(), cent.data_ptr(), n_it.data_ptr(), B, N, K, D,
                                         _dtype_code(x), int(max_iters), float(tol), w["ws"].data_ptr(), w["ws"].numel(), _stream())
        _check(rc, "svg_kmeans_loop_strided")
        return labels, cent, counts, n_it, sorted_idx
    rc = lib.svg_kmeans_loop(x.data_ptr(), _ptr(xsq), c_init.data_ptr(), w["ca"].data_ptr(), w["cb"].data_ptr(), labels.data_ptr(),
                             counts.data_ptr(), sorted_idx.data_ptr(), cent.data_ptr(), n_it.data_ptr(), B, N, K, D, _dtype_code(x),
                             int(max_iters), float(tol), w["ws"].data_ptr(), w["ws"].numel(), _stream())
    _check(rc, "svg_kmeans_loop")
    return labels, cent, counts, n_it, sorted_idx


def identify_dynamic_map(qc: torch.Tensor, kc: torch.Tensor, k_sizes: torch.Tensor, top_p: float,
                         preserve_length: int) -> torch.Tensor:
    """qc [BH, QC, D], kc [BH, KC, D], k_sizes int32 [BH, KC] -> bool [BH, QC, KC]"""
    lib = load()
    _dev(qc, kc, k_sizes)
    BH, QC, D = qc.shape
    KC = kc.shape[1]
    assert k_sizes.dtype == torch.int32 and k_sizes.shape == (BH, KC)
    out = torch.empty((BH, QC, KC), dtype=torch.uint8, device=qc.device)
    rc = lib.svg_identify_dynamic_map(qc.data_ptr(), kc.data_ptr(), k_sizes.data_ptr(), out.data_ptr(), BH, QC, KC, D,
                                      _dtype_code(qc), float(top_p), int(preserve_length), _stream())
    _check(rc, "svg_identify_dynamic_map")
    return out.view(torch.bool)


def map_density(block_map: torch.Tensor, q_sizes: torch.Tensor, k_sizes: torch.Tensor) -> torch.Tensor:
    lib = load()
    _dev(block_map, q_sizes, k_sizes)
    BH, QB, KB = block_map.shape
    out = torch.empty((BH,), dtype=torch.float32, device=block_map.device)
    rc = lib.svg_map_density(block_map.data_ptr(), q_sizes.data_ptr(), k_sizes.data_ptr(), out.data_ptr(), BH, QB, KB,
                             _stream())
    _check(rc, "svg_map_density")
    return out


def debug_pp_trace():
    """Cycle trace of the ping-pong attention schedule (svg_debug_pp_trace): dict wave -> 8 tick sums, tiles, loop ticks."""
    buf = (C.c_uint64 * 104)()
    torch.cuda.synchronize()
    _check(load().svg_debug_pp_trace(C.cast(buf, C.c_void_p)), "svg_debug_pp_trace")
    v = list(buf)
    return {"waves": [v[8 * w: 8 * w + 8] for w in range(8)], "sv": [v[72 + 4 * w: 72 + 4 * w + 3] for w in range(8)],
            "tiles": v[64], "loop_ticks": v[65]}


def debug_wg_trace(n_workgroups: int):
    """Launch timeline of the traced two-phase kernel (svg_debug_wg_trace): int64 array [n, 6] =
    [entry, loop start, loop end, exit (s_memtime ticks), HW_ID, XCC_ID] per workgroup."""
    import numpy as np

    buf = np.zeros((n_workgroups, 6), dtype=np.uint64)
    torch.cuda.synchronize()
    _check(load().svg_debug_wg_trace(C.c_void_p(buf.ctypes.data), n_workgroups), "svg_debug_wg_trace")
    return buf


# ---- pre-attention prologue (svg/kernels/csrc/ops.h of the reference) ----
def _qk4(q, k):
    assert q.dim() == 4 and k.dim() == 4 and q.is_contiguous() and k.is_contiguous(), "q, k: contiguous [bsz, H, S, D]"
    assert q.shape[0] == k.shape[0] and q.shape[2] == k.shape[2] and q.shape[3] == k.shape[3] and q.dtype == k.dtype
    return q.shape[0], q.shape[1], k.shape[1], q.shape[2], q.shape[3]


def rms_norm_forward(x: torch.Tensor, weight: torch.Tensor, eps: float = 1e-5) -> None:
    """in place; x [m, n] (ref: ops.h:52-78)"""
    lib = load()
    _dev(x, weight)
    assert x.dim() == 2 and x.is_contiguous() and weight.shape == (x.shape[1],) and weight.dtype == x.dtype
    _check(lib.svg_rms_norm_forward(x.data_ptr(), weight.data_ptr(), x.shape[0], x.shape[1], _dtype_code(x), float(eps),
                                    _stream()), "svg_rms_norm_forward")


def layer_norm_forward(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> None:
    """in place; x [m, n], eps = 1e-5 (ref: ops.h:19-44)"""
    lib = load()
    _dev(x, weight, bias)
    assert x.dim() == 2 and x.is_contiguous() and weight.shape == (x.shape[1],) and bias.shape == weight.shape
    assert weight.dtype == x.dtype and bias.dtype == x.dtype
    _check(lib.svg_layer_norm_forward(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), x.shape[0], x.shape[1],
                                      _dtype_code(x), _stream()), "svg_layer_norm_forward")


def _rope(fn_name: str, q, k, a, b, len_text_prompt: int, half: bool) -> None:
    lib = load()
    _dev(q, k, a, b)
    bsz, Hq, Hkv, S, D = _qk4(q, k)
    valid = S - int(len_text_prompt)
    assert valid > 0, "len_text_prompt must be smaller than the sequence"
    cols = D // 2 if half else D
    assert a.dtype == torch.float32 and b.dtype == torch.float32 and a.is_contiguous() and b.is_contiguous()
    assert a.shape == (valid, cols) and b.shape == (valid, cols), f"rope tables must be [{valid}, {cols}]"
    _check(getattr(lib, fn_name)(q.data_ptr(), k.data_ptr(), a.data_ptr(), b.data_ptr(), bsz, Hq, Hkv, S, D, _dtype_code(q),
                                 int(len_text_prompt), _stream()), fn_name)


def apply_qk_rope_inplace_cossin(q, k, cos, sin, len_text_prompt: int) -> None:
    """ref: ops.h:80-136 — the FIRST len_text_prompt positions are skipped"""
    _rope("svg_apply_qk_rope_inplace_cossin", q, k, cos, sin, len_text_prompt, False)


def apply_qk_rope_inplace_cossin_txtlast(q, k, cos, sin, len_text_prompt: int) -> None:
    """ref: ops.h:138-196 — the LAST len_text_prompt positions are skipped"""
    _rope("svg_apply_qk_rope_inplace_cossin_txtlast", q, k, cos, sin, len_text_prompt, False)


def apply_qk_rope_inplace_cossin_complex(q, k, freqs_real, freqs_imag, len_text_prompt: int) -> None:
    """ref: ops.h:198-260 — complex multiply in fp64, tables [S - len_text_prompt, D / 2]"""
    _rope("svg_apply_qk_rope_inplace_cossin_complex", q, k, freqs_real, freqs_imag, len_text_prompt, True)


def qk_norm_rope(q, k=None, norm_kind: int = 0, q_weight=None, q_bias=None, k_weight=None, k_bias=None, eps: float = 1e-5,
                 rope_kind: int = 0, cos=None, sin=None, rope_lo: int = 0, rope_hi: Optional[int] = None,
                 q_scale: float = 1.0) -> None:
    """Fused in-place normalisation + rotary embedding of q and (optionally) k in one pass (svg_qk_norm_rope).
    q_scale != 1: folded into the last rounding of q (svg_qk_norm_rope_qscale; softmax_q_scale(D) for band_attention(q_prescaled=True))."""
    lib = load()
    _dev(q, k, q_weight, q_bias, k_weight, k_bias, cos, sin)
    if k is not None:
        bsz, Hq, Hkv, S, D = _qk4(q, k)
    else:
        assert q.dim() == 4 and q.is_contiguous()
        (bsz, Hq, S, D), Hkv = q.shape, 0
    rope_hi = S if rope_hi is None else rope_hi
    if rope_kind:
        cols = D // 2 if rope_kind == 2 else D
        assert cos.dtype == torch.float32 and sin.dtype == torch.float32 and cos.is_contiguous() and sin.is_contiguous()
        assert cos.shape == (rope_hi - rope_lo, cols) and sin.shape == cos.shape
    for w in (q_weight, q_bias, k_weight, k_bias):
        assert w is None or (w.dtype == q.dtype and w.shape == (D,) and w.is_contiguous())
    _check(lib.svg_qk_norm_rope_qscale(q.data_ptr(), _ptr(k), bsz, Hq, Hkv, S, D, _dtype_code(q), int(norm_kind), _ptr(q_weight),
                                       _ptr(q_bias), _ptr(k_weight), _ptr(k_bias), float(eps), int(rope_kind), _ptr(cos), _ptr(sin),
                                       int(rope_lo), int(rope_hi), float(q_scale), _stream()), "svg_qk_norm_rope")


def qk_norm_rope_transpose(q_in, k_in, heads_q: int, heads_k: int, norm_kind: int = 0, q_weight=None, q_bias=None, k_weight=None,
                           k_bias=None, eps: float = 1e-5, rope_kind: int = 0, cos=None, sin=None, rope_lo: int = 0,
                           rope_hi: Optional[int] = None, q_scale: float = 1.0):
    """q_in [bsz, S, Hq * D] (k_in [bsz, S, Hkv * D] or None) token-major -> new head-major tensors [bsz, H, S, D] with
    normalisation + rotary embedding applied in the same pass (svg_qk_norm_rope_transpose); q_scale as in qk_norm_rope."""
    lib = load()
    _dev(q_in, k_in, q_weight, q_bias, k_weight, k_bias, cos, sin)
    bsz, S, HD = q_in.shape
    D = HD // heads_q
    q_out = torch.empty((bsz, heads_q, S, D), dtype=q_in.dtype, device=q_in.device)
    k_out = None
    if k_in is not None:
        assert k_in.shape == (bsz, S, heads_k * D) and k_in.dtype == q_in.dtype
        k_out = torch.empty((bsz, heads_k, S, D), dtype=q_in.dtype, device=q_in.device)
    rope_hi = S if rope_hi is None else rope_hi
    if rope_kind:
        cols = D // 2 if rope_kind == 2 else D
        assert cos.dtype == torch.float32 and sin.dtype == torch.float32 and cos.shape == (rope_hi - rope_lo, cols) and sin.shape == cos.shape
    _check(lib.svg_qk_norm_rope_transpose_qscale(q_in.data_ptr(), _ptr(k_in), q_out.data_ptr(), _ptr(k_out), bsz, heads_q, heads_k, S,
                                                 D, _dtype_code(q_in), int(norm_kind), _ptr(q_weight), _ptr(q_bias), _ptr(k_weight),
                                                 _ptr(k_bias), float(eps), int(rope_kind), _ptr(cos), _ptr(sin), int(rope_lo),
                                                 int(rope_hi), float(q_scale), _stream()), "svg_qk_norm_rope_transpose")
    return q_out, k_out


def rmsnorm_rope_transpose(q_in, k_in, v_in, heads: int, q_weight=None, k_weight=None, eps: float = 1e-6, rope_kind: int = 0, cos=None,
                           sin=None, rope_lo: int = 0, rope_hi: Optional[int] = None, q_scale: float = 1.0):
    """The Wan 2.1 prologue in one pass (svg_rmsnorm_rope_transpose): q_in / k_in / v_in [bsz, S, heads * D] token-major (k_in, v_in may be
    None) -> (q, k, v) head-major [bsz, heads, S, D]; q and k RMS-normalised ACROSS ALL HEADS (the reference's Triton form, weights [heads * D])
    and rotated, v transposed.  Bit-identical to rmsnorm_forward -> qk_norm_rope_transpose(norm 0, rope) and a transpose of v."""
    lib = load()
    _dev(q_in, k_in, v_in, q_weight, k_weight, cos, sin)
    bsz, S, HD = q_in.shape
    D = HD // heads
    assert q_in.is_contiguous() and all(t is None or (t.shape == q_in.shape and t.dtype == q_in.dtype and t.is_contiguous()) for t in (k_in, v_in))
    mk = lambda t: None if t is None else torch.empty((bsz, heads, S, D), dtype=q_in.dtype, device=q_in.device)  # noqa: E731
    q_out, k_out, v_out = mk(q_in), mk(k_in), mk(v_in)
    w = q_weight if q_weight is not None else k_weight
    wdt = _GLUE_DT[w.dtype] if w is not None else 2
    for t in (q_weight, k_weight):
        assert t is None or (t.shape == (HD,) and t.is_contiguous() and _GLUE_DT[t.dtype] == wdt)
    rope_hi = S if rope_hi is None else rope_hi
    if rope_kind:
        cols = D // 2 if rope_kind == 2 else D
        assert cos.dtype == torch.float32 and sin.dtype == torch.float32 and cos.shape == (rope_hi - rope_lo, cols) and sin.shape == cos.shape
    _check(lib.svg_rmsnorm_rope_transpose(q_in.data_ptr(), _ptr(k_in), _ptr(v_in), q_out.data_ptr(), _ptr(k_out), _ptr(v_out), bsz, heads, S, D,
                                          _dtype_code(q_in), _ptr(q_weight), _ptr(k_weight), wdt, float(eps), int(rope_kind), _ptr(cos),
                                          _ptr(sin), int(rope_lo), int(rope_hi), float(q_scale), _stream()), "svg_rmsnorm_rope_transpose")
    return q_out, k_out, v_out


# ---- transformer-block glue (svg/kernels/triton/{layernorm,modulate}.py of the reference) ----
_GLUE_DT = {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}


def _rows(x: torch.Tensor):
    assert x.dim() in (2, 3) and x.is_contiguous(), "expected contiguous [M, N] or [B, S, N]"
    N = x.shape[-1]
    return x.numel() // N, N, (x.shape[1] if x.dim() == 3 else x.numel() // N)


def _per_batch(t: torch.Tensor, x: torch.Tensor, N: int) -> torch.Tensor:
    """scale / shift / gate: fp32, one row of N per batch element ([N], [1, N], [B, 1, N], [B, N])"""
    B = x.shape[0] if x.dim() == 3 else 1
    t = t.detach().to(torch.float32).reshape(-1, N)
    if t.shape[0] == 1 and B > 1:
        t = t.expand(B, N)
    assert t.shape == (B, N), f"modulation tensor must broadcast to [{B}, {N}]"
    return t.contiguous()


def layernorm_forward(x, weight=None, bias=None, eps: float = 1e-5, out_dtype=torch.float32, reference_padding: bool = False) -> torch.Tensor:
    """fp32 LayerNorm of the rows of x.  reference_padding: the variance of the reference's Triton kernels as they are — the zero
    padding of a row up to the next power of two counted in (svg_layernorm_forward_ex; include/svg_attn.h) — instead of LayerNorm's."""
    lib = load()
    _dev(x, weight, bias)
    M, N, _ = _rows(x)
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    wdt = _GLUE_DT[weight.dtype] if weight is not None else 2
    if weight is not None:
        assert bias is not None and weight.shape == (N,) and bias.shape == (N,) and bias.dtype == weight.dtype
    _check(lib.svg_layernorm_forward_ex(x.data_ptr(), y.data_ptr(), _ptr(weight), _ptr(bias), M, N, _GLUE_DT[x.dtype],
                                        _GLUE_DT[out_dtype], wdt, float(eps), int(bool(reference_padding)), _stream()),
           "svg_layernorm_forward_ex")
    return y


def rmsnorm_forward(x, weight=None, eps: float = 1e-6, out_dtype=None) -> torch.Tensor:
    """RMSNorm of the rows of x ([M, N] or [B, S, N]) in the reference's Triton form: fp32, x * rstd * w, one rounding
    (svg_rmsnorm_forward); out_dtype defaults to x.dtype."""
    lib = load()
    _dev(x, weight)
    M, N, _ = _rows(x)
    out_dtype = x.dtype if out_dtype is None else out_dtype
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    wdt = _GLUE_DT[weight.dtype] if weight is not None else 2
    if weight is not None:
        assert weight.shape == (N,) and weight.is_contiguous()
    _check(lib.svg_rmsnorm_forward(x.data_ptr(), y.data_ptr(), _ptr(weight), M, N, _GLUE_DT[x.dtype], _GLUE_DT[out_dtype], wdt,
                                   float(eps), _stream()), "svg_rmsnorm_forward")
    return y


def modulate_shift_forward(x, scale, shift, out_dtype=torch.float32) -> torch.Tensor:
    lib = load()
    _dev(x)
    M, N, rpb = _rows(x)
    sc, sh = _per_batch(scale, x, N), _per_batch(shift, x, N)
    _dev(sc, sh)
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    _check(lib.svg_modulate_shift_forward(x.data_ptr(), y.data_ptr(), sc.data_ptr(), sh.data_ptr(), M, N, rpb, _GLUE_DT[x.dtype],
                                          _GLUE_DT[out_dtype], _stream()), "svg_modulate_shift_forward")
    return y


def modulate_gate_residual_forward(residual, x, gate, out_dtype=torch.float32) -> torch.Tensor:
    lib = load()
    _dev(residual, x)
    assert residual.shape == x.shape
    M, N, rpb = _rows(x)
    _rows(residual)
    g = _per_batch(gate, x, N)
    _dev(g)
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    _check(lib.svg_modulate_gate_residual_forward(residual.data_ptr(), x.data_ptr(), g.data_ptr(), y.data_ptr(), M, N, rpb,
                                                  _GLUE_DT[residual.dtype], _GLUE_DT[x.dtype], _GLUE_DT[out_dtype], _stream()),
           "svg_modulate_gate_residual_forward")
    return y


def layernorm_modulate_forward(x, weight=None, bias=None, scale=None, shift=None, eps: float = 1e-5, out_dtype=None,
                               reference_padding: bool = False) -> torch.Tensor:
    """Fused fp32 LayerNorm (+ affine) (+ y * (1 + scale) + shift), one pass; out_dtype defaults to x.dtype.
    reference_padding: see layernorm_forward."""
    lib = load()
    _dev(x, weight, bias)
    M, N, rpb = _rows(x)
    out_dtype = x.dtype if out_dtype is None else out_dtype
    sc = _per_batch(scale, x, N) if scale is not None else None
    sh = _per_batch(shift, x, N) if shift is not None else None
    _dev(sc, sh)
    wdt = _GLUE_DT[weight.dtype] if weight is not None else 2
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    _check(lib.svg_layernorm_modulate_forward_ex(x.data_ptr(), y.data_ptr(), _ptr(weight), _ptr(bias), _ptr(sc), _ptr(sh), M, N, rpb,
                                                 _GLUE_DT[x.dtype], _GLUE_DT[out_dtype], wdt, float(eps), int(bool(reference_padding)),
                                                 _stream()),
           "svg_layernorm_modulate_forward_ex")
    return y


def bsr_to_block_map(indptr: torch.Tensor, indices: torch.Tensor, MB: int, NB: int, row_block: int, col_block: int,
                     len_text: int, heads: int):
    """BSR over uniform blocks -> (block_map uint8 [heads, MB+1, NB+1], q_sizes int32 [heads, MB+1], k_sizes int32 [heads, NB+1])
    with the text block in front (svg_bsr_to_block_map)."""
    lib = load()
    _dev(indptr, indices)
    assert indptr.dtype == torch.int32 and indices.dtype == torch.int32 and indptr.numel() == MB + 1
    dev = indptr.device
    bm = torch.empty((heads, MB + 1, NB + 1), dtype=torch.uint8, device=dev)
    qs = torch.empty((heads, MB + 1), dtype=torch.int32, device=dev)
    ks = torch.empty((heads, NB + 1), dtype=torch.int32, device=dev)
    _check(lib.svg_bsr_to_block_map(indptr.data_ptr(), indices.data_ptr(), MB, NB, row_block, col_block, len_text, heads,
                                    bm.data_ptr(), qs.data_ptr(), ks.data_ptr(), _stream()), "svg_bsr_to_block_map")
    return bm, qs, ks
