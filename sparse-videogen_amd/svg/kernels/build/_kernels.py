"""`_kernels` — the five-function native module of the reference (PYBIND11_MODULE, ref: svg/kernels/csrc/ops.cu:4-11),
served by libsvgattn (csrc/prologue.hip) through the C ABI.

Same names, argument order and in-place semantics as the reference's pybind functions (ref: svg/kernels/csrc/ops.h:19-260):
tensors are modified in place, nothing is returned, tensors must be contiguous GPU tensors (a CPU tensor raises
RuntimeError: there is no CPU fallback, exactly like the CUDA extension).  The reference's callers put this directory on
sys.path and `import _kernels` (ref: svg/models/wan/attention.py:43-47, svg/kernels/test/test_rms_norm.py:7); both that and
`from svg.kernels.build import _kernels` work here.
"""
from svg import _native as _nat


def layer_norm_forward(input, gemma, beta):
    """Layer norm with bias and learned weight, eps = 1e-5, in place on input [m, n] (ref: ops.h:19-44)."""
    _nat.layer_norm_forward(input, gemma, beta)


def rms_norm_forward(input, gemma, eps):
    """RMS norm with learned weight, in place on input [m, n] (ref: ops.h:52-78)."""
    _nat.rms_norm_forward(input, gemma, eps)


def apply_qk_rope_inplace_cossin(q, k, cos, sin, len_text_prompt):
    """Rotary embedding with cos / sin caches [S - len_text_prompt, D], in place on q, k [bsz, H, S, D]; the FIRST
    len_text_prompt positions are text and are skipped (ref: ops.h:80-136)."""
    _nat.apply_qk_rope_inplace_cossin(q, k, cos, sin, len_text_prompt)


def apply_qk_rope_inplace_cossin_txtlast(q, k, cos, sin, len_text_prompt):
    """The same with the text modality after the video: the LAST len_text_prompt positions are skipped (ref: ops.h:138-196)."""
    _nat.apply_qk_rope_inplace_cossin_txtlast(q, k, cos, sin, len_text_prompt)


def apply_qk_rope_inplace_cossin_complex(q, k, freqs_real, freqs_imag, len_text_prompt):
    """Complex rotary embedding, tables [S - len_text_prompt, D / 2], multiplied in fp64 (ref: ops.h:198-260)."""
    _nat.apply_qk_rope_inplace_cossin_complex(q, k, freqs_real, freqs_imag, len_text_prompt)
