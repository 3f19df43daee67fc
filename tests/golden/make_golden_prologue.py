#!/usr/bin/env python3
"""Golden vectors for the pre-attention prologue (QK norm + RoPE), produced by the REFERENCE's own torch reference
functions: the `ref_host_*` / `replica_host_*` functions are extracted (ast) from /root/reference/svg/kernels/test/test_*.py —
those files `import _kernels` (the CUDA extension, not buildable here) at module level, so only the function bodies are
executed.  Run in the build container (needs /root/reference); writes tests/golden/prologue_golden.npz.

    python tests/golden/make_golden_prologue.py
"""
import ast
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference/svg/kernels/test")
OUT = Path(__file__).resolve().parent / "prologue_golden.npz"


def load_functions(fname, names):
    src = (REF / fname).read_text()
    tree = ast.parse(src)
    ns = {"torch": torch, "Tuple": tuple, "List": list}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            node.decorator_list = []
            exec(compile(ast.Module([node], []), str(REF / fname), "exec"), ns)
    return [ns[n] for n in names]


def bits(t):  # bf16 / fp16 -> int16 view for lossless storage
    return t.contiguous().view(torch.int16).numpy()


def main():
    (ref_rms, replica_rms) = load_functions("test_rms_norm.py", ["ref_host_rms_norm", "replica_host_rms_norm"])
    (ref_ln,) = load_functions("test_layer_norm.py", ["ref_host_layer_norm"])
    (ref_rope,) = load_functions("test_apply_rope.py", ["ref_host_apply_rope"])
    (ref_rope_last,) = load_functions("test_apply_rope_txtlast.py", ["ref_host_apply_rope"])
    (ref_cplx,) = load_functions("test_apply_rope_complex.py", ["ref_host_apply_rope_complex"])
    g = torch.Generator().manual_seed(20250924)
    out = {}
    # norms: (m, n) pairs from the reference's grid (test_rms_norm.py:38, test_layer_norm.py:32)
    for m, n in [(7, 32), (31, 64), (95, 128), (128, 256)]:
        x = torch.randn(m, n, generator=g).to(torch.bfloat16)
        w = torch.randn(n, generator=g).to(torch.bfloat16)
        b = torch.randn(n, generator=g).to(torch.bfloat16)
        tag = f"{m}x{n}"
        out[f"norm_x_{tag}"], out[f"norm_w_{tag}"], out[f"norm_b_{tag}"] = bits(x), bits(w), bits(b)
        out[f"rms_ref_{tag}"] = bits(ref_rms(x, w).to(torch.bfloat16))          # torch.nn.functional.rms_norm
        out[f"rms_replica_{tag}"] = bits(replica_rms(x, w).to(torch.bfloat16))  # diffusers-style rounding points
        out[f"ln_ref_{tag}"] = bits(ref_ln(x, w, b))
    # rope: (bsz, H, S, D, L) from the reference's grid (test_apply_rope.py:39), smallest sequence length
    for bsz, H, S, D, L in [(1, 2, 151, 64, 15), (2, 1, 151, 128, 35), (1, 1, 151, 256, 77)]:
        tag = f"{bsz}_{H}_{S}_{D}_{L}"
        q = torch.randn(bsz, H, S, D, generator=g).to(torch.bfloat16)
        cos = torch.randn(S - L, D, generator=g)
        sin = torch.randn(S - L, D, generator=g)
        out[f"rope_q_{tag}"], out[f"rope_cos_{tag}"], out[f"rope_sin_{tag}"] = bits(q), cos.numpy(), sin.numpy()
        out[f"rope_first_{tag}"] = bits(ref_rope(q[:, :, L:, :], cos, sin))        # text first: positions L.. rotated
        out[f"rope_last_{tag}"] = bits(ref_rope_last(q[:, :, :-L, :], cos, sin))    # text last: positions ..S-L rotated
        qh = torch.randn(bsz, H, S, D, generator=g).to(torch.float16)
        fr = torch.randn(S - L, D // 2, generator=g)
        fi = torch.randn(S - L, D // 2, generator=g)
        out[f"cplx_q_{tag}"], out[f"cplx_fr_{tag}"], out[f"cplx_fi_{tag}"] = bits(qh), fr.numpy(), fi.numpy()
        out[f"cplx_out_{tag}"] = bits(ref_cplx(qh[:, :, L:, :], torch.complex(fr, fi)))
        qb = qh.to(torch.bfloat16)
        out[f"cplx_out_bf16_{tag}"] = bits(ref_cplx(qb[:, :, L:, :], torch.complex(fr, fi)))
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} ({OUT.stat().st_size / 1024:.0f} KB, {len(out)} arrays)")


if __name__ == "__main__":
    main()
