#!/usr/bin/env python3
"""Golden element masks for the uniform-block (BSR) attention ops, produced by the REFERENCE's own mask generators:
`ref_gen_temporal_mask`, `ref_gen_spatial_mask` and `gen_mask_block2element` are extracted (ast) from
/root/reference/svg/kernels/test/test_sparse_attn.py (the file imports flashinfer at module level; only these pure-numpy function
bodies are executed, with `.cuda()` dropped).  Writes tests/golden/bsr_golden.npz (bit-packed masks).
    python tests/golden/make_golden_bsr.py"""
import ast
from pathlib import Path

import numpy as np
import torch

SRC = Path("/root/reference/svg/kernels/test/test_sparse_attn.py")
OUT = Path(__file__).resolve().parent / "bsr_golden.npz"
NAMES = ["ref_gen_temporal_mask", "ref_gen_spatial_mask", "gen_mask_block2element"]


def main():
    src = SRC.read_text().replace(".cuda()", "")
    ns = {"np": np, "torch": torch, "Tuple": tuple}
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name in NAMES:
            exec(compile(ast.Module([node], []), str(SRC), "exec"), ns)
    out = {}
    P = 40   # tokens per frame (the reference uses 3600; the generators are size-agnostic, P % 10 == 0)
    for F, L in [(5, 16), (13, 77), (5, 0)]:
        for mul in (0.5, 1, 1.4, 1.8):
            bm = ns["ref_gen_temporal_mask"](F, P, mul)
            em = ns["gen_mask_block2element"](bm, (P // 10, P // 10), L)
            out[f"temporal_{F}_{L}_{mul}"] = np.packbits(em.numpy())
        for mul in (0, 1, 2):
            bm = ns["ref_gen_spatial_mask"](F, P, mul)
            em = ns["gen_mask_block2element"](bm, (P, P), L)
            out[f"spatial_{F}_{L}_{mul}"] = np.packbits(em.numpy())
    # the reference's own grid (test_sparse_attn.py:160-161, 205-206): F = 21, P = 3600 — block masks only (the element mask is
    # 75616^2); tests/test_gpu_bsr.py expands the rows it checks
    out["blk_spatial_21_3600_2"] = (ns["ref_gen_spatial_mask"](21, 3600, 2) >= 0)
    out["blk_temporal_21_3600_1.8"] = (ns["ref_gen_temporal_mask"](21, 3600, 1.8) >= 0)
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} ({OUT.stat().st_size / 1024:.0f} KB, {len(out)} masks)")


if __name__ == "__main__":
    main()
