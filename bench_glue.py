#!/usr/bin/env python3
"""Secondary benchmark: Wan 2.1 720p block glue on one MI355X, hidden states [1, 75600, 5120] bf16.
HBM-bound: fused LayerNorm + modulate = 4 B per element (read 2, write 2); the reference's two kernels move 12 B per element
(fp32 intermediate); gate-residual = 6 B per element.  Prints one JSON line (HIP events).
    python bench_glue.py [--steps K] [--warmup W]"""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    from svg import _native as nat
    nat.load()
    dev = torch.device("cuda", 0)
    B, S, N = 1, 75600, 5120
    x = torch.randn(B, S, N, device=dev, dtype=torch.bfloat16)
    att = torch.randn(B, S, N, device=dev, dtype=torch.bfloat16)
    sc, sh, g = (torch.randn(B, 1, N, device=dev) * 0.2 for _ in range(3))

    def timeit(fn):
        for _ in range(a.warmup):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.steps

    t_f = timeit(lambda: nat.layernorm_modulate_forward(x, None, None, sc, sh, 1e-6))
    t_2 = timeit(lambda: nat.modulate_shift_forward(nat.layernorm_forward(x, None, None, 1e-6), sc, sh, torch.bfloat16))
    t_g = timeit(lambda: nat.modulate_gate_residual_forward(x, att, g, torch.bfloat16))
    t_t = timeit(lambda: (torch.nn.functional.layer_norm(x.float(), (N,), None, None, 1e-6) * (1 + sc) + sh).to(torch.bfloat16))
    el = B * S * N
    out = {"metric": "block_glue_GBps", "value": round(4 * el / t_f / 1e6, 1), "unit": "GB/s", "n_gpus": 1, "steps": a.steps,
           "warmup": a.warmup, "ms_per_step": round(t_f, 4), "higher_is_better": True, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": f"Wan 2.1 720p fused fp32 LayerNorm + modulate, hidden [{B}, {S}, {N}] bf16"},
           "roofline": {"bound": "hbm", "kernel": "row_glue_kernel<10>", "achieved": round(4 * el / t_f / 1e6, 1), "peak": 8000.0,
                        "unit": "GB/s", "frac": round(4 * el / t_f / 1e6 / 8000.0, 4), "algorithmic_bytes": 4 * el, "traffic": None},
           "two_kernels_like_reference": {"ms": round(t_2, 4), "bytes_moved": 12 * el, "GBps_moved": round(12 * el / t_2 / 1e6, 1)},
           "torch_eager": {"ms": round(t_t, 4)},
           "gate_residual": {"ms": round(t_g, 4), "GBps": round(6 * el / t_g / 1e6, 1)}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
