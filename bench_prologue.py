#!/usr/bin/env python3
"""Secondary benchmark: the pre-attention prologue of one HunyuanVideo 720p layer-call on one MI355X —
in-place QK RMSNorm + text-last RoPE on q, k [1, 24, 119056, 128] bf16.  HBM-bound: algorithmic bytes = every element of q and
k read once and written once = 4 * H * S * D * 2 B = 2.93 GB for the fused pass; the reference's pipeline
(`rms_norm_forward` x2 + `apply_qk_rope_inplace_cossin_txtlast`, svg/models/hyvideo/attention.py:162-178) moves twice that.
Prints one JSON line (HIP events).
    python bench_prologue.py [--steps K] [--warmup W] [--workload hy720p|wan720p|cog]
"""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))

import torch  # noqa: E402

WORK = {  # bsz, H, S, D, text, norm kind, rope kind ('last' | 'first' | 'complex')
    "hy720p": (1, 24, 119056, 128, 256, 1, "last"),
    "wan720p": (1, 40, 75600, 128, 0, 0, "complex"),
    "cog": (2, 48, 17776, 64, 226, 2, "first"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="hy720p", choices=sorted(WORK))
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    from svg import _native as nat

    nat.load()
    dev = torch.device("cuda", 0)
    bsz, H, S, D, L, norm, rope = WORK[a.workload]
    dt = torch.bfloat16
    q = torch.randn(bsz, H, S, D, device=dev, dtype=dt)
    k = torch.randn(bsz, H, S, D, device=dev, dtype=dt)
    w = [torch.randn(D, device=dev).to(dt) for _ in range(4)]
    cols = D // 2 if rope == "complex" else D
    cs, sn = torch.randn(S - L, cols, device=dev), torch.randn(S - L, cols, device=dev)
    lo, hi = (0, S - L) if rope == "last" else (L, S)
    rk = 2 if rope == "complex" else 1

    def fused():
        nat.qk_norm_rope(q, k, norm, w[0], w[1] if norm == 2 else None, w[2], w[3] if norm == 2 else None, 1e-6, rk, cs, sn, lo, hi)

    def separate():
        if norm == 1:
            nat.rms_norm_forward(q.view(-1, D), w[0], 1e-6)
            nat.rms_norm_forward(k.view(-1, D), w[2], 1e-6)
        elif norm == 2:
            nat.layer_norm_forward(q.view(-1, D), w[0], w[1])
            nat.layer_norm_forward(k.view(-1, D), w[2], w[3])
        {"last": nat.apply_qk_rope_inplace_cossin_txtlast, "first": nat.apply_qk_rope_inplace_cossin,
         "complex": nat.apply_qk_rope_inplace_cossin_complex}[rope](q, k, cs, sn, L)

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

    # the whole pre-attention data movement of a layer-call: projection outputs [bsz, S, H*D] -> head-major q, k (norm + rope) and v
    qin, kin, vin = (torch.randn(bsz, S, H * D, device=dev, dtype=dt) for _ in range(3))

    def transposing():
        nat.qk_norm_rope_transpose(qin, kin, H, H, norm, w[0], w[1] if norm == 2 else None, w[2], w[3] if norm == 2 else None,
                                   1e-6, rk, cs, sn, lo, hi)
        nat.qk_norm_rope_transpose(vin, None, H, 0)

    def reference_pipeline():
        qq, kk, _ = (x.unflatten(2, (H, D)).transpose(1, 2).contiguous() for x in (qin, kin, vin))
        if norm == 1:
            nat.rms_norm_forward(qq.view(-1, D), w[0], 1e-6)
            nat.rms_norm_forward(kk.view(-1, D), w[2], 1e-6)
        elif norm == 2:
            nat.layer_norm_forward(qq.view(-1, D), w[0], w[1])
            nat.layer_norm_forward(kk.view(-1, D), w[2], w[3])
        {"last": nat.apply_qk_rope_inplace_cossin_txtlast, "first": nat.apply_qk_rope_inplace_cossin,
         "complex": nat.apply_qk_rope_inplace_cossin_complex}[rope](qq, kk, cs, sn, L)

    t_f, t_s = timeit(fused), timeit(separate)
    t_t, t_r = timeit(transposing), timeit(reference_pipeline)
    qk_bytes = 2 * 2 * bsz * H * S * D * 2             # q and k, read + write
    table_bytes = 2 * (S - L) * cols * 4
    alg = qk_bytes + table_bytes
    out = {
        "metric": "qk_prologue_GBps", "value": round(alg / t_f / 1e6, 1), "unit": "GB/s", "n_gpus": 1, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(t_f, 4), "higher_is_better": True, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{a.workload} QK prologue: norm kind {norm} + rope {rope}, bsz={bsz} H={H} S={S} D={D} text={L}, fused one-pass svg_qk_norm_rope"},
        "roofline": {"bound": "hbm", "kernel": f"qk_prologue_kernel<bf16,{D}>", "achieved": round(alg / t_f / 1e6, 1), "peak": 8000.0,
                     "unit": "GB/s", "frac": round(alg / t_f / 1e6 / 8000.0, 4), "algorithmic_bytes": alg, "traffic": None},
        "separate_passes": {"ms": round(t_s, 4), "calls": "norm(q), norm(k), rope(q, k)" if norm else "rope(q, k)",
                            "bytes_moved": (2 if norm else 1) * qk_bytes + table_bytes,
                            "GBps_moved": round(((2 if norm else 1) * qk_bytes + table_bytes) / t_s / 1e6, 1)},
        "speedup_fused_vs_separate": round(t_s / t_f, 3),
        "with_transpose": {"ms_one_pass": round(t_t, 4), "GBps": round(3 * qk_bytes / 2 / t_t / 1e6, 1),
                           "ms_reference_pipeline": round(t_r, 4),
                           "what": "q, k, v [bsz, S, H*D] -> [bsz, H, S, D] incl. norm + rope: svg_qk_norm_rope_transpose x2 vs "
                                   "3 x transpose().contiguous() + the reference's three prologue calls", "speedup": round(t_r / t_t, 3)},
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
