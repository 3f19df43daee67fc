#!/usr/bin/env python3
"""Per-shape GEMM rates of the two denoise-step stacks (bench_step.py) with torch's default hipBLASLt heuristics and with TunableOp's pick
among the hipBLASLt / rocBLAS solutions (PYTORCH_TUNABLEOP_*): what the "other 40 %" of a step could gain from solution selection alone.
No GEMM kernel is written here — the step's GEMMs stay library calls; this only measures which library solution they should take.

    python tools/tune_gemms.py [--out sparse-videogen_amd/tuning/tunableop_mi355x.csv] [--shapes hy,wan]

Prints one JSON line per (shape, op): {"shape": "MxNxK", "op": "mm|addmm|addmm_gelu", "default_tflops", "tuned_tflops"} and writes the results file
TunableOp reads back (PYTORCH_TUNABLEOP_FILENAME; validated by TunableOp against the library versions it was made with)."""
import argparse
import json
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="gpurun_out/tunableop_mi355x.csv")
ap.add_argument("--shapes", default="hy,wan")
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()

import torch  # noqa: E402

SHAPES = {
    # (M, N, K, op): y[M, N] = x[M, K] @ w[N, K]^T (+ bias) (+ gelu)
    "hy": [(118800, 3072, 3072, "mm"), (256, 3072, 3072, "mm"), (119056, 3072, 3072, "mm"), (118800, 12288, 3072, "mm_gelu"),
           (118800, 3072, 12288, "mm"), (119056, 12288, 3072, "mm_gelu"), (119056, 3072, 12288, "mm")],
    "wan": [(75600, 5120, 5120, "addmm"), (512, 5120, 5120, "addmm"), (75600, 13824, 5120, "addmm_gelu"), (75600, 5120, 13824, "addmm")],
}
dev = torch.device("cuda", 0)


def run(op, x, w, b, y):
    if op == "mm":
        return torch.mm(x, w.t(), out=y)
    if op == "addmm":
        return torch.addmm(b, x, w.t(), out=y)
    return torch._addmm_activation(b, x, w.t(), use_gelu=True)       # mm_gelu (zero bias) / addmm_gelu


def rate(op, M, N, K, reps):
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    b = torch.zeros(N, device=dev, dtype=torch.bfloat16) if op == "mm_gelu" else torch.randn(N, device=dev, dtype=torch.bfloat16)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    run(op, x, w, b, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run(op, x, w, b, y)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * M * N * K * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12


todo = [s for g in a.shapes.split(",") for s in SHAPES[g]]
base = {}
torch.cuda.tunable.enable(False)
for M, N, K, op in todo:
    base[(M, N, K, op)] = rate(op, M, N, K, a.reps)
os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
torch.cuda.tunable.enable(True)
torch.cuda.tunable.tuning_enable(True)
torch.cuda.tunable.set_filename(a.out)
torch.cuda.tunable.set_max_tuning_duration(200)       # ms per candidate solution
torch.cuda.tunable.set_max_tuning_iterations(20)
for M, N, K, op in todo:
    t0 = time.time()
    r = rate(op, M, N, K, a.reps)          # the first call tunes
    print(json.dumps({"shape": f"{M}x{N}x{K}", "op": op, "default_tflops": round(base[(M, N, K, op)], 1), "tuned_tflops": round(r, 1),
                      "gain": round(r / base[(M, N, K, op)], 4), "tuning_s": round(time.time() - t0, 1)}), flush=True)
print(json.dumps({"results_file": a.out + " (written by TunableOp at exit)", "validators": torch.cuda.tunable.get_validators()}))
