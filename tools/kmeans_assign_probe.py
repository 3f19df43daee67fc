#!/usr/bin/env python3
"""svg_kmeans_assign alone on the GPU at the Wan 2.1 720p size: time vs K (tiles of 64 centroids) and vs N -> the per-tile rate in steady state and
the fixed part per launch, against the MFMA floor at the granted clock.    python tools/kmeans_assign_probe.py"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from svg import _native as nat  # noqa: E402
import bench_svg2  # noqa: E402

nat.load()
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(0)
B, N, D = 40, 75600, 128
x = bench_svg2.clustered(B, N, D, 64, dev, gen)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for K in (64, 128, 256, 320, 512, 1000, 1024, 2048, 4096):
    c = x[:, :K].contiguous()
    ms = timed(lambda: nat.kmeans_assign(x, c))
    tiles = (K + 63) // 64
    print(json.dumps({"B": B, "N": N, "K": K, "tiles": tiles, "ms": round(ms, 4), "tflops": round(2.0 * B * N * K * D / ms / 1e9, 1),
                      "ms_per_tile": round(ms / tiles, 4)}), flush=True)
# granted shader clock while the assignment runs back to back (svg_debug_clock_probe: one sleeping wave beside the launches)
try:
    c = x[:, :4096].contiguous()
    probe = nat.ClockProbe(dev)
    probe.start(max_ms=5000)
    for _ in range(20):
        nat.kmeans_assign(x, c)
    probe.arm_stop()
    print(json.dumps({"sclk_mhz_during_assign_K4096": probe.result()}), flush=True)
except Exception as e:  # noqa: BLE001
    print(json.dumps({"sclk_probe_error": str(e)[:200]}), flush=True)
for n in (9450, 18900, 37800):
    xs = x[:, :n].contiguous()
    c = xs[:, :1000].contiguous()
    ms = timed(lambda: nat.kmeans_assign(xs, c))
    print(json.dumps({"B": B, "N": n, "K": 1000, "ms": round(ms, 4), "tflops": round(2.0 * B * n * 1000 * D / ms / 1e9, 1)}), flush=True)
