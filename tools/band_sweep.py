#!/usr/bin/env python3
"""Band-width sweep of the band-attention kernel at the HunyuanVideo 720p geometry: time = a * work + b * launch overhead.
Separates the per-tile rate of the kernel from what a workgroup pays outside its tile loop (Q load, pipeline fill,
epilogue, launch tail).  python tools/band_sweep.py [--heads spatial|temporal|alt] [--variant V]"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
import torch  # noqa: E402

from svg import _native as nat  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--heads", default="spatial")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--H", type=int, default=24)
    a = ap.parse_args()
    H, D, F_, P_, ctx, L = a.H, 128, 33, 3600, 256, 64
    V = F_ * P_
    S = V + ctx
    dev = torch.device("cuda:0")
    q, k, v = (torch.randn(1, H, S, D, device=dev, dtype=torch.bfloat16) for _ in range(3))
    o = torch.empty_like(q)
    pat = {"alt": lambda h: h % 2, "spatial": lambda h: 0, "temporal": lambda h: 1}[a.heads]
    best = torch.tensor([[pat(h) for h in range(H)]], device=dev, dtype=torch.int64)
    rows = []
    for band in (1024, 2048, 4096, 8192, 15616, 31232, 62464, S + 1):
        if band > S:
            mask = nat.BandMask(real_len=V + L, band=S + 1, colfull_lo=0, colfull_hi=0, rowfull_lo=0, rowfull_hi=0)
            pairs = (V + L) ** 2 + (ctx - L) ** 2
        else:
            mask = nat.BandMask(real_len=V + L, band=band, colfull_lo=V, colfull_hi=V + L, rowfull_lo=V, rowfull_hi=V + L)
            tf = min(band, V)
            pairs = V * (2 * tf - 1) - tf * (tf - 1) + 2 * V * L + L * L + (ctx - L) ** 2

        def run():
            nat.band_attention(q, k, v, mask, head_perm_flag=best, vid0=0, num_frame=F_, frame_size=P_, variant=a.variant, out=o)

        run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = min(ts)
        tf_ = 4.0 * D * H * pairs / 1e12
        rows.append((band, tf_, ms))
        print(f"band {band:7d}  work {tf_:8.2f} TFLOP  {ms:8.3f} ms  {tf_ / ms:6.3f} PFLOP/s", flush=True)
    # least squares  ms = a * TF + b
    n = len(rows)
    sx = sum(r[1] for r in rows)
    sy = sum(r[2] for r in rows)
    sxx = sum(r[1] ** 2 for r in rows)
    sxy = sum(r[1] * r[2] for r in rows)
    a_ = (n * sxy - sx * sy) / (n * sxx - sx * sx)
    b_ = (sy - a_ * sx) / n
    print(f"fit: ms = {a_:.4f} * TFLOP + {b_:.3f}   (asymptotic {1 / a_:.3f} PFLOP/s, fixed {b_:.2f} ms per launch)")


if __name__ == "__main__":
    main()
