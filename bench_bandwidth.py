#!/usr/bin/env python3
"""HBM-bound kernels of the path (SURVEY §8 d: placement, token permutation, label sort): achieved bytes/s against the measured
copy rate of this GPU.  python bench_bandwidth.py  -> one JSON line.
Algorithmic bytes = every tensor row read once + written once (placement of q, k, v: 6 * H * S * D * 2 B = 4.39 GB at Hunyuan 720p)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
import torch  # noqa: E402
from svg import _native as nat  # noqa: E402

PEAK_HBM_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)


def main():
    dev = torch.device("cuda", 0)
    H, D, F_, P_, ctx = 24, 128, 33, 3600, 256
    S = F_ * P_ + ctx
    q, k, v = (torch.randn(1, H, S, D, device=dev, dtype=torch.bfloat16) for _ in range(3))
    qo, ko, vo = (torch.empty_like(q) for _ in range(3))
    best = torch.tensor([[h % 2 for h in range(H)]], device=dev)
    row_bytes = H * S * D * 2
    out = {"metric": "hbm_bound_kernels", "workload": f"HunyuanVideo 720p cfg=1 H={H} S={S} D={D} bf16", "peak_GBs": PEAK_HBM_GBS, "kernels": {}}

    def rec(name, ms, nbytes):
        out["kernels"][name] = {"ms": round(ms, 3), "GBs": round(nbytes / ms / 1e6, 1), "frac_of_8TBs": round(nbytes / ms / 1e6 / PEAK_HBM_GBS, 3),
                                "algorithmic_bytes": nbytes}

    ms = timeit(lambda: qo.copy_(q))
    rec("torch copy (reference point)", ms, 2 * row_bytes)
    ms = timeit(lambda: nat.head_placement([q, k, v], [qo, ko, vo], best, ctx, F_, P_, False, inverse=False))
    rec("placement q,k,v (svg_head_placement)", ms, 6 * row_bytes)
    ms = timeit(lambda: nat.head_placement([q], [qo], best, ctx, F_, P_, False, inverse=True))
    rec("inverse placement of o", ms, 2 * row_bytes)
    # SVG2: Wan 720p token permutation by k-means labels
    Hw, Sw, K = 40, 75600, 1000
    x = torch.randn(Hw, Sw, D, device=dev, dtype=torch.bfloat16)
    labels = torch.randint(0, K, (Hw, Sw), device=dev, dtype=torch.int32)
    sidx, counts = nat.argsort_labels(labels, K)
    ms = timeit(lambda: nat.argsort_labels(labels, K))
    rec("stable label sort (svg_argsort_labels), Wan 720p K=1000", ms, Hw * Sw * 4 * 2)
    wbytes = Hw * Sw * D * 2
    ms = timeit(lambda: nat.permute_rows(x, sidx))
    rec("permute rows (gather)", ms, 2 * wbytes + Hw * Sw * 4)
    y = nat.permute_rows(x, sidx)
    ms = timeit(lambda: nat.permute_rows(y, sidx, inverse=True))
    rec("inverse permute rows (scatter)", ms, 2 * wbytes + Hw * Sw * 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
