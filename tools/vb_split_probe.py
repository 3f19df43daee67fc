"""Cost of a q tile of the variable-block kernel as a function of its rows (kernel ms, HIP events): uniform q-clusters of R rows
(R = 64, 128, 192, 256), uniform 128-row k-clusters, density 0.25, the same number of q tiles in every case — what the key-split form
of the two-phase body (attn_body_pp2_ctx KSPLIT, tiles with <= 128 rows) buys per tile.  Compare builds with SVG_ATTN_LIB."""
import importlib
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
nat = importlib.import_module("sparse-videogen_amd.svg._native")


def run(R, H=8, QB=288, KB=576, D=128, density=0.25, steps=5):
    torch.manual_seed(0)
    Sq, Skv = QB * R, KB * 128
    q = torch.randn(H, Sq, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(H, Skv, D, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(H, Skv, D, device="cuda", dtype=torch.bfloat16)
    bm = torch.rand(H, QB, KB, device="cuda") < density
    qs = torch.full((H, QB), R, dtype=torch.int32, device="cuda")
    ks = torch.full((H, KB), 128, dtype=torch.int32, device="cuda")
    for _ in range(2):
        nat.varblock_attention(q, k, v, bm, qs, ks)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(steps):
        nat.varblock_attention(q, k, v, bm, qs, ks)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / steps
    flop = 4.0 * D * R * 128 * float(bm.sum())
    return ms, flop / ms / 1e9


if __name__ == "__main__":
    for R in (64, 128, 192, 256):
        ms, tf = run(R)
        print(f"rows/tile {R:4d}: {ms:7.3f} ms  {tf:7.1f} TFLOP/s on real rows")
    # the same number of tile iterations with K / V of a head resident in L2 (32 k-clusters = 1 MB of K + V, every block active)
    for R in (64, 128, 192, 256):
        ms, tf = run(R, QB=2592, KB=32, density=1.0)
        print(f"L2-resident K/V, rows/tile {R:4d}: {ms:7.3f} ms  {tf:7.1f} TFLOP/s on real rows")
