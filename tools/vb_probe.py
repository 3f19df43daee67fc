#!/usr/bin/env python3
"""Where the distance between the SVG2 attention kernel (0.42 of the bf16 MFMA peak at Wan 2.1 720p) and the band kernel (0.51) comes from:
svg_varblock_attention on synthetic plans at the Wan 720p size (H = 40, D = 128, S ~ 75600, KB = 1000, density ~ 0.25) that switch ONE property at
a time —

    uniform        q-clusters of exactly 256 rows (every q-tile full), k-clusters of 75 / 76 keys, each block-row a random quarter of the key blocks,
                   rows in place (no index arrays): the kernel body and the run-list walk alone
    uniform_idx    the same plan with the rows behind a random permutation (q_row_idx / kv_row_idx): + the 256-byte row gather
    uniform_nbr    uniform, but neighbouring block-rows select nearly the same key blocks (what k-means clusters of one neighbourhood do)
    uniformNNN     q-clusters of NNN rows each (64, 96, 128, 160, 192, 224): ONE partially filled q-tile per block-row — the cost of a tile iteration
                   as a function of its fill (run with --variant 6: no remainder packing)
    ragged         the REAL cluster sizes of the bench pipeline (252 +- 124 rows) with the random quarter map, rows in place: + q-tile fill, packing
    real_noidx     the bench pipeline's sizes and map, q / k / v permuted beforehand (the reference's pipeline), rows in place
    real           the bench pipeline as shipped (fused gather / scatter)

and prints for each: ms, algorithmic TFLOP/s (4 D sum n(Q_i) n(K_j)), EXECUTED TFLOP/s (rows rounded up to the 32 of a wave, keys to the 64 of a
tile: what the matrix pipe is asked to do) and the wave-level fill.  `--variant` as bench_svg2.py (6: no remainder packing).

    python tools/vb_probe.py [--cases uniform,uniform_idx,...] [--variant -1] [--reps 3]
"""
import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

H, D, KB = 40, 128, 1000


def executed(dmap, qs, ks):
    """(algorithmic pairs, executed pairs at wave x tile granularity, q-tile x key-tile iterations without packing)"""
    act = dmap.bool() & (ks > 0)[:, None, :]
    keys = (act.double() * ks[:, None, :].double()).sum(-1)                  # [H, QB]
    alg = (keys * qs.double()).sum().item()
    rows32 = ((qs + 31) // 32 * 32).double()
    keys64 = torch.ceil(keys / 64) * 64
    exe = (rows32 * keys64).sum().item()
    iters = (((qs + 255) // 256).double() * keys64 / 64).sum().item()
    return alg, exe, iters


def random_quarter_map(QB, gen, dev, neighbours=False):
    if not neighbours:
        return (torch.rand(H, QB, KB, device=dev, generator=gen) < 0.25).to(torch.uint8)
    # block-rows in groups of 8 share a base selection and flip 3 % of it
    base = torch.rand(H, (QB + 7) // 8, KB, device=dev, generator=gen) < 0.25
    m = base.repeat_interleave(8, dim=1)[:, :QB]
    flip = torch.rand(H, QB, KB, device=dev, generator=gen) < 0.03
    return (m ^ flip).to(torch.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="uniform,uniform_idx,uniform_nbr,ragged,real_noidx,real")
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    from svg import _native as nat
    from svg.kmeans_utils import identify_dynamic_map
    from svg.models import _core

    import bench_svg2

    nat.load()
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(0)
    S = 75600
    q = bench_svg2.clustered(H, S, D, 64, dev, gen)
    k = bench_svg2.clustered(H, S, D, 64, dev, gen)
    v = torch.randn(H, S, D, device=dev, dtype=torch.bfloat16, generator=gen)
    # the bench pipeline's plan (50 + 2 k-means iterations from the first rows, top-p map)
    store = _core.CentroidStore()
    _core.kmeans_clustering(store, 0, q[None], k[None], 300, KB, 50, 2)
    (ql, qc, qs_r, _, qidx), (kl, kc, ks_r, _, kidx) = _core.kmeans_clustering(store, 0, q[None], k[None], 300, KB, 50, 2)
    qs_r, ks_r = qs_r.view(H, 300).contiguous(), ks_r.view(H, KB).contiguous()
    dmap_r = identify_dynamic_map(qc.view(1, H, 300, D), kc.view(1, H, KB, D), qs_r[None], ks_r[None], 0.9, 0.1).view(H, 300, KB).contiguous()
    qidx, kidx = qidx.contiguous(), kidx.contiguous()

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    for case in a.cases.split(","):
        qq, kk, vv, qi, ki = q, k, v, None, None
        if case.startswith("uniform"):
            # uniform / uniform_idx / uniform_nbr: q-clusters of 256 rows; uniformNNN: of NNN rows (one partially filled q-tile each: what a
            # tile iteration costs as a function of its fill — packing off, see --variant 6)
            rows = int(case[len("uniform"):]) if case[len("uniform"):].isdigit() else 256
            QB = S // rows
            Sq = QB * rows
            qs = torch.full((H, QB), rows, dtype=torch.int32, device=dev)
            ks = torch.full((H, KB), Sq // KB, dtype=torch.int32, device=dev)
            ks[:, : Sq - (Sq // KB) * KB] += 1
            dmap = random_quarter_map(QB, gen, dev, neighbours=case == "uniform_nbr")
            qq, kk, vv = q[:, :Sq].contiguous(), k[:, :Sq].contiguous(), v[:, :Sq].contiguous()
            if case == "uniform_idx":
                qi = torch.stack([torch.randperm(Sq, device=dev, generator=gen) for _ in range(H)]).to(torch.int32)
                ki = torch.stack([torch.randperm(Sq, device=dev, generator=gen) for _ in range(H)]).to(torch.int32)
        elif case == "ragged":
            qs, ks, dmap = qs_r, ks_r, random_quarter_map(300, gen, dev)
        elif case == "real_noidx":
            qs, ks, dmap = qs_r, ks_r, dmap_r
            qq, kk, vv = nat.permute_rows(q, qidx), nat.permute_rows(k, kidx), nat.permute_rows(v, kidx)
        elif case == "real":
            qs, ks, dmap, qi, ki = qs_r, ks_r, dmap_r, qidx, kidx
        else:
            raise SystemExit(f"unknown case {case}")
        ms = timed(lambda: nat.varblock_attention(qq, kk, vv, dmap, qs, ks, q_row_idx=qi, kv_row_idx=ki, variant=a.variant))
        alg, exe, iters = executed(dmap, qs, ks)
        print(json.dumps({"case": case, "variant": a.variant, "ms": round(ms, 3), "alg_tflops": round(4 * D * alg / ms / 1e9, 1),
                          "executed_tflops": round(4 * D * exe / ms / 1e9, 1), "wave_fill": round(alg / exe, 4),
                          "qtile_iters_nopack": int(iters), "density": round(alg / float(qq.shape[1]) ** 2 / H, 4)}), flush=True)


if __name__ == "__main__":
    main()
