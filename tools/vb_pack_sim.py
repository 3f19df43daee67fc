#!/usr/bin/env python3
"""CPU model of the SVG2 attention launch at the bench geometry: how many (q-tile, 64-key tile) iterations svg_varblock_attention
walks under different ways of filling its 256-row q-tiles, counted from the cluster sizes and the block map alone (no GPU, no library).

The kernel's time is iterations x time per iteration, and the time per iteration is the band kernel's (Wan 720p: 4 185 845 iterations in
28.3 ms = 148 M/s, profiles/r04j_vb_pack_probe_m16.txt; the band kernel at HunyuanVideo 720p: 149 M/s) — the distance between SVG2's
0.42 and SVG1's 0.50 of the MFMA peak is q-tile fill, not the gather.  This script puts numbers on the fill:

    python tools/vb_pack_sim.py [heads=2] [workload=wan720p]

Data and pipeline as bench_svg2.py (64-mode Gaussian mixture, k-means 50 + 2 iterations from the first K rows, top-p 0.9 / min_kc_ratio
0.1) through the oracle's restatements (oracle/svg_oracle.py: batch_kmeans_euclid, identify_dynamic_map, varblock_pair_partners), on
`heads` heads (a few minutes of CPU per head at Wan 720p).  Printed per head and summed:
  * block-rows, rows per q-tile ("fill"), iterations without packing (variant 6), with the shipped remainder packing (variant 3:
    pairs by common key blocks, 3 handshake rounds), and the floor (every tile full);
  * what-ifs for round 5: more handshake rounds; greedy pairing of what the handshake leaves; remainders of <= 128 rows run as two
    INDEPENDENT half tiles of one workgroup (waves 0-3 / 4-7 each on its own block-row and key list: cost max instead of sum, no key
    overlap needed)."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import svg_oracle as O  # noqa: E402

WORKLOADS = {"wan720p": (40, 128, 21 * 3600, 300, 1000), "hy720p": (24, 128, 33 * 3600, 400, 1000), "small": (4, 128, 5000, 40, 100)}
BM, BN = 256, 64


def clustered(N, D, modes, gen, spread=0.35):
    centers = torch.randn(modes, D, generator=gen) * 1.5
    lab = torch.randint(0, modes, (N,), generator=gen)
    return (centers[lab] + spread * torch.randn(N, D, generator=gen)).to(torch.bfloat16)


def kmeans(x, K):
    """50 iterations from the first K rows, then 2 warm-started ones: what the bench's second kmeans_clustering call returns"""
    x = x[None]
    lab, c, cnt, _ = O.batch_kmeans_euclid(x, K, max_iters=50, init_centroids=x[:, :K].clone())
    lab, c, cnt, _ = O.batch_kmeans_euclid(x, K, max_iters=2, init_centroids=c)
    cnt = torch.bincount(lab[0], minlength=K).to(torch.int32)     # sizes of the returned labels
    return c[0], cnt


def iterations(qs, kt, partner=None):
    """q-tile x key-tile iterations.  qs [QB] rows per block-row, kt [QB] key tiles per block-row, partner: varblock_pair_partners row;
    a packed last tile walks the union of both key lists (kt_union[(i, j)])"""
    full = (qs // BM) * kt
    rem = ((qs % BM) > 0).long() * kt
    return int(full.sum()), rem


def main():
    heads = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    wl = sys.argv[2] if len(sys.argv) > 2 else "wan720p"
    H, D, S, QC, KC = WORKLOADS[wl]
    tot = {}
    for h in range(heads):
        t0 = time.time()
        gen = torch.Generator().manual_seed(100 + h)
        q, k = clustered(S, D, 64, gen), clustered(S, D, 64, gen)
        qc, qs = kmeans(q, QC)
        kc, ks = kmeans(k, KC)
        dmap = O.identify_dynamic_map(qc[None, None], kc[None, None], qs[None, None], ks[None, None], 0.9, 0.1)[0, 0]   # [QC, KC]
        act = dmap & (ks > 0)[None]
        qs64, ks64 = qs.long(), ks.long()
        keys = (act.long() * ks64[None]).sum(1)                  # active keys per block-row (one consecutive run list)
        kt = (keys + BN - 1) // BN
        live = qs64 > 0
        tiles = ((qs64 + BM - 1) // BM)
        rem = qs64 % BM
        n_full = qs64 // BM
        it_full = int((n_full * kt).sum())
        it_rem = (rem > 0).long() * kt
        it_nopack = it_full + int(it_rem.sum())
        floor_it = float((qs64.double() / BM * kt.double()).sum())

        def union_tiles(i, j):
            u = act[i] | act[j]
            return int(((u.long() * ks64).sum() + BN - 1) // BN)

        def packed_cost(partner):
            c = it_full
            for i in range(QC):
                p = int(partner[i])
                if rem[i] == 0 or p == -2:
                    continue
                c += union_tiles(i, p) if p >= 0 else int(kt[i])
            return c

        res = {"block_rows": int(live.sum()), "q_tiles": int(tiles.sum()), "fill": float(qs64.sum()) / float(tiles.sum() * BM),
               "iter_nopack": it_nopack, "iter_floor": floor_it}
        for rounds in (3, 8):
            partner = O.varblock_pair_partners(dmap[None], qs[None], ks[None], BM, rounds, 8)[0]
            res[f"iter_pack_r{rounds}"] = packed_cost(partner)
            res[f"pairs_r{rounds}"] = int((partner >= 0).sum())
        # greedy completion: what three rounds leave, best remaining partner by saved iterations (exhaustive, descending)
        partner = O.varblock_pair_partners(dmap[None], qs[None], ks[None], BM, 3, 8)[0].clone()
        free = [i for i in range(QC) if rem[i] > 0 and int(partner[i]) == -1]
        cand = []
        for a_i, i in enumerate(free):
            for j in free[a_i + 1:]:
                if rem[i] + rem[j] <= BM:
                    save = int(kt[i]) + int(kt[j]) - union_tiles(i, j)
                    if save > 10:
                        cand.append((save, i, j))
        cand.sort(reverse=True)
        used = set()
        for save, i, j in cand:
            if i in used or j in used:
                continue
            used.update((i, j))
            partner[i], partner[j] = j, -2
        res["iter_pack_r3_greedy"] = packed_cost(partner)
        res["pairs_r3_greedy"] = int((partner >= 0).sum())
        # half tiles: after the shipped packing, unpaired remainders of <= 128 rows run two to a workgroup, each half on its own key list
        partner3 = O.varblock_pair_partners(dmap[None], qs[None], ks[None], BM, 3, 8)[0]
        small = sorted((int(kt[i]) for i in range(QC) if 0 < rem[i] <= BM // 2 and int(partner3[i]) == -1), reverse=True)
        saved = sum(small[1::2])                                  # longest-first pairing: the shorter of each pair rides along
        res["iter_pack_r3_halftiles"] = res["iter_pack_r3"] - saved
        res["halftile_candidates"] = len(small)
        # where the empty slots are after the shipped packing: iterations by rows of the (possibly packed) last tile
        hist = {64: 0, 128: 0, 192: 0, 256: 0}
        for i in range(QC):
            p3 = int(partner3[i])
            if rem[i] == 0 or p3 == -2:
                continue
            rows, cost = (int(rem[i] + rem[p3]), union_tiles(i, p3)) if p3 >= 0 else (int(rem[i]), int(kt[i]))
            hist[min(b for b in hist if rows <= b)] += cost
        for b, c in hist.items():
            res[f"iter_last_tiles_le{b}"] = c
        res["iter_full_tiles"] = it_full
        for k2, v in res.items():
            tot[k2] = tot.get(k2, 0) + v
        print(f"head {h} ({time.time() - t0:.0f} s): rows/cluster {qs64.float().mean():.0f} +- {qs64.float().std():.0f}, key tiles/block-row "
              f"{kt.float().mean():.0f}, density {float((act.long() * ks64[None]).sum(1).double().mul(qs64.double()).sum()) / S / S:.3f}  "
              + "  ".join(f"{a}={b:.4g}" if isinstance(b, float) else f"{a}={b}" for a, b in res.items()), flush=True)
    n = heads
    print(f"\nmean of {n} heads, scaled to {H} heads:")
    base = tot["iter_pack_r3"]
    for k2 in ("iter_nopack", "iter_pack_r3", "iter_pack_r8", "iter_pack_r3_greedy", "iter_pack_r3_halftiles", "iter_floor"):
        print(f"  {k2:26s} {tot[k2] / n * H:12.0f}   {tot[k2] / base:6.3f} of the shipped packing")
    print("  iterations of the shipped packing by rows in the tile: " + ", ".join(
        f"<= {b}: {tot[f'iter_last_tiles_le{b}'] / base:.3f}" for b in (64, 128, 192, 256)) + f", full tiles {tot['iter_full_tiles'] / base:.3f}")
    print(f"  fill (rows per q-tile slot) {tot['fill'] / n:.3f}; pairs per head: 3 rounds {tot['pairs_r3'] / n:.0f}, 8 rounds {tot['pairs_r8'] / n:.0f}, "
          f"+ greedy {tot['pairs_r3_greedy'] / n:.0f}; half-tile candidates per head {tot['halftile_candidates'] / n:.0f}")
    print("  (measured on the GPU, 40 heads: 5 076 420 without / 4 185 845 with the shipped packing — profiles/r04j_vb_pack_probe_m16.txt)")


if __name__ == "__main__":
    main()
