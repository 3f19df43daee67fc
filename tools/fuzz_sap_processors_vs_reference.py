#!/usr/bin/env python3
"""Randomised cross-check of the SVG2 (semantic-aware permutation) layer-call: `attention_core_logic` of the reference's
Hunyuan_SAPAttn_Processor2_0 and WanAttn_SAPAttn_Processor executed as they are in the build container — flash-kmeans on both Triton
kernels (interpreted) from warm-start centroids with the reference's stopping rule, identify_dynamic_map, the Triton permutation, the
prompt / unused-prompt pseudo clusters (Hunyuan), the variable-block attention (flashinfer is third-party and GPU-only: the reference's
own Triton kernel for the operator stands in, as in tests/golden/make_golden_triton.py section 8), the inverse permutation — against
the oracle's composition of the same call, in float32, on random geometries: frame count, ragged frame size, text / prompt length,
heads, head size, cluster counts, top-p, min_kc_ratio, iteration cap.

The data are ragged, well-separated modes with one warm-start centroid inside each (plus, sometimes, spare centroids far away that stay
EMPTY): no point sits on a decision boundary, so the comparison is about the plumbing across geometries, not about which side of an
fp32 rounding a near-tie falls (tools/fuzz_oracle_vs_triton.py counts those at kernel level).

    python tools/fuzz_sap_processors_vs_reference.py [--trials 16] > profiles/<round>_fuzz_sap_processors_vs_reference.txt"""
import argparse
import json
import os
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests" / "golden"))
import make_golden_triton as MT  # noqa: E402  (sets TRITON_INTERPRET=1 before triton is imported)

import torch  # noqa: E402

from oracle import svg_oracle as O  # noqa: E402

MG = MT.MG


def oracle_layer_call(q, k, v, geo, init_q, init_k, top_p, min_ratio, iters):
    """the oracle's statement of the SVG2 layer-call (ref: hyvideo/attention.py:715-804, wan/attention.py:500-556) in the original row
    order (permutation and inverse permutation cancel) -> (out [H, S, D], density [H], centroids q, centroids k, iterations q / k)"""
    H, D, F_, P_, ctx, L, QC, KC = geo
    V, S = F_ * P_, F_ * P_ + ctx
    ql, cq, qs, nq = O.batch_kmeans_euclid(q[0, :, :V], QC, max_iters=iters, init_centroids=init_q.clone())
    kl, ck, ks, nk = O.batch_kmeans_euclid(k[0, :, :V], KC, max_iters=iters, init_centroids=init_k.clone())
    dmap = O.identify_dynamic_map(cq[None], ck[None], qs[None].long(), ks[None].long(), top_p, min_ratio)
    q_lab, k_lab, q_sz, k_sz = ql, kl, qs[None].long(), ks[None].long()
    if ctx:
        dmap, q_sz, k_sz, _ = O.dynamic_map_post_processing(dmap, q_sz, k_sz, torch.zeros(H, V, dtype=torch.long), V, ctx, L)
        q_lab = torch.cat([ql, torch.cat([torch.full((L,), QC), torch.full((ctx - L,), QC + 1)]).expand(H, -1)], 1)
        k_lab = torch.cat([kl, torch.cat([torch.full((L,), KC), torch.full((ctx - L,), KC + 1)]).expand(H, -1)], 1)
    out = torch.zeros(H, S, D)
    for h in range(H):
        em = dmap[0, h][q_lab[h]][:, k_lab[h]]
        out[h] = O.masked_attention(q[0, h], k[0, h], v[0, h], em)
    return out, O.density_calculation(dmap, q_sz, k_sz)[0], cq, ck, int(nq), int(nk)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=16)
    ap.add_argument("--seed", type=int, default=2718)
    args = ap.parse_args()
    MG.install_stubs()
    MG._stub("diffusers.models.normalization", RMSNorm=type("RMSNorm", (), {}))
    sys.path.insert(0, MG.REF)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.Tensor.is_cuda = property(lambda self: True)
    import svg.kmeans_utils as KU
    import svg.models.hyvideo.attention as hy_attn
    import svg.models.wan.attention as wan_attn

    KU._euclid_assign_kernel = MT.ExplicitConfig(KU._euclid_assign_kernel, BLOCK_N=64, BLOCK_K=64)
    KU._euclid_iter_compiled = KU._euclid_iter

    def own_triton(q, k, v, m, qc, kc, is_cpu=False):
        return KU.dynamic_block_sparse_fwd_triton(q.contiguous(), k.contiguous(), v.contiguous(), m, qc, kc)

    hy_attn.dynamic_block_sparse_fwd_flashinfer = own_triton
    wan_attn.dynamic_block_sparse_fwd_flashinfer = own_triton
    gen = torch.Generator().manual_seed(args.seed)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=gen))

    def rf(lo, hi):
        return lo + (hi - lo) * float(torch.rand(1, generator=gen))

    counts, notes = {}, {"empty": 0, "converged": 0, "calls": 0, "dens": []}

    def ok(name, cond, detail=""):
        c = counts.setdefault(name, [0, 0])
        c[0] += 1
        if not cond:
            c[1] += 1
            print(f"MISMATCH {name}: {detail}")

    for trial in range(args.trials):
        for model in ("hy", "wan"):
            H, D = ri(1, 3), (32, 64)[ri(0, 1)]
            F_, P_ = ri(2, 5), ri(16, 70)
            ctx = ri(2, 40) if model == "hy" else 0
            L = ri(1, ctx) if ctx else 0
            QC, KC = ri(2, 9), ri(2, 12)
            top_p, min_ratio, iters = rf(0.4, 0.95), (0.0 if ri(0, 1) else rf(0.05, 0.4)), ri(1, 6)
            V, S = F_ * P_, F_ * P_ + ctx
            while V < 64:                                      # (the reference's attention wrapper needs S >= 64 or a power of two)
                P_ += 8
                V, S = F_ * P_, F_ * P_ + ctx
            spare_q, spare_k = ri(0, 1), ri(0, 2)             # centroids far from every point: their clusters stay empty

            def modes(n_modes, spare):
                live = max(1, n_modes - spare)
                centers = torch.randn(H, live, D, generator=gen) * 1.2
                lab = torch.randint(0, live, (H, S), generator=gen)
                lab[:, :live] = torch.arange(live)
                x = torch.gather(centers, 1, lab[..., None].expand(-1, -1, D)) + 0.12 * torch.randn(H, S, D, generator=gen)
                init = torch.cat([x[:, :live].clone(), 40.0 + torch.randn(H, n_modes - live, D, generator=gen)], 1)
                return x[None], init

            (q, init_q), (k, init_k) = modes(QC, spare_q), modes(KC, spare_k)
            v = torch.randn(1, H, S, D, generator=gen)
            log = tempfile.NamedTemporaryFile("w", suffix=".jsonl", delete=False)
            log.close()
            if model == "hy":
                proc = hy_attn.Hunyuan_SAPAttn_Processor2_0(0)
                proc.centroids_init, proc.q_centroids, proc.k_centroids = {0: True}, {0: init_q.clone()}, {0: init_k.clone()}
                proc.prompt_length = L
            else:
                proc = wan_attn.WanAttn_SAPAttn_Processor(0)
                proc.centroids_init, proc.q_centroids, proc.k_centroids = True, init_q.clone(), init_k.clone()
            proc.context_length, proc.num_frame, proc.frame_size = ctx, F_, P_
            proc.num_q_centroids, proc.num_k_centroids, proc.top_p_kmeans, proc.min_kc_ratio = QC, KC, top_p, min_ratio
            proc.kmeans_iter_init, proc.kmeans_iter_step, proc.first_layers_fp, proc.first_times_fp = 0, iters, 0, 1.0
            proc.logging_file = log.name
            ts = torch.tensor([0.5])
            if model == "hy":
                o = proc.attention_core_logic(q.clone(), k.clone(), v.clone(), ts, 0, None)
                cq, ck = proc.q_centroids[0], proc.k_centroids[0]
            else:
                o = proc.attention_core_logic(q.clone(), k.clone(), v.clone(), ts)
                cq, ck = proc.q_centroids, proc.k_centroids
            entry = json.loads(open(log.name).read().strip().splitlines()[-1])
            os.unlink(log.name)
            geo = (H, D, F_, P_, ctx, L, QC, KC)
            out, dens, mcq, mck, nq, nk = oracle_layer_call(q, k, v, geo, init_q, init_k, top_p, min_ratio, iters)
            det = (model, geo, round(top_p, 3), round(min_ratio, 3), iters)
            ok(f"block-map density {model}", torch.allclose(dens.double(), torch.tensor(entry["density"], dtype=torch.float64).reshape(-1), atol=1e-6), det)
            ok(f"centroids {model}", torch.allclose(mcq, cq, rtol=1e-5, atol=1e-5) and torch.allclose(mck, ck, rtol=1e-5, atol=1e-5), det)
            e = ((out - o[0]).norm() / o[0].norm()).item()
            ok(f"attention_core_logic output {model}", e < 5e-6 and torch.allclose(out, o[0], atol=3e-5, rtol=3e-5), det + (e,))
            notes["calls"] += 1
            notes["empty"] += int(spare_q + spare_k > 0)
            notes["converged"] += int(nq < iters or nk < iters)
            notes["dens"].append(float(dens.mean()))

    print(f"# fuzz of the oracle's statement of the SVG2 layer-call against the reference's SAP processors executed as they are (float32): {args.trials} random geometries x 2 models, seed {args.seed}")
    print("| check | comparisons | mismatches |\n|---|---|---|")
    bad_total = 0
    for name, (n, bad) in counts.items():
        print(f"| {name} | {n} | {bad} |")
        bad_total += bad
    d = notes["dens"]
    print(f"\n{notes['calls']} layer-calls: {notes['empty']} with clusters that stay empty, {notes['converged']} where a k-means loop left on its tolerance before the iteration cap; "
          f"block-map density {min(d):.2f} .. {max(d):.2f} (mean {sum(d) / len(d):.2f})")
    print("RESULT:", "all equal" if bad_total == 0 else f"{bad_total} MISMATCHES")
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
