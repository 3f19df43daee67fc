#!/usr/bin/env python3
"""Kernel-level pin of the reference's Triton kernels: EXECUTES the reference's own `@triton.jit` sources with Triton's interpreter
(TRITON_INTERPRET=1: the kernel body runs on the CPU, tl.* ops on numpy arrays) through the reference's own Python wrappers, and
stores inputs + outputs as fixtures (tests/golden/triton_golden.npz, checked by tests/test_triton_golden.py against the oracle).

What that pins that the loop-level fixtures (make_golden_kmeans.py: launches REPLACED by torch statements) could not:
  * flash-kmeans assignment  `_euclid_assign_kernel`            svg/kmeans_utils.py:464-554   (chunked strict-'<' update, first-index
        argmin inside a chunk, clamp at 0, masking of the ragged last chunk / tile, int64 output of the wrapper :562-627)
  * flash-kmeans update      `_centroid_update_chunk_kernel`    svg/kmeans_utils.py:258-322 + host half :375-421
  * the whole loop           `batch_kmeans_Euclid`              svg/kmeans_utils.py:685-733 on BOTH real kernels
  * variable-block attention `_dynamic_block_sparse_fwd_kernel` svg/kmeans_utils.py:1001-1317 (the reference's Triton statement of the
        SVG2 attention; the flashinfer kernel that replaces it in production is third-party and stays anchored on the reference's test)
  * head placement           `*_sparse_head_placement_kernel`, `*_hidden_states_placement_kernel`  svg/models/{hyvideo,wan,cog}/placement.py
  * token permutation        `_permute_kernel`, `_inverse_permute_kernel`   svg/kernels/triton/permute.py
  * block glue               RMSNorm / LayerNorm / modulate kernels         svg/kernels/triton/{rmsnorm,layernorm,modulate}.py
and, on top of the kernels, the reference's PROCESSORS and its Wan block forward as they are (sections 8-16): `attention_core_logic` of the SAP and SVG1 processors, the
Wan uniform-block mask generator, and the whole `__call__` of the Wan / Hunyuan (double-, single-stream) / CogVideoX SVG processors.
Limits, stated: the interpreter of the Triton in this image (3.6.0) mis-handles bfloat16 (numpy has no such type; a 16 x 16
bf16 `tl.dot` returns garbage), so the fixtures are float32 and float16 — the dtype-independent structure of every kernel is pinned,
the bf16 rounding points of the two k-means norms stay a restatement (svg_oracle.kmeans_xsq / kmeans_csq); `triton.autotune`
needs a GPU to time its candidates, so the assign kernel is launched on its `.fn` with each tile configuration of the reference's
list given explicitly (the result must not depend on it, and does not).  Atomic float adds are order-free only up to fp32
rounding: update results are compared with a tolerance of one ulp of the stored dtype.

    python tests/golden/make_golden_triton.py        (a few minutes: the interpreter runs program by program)"""
import os
import sys
from pathlib import Path

os.environ["TRITON_INTERPRET"] = "1"          # before triton is imported

import numpy as np  # noqa: E402
import torch  # noqa: E402

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_golden as MG  # noqa: E402  (stubs for the third-party modules the reference imports)


class ExplicitConfig:
    """stand-in for a triton.autotune object: launches the jitted function underneath with ONE explicit configuration"""

    def __init__(self, auto, **meta):
        self.fn, self.meta = auto.fn, meta

    def __getitem__(self, grid):
        def launch(*args, **kw):
            return self.fn[grid(self.meta) if callable(grid) else grid](*args, **kw, **self.meta)

        return launch


def clustered(B, N, D, modes, g, spread=0.3, scale=2.0):
    centers = torch.randn(B, modes, D, generator=g) * scale
    lab = torch.randint(0, modes, (B, N), generator=g)
    return torch.gather(centers, 1, lab[..., None].expand(-1, -1, D)) + spread * torch.randn(B, N, D, generator=g)


def main():
    MG.install_stubs()
    sys.path.insert(0, MG.REF)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.Tensor.is_cuda = property(lambda self: True)      # the wrappers assert .is_cuda
    import svg.kmeans_utils as KU

    out = {}
    auto = KU._euclid_assign_kernel

    # ---------------- 1. assignment kernel ----------------
    cases = {
        # tag: (dtype, B, N, D, K, [(BLOCK_N, BLOCK_K), ...])
        "as_a": (torch.float32, 2, 333, 64, 37, [(32, 32), (64, 128), (128, 64)]),
        "as_b": (torch.float32, 1, 300, 128, 100, [(64, 32), (128, 128)]),
        "as_c": (torch.float16, 2, 333, 64, 37, [(32, 32), (128, 64)]),
        "as_d": (torch.float16, 1, 300, 128, 70, [(64, 64)]),
    }
    for tag, (dt, B, N, D, K, cfgs) in cases.items():
        g = torch.Generator().manual_seed(len(tag) + B * N)
        x = clustered(B, N, D, 9, g).to(dt)
        c = x[:, :K].clone()
        # exact ties: duplicated centroids inside one chunk (first index wins) and across chunks (the earlier chunk wins: strict '<')
        c[:, 5] = c[:, 2]
        if K > 40:
            c[:, 36] = c[:, 3]
            c[:, K - 1] = c[:, 1]
        x_sq = (x ** 2).sum(dim=-1)                      # ref: batch_kmeans_Euclid :704
        res = []
        for bn, bk in cfgs:
            KU._euclid_assign_kernel = ExplicitConfig(auto, BLOCK_N=bn, BLOCK_K=bk)
            ids = KU.euclid_assign_triton(x, c, x_sq)
            assert ids.dtype == torch.int64 and ids.shape == (B, N)
            res.append(ids)
        agree = all(torch.equal(res[0], r) for r in res[1:])
        print(f"{tag}: {dt} B={B} N={N} D={D} K={K}: {len(cfgs)} tile configurations agree: {agree}")
        out[f"{tag}_x"], out[f"{tag}_c"], out[f"{tag}_ids"] = x.numpy(), c.numpy(), res[0].numpy().astype(np.int32)
        out[f"{tag}_meta"] = np.array([{torch.float32: 2, torch.float16: 1}[dt], int(agree)], dtype=np.int64)
    KU._euclid_assign_kernel = ExplicitConfig(auto, BLOCK_N=64, BLOCK_K=64)

    # ---------------- 2. centroid update (sorted, chunked) ----------------
    for tag, (dt, B, N, D, K, block_n) in {"up_a": (torch.float32, 2, 400, 64, 23, 256), "up_b": (torch.float16, 1, 500, 128, 40, 64)}.items():
        g = torch.Generator().manual_seed(7 + N)
        x = clustered(B, N, D, 7, g).to(dt)
        ids = torch.randint(0, K, (B, N), generator=g)
        ids[ids == 4] = 5                                   # cluster 4 is empty: keeps its old centroid (:416-418)
        ids[:, :3] = K - 1
        old = torch.randn(B, K, D, generator=g).to(dt)
        cent, cnt = KU.triton_centroid_update_sorted_euclid(x, ids, old, BLOCK_N=block_n)
        assert cent.dtype == dt and cnt.dtype == torch.int32
        out[f"{tag}_x"], out[f"{tag}_ids"], out[f"{tag}_old"] = x.numpy(), ids.numpy().astype(np.int32), old.numpy()
        out[f"{tag}_cent"], out[f"{tag}_cnt"] = cent.numpy(), cnt.numpy()
        out[f"{tag}_meta"] = np.array([{torch.float32: 2, torch.float16: 1}[dt]], dtype=np.int64)
        print(f"{tag}: update {dt} B={B} N={N} K={K}: empty clusters {int((cnt == 0).sum())}")

    # ---------------- 3. the loop on both real kernels ----------------
    KU._euclid_iter_compiled = KU._euclid_iter
    for tag, (dt, B, N, D, K, iters) in {"lp_a": (torch.float16, 2, 400, 64, 10, 4), "lp_b": (torch.float32, 1, 300, 64, 8, 30)}.items():
        g = torch.Generator().manual_seed(3 + N)
        x = clustered(B, N, D, 6, g).to(dt)
        init = x[:, :K].clone()
        if tag == "lp_a":
            init[:, -2:] = 20.0                             # far-away seeds stay empty (20^2 x 64 still fits fp16)
        ids, cent, sizes, n_it = KU.batch_kmeans_Euclid(x, K, max_iters=iters, tol=1e-4, init_centroids=init.clone())
        out[f"{tag}_x"], out[f"{tag}_init"] = x.numpy(), init.numpy()
        out[f"{tag}_ids"], out[f"{tag}_cent"], out[f"{tag}_sizes"] = ids.numpy().astype(np.int32), cent.numpy(), sizes.numpy().astype(np.int32)
        out[f"{tag}_meta"] = np.array([{torch.float32: 2, torch.float16: 1}[dt], K, iters, int(n_it)], dtype=np.int64)
        print(f"{tag}: loop {dt}: {int(n_it)} iterations of at most {iters}, sizes {sizes[0].tolist()}")

    # ---------------- 4. variable-block attention: the reference's Triton statement ----------------
    for tag, (dt, H, D, qsz, ksz, seed) in {
        "vb_a": (torch.float16, 2, 64, [70, 1, 129, 0, 56], [33, 64, 0, 100, 59], 11),      # ragged + EMPTY clusters on both sides
        "vb_b": (torch.float32, 1, 32, [40, 88, 72], [128, 8, 64], 12),
    }.items():
        g = torch.Generator().manual_seed(seed)
        S = sum(qsz)
        assert S == sum(ksz)
        q, k, v = (torch.randn(1, H, S, D, generator=g).to(dt) for _ in range(3))
        qc = torch.tensor(qsz).expand(1, H, -1).contiguous()
        kc = torch.tensor(ksz).expand(1, H, -1).contiguous()
        dmap = torch.rand(1, H, len(qsz), len(ksz), generator=g) < 0.55
        for h in range(H):                                 # every q block with rows sees at least one key block with rows
            for i in range(len(qsz)):
                if qsz[i] and not any(bool(dmap[0, h, i, j]) and ksz[j] for j in range(len(ksz))):
                    dmap[0, h, i, max(range(len(ksz)), key=lambda j: ksz[j])] = True
        o = KU.dynamic_block_sparse_fwd_triton(q, k, v, dmap, qc, kc)
        o_torch = KU.dynamic_block_sparse_fwd_torch(q, k, v, dmap, qc, kc)      # the reference's own torch statement, same inputs
        out[f"{tag}_q"], out[f"{tag}_k"], out[f"{tag}_v"] = q.numpy(), k.numpy(), v.numpy()
        out[f"{tag}_map"], out[f"{tag}_qc"], out[f"{tag}_kc"] = dmap.numpy(), qc.numpy().astype(np.int32), kc.numpy().astype(np.int32)
        out[f"{tag}_o"], out[f"{tag}_o_torch"] = o.numpy(), o_torch.numpy()
        print(f"{tag}: varblock {dt} S={S}: triton vs the reference's torch statement max abs diff {float((o.float() - o_torch.float()).abs().max()):.2e}")

    # ---------------- 5. head placement kernels (the reference ships a torch reference beside each: both are stored) ----------------
    import svg.models.cog.placement as cog_pl
    import svg.models.hyvideo.placement as hy_pl
    import svg.models.wan.placement as wan_pl

    for tag, (mod, fwd, inv, ctx, F_, P_) in {
        "pl_hy": (hy_pl, "hunyuan_sparse_head_placement", "hunyuan_hidden_states_placement", 20, 5, 37),
        "pl_wan": (wan_pl, "wan_sparse_head_placement", "wan_hidden_states_placement", 0, 4, 50),
        "pl_cog": (cog_pl, "sparse_head_placement", "hidden_states_placement", 17, 3, 61),
    }.items():
        g = torch.Generator().manual_seed(len(tag))
        cfg, H, D = 2, 3, 16
        S = ctx + F_ * P_
        q, k, v = (torch.randn(cfg, H, S, D, generator=g).to(torch.float16) for _ in range(3))
        best = torch.randint(0, 2, (cfg, H), generator=g).to(torch.int32)
        best[0, 0], best[0, 1] = 0, 1
        qo, ko, vo = (torch.zeros_like(q) for _ in range(3))
        getattr(mod, fwd)(q, k, v, qo, ko, vo, best, ctx, F_, P_)
        back = torch.zeros_like(q)
        getattr(mod, inv)(qo, back, best, ctx, F_, P_)
        assert torch.equal(back, q)                                        # (the round trip is asserted here, not stored)
        out[f"{tag}_q"], out[f"{tag}_best"], out[f"{tag}_qo"] = q.numpy(), best.numpy(), qo.numpy()
        if tag == "pl_hy":                                                 # k and v go through the same index arithmetic: one model keeps them
            out[f"{tag}_k"], out[f"{tag}_v"], out[f"{tag}_ko"], out[f"{tag}_vo"] = k.numpy(), v.numpy(), ko.numpy(), vo.numpy()
        out[f"{tag}_geo"] = np.array([ctx, F_, P_], dtype=np.int64)
        print(f"{tag}: placement S={S}: inverse(placement(q)) == q: {torch.equal(back, q)}")

    # ---------------- 6. token permutation by labels ----------------
    import svg.kernels.triton.permute as TP

    g = torch.Generator().manual_seed(21)
    B, H, S, D = 1, 3, 203, 32
    x = torch.randn(B, H, S, D, generator=g).to(torch.float16)
    labels = torch.randint(0, 9, (B, H, S), generator=g)
    sidx = torch.stack([torch.argsort(labels[0, h], stable=True) for h in range(H)])[None]    # (torch.argsort without stable=True is
    xp, sidx_out = TP.permute_tensor_by_labels_triton(x, None, 2, sorted_indices=sidx)         #  not reproducible: indices are passed in)
    xb = TP.apply_inverse_permutation_triton(xp, sidx_out.reshape(B, H, S), 2)
    assert torch.equal(xb, x)
    out["pm_x"], out["pm_labels"], out["pm_sidx"], out["pm_xp"] = x.numpy(), labels.numpy().astype(np.int32), sidx.numpy().astype(np.int32), xp.numpy()
    print(f"pm: permute S={S}: inverse(permute(x)) == x: {torch.equal(xb, x)}")

    # ---------------- 7. block glue: RMSNorm, LayerNorm (with / without affine), modulate ----------------
    from svg.kernels.triton.layernorm import triton_layernorm_forward
    from svg.kernels.triton.modulate import triton_modulate_gate_residual_forward, triton_modulate_shift_forward
    from svg.kernels.triton.rmsnorm import triton_rmsnorm_forward

    g = torch.Generator().manual_seed(31)
    M, N = 5, 1536                                         # Wan 1.3B hidden size: not a power of two (N2 = 2048, masked), BLOCK_M = 1
    for dt, sfx in ((torch.float16, "h"), (torch.float32, "f")):
        x = (torch.randn(1, M, N, generator=g) * 1.7 + 0.3).to(dt)
        w, b = (torch.randn(N, generator=g) * 0.2 + 1).float(), (torch.randn(N, generator=g) * 0.1).float()
        scale, shift, gate = (torch.randn(1, 1, N, generator=g) * 0.3).float(), torch.randn(1, 1, N, generator=g).float(), torch.randn(1, 1, N, generator=g).float()
        ln_p = triton_layernorm_forward(x, w, b, 1e-6, True)
        ln_n = triton_layernorm_forward(x, None, None, 1e-6, False)
        ms = triton_modulate_shift_forward(ln_n, scale, shift, output_dtype=dt)
        att = torch.randn(1, M, N, generator=g).to(dt)
        gr = triton_modulate_gate_residual_forward(x, att, gate, output_dtype=dt)
        rms = triton_rmsnorm_forward(x.reshape(M, N).contiguous(), w.to(dt), 1e-6)
        rms_y = rms[0] if isinstance(rms, (tuple, list)) else rms
        for name, t in (("x", x), ("w", w), ("b", b), ("scale", scale), ("shift", shift), ("gate", gate), ("att", att), ("ln_p", ln_p),
                        ("ln_n", ln_n), ("ms", ms), ("gr", gr), ("rms", rms_y)):
            out[f"gl_{sfx}_{name}"] = t.numpy()
        print(f"gl_{sfx}: glue {dt} M={M} N={N}: layernorm out {ln_p.dtype}, modulate out {ms.dtype}, rmsnorm out {rms_y.dtype}")

    # ---------------- 8. SVG2 layer-call of the reference's PROCESSORS on the reference's kernels ----------------
    # attention_core_logic of Hunyuan_SAPAttn_Processor2_0 (hyvideo/attention.py:715-804) and WanAttn_SAPAttn_Processor (wan/attention.py:500-556)
    # as they are: k-means (both Triton kernels) from given centroids, identify_dynamic_map, Triton permutation, the prompt / unused-prompt
    # post-processing (Hunyuan), inverse permutation.  The only substitution: `dynamic_block_sparse_fwd_flashinfer` (third-party, GPU-only) ->
    # the reference's own Triton kernel for the same operator, `dynamic_block_sparse_fwd_triton`.
    import json
    import tempfile

    class FP32LayerNorm(torch.nn.LayerNorm):
        """diffusers.models.normalization.FP32LayerNorm by its published definition (diffusers is absent): layer_norm of the input cast to
        fp32 with fp32 parameters, cast back.  Only the torch fall-back branch of the Wan block (section 14) calls it."""

        def forward(self, inputs):
            return torch.nn.functional.layer_norm(inputs.float(), self.normalized_shape, None if self.weight is None else self.weight.float(),
                                                  None if self.bias is None else self.bias.float(), self.eps).to(inputs.dtype)

    MG._stub("diffusers.models.normalization", RMSNorm=type("RMSNorm", (), {}), FP32LayerNorm=FP32LayerNorm)   # (wan/attention.py:11, wan/custom_models.py:5)
    import svg.models.hyvideo.attention as hy_attn
    import svg.models.wan.attention as wan_attn

    def own_triton(q, k, v, m, qc, kc, is_cpu=False):
        return KU.dynamic_block_sparse_fwd_triton(q.contiguous(), k.contiguous(), v.contiguous(), m, qc, kc)

    hy_attn.dynamic_block_sparse_fwd_flashinfer = own_triton
    wan_attn.dynamic_block_sparse_fwd_flashinfer = own_triton
    for tag, (model, H, D, F_, P_, ctx, L, QC, KC) in {"sap_hy": ("hy", 2, 64, 4, 48, 32, 20, 6, 8), "sap_wan": ("wan", 2, 64, 3, 64, 0, 0, 5, 7)}.items():
        g = torch.Generator().manual_seed(41 + QC)
        V, S = F_ * P_, F_ * P_ + ctx
        # well-separated modes, one warm-start centroid inside every mode: no point sits near a decision boundary, so the labels do not
        # depend on how the centroid norms are rounded (the fp16 reduction of the Triton kernel vs an fp32 one) and the layer-call is
        # comparable across implementations; mode sizes are ragged
        def modes(n_modes):
            centers = torch.randn(H, n_modes, D, generator=g) * 0.9
            lab = torch.randint(0, n_modes, (H, S), generator=g)
            lab[:, :n_modes] = torch.arange(n_modes)                                 # every mode has at least its seed point
            x = torch.gather(centers, 1, lab[..., None].expand(-1, -1, D)) + 0.15 * torch.randn(H, S, D, generator=g)
            return x.to(torch.float16)[None]

        q, k = modes(QC), modes(KC)
        v = torch.randn(1, H, S, D, generator=g).to(torch.float16)
        init_q, init_k = q[0, :, :QC].clone(), k[0, :, :KC].clone()                 # [H, QC, D] warm-start centroids: the seed points
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
        proc.num_q_centroids, proc.num_k_centroids, proc.top_p_kmeans, proc.min_kc_ratio = QC, KC, 0.8, 0.1
        proc.kmeans_iter_init, proc.kmeans_iter_step, proc.first_layers_fp, proc.first_times_fp = 0, 2, 0, 1.0
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
        out[f"{tag}_q"], out[f"{tag}_k"], out[f"{tag}_v"], out[f"{tag}_o"] = q.numpy(), k.numpy(), v.numpy(), o.numpy()
        out[f"{tag}_init_q"], out[f"{tag}_init_k"], out[f"{tag}_cq"], out[f"{tag}_ck"] = init_q.numpy(), init_k.numpy(), cq.numpy(), ck.numpy()
        out[f"{tag}_density"] = np.array(entry["density"], dtype=np.float64)
        out[f"{tag}_geo"] = np.array([H, D, F_, P_, ctx, L, QC, KC], dtype=np.int64)
        print(f"{tag}: processor layer-call S={S}: density per head {np.round(out[tag + '_density'], 3).tolist()}, output finite {bool(torch.isfinite(o.float()).all())}")

    # ---------------- 9. SVG1 layer-call of the reference's processors (Hunyuan, Wan, CogVideoX) ----------------
    # attention_core_logic of Hunyuan_SVGAttn_Processor2_0 (hyvideo/attention.py:473-524), WanAttn_SVGAttn_Processor2_0 (wan/attention.py:284-330)
    # and CogVideoX_SparseAttn_Processor2_0 (cog/attention.py:164-193) as they are: sample_mse on the two profiling masks, argmin, the Triton
    # head placement, torch flex_attention (eager, CPU) under the BlockMask of the model's mask_mod, the Triton inverse placement.  Heads are
    # built so that the online profiler's choice is unambiguous: one head attends its neighbours in frame-major order (spatial), one in
    # token-major order (temporal) — whatever rows are sampled.
    import math

    from torch.nn.attention.flex_attention import create_block_mask

    import svg.models.cog.attention as cog_attn
    import svg.models.cog.utils as cog_u
    import svg.models.hyvideo.utils as hy_u
    import svg.models.wan.utils as wan_u

    for tag, (H, D, F_, P_, ctx, L, mul) in {"svg1": (2, 64, 4, 128, 32, 20, 1.4), "svg1_wan": (2, 64, 4, 128, 0, 0, 0.9), "svg1_cog": (2, 64, 3, 128, 32, 0, 1.4)}.items():
        V, S = F_ * P_, F_ * P_ + ctx
        g = torch.Generator().manual_seed(77 + ctx + F_)
        i = torch.arange(V)
        pos = {0: i.float(), 1: ((i % P_) * F_ + i // P_).float()}              # frame-major / token-major position of video token i
        freqs = torch.arange(1, D // 2 + 1).float()

        def features(kind):                                                      # nearby positions -> nearly parallel vectors
            ang = 2 * math.pi * pos[kind][:, None] * freqs[None, :] / (4.0 * V)
            return torch.cat([torch.cos(ang), torch.sin(ang)], 1) * 1.6

        vid0 = ctx if tag == "svg1_cog" else 0                                    # CogVideoX: text first
        q = torch.randn(1, H, S, D, generator=g) * 0.3                            # (text rows stay noise)
        for h, kind in enumerate((0, 1)):
            q[0, h, vid0:vid0 + V] = features(kind)
        q = (q + 0.05 * torch.randn(1, H, S, D, generator=g)).to(torch.float16).float()   # fp16-representable values, fp32 arithmetic
        k = (q + 0.05 * torch.randn(1, H, S, D, generator=g)).to(torch.float16).float()
        v = torch.randn(1, H, S, D, generator=g).to(torch.float16).float()
        if tag == "svg1":
            cls = hy_attn.Hunyuan_SVGAttn_Processor2_0
            cls.prompt_length, cls.sample_mse_max_row, cls.first_times_fp = L, V, 1.0
            cls.attention_masks = [hy_u.get_attention_mask("spatial", V, ctx, F_, P_), hy_u.get_attention_mask("temporal", V, ctx, F_, P_, device="cpu")]
            mask_mod = hy_u.generate_temporal_head_mask_mod(ctx, L, F_, P_, mul=mul)
        elif tag == "svg1_wan":
            cls = wan_attn.WanAttn_SVGAttn_Processor2_0
            cls.sample_mse_max_row, cls.first_times_fp = V, 1.0
            cls.attention_masks = [wan_u.get_attention_mask("spatial", V, 0, F_, P_), wan_u.get_attention_mask("temporal", V, 0, F_, P_)]
            mask_mod = wan_u.generate_temporal_head_mask_mod(0, 0, F_, P_, mul=mul)
        else:
            cls = cog_attn.CogVideoX_SparseAttn_Processor2_0
            cls.first_times_fp = 0.0                                              # Cog: dense iff timestep > 1000 * (1 - first_times_fp)
            cls.attention_masks = [cog_u.get_attention_mask("spatial", ctx, F_, P_), cog_u.get_attention_mask("temporal", ctx, F_, P_)]
            mask_mod = cog_u.generate_temporal_head_mask_mod(ctx, F_, P_, mul=mul)
        cls.context_length, cls.num_frame, cls.frame_size, cls.num_sampled_rows, cls.first_layers_fp = ctx, F_, P_, 32, 0
        cls.block_mask = create_block_mask(mask_mod, None, None, S, S, device="cpu")
        proc = cls(0)
        seen = {}
        orig_mse = proc.sample_mse

        def spy(qq, kk, vv, _orig=orig_mse, _seen=seen):
            _seen["mse"] = _orig(qq, kk, vv)
            return _seen["mse"]

        proc.sample_mse = spy
        torch.manual_seed(5)
        ts = torch.tensor([0.5])
        o = proc.attention_core_logic(q.clone(), k.clone(), v.clone(), ts, 0, None) if tag == "svg1" else proc.attention_core_logic(q.clone(), k.clone(), v.clone(), ts)
        mse = seen["mse"].float()
        best = mse.argmin(0)
        margin = (mse.max(0).values / mse.min(0).values).min().item()
        torch.manual_seed(5)                                                      # the rows sample_mse drew: its randint is the first draw after the seed
        out[f"{tag}_rows"] = torch.randint(low=0, high=S if tag == "svg1_cog" else V, size=(32,)).numpy()
        out[f"{tag}_q"], out[f"{tag}_k"], out[f"{tag}_v"] = (t.to(torch.float16).numpy() for t in (q, k, v))
        out[f"{tag}_o"], out[f"{tag}_best"], out[f"{tag}_mse"] = o.to(torch.float16).numpy(), best.numpy(), mse.numpy()
        out[f"{tag}_geo"] = np.array([H, D, F_, P_, ctx, L], dtype=np.int64)
        out[f"{tag}_mul"] = np.float64(mul)
        n_txt = int((torch.from_numpy(out[f"{tag}_rows"]) < vid0).sum()) if tag == "svg1_cog" else 0
        print(f"{tag}: SVG1 processor S={S}: best_mask_idx {best.tolist()}, worst MSE ratio between the two masks {margin:.1f}x"
              + (f"; {n_txt} sampled TEXT rows -> NaN under the temporal profiling mask (no key allowed, cog/utils.py get_attention_mask) -> argmin picks it"
                 if n_txt else ""))

    # ---------------- 10. the Wan uniform-block (BSR) op's mask generator ----------------
    # svg/kernels/ops/attention_ops_wan.py: get_factor (block size = largest divisor of the frame size below 256) and ref_gen_temporal_mask
    import types

    if "matplotlib" not in sys.modules:
        try:
            import matplotlib  # noqa: F401
        except ImportError:
            MG._stub("matplotlib", pyplot=types.SimpleNamespace())
            MG._stub("matplotlib.pyplot")
    import svg.kernels.ops.attention_ops_wan as ops_wan

    for F_, P_, mul in ((5, 40, 1.4), (4, 150, 0.6), (3, 96, 2.0), (21, 3600, 1.8)):
        bs = ops_wan.get_factor(F_, P_)
        blk = ops_wan.ref_gen_temporal_mask(F_, P_, mul) != -1
        out[f"wbsr_{F_}_{P_}_{mul}"] = np.packbits(blk.reshape(-1))
        out[f"wbsr_{F_}_{P_}_{mul}_bs"] = np.int64(bs)
    out["wbsr_factors"] = np.array([[p_, ops_wan.get_factor(1, p_)] for p_ in (40, 96, 150, 255, 256, 257, 1350, 3600, 4080, 509)], dtype=np.int64)
    print("wbsr: block sizes", {int(a): int(b) for a, b in out["wbsr_factors"]})

    # ---------------- 11. the whole `__call__` of the reference's Wan SVG processor ----------------
    # WanAttn_SVGAttn_Processor2_0.__call__ (wan/attention.py:151-208) on a duck-typed attention module (tests/standins.py): q / k / v projections,
    # the Triton RMSNorm across heads (interpreted), head split, the torch RoPE fall-back (complex multiply in fp64, :58-66), attention_core_logic
    # as in section 9, output projection.  q and k projections are the identity so that the two heads keep the spatial / temporal structure
    # of the hidden states (unambiguous profiler choice); weights and inputs are fp16-representable, the arithmetic is the reference's fp32.
    sys.path.insert(0, str(HERE.parent))
    import standins

    wan_attn.DiffusersRMSNorm = standins.RMSNorm           # `isinstance(attn.norm_q, DiffusersRMSNorm)` (:108)
    heads, hd, F_, P_, mul = 2, 64, 4, 128, 0.9
    dim, S = heads * hd, F_ * P_
    g = torch.Generator().manual_seed(123)
    attn = standins.Attention(dim, heads, qk_norm="rms", across_heads=True, dtype=torch.float32)
    with torch.no_grad():
        for lin in (attn.to_q, attn.to_k):
            lin.weight.copy_(torch.eye(dim))
            lin.bias.zero_()
        attn.to_v.weight.copy_((torch.randn(dim, dim, generator=g) / dim ** 0.5).half().float())
        attn.to_v.bias.copy_((torch.randn(dim, generator=g) * 0.1).half().float())
        attn.to_out[0].weight.copy_((torch.randn(dim, dim, generator=g) / dim ** 0.5).half().float())
        attn.to_out[0].bias.copy_((torch.randn(dim, generator=g) * 0.1).half().float())
        # per head: 48 structure channels (q / k see them), 16 noise channels the q / k norm weights all but switch off — v, a random
        # projection of everything, is dominated by them and so differs from one position to the next (a wrong mask costs a large MSE)
        chan_w = torch.cat([torch.full((48,), 1.6), torch.full((16,), 0.03)]).repeat(heads)
        attn.norm_q.weight.copy_((chan_w * (1 + 0.1 * torch.randn(dim, generator=g))).half().float())
        attn.norm_k.weight.copy_((chan_w * (1 + 0.1 * torch.randn(dim, generator=g))).half().float())
    i = torch.arange(S)
    pos = {0: i.float(), 1: ((i % P_) * F_ + i // P_).float()}
    freqs = torch.arange(1, 25).float()
    feats = []
    for kind in (0, 1):
        ang = 2 * math.pi * pos[kind][:, None] * freqs[None, :] / (4.0 * S)
        feats.append(torch.cat([torch.cos(ang) * 2.2, torch.sin(ang) * 2.2, 1.5 * torch.randn(S, 16, generator=g)], 1))
    hidden = (torch.cat(feats, 1)[None] + 0.05 * torch.randn(1, S, dim, generator=g)).half().float()
    rope_ang = 0.03 * torch.rand(S, hd // 2, generator=g)                                  # small rotations: the structure survives
    cls = wan_attn.WanAttn_SVGAttn_Processor2_0
    cls.context_length, cls.num_frame, cls.frame_size, cls.num_sampled_rows, cls.sample_mse_max_row = 0, F_, P_, 32, S
    cls.first_layers_fp, cls.first_times_fp = 0, 1.0
    cls.attention_masks = [wan_u.get_attention_mask("spatial", S, 0, F_, P_), wan_u.get_attention_mask("temporal", S, 0, F_, P_)]
    cls.block_mask = create_block_mask(wan_u.generate_temporal_head_mask_mod(0, 0, F_, P_, mul=mul), None, None, S, S, device="cpu")
    proc = cls(0)
    seen = {}
    orig_mse = proc.sample_mse
    proc.sample_mse = lambda a, b, c: seen.setdefault("mse", orig_mse(a, b, c))
    torch.manual_seed(5)
    with torch.no_grad():
        o = proc(attn, hidden, rotary_emb=torch.polar(torch.ones_like(rope_ang).double(), rope_ang.double())[None, None], timestep=torch.tensor([0.5]))
    best = seen["mse"].float().argmin(0)
    out["call_wan_hidden"], out["call_wan_o"], out["call_wan_best"] = hidden.half().numpy(), o.half().numpy(), best.numpy()
    out["call_wan_rope_ang"] = rope_ang.numpy()
    for n, t in (("wv", attn.to_v.weight), ("bv", attn.to_v.bias), ("wo", attn.to_out[0].weight), ("bo", attn.to_out[0].bias),
                 ("nq", attn.norm_q.weight), ("nk", attn.norm_k.weight)):
        out[f"call_wan_{n}"] = t.detach().half().numpy()
    out["call_wan_geo"] = np.array([heads, hd, F_, P_], dtype=np.int64)
    out["call_wan_mul"] = np.float64(mul)
    m_ = seen["mse"].float()
    print(f"call_wan: Wan SVG processor __call__ S={S}: best_mask_idx {best.tolist()}, MSE ratio {(m_.max(0).values / m_.min(0).values).min().item():.1f}x, out {tuple(o.shape)}")

    # ---------------- 12. the whole `__call__` of the reference's Hunyuan SVG processor: double-stream and single-stream block ----------------
    # Hunyuan_SVGAttn_Processor2_0.__call__ (hyvideo/attention.py:328-374) with the reference's own fall-backs for the CUDA extension that is
    # not built here (:196-224: the norm modules' forward, diffusers' apply_rotary_emb on the video rows).  diffusers is absent (a dependency
    # of the reference, not vendored): for its `apply_rotary_emb(x, (cos, sin))` the REFERENCE'S OWN statement of that function is loaded from
    # the reference's kernel test (ref_host_apply_rope, svg/kernels/test/test_apply_rope.py:24-37, "Ref: diffusers/models/embeddings.py").
    # Everything else is the reference's code as it is: projections, per-head RMSNorm, RoPE on the video rows only, the text stream's own
    # projections and norms (double) or the concatenated sequence (single), attention_core_logic as in section 9, the split and the output
    # projections.
    import importlib.util

    if "_kernels" not in sys.modules:
        MG._stub("_kernels")         # the test module imports the CUDA extension at the top; only its host reference is used (stubbed AFTER
                                     # the processors were imported: they took their fall-back branch, ENABLE_FAST_KERNEL False)
    spec = importlib.util.spec_from_file_location("ref_test_apply_rope", Path(MG.REF) / "svg/kernels/test/test_apply_rope.py")
    ref_rope = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_rope)
    hy_attn.apply_rotary_emb = lambda x, freqs_cis: ref_rope.ref_host_apply_rope(x, freqs_cis[0], freqs_cis[1])
    heads, hd, F_, P_, ctx, L, mul = 2, 64, 4, 128, 32, 20, 1.4
    dim, V = heads * hd, F_ * P_
    S = V + ctx
    for tag, single in (("call_hyd", False), ("call_hys", True)):
        g = torch.Generator().manual_seed(321 + single)
        attn = standins.Attention(dim, heads, qk_norm="rms", added_kv=not single, dtype=torch.float32)

        def h16(*shape, s=1.0):
            return (torch.randn(*shape, generator=g) * s).half().float()

        with torch.no_grad():
            for lin in (attn.to_q, attn.to_k):
                lin.weight.copy_(torch.eye(dim))
                lin.bias.zero_()
            attn.to_v.weight.copy_(h16(dim, dim, s=dim ** -0.5)), attn.to_v.bias.copy_(h16(dim, s=0.1))
            chan_w = torch.cat([torch.full((48,), 1.6), torch.full((16,), 0.03)])
            attn.norm_q.weight.copy_((chan_w * (1 + 0.1 * torch.randn(hd, generator=g))).half().float())
            attn.norm_k.weight.copy_((chan_w * (1 + 0.1 * torch.randn(hd, generator=g))).half().float())
            if single:
                attn.to_out = None                               # HunyuanVideo's single-stream blocks are `pre_only`: no output projection
            else:
                attn.to_out[0].weight.copy_(h16(dim, dim, s=dim ** -0.5)), attn.to_out[0].bias.copy_(h16(dim, s=0.1))
                for lin in (attn.add_q_proj, attn.add_k_proj, attn.add_v_proj, attn.to_add_out):
                    lin.weight.copy_(h16(dim, dim, s=dim ** -0.5)), lin.bias.copy_(h16(dim, s=0.1))
                attn.norm_added_q.weight.copy_((1 + 0.1 * torch.randn(hd, generator=g)).half().float())
                attn.norm_added_k.weight.copy_((1 + 0.1 * torch.randn(hd, generator=g)).half().float())
        i = torch.arange(V)
        pos = {0: i.float(), 1: ((i % P_) * F_ + i // P_).float()}
        freqs = torch.arange(1, 25).float()
        feats = []
        for kind in (0, 1):
            ang = 2 * math.pi * pos[kind][:, None] * freqs[None, :] / (4.0 * V)
            feats.append(torch.cat([torch.cos(ang) * 2.2, torch.sin(ang) * 2.2, 1.5 * torch.randn(V, 16, generator=g)], 1))
        hidden = (torch.cat(feats, 1)[None] + 0.05 * torch.randn(1, V, dim, generator=g)).half().float()
        enc = h16(1, ctx, dim)
        rope_ang = 0.03 * torch.rand(V, hd // 2, generator=g)
        rope = (rope_ang.cos().repeat_interleave(2, -1), rope_ang.sin().repeat_interleave(2, -1))    # [V, hd], pair (2i, 2i+1) shares an angle
        amask = torch.zeros(S, dtype=torch.bool)
        amask[:V + L] = True
        cls = hy_attn.Hunyuan_SVGAttn_Processor2_0
        cls.context_length, cls.num_frame, cls.frame_size, cls.num_sampled_rows, cls.sample_mse_max_row = ctx, F_, P_, 32, V
        cls.prompt_length, cls.first_layers_fp, cls.first_times_fp = L, 0, 1.0
        cls.attention_masks = [hy_u.get_attention_mask("spatial", V, ctx, F_, P_), hy_u.get_attention_mask("temporal", V, ctx, F_, P_, device="cpu")]
        cls.block_mask = create_block_mask(hy_u.generate_temporal_head_mask_mod(ctx, L, F_, P_, mul=mul), None, None, S, S, device="cpu")
        proc = cls(0)
        seen = {}
        orig_mse = proc.sample_mse
        proc.sample_mse = lambda a, b, c, _o=orig_mse, _s=seen: _s.setdefault("mse", _o(a, b, c))
        torch.manual_seed(5)
        with torch.no_grad():
            o_h, o_e = proc(attn, hidden, encoder_hidden_states=enc, attention_mask=amask, image_rotary_emb=rope, timestep=torch.tensor([0.5]))
        m_ = seen["mse"].float()
        best = m_.argmin(0)
        out[f"{tag}_hidden"], out[f"{tag}_enc"] = hidden.half().numpy(), enc.half().numpy()
        out[f"{tag}_o_h"], out[f"{tag}_o_e"], out[f"{tag}_best"] = o_h.half().numpy(), o_e.half().numpy(), best.numpy()
        out[f"{tag}_rope_ang"] = rope_ang.numpy()
        names = [("wv", attn.to_v.weight), ("bv", attn.to_v.bias), ("nq", attn.norm_q.weight), ("nk", attn.norm_k.weight)]
        if not single:
            names += [("wo", attn.to_out[0].weight), ("bo", attn.to_out[0].bias), ("naq", attn.norm_added_q.weight), ("nak", attn.norm_added_k.weight)]
            for n, lin in (("aq", attn.add_q_proj), ("ak", attn.add_k_proj), ("av", attn.add_v_proj), ("ao", attn.to_add_out)):
                names += [("w" + n, lin.weight), ("b" + n, lin.bias)]
        for n, t in names:
            out[f"{tag}_{n}"] = t.detach().half().numpy()
        out[f"{tag}_geo"] = np.array([heads, hd, F_, P_, ctx, L], dtype=np.int64)
        out[f"{tag}_mul"] = np.float64(mul)
        print(f"{tag}: Hunyuan SVG processor __call__ ({'single' if single else 'double'}-stream) S={S}: best_mask_idx {best.tolist()}, "
              f"MSE ratio {(m_.max(0).values / m_.min(0).values).min().item():.1f}x, out {tuple(o_h.shape)} + {tuple(o_e.shape)}")

    # ---------------- 13. the whole `__call__` of the reference's CogVideoX SVG processor ----------------
    # CogVideoX_SparseAttn_Processor2_0.__call__ (cog/attention.py:199-224) with its own fall-backs (:40-50: the LayerNorm modules' forward over
    # head_dim, apply_rotary_emb — the reference's ref_host_apply_rope again — on the video rows, which come AFTER the text).  Two calls on the
    # same module: one under a seed whose 32 profiler rows are all video rows (the structure of the heads decides: spatial, temporal), one
    # under a seed that draws a text row (NaN under the temporal profiling mask -> argmin sends EVERY head temporal; the quirk of section 9,
    # here through the whole call).  `sample_mse` draws with the CPU generator right after the seed, so a caller that seeds the same way
    # profiles the same rows.
    cog_attn.apply_rotary_emb = hy_attn.apply_rotary_emb
    heads, hd, F_, P_, ctx, mul = 2, 64, 6, 128, 32, 1.4
    dim, V = heads * hd, F_ * P_
    S = V + ctx
    g = torch.Generator().manual_seed(4321)
    attn = standins.Attention(dim, heads, qk_norm="layer", dtype=torch.float32)

    def h16(*shape, s=1.0):
        return (torch.randn(*shape, generator=g) * s).half().float()

    with torch.no_grad():
        for lin in (attn.to_q, attn.to_k):
            lin.weight.copy_(torch.eye(dim))
            lin.bias.zero_()
        attn.to_v.weight.copy_(h16(dim, dim, s=dim ** -0.5)), attn.to_v.bias.copy_(h16(dim, s=0.1))
        attn.to_out[0].weight.copy_(h16(dim, dim, s=dim ** -0.5)), attn.to_out[0].bias.copy_(h16(dim, s=0.1))
        chan_w = torch.cat([torch.full((48,), 1.6), torch.full((16,), 0.03)])
        for nm in (attn.norm_q, attn.norm_k):
            nm.weight.copy_((chan_w * (1 + 0.1 * torch.randn(hd, generator=g))).half().float())
            nm.bias.copy_(h16(hd, s=0.02))
    i = torch.arange(V)
    pos = {0: i.float(), 1: ((i % P_) * F_ + i // P_).float()}
    freqs = torch.arange(1, 25).float()
    feats = []
    for kind in (0, 1):
        ang = 2 * math.pi * pos[kind][:, None] * freqs[None, :] / (4.0 * V)
        feats.append(torch.cat([torch.cos(ang) * 2.2, torch.sin(ang) * 2.2, 1.5 * torch.randn(V, 16, generator=g)], 1))
    hidden = (torch.cat(feats, 1)[None] + 0.05 * torch.randn(1, V, dim, generator=g)).half().float()
    enc = h16(1, ctx, dim)
    rope_ang = 0.03 * torch.rand(V, hd // 2, generator=g)
    rope = (rope_ang.cos().repeat_interleave(2, -1), rope_ang.sin().repeat_interleave(2, -1))
    cls = cog_attn.CogVideoX_SparseAttn_Processor2_0
    cls.context_length, cls.num_frame, cls.frame_size, cls.num_sampled_rows, cls.first_layers_fp, cls.first_times_fp = ctx, F_, P_, 32, 0, 0.0
    cls.attention_masks = [cog_u.get_attention_mask("spatial", ctx, F_, P_), cog_u.get_attention_mask("temporal", ctx, F_, P_)]
    cls.block_mask = create_block_mask(cog_u.generate_temporal_head_mask_mod(ctx, F_, P_, mul=mul), None, None, S, S, device="cpu")

    def drawn(seed):
        torch.manual_seed(seed)
        return torch.randint(low=0, high=S, size=(32,))

    seed_video = next(sd for sd in range(1000) if int((drawn(sd) < ctx).sum()) == 0)
    seed_text = next(sd for sd in range(1000) if int((drawn(sd) < ctx).sum()) > 0)
    for nm, seed in (("v", seed_video), ("t", seed_text)):
        proc = cls(0)
        seen = {}
        orig_mse = proc.sample_mse
        proc.sample_mse = lambda a, b, c, _o=orig_mse, _s=seen: _s.setdefault("mse", _o(a, b, c))
        torch.manual_seed(seed)
        with torch.no_grad():
            o_h, o_e = proc(attn, hidden.clone(), enc.clone(), image_rotary_emb=rope, timestep=torch.tensor([0.5]))
        m_ = seen["mse"].float()
        best = m_.argmin(0)
        out[f"call_cog_{nm}_o_h"], out[f"call_cog_{nm}_o_e"], out[f"call_cog_{nm}_best"] = o_h.half().numpy(), o_e.half().numpy(), best.numpy()
        out[f"call_cog_{nm}_seed"] = np.int64(seed)
        print(f"call_cog[{nm}]: CogVideoX SVG processor __call__ S={S}, seed {seed} ({int((drawn(seed) < ctx).sum())} text rows drawn): best_mask_idx "
              f"{best.tolist()}, MSE ratio {(m_.max(0).values / m_.min(0).values).min().item():.1f}x, out {tuple(o_h.shape)} + {tuple(o_e.shape)}")
    out["call_cog_hidden"], out["call_cog_enc"], out["call_cog_rope_ang"] = hidden.half().numpy(), enc.half().numpy(), rope_ang.numpy()
    for n, t in (("wv", attn.to_v.weight), ("bv", attn.to_v.bias), ("wo", attn.to_out[0].weight), ("bo", attn.to_out[0].bias),
                 ("nq", attn.norm_q.weight), ("nqb", attn.norm_q.bias), ("nk", attn.norm_k.weight), ("nkb", attn.norm_k.bias)):
        out[f"call_cog_{n}"] = t.detach().half().numpy()
    out["call_cog_geo"] = np.array([heads, hd, F_, P_, ctx], dtype=np.int64)
    out["call_cog_mul"] = np.float64(mul)

    # ---------------- 14. the reference's Wan transformer BLOCK forward, both branches ----------------
    # WanTransformerBlock_Sparse.forward (wan/custom_models.py:23-111) called as a plain function on a duck-typed block (scale_shift_table,
    # norm1/2/3, attn1/attn2/ffn): once with ENABLE_FAST_KERNEL (its Triton LayerNorm / modulate-shift / gate-residual kernels, interpreted)
    # and once on its torch fall-back.  Hidden size 192 — not a power of two, rows with a non-zero mean — so the two branches DIFFER: the
    # Triton LayerNorm counts its zero padding (to 256) in the variance (section 7's finding), the fall-back is FP32LayerNorm.  attn1 /
    # attn2 / ffn are fixed linear stand-ins (what they compute is not the block's business); diffusers' base classes are empty stubs.
    MG._stub("diffusers.models.modeling_outputs", Transformer2DModelOutput=object)
    MG._stub("diffusers.models.transformers")
    MG._stub("diffusers.models.transformers.transformer_wan", WanTransformer3DModel=type("WanTransformer3DModel", (), {}),
             WanTransformerBlock=type("WanTransformerBlock", (), {}))
    MG._stub("diffusers.utils", USE_PEFT_BACKEND=False, scale_lora_layers=lambda *a, **k: None, unscale_lora_layers=lambda *a, **k: None)
    import svg.models.wan.custom_models as wan_cm

    # Batch 1: the reference's modulate kernels load ONE scale / shift / gate vector (`tl.load(SCALE + cols)`, kernels/triton/modulate.py:35-36,113)
    # — with a [B, 1, C] modulation they apply batch 0's to every batch (the fast path is a batch-1 path, which is how the Wan pipeline calls it).
    C, S_, B_ = 192, 64, 1     # B_ * S_ a multiple of 32: for hidden sizes <= 512 the reference's kernels take 32 rows per program WITHOUT a row
                               # mask (layernorm.py:27-35,60-61,78-81) and would read and WRITE past the last row otherwise
    g = torch.Generator().manual_seed(99)

    def h16(*shape, s=1.0):
        return (torch.randn(*shape, generator=g) * s).half().float()

    lin = {n: (h16(C, C, s=C ** -0.5), h16(C, s=0.1)) for n in ("attn1", "attn2", "ffn")}
    blk = types.SimpleNamespace(
        scale_shift_table=h16(1, 6, C, s=C ** -0.5),
        norm1=FP32LayerNorm(C, eps=1e-6, elementwise_affine=False), norm3=FP32LayerNorm(C, eps=1e-6, elementwise_affine=False),
        norm2=FP32LayerNorm(C, eps=1e-6, elementwise_affine=True),
        attn1=lambda hidden_states, rotary_emb=None, timestep=None: torch.roll(hidden_states, 1, 1) @ lin["attn1"][0].T + lin["attn1"][1],
        attn2=lambda hidden_states, encoder_hidden_states=None: hidden_states @ lin["attn2"][0].T + lin["attn2"][1] + encoder_hidden_states.mean(1, keepdim=True),
        ffn=lambda x: torch.tanh(x @ lin["ffn"][0].T + lin["ffn"][1]))
    # The affine Triton kernel loads W and B WITHOUT a mask over the padded width (`tl.load(W + cols)`, kernels/triton/layernorm.py:52-53):
    # for a hidden size that is not a power of two it reads past both vectors (the values only reach masked-off columns).  On a GPU that
    # goes unnoticed; the interpreter segfaults on it.  Here the two vectors are views into 256-element buffers, so the reads stay in bounds.
    pad_w, pad_b = torch.zeros(256), torch.zeros(256)
    blk.norm2.weight, blk.norm2.bias = torch.nn.Parameter(pad_w[:C]), torch.nn.Parameter(pad_b[:C])
    with torch.no_grad():
        blk.norm2.weight.copy_((1 + h16(C, s=0.2)).half().float()), blk.norm2.bias.copy_(h16(C, s=0.1))
    hidden = (h16(B_, S_, C, s=1.3) + 0.9 * torch.randn(B_, S_, 1, generator=g)).half().float()      # per-row mean offset
    enc, temb = h16(B_, 5, C), h16(B_, 6, C, s=0.3)
    fwd = wan_cm.WanTransformerBlock_Sparse.forward
    with torch.no_grad():
        for name, fast in (("fast", True), ("torch", False)):
            wan_cm.ENABLE_FAST_KERNEL = fast
            out[f"blk_{name}_out"] = fwd(blk, hidden.clone(), enc, temb, None, timestep=0).numpy()
    d = np.abs(out["blk_fast_out"] - out["blk_torch_out"])
    out["blk_hidden"], out["blk_enc"], out["blk_temb"], out["blk_table"] = hidden.half().numpy(), enc.half().numpy(), temb.half().numpy(), blk.scale_shift_table.half().numpy()
    out["blk_n2w"], out["blk_n2b"] = blk.norm2.weight.detach().half().numpy(), blk.norm2.bias.detach().half().numpy()
    for n, (w_, b_) in lin.items():
        out[f"blk_{n}_w"], out[f"blk_{n}_b"] = w_.half().numpy(), b_.half().numpy()
    print(f"blk: Wan block forward C={C}: Triton branch vs torch branch differ by up to {d.max():.3f} (mean {d.mean():.4f}) — the padded variance")

    # ---------------- 15. the whole `__call__` of the reference's Cosmos SVG processor: self attention (sparse) and cross attention ----------------
    # Cosmos_SVG_AttnProcessor2_0.__call__ (cosmos/attention.py:73-124): projections, head split, per-head norm modules, diffusers' apply_rotary_emb
    # in its HALF-SPLIT form (use_real_unbind_dim=-2: the channel halves are the real / imaginary parts), attention_core_logic (the Wan one:
    # cosmos/utils.py is wan/utils.py), output projection; with `timestep=None` and encoder states it is the block's cross attention (torch
    # SDPA, no RoPE given).  diffusers is absent and the reference has no statement of its own for the -2 form (its kernel tests cover -1):
    # restated here from diffusers' published definition — x_real, x_imag = x.reshape(..., 2, D/2).unbind(-2); rot = cat(-x_imag, x_real).
    import svg.models.cosmos.attention as cos_attn
    import svg.models.cosmos.utils as cos_u

    def diffusers_rope(x, freqs_cis, use_real=True, use_real_unbind_dim=-1):
        assert use_real and use_real_unbind_dim == -2
        cos, sin = freqs_cis
        cos, sin = cos[None, None], sin[None, None]
        x_real, x_imag = x.reshape(*x.shape[:-1], 2, -1).unbind(-2)
        x_rot = torch.cat([-x_imag, x_real], dim=-1)
        return (x.float() * cos + x_rot.float() * sin).to(x.dtype)

    cos_attn.apply_rotary_emb = diffusers_rope
    heads, hd, F_, P_, mul = 2, 64, 4, 128, 0.9
    dim, S = heads * hd, F_ * P_
    g = torch.Generator().manual_seed(777)
    attn = standins.Attention(dim, heads, qk_norm="rms", dtype=torch.float32)

    def h16(*shape, s=1.0):
        return (torch.randn(*shape, generator=g) * s).half().float()

    with torch.no_grad():
        for lin in (attn.to_q, attn.to_k):
            lin.weight.copy_(torch.eye(dim))
            lin.bias.zero_()
        attn.to_v.weight.copy_(h16(dim, dim, s=dim ** -0.5)), attn.to_v.bias.copy_(h16(dim, s=0.1))
        attn.to_out[0].weight.copy_(h16(dim, dim, s=dim ** -0.5)), attn.to_out[0].bias.copy_(h16(dim, s=0.1))
        # channel layout of a head for the half-split RoPE: [24 cos | 8 noise | 24 sin | 8 noise] — pair (c, c + 32) rotates together
        chan_w = torch.cat([torch.full((24,), 1.6), torch.full((8,), 0.03)]).repeat(2)
        attn.norm_q.weight.copy_((chan_w * (1 + 0.1 * torch.randn(hd, generator=g))).half().float())
        attn.norm_k.weight.copy_((chan_w * (1 + 0.1 * torch.randn(hd, generator=g))).half().float())
    i = torch.arange(S)
    pos = {0: i.float(), 1: ((i % P_) * F_ + i // P_).float()}
    freqs = torch.arange(1, 25).float()
    feats = []
    for kind in (0, 1):
        ang = 2 * math.pi * pos[kind][:, None] * freqs[None, :] / (4.0 * S)
        feats.append(torch.cat([torch.cos(ang) * 2.2, 1.5 * torch.randn(S, 8, generator=g), torch.sin(ang) * 2.2, 1.5 * torch.randn(S, 8, generator=g)], 1))
    hidden = (torch.cat(feats, 1)[None] + 0.05 * torch.randn(1, S, dim, generator=g)).half().float()
    rope_ang = 0.03 * torch.rand(S, hd // 2, generator=g)
    rope = (torch.cat([rope_ang.cos()] * 2, -1), torch.cat([rope_ang.sin()] * 2, -1))            # [S, hd]: both halves share the angles
    cls = cos_attn.Cosmos_SVG_AttnProcessor2_0
    cls.context_length, cls.num_frame, cls.frame_size, cls.num_sampled_rows, cls.sample_mse_max_row = 0, F_, P_, 32, S
    cls.first_layers_fp, cls.first_times_fp = 0, 1.0
    cls.attention_masks = [cos_u.get_attention_mask("spatial", S, 0, F_, P_), cos_u.get_attention_mask("temporal", S, 0, F_, P_)]
    cls.block_mask = create_block_mask(cos_u.generate_temporal_head_mask_mod(0, 0, F_, P_, mul=mul), None, None, S, S, device="cpu")
    proc = cls(0)
    seen = {}
    orig_mse = proc.sample_mse
    proc.sample_mse = lambda a, b, c: seen.setdefault("mse", orig_mse(a, b, c))
    torch.manual_seed(5)
    with torch.no_grad():
        o = proc(attn, hidden, image_rotary_emb=rope, timestep=torch.tensor([0.5]))
        enc = h16(1, 40, dim)
        o_cross = proc(attn, hidden, encoder_hidden_states=enc)                                  # timestep None: cross attention
    m_ = seen["mse"].float()
    best = m_.argmin(0)
    out["call_cos_hidden"], out["call_cos_o"], out["call_cos_best"], out["call_cos_rope_ang"] = hidden.half().numpy(), o.half().numpy(), best.numpy(), rope_ang.numpy()
    out["call_cos_enc"], out["call_cos_o_cross"] = enc.half().numpy(), o_cross.half().numpy()
    for n, t in (("wv", attn.to_v.weight), ("bv", attn.to_v.bias), ("wo", attn.to_out[0].weight), ("bo", attn.to_out[0].bias),
                 ("nq", attn.norm_q.weight), ("nk", attn.norm_k.weight)):
        out[f"call_cos_{n}"] = t.detach().half().numpy()
    out["call_cos_geo"] = np.array([heads, hd, F_, P_], dtype=np.int64)
    out["call_cos_mul"] = np.float64(mul)
    print(f"call_cos: Cosmos SVG processor __call__ S={S}: best_mask_idx {best.tolist()}, MSE ratio {(m_.max(0).values / m_.min(0).values).min().item():.1f}x, "
          f"out {tuple(o.shape)}; cross attention out {tuple(o_cross.shape)}")

    # ---------------- 16. the reference's Wan processor as CROSS attention: text (T2V) and text + CLIP image tokens (I2V) ----------------
    # WanAttn_SVGAttn_Processor2_0.__call__ with encoder states and `timestep=None` (wan/attention.py:151-208): q from the video tokens, k / v from
    # the encoder tokens, the Triton RMSNorm across heads on q and k (interpreted), torch SDPA; with `add_k_proj` (I2V) the first 257 encoder
    # tokens are the image branch — its own k / v projections, `norm_added_k` by the MODULE's forward, a second SDPA with the same q, the two
    # results added before the output projection.
    heads, hd = 2, 64
    dim, S_v, n_txt = heads * hd, 96, 64      # (row counts in multiples of 32: below 513 columns the Triton RMSNorm takes 32 rows per program without a row mask)
    g = torch.Generator().manual_seed(2468)

    def h16(*shape, s=1.0):
        return (torch.randn(*shape, generator=g) * s).half().float()

    for tag, i2v in (("xwan_t2v", False), ("xwan_i2v", True)):
        attn = standins.Attention(dim, heads, qk_norm="rms", across_heads=True, added_kv=i2v, dtype=torch.float32)
        mods = [("q", attn.to_q), ("k", attn.to_k), ("v", attn.to_v), ("o", attn.to_out[0])]
        if i2v:
            attn.norm_added_k = standins.RMSNorm(dim)                     # Wan: across heads, like norm_k
            mods += [("ak", attn.add_k_proj), ("av", attn.add_v_proj)]
        with torch.no_grad():
            for n, lin in mods:
                lin.weight.copy_(h16(dim, dim, s=dim ** -0.5)), lin.bias.copy_(h16(dim, s=0.1))
                out[f"{tag}_w{n}"], out[f"{tag}_b{n}"] = lin.weight.detach().half().numpy(), lin.bias.detach().half().numpy()
            norms = [("nq", attn.norm_q), ("nk", attn.norm_k)] + ([("nak", attn.norm_added_k)] if i2v else [])
            for n, nm in norms:
                nm.weight.copy_((1 + h16(dim, s=0.2)).half().float())
                out[f"{tag}_{n}"] = nm.weight.detach().half().numpy()
        hidden, enc = h16(1, S_v, dim), h16(1, n_txt + (257 if i2v else 0), dim)
        proc = wan_attn.WanAttn_SVGAttn_Processor2_0(0)
        with torch.no_grad():
            o = proc(attn, hidden, encoder_hidden_states=enc)
        out[f"{tag}_hidden"], out[f"{tag}_enc"], out[f"{tag}_o"] = hidden.half().numpy(), enc.half().numpy(), o.half().numpy()
        out[f"{tag}_geo"] = np.array([heads, hd], dtype=np.int64)
        print(f"{tag}: Wan cross attention ({'I2V: 257 image tokens + ' if i2v else ''}{n_txt} text tokens): out {tuple(o.shape)}, finite {bool(torch.isfinite(o).all())}")

    p = HERE / "triton_golden.npz"
    np.savez_compressed(p, **out)
    print(f"wrote {p} ({p.stat().st_size / 1024:.0f} KB)")


if __name__ == "__main__":
    main()
