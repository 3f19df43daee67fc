#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REFERENCE's own code (read-only tree at
/root/reference) on CPU in this container.  The fixtures pin oracle/svg_oracle.py; they travel with the repo,
the reference does not.  Re-run:  python tests/golden/make_golden.py

What can be executed from the reference here (everything else is GPU-only Triton / flashinfer / flash-attn):
  * svg.models.{hyvideo,wan,cog}.utils.generate_temporal_head_mask_mod  + torch flex_attention eager on CPU
  * svg.models.{hyvideo,wan,cog}.utils.get_attention_mask  (with Tensor.cuda patched to identity)
  * svg.models.hyvideo.attention.Hunyuan_SVGAttn_Processor2_0.sample_mse
  * svg.models.{hyvideo,cog}.placement.ref_* (torch reference placement)
  * svg.kmeans_utils: permute_tensor_by_labels, apply_inverse_permutation, weighted_softmax, identify_dynamic_map,
    dynamic_block_sparse_fwd_torch, density_calculation
  * svg.models.hyvideo.attention.Hunyuan_SAPAttn_Processor2_0.dynamic_map_post_processing
Third-party modules absent here (flashinfer, cuvs, flash_attn, diffusers, loguru, termcolor) are stubbed: none of
the functions above touches them.
"""
import hashlib
import sys
import types
from pathlib import Path

import numpy as np
import torch

REF = "/root/reference"
OUT = Path(__file__).resolve().parent


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    class _Any:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return None

        def __getattr__(self, n):
            return _Any()

    fi = _stub("flashinfer", sparse=_Any(), prefill=_Any(), BlockSparseAttentionWrapper=_Any,
               single_prefill_with_kv_cache=_Any(), merge_state=_Any())
    _stub("flashinfer.sparse", VariableBlockSparseAttentionWrapper=_Any)
    _stub("cuvs")
    _stub("cuvs.cluster")
    _stub("cuvs.cluster.kmeans", KMeansParams=_Any, fit=_Any())
    _stub("flash_attn", flash_attn_varlen_func=_Any())
    _stub("flash_attn.flash_attn_interface", flash_attn_varlen_func=_Any())
    _stub("loguru", logger=_Any())
    _stub("termcolor", colored=lambda s, *a, **k: s)
    _stub("diffusers")
    _stub("diffusers.models")
    _stub("diffusers.models.attention", Attention=_Any)
    _stub("diffusers.models.attention_processor", Attention=_Any)
    _stub("diffusers.models.embeddings", apply_rotary_emb=_Any())
    _stub("diffusers.pipelines")
    _stub("diffusers.pipelines.hunyuan_video")
    _stub("diffusers.pipelines.hunyuan_video.pipeline_hunyuan_video", DEFAULT_PROMPT_TEMPLATE={"template": "{}"})
    return fi


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()


def bits(m: torch.Tensor) -> np.ndarray:
    return np.packbits(m.to(torch.bool).numpy().reshape(-1))


def main():
    install_stubs()
    sys.path.insert(0, REF)
    torch.Tensor.cuda = lambda self, *a, **k: self  # get_attention_mask hard-codes .cuda()
    from torch.nn.attention.flex_attention import create_block_mask, flex_attention

    import svg.kmeans_utils as KU
    import svg.models.cog.placement as cog_pl
    import svg.models.cog.utils as cog_u
    import svg.models.hyvideo.attention as hy_attn
    import svg.models.hyvideo.placement as hy_pl
    import svg.models.hyvideo.utils as hy_u
    import svg.models.wan.utils as wan_u

    g = {}

    # ---------------- 1. sparsity_to_width at the production geometries ----------------
    g["width_hy_025"] = np.float64(hy_u.sparsity_to_width(0.25, 256, 33, 3600))
    g["width_wan_030"] = np.float64(wan_u.sparsity_to_width(0.30, 0, 21, 3600))
    g["width_cog_025"] = np.float64(cog_u.sparsity_to_width(0.25, 226, 13, 1350))

    # ---------------- 2. mask_mod masks + flex_attention on CPU (small geometry) ----------------
    F_, P_, ctx, L = 4, 140, 16, 9
    S = F_ * P_ + ctx
    mul = 1.9  # -> floor(266/128)*128 = 256 (hy/cog), ceil -> 384 (wan)
    qi = torch.arange(S)[:, None]
    ki = torch.arange(S)[None, :]
    mm_hy = hy_u.generate_temporal_head_mask_mod(ctx, L, F_, P_, mul=mul)
    mm_cog = cog_u.generate_temporal_head_mask_mod(ctx, F_, P_, mul=mul)
    mm_cog_sink = cog_u.generate_temporal_head_mask_mod(ctx, F_, P_, mul=mul, attn_sink=True)
    Sw = F_ * P_
    mm_wan = wan_u.generate_temporal_head_mask_mod(0, 0, F_, P_, mul=mul)
    g["mask_geom"] = np.array([F_, P_, ctx, L, S, Sw], dtype=np.int64)
    g["mask_mul"] = np.float64(mul)
    g["mask_hy"] = bits(mm_hy(0, 0, qi, ki))
    g["mask_cog"] = bits(mm_cog(0, 0, qi, ki))
    g["mask_cog_sink"] = bits(mm_cog_sink(0, 0, qi, ki))
    g["mask_wan"] = bits(mm_wan(0, 0, torch.arange(Sw)[:, None], torch.arange(Sw)[None, :]))

    torch.manual_seed(0)
    H, D = 2, 64
    q = torch.randn(1, H, S, D)
    k = torch.randn(1, H, S, D)
    v = torch.randn(1, H, S, D)
    bm = create_block_mask(mm_hy, None, None, S, S, device="cpu")
    g["flex_hy_out"] = flex_attention(q, k, v, block_mask=bm).to(torch.float16).numpy()
    bmw = create_block_mask(mm_wan, None, None, Sw, Sw, device="cpu")
    g["flex_wan_out"] = flex_attention(q[:, :, :Sw], k[:, :, :Sw], v[:, :, :Sw], block_mask=bmw).to(torch.float16).numpy()
    bmc = create_block_mask(mm_cog, None, None, S, S, device="cpu")
    g["flex_cog_out"] = flex_attention(q, k, v, block_mask=bmc).to(torch.float16).numpy()

    # ---------------- 3. profiling masks + sample_mse (Hunyuan processor method) ----------------
    sp = hy_u.get_attention_mask("spatial", S, ctx, F_, P_)
    tp = hy_u.get_attention_mask("temporal", S, ctx, F_, P_, device="cpu")
    g["prof_hy_spatial"] = bits(sp != 0)
    g["prof_hy_temporal"] = bits(tp != 0)
    spw = wan_u.get_attention_mask("spatial", Sw, 0, F_, P_)
    tpw = wan_u.get_attention_mask("temporal", Sw, 0, F_, P_)
    g["prof_wan_spatial"] = bits(spw != 0)
    g["prof_wan_temporal"] = bits(tpw != 0)
    spc = cog_u.get_attention_mask("spatial", ctx, F_, P_)
    tpc = cog_u.get_attention_mask("temporal", ctx, F_, P_)
    g["prof_cog_spatial"] = bits(spc != 0)
    g["prof_cog_temporal"] = bits(tpc != 0)

    proc_cls = hy_attn.Hunyuan_SVGAttn_Processor2_0
    proc_cls.num_sampled_rows = 16
    proc_cls.sample_mse_max_row = S - ctx  # rows are drawn with CPU RNG: torch.randint(0, max_row, (n,))
    proc_cls.attention_masks = [sp, tp]
    proc = proc_cls(0)
    qb, kb, vb = (t.to(torch.bfloat16) for t in (q, k, v))
    # structured data so that the two masks give clearly different errors
    torch.manual_seed(123)
    mses = proc.sample_mse(qb, kb, vb)
    torch.manual_seed(123)
    rows = torch.randint(low=0, high=S - ctx, size=(16,))
    g["mse_rows"] = rows.numpy()
    g["mse_hy_bf16"] = mses.float().numpy()

    # ---------------- 4. placement (bit-exact copies -> hashes) ----------------
    torch.manual_seed(1)
    cfg, Hh, Dp = 2, 3, 64
    xq = torch.randn(cfg, Hh, S, Dp).to(torch.bfloat16)
    xk = torch.randn(cfg, Hh, S, Dp).to(torch.bfloat16)
    xv = torch.randn(cfg, Hh, S, Dp).to(torch.bfloat16)
    best = torch.tensor([[0, 1, 1], [1, 0, 1]])
    qo, ko, vo = hy_pl.ref_hunyuan_sparse_head_placement(xq, xk, xv, best, ctx, F_, P_)
    g["place_best"] = best.numpy()
    g["place_hy_fwd_sha"] = np.array([sha(qo), sha(ko), sha(vo)])
    hs_out = torch.zeros_like(xq)
    hy_pl.ref_hunyuan_hidden_states_placement(xq, hs_out, best, ctx, F_, P_)
    g["place_hy_inv_sha"] = np.array([sha(hs_out)])
    qo, ko, vo = cog_pl.ref_sparse_head_placement(xq, xk, xv, best, ctx, F_, P_)
    g["place_cog_fwd_sha"] = np.array([sha(qo), sha(ko), sha(vo)])
    hs_out = torch.zeros_like(xq)
    cog_pl.ref_hidden_states_placement(xq, hs_out, best, ctx, F_, P_)
    g["place_cog_inv_sha"] = np.array([sha(hs_out)])

    # ---------------- 5. permutation (canonicalised to the stable order) ----------------
    torch.manual_seed(2)
    Bp, Hp, Sp, Dq = 1, 3, 500, 32
    t = torch.randn(Bp, Hp, Sp, Dq).to(torch.bfloat16)
    labels = torch.randint(0, 17, (Bp * Hp, Sp))
    perm, sidx = KU.permute_tensor_by_labels(t, labels.view(Bp, Hp, Sp), dim=2)
    # canonical within-cluster order: sort the reference's indices inside every cluster segment
    sidx = sidx.view(Bp * Hp, Sp)
    canon = torch.empty_like(sidx)
    for b in range(Bp * Hp):
        lab_sorted = labels[b][sidx[b]]
        key = lab_sorted * Sp + sidx[b]
        canon[b] = sidx[b][torch.argsort(key)]
    g["perm_labels"] = labels.numpy()
    g["perm_canon_idx"] = canon.numpy().astype(np.int32)
    back = KU.apply_inverse_permutation(perm, sidx.view(Bp, Hp, Sp), dim=2)
    g["perm_roundtrip_ok"] = np.array([bool(torch.equal(back, t))])

    # ---------------- 6. weighted_softmax / identify_dynamic_map / density / dynamic_block_sparse_fwd_torch -------------
    torch.manual_seed(3)
    B2, H2, QC, KC, D2 = 1, 2, 12, 40, 64
    qc = torch.randn(B2, H2, QC, D2)
    kc = torch.randn(B2, H2, KC, D2)
    ksz = torch.randint(1, 30, (B2, H2, KC), dtype=torch.int32)
    qsz = torch.randint(1, 30, (B2, H2, QC), dtype=torch.int32)
    scores = torch.matmul(qc, kc.transpose(-2, -1)) / (D2 ** 0.5)
    g["ws_out"] = KU.weighted_softmax(scores, ksz.unsqueeze(-2).float()).numpy()
    for p_, r_ in ((0.9, 0.1), (0.5, 0.0)):
        m = KU.identify_dynamic_map(qc, kc, qsz, ksz, p_, r_)
        g[f"dynmap_fp32_p{int(p_ * 100)}"] = bits(m)
    mb = KU.identify_dynamic_map(qc.bfloat16(), kc.bfloat16(), qsz, ksz, 0.9, 0.1)
    g["dynmap_bf16_p90"] = bits(mb)
    g["dyn_inputs_qc"] = qc.numpy()
    g["dyn_inputs_kc"] = kc.numpy()
    g["dyn_inputs_ksz"] = ksz.numpy()
    g["dyn_inputs_qsz"] = qsz.numpy()
    g["density"] = KU.density_calculation(m, qsz, ksz).numpy()

    torch.manual_seed(4)
    Sq = int(qsz[0, 0].sum())
    # sizes must sum to the same S for q and k blocks of a head: rebuild partitions of a common S
    Sv = 384

    def partition(n, parts, gen):
        cuts = torch.sort(torch.randperm(n - 1, generator=gen)[: parts - 1] + 1).values
        return torch.diff(torch.cat([torch.tensor([0]), cuts, torch.tensor([n])])).to(torch.int32)

    gen = torch.Generator().manual_seed(5)
    qsz2 = torch.stack([partition(Sv, 7, gen) for _ in range(2)])[None]
    ksz2 = torch.stack([partition(Sv, 19, gen) for _ in range(2)])[None]
    qsz2[0, 1, 3] += qsz2[0, 1, 4]
    qsz2[0, 1, 4] = 0  # an empty query block
    ksz2[0, 0, 5] += ksz2[0, 0, 6]
    ksz2[0, 0, 6] = 0  # an empty key block
    dmap = torch.rand(1, 2, 7, 19, generator=gen) > 0.55
    dmap[0, 0, 2, :] = False  # a query block with no active key block -> zeros
    q3 = torch.randn(1, 2, Sv, 64, generator=gen)
    k3 = torch.randn(1, 2, Sv, 64, generator=gen)
    v3 = torch.randn(1, 2, Sv, 64, generator=gen)
    o3 = KU.dynamic_block_sparse_fwd_torch(q3, k3, v3, dmap, qsz2, ksz2)
    g["vb_q"], g["vb_k"], g["vb_v"] = q3.numpy(), k3.numpy(), v3.numpy()
    g["vb_map"] = dmap.numpy()
    g["vb_qsz"], g["vb_ksz"] = qsz2.numpy(), ksz2.numpy()
    g["vb_out"] = o3.numpy()

    # ---------------- 7. dynamic_map_post_processing (Hunyuan) ----------------
    sap = hy_attn.Hunyuan_SAPAttn_Processor2_0(0)
    vid_len, ctxl, pl = 40, 8, 5
    dm = torch.rand(1, 2, 3, 4) > 0.5
    qs = torch.tensor([[[10, 20, 10], [5, 5, 30]]], dtype=torch.int32)
    ks = torch.tensor([[[10, 10, 10, 10], [1, 2, 3, 34]]], dtype=torch.int32)
    qsi = torch.stack([torch.randperm(vid_len) for _ in range(2)]).to(torch.int32)
    dummy = torch.zeros(1, 2, vid_len + ctxl, 4)
    _, _, _, dm2, qs2, ks2, qsi2 = sap.dynamic_map_post_processing(
        dummy[:, :, :vid_len], dummy[:, :, :vid_len], dummy[:, :, :vid_len], dummy.clone(), dummy.clone(), dummy.clone(),
        dm, qs, ks, qsi, vid_len, ctxl, pl, ctxl - pl)
    g["pp_in_map"], g["pp_in_qs"], g["pp_in_ks"], g["pp_in_qsi"] = dm.numpy(), qs.numpy(), ks.numpy(), qsi.numpy()
    g["pp_geom"] = np.array([vid_len, ctxl, pl])
    g["pp_out_map"], g["pp_out_qs"], g["pp_out_ks"] = dm2.numpy(), qs2.numpy(), ks2.numpy()
    g["pp_out_qsi"] = qsi2.numpy()

    np.savez_compressed(OUT / "reference_golden.npz", **g)
    size = (OUT / "reference_golden.npz").stat().st_size
    print(f"wrote {OUT / 'reference_golden.npz'} ({size / 1e3:.0f} kB, {len(g)} arrays)")


if __name__ == "__main__":
    main()
