"""GPU end-to-end tests of the attention processors (the drop-in boundary): stand-in diffusers modules, small
geometries, results compared with the same flow restated with torch + the CPU oracle."""
import math

import pytest
import torch

from oracle import svg_oracle as O
from standins import Attention, Block, Pipe, Transformer

pytestmark = pytest.mark.gpu
DT = torch.bfloat16


def rope_tables(n, d):
    pos = torch.arange(n)[:, None].float()
    inv = 1.0 / (10000 ** (torch.arange(0, d, 2).float() / d))
    ang = (pos * inv[None]).repeat_interleave(2, dim=1)
    return ang.cos(), ang.sin()


def hy_reference(attn, hidden, enc, rope, best, single, geo, prm):
    """Plain-torch statement of Hunyuan_SVGAttn_Processor2_0.__call__ with the oracle attention (fp32)."""
    from svg.models.hyvideo.attention import apply_rotary_emb

    ctx, F_, P_ = geo
    a = attn.float()
    x = torch.cat([hidden, enc], dim=1).float() if single else hidden.float()
    q, k, v = (m(x).unflatten(2, (attn.heads, -1)).transpose(1, 2) for m in (a.to_q, a.to_k, a.to_v))
    q, k = a.norm_q(q), a.norm_k(k)
    n_txt = enc.shape[1]
    if single:
        q = torch.cat([apply_rotary_emb(q[:, :, :-n_txt], rope), q[:, :, -n_txt:]], 2)
        k = torch.cat([apply_rotary_emb(k[:, :, :-n_txt], rope), k[:, :, -n_txt:]], 2)
    else:
        q, k = apply_rotary_emb(q, rope), apply_rotary_emb(k, rope)
        e = enc.float()
        eq, ek, ev = (m(e).unflatten(2, (attn.heads, -1)).transpose(1, 2) for m in (a.add_q_proj, a.add_k_proj, a.add_v_proj))
        eq, ek = a.norm_added_q(eq), a.norm_added_k(ek)
        q, k, v = torch.cat([q, eq], 2), torch.cat([k, ek], 2), torch.cat([v, ev], 2)
    q, k, v = (t.to(DT) for t in (q, k, v))  # the kernel sees bf16 inputs
    S = q.shape[2]
    qp, kp, vp = (O.head_placement(t, best, ctx, F_, P_) for t in (q, k, v))
    o = O.head_placement(O.masked_attention(qp, kp, vp, O.band_mask(S, *prm)), best, ctx, F_, P_, inverse=True)
    o = o.transpose(1, 2).flatten(2, 3)
    h, e2 = o[:, :-n_txt], o[:, -n_txt:]
    h = a.to_out[0](h)
    if not single:
        e2 = a.to_add_out(e2)
    attn.to(DT)
    return h, e2


def test_hunyuan_svg_processor_end_to_end():
    from svg.models import _core
    from svg.models.hyvideo.inference import replace_hyvideo_attention

    torch.manual_seed(0)
    heads, hd = 4, 128
    dim = heads * hd
    blocks = [Block(Attention(dim, heads, added_kv=True, dtype=DT), "attn"), Block(Attention(dim, heads, dtype=DT), "attn")]
    tr = Transformer(blocks[:1], "transformer_blocks")
    tr.single_transformer_blocks = torch.nn.ModuleList(blocks[1:])
    pipe = Pipe(tr)
    tr.cuda()
    L = 21
    cls = replace_hyvideo_attention(pipe, 160, 320, 17, L, first_layers_fp=0, first_times_fp=900.0, pattern="SVG",
                                    num_sampled_rows=32, sparsity=0.45)
    ctx, F_, P_ = cls.context_length, cls.num_frame, cls.frame_size
    assert (ctx, F_, P_) == (256, 5, 200)
    cls.sample_mse_max_row = F_ * P_
    V = F_ * P_
    hidden = (torch.randn(1, V, dim) * 0.3).to(DT).cuda()
    enc = (torch.randn(1, ctx, dim) * 0.3).to(DT).cuda()
    amask = torch.zeros(1, V + ctx, dtype=torch.bool)
    amask[:, : V + L] = True
    rope = rope_tables(V, hd)
    prm = cls.block_mask.as_tuple()
    for blk, single in ((blocks[0], False), (blocks[1], True)):
        with torch.no_grad():
            h, e = blk.attn(hidden, encoder_hidden_states=enc, attention_mask=amask.cuda(), image_rotary_emb=rope,
                            timestep=torch.tensor([100.0]))
            best = blk.attn.processor.last_best_mask_idx.cpu()
            blk.attn.cpu()
            rh, re = hy_reference(blk.attn, hidden.cpu(), enc.cpu(), rope, best, single, (ctx, F_, P_), prm)
            blk.attn.cuda()
        torch.testing.assert_close(h.float().cpu(), rh, atol=3e-2, rtol=3e-2)
        torch.testing.assert_close(e.float().cpu(), re, atol=3e-2, rtol=3e-2)
        # dense warm-up branch (two segments through the attention_mask) against torch
        with torch.no_grad():
            hd_, _ = blk.attn(hidden, encoder_hidden_states=enc, attention_mask=amask.cuda(), image_rotary_emb=rope,
                              timestep=torch.tensor([950.0]))
        assert torch.isfinite(hd_.float()).all()
    # fused and materialised placement give bit-identical processor outputs
    cls.fused_placement = False
    torch.manual_seed(5)
    with torch.no_grad():
        h2, _ = blocks[1].attn(hidden, encoder_hidden_states=enc, attention_mask=amask.cuda(), image_rotary_emb=rope,
                               timestep=torch.tensor([100.0]))
    cls.fused_placement = True
    torch.manual_seed(5)
    with torch.no_grad():
        h3, _ = blocks[1].attn(hidden, encoder_hidden_states=enc, attention_mask=amask.cuda(), image_rotary_emb=rope,
                               timestep=torch.tensor([100.0]))
    assert torch.equal(h2, h3)
    # device-side dense / sparse switch (GPU timestep tensor): the same result as the host-side decision on both kinds of step
    for t in (100.0, 950.0):
        outs = []
        for dev_switch, ts in ((False, torch.tensor([t])), (True, torch.tensor([t]).cuda())):
            cls.device_switch = dev_switch
            torch.manual_seed(11)
            _core.reseed_switch_generator(11)   # the switched path draws its rows from its own generator
            state = torch.get_rng_state()
            with torch.no_grad():
                hh, ee = blocks[1].attn(hidden, encoder_hidden_states=enc, attention_mask=amask.cuda(), image_rotary_emb=rope,
                                        timestep=ts)
            outs.append((hh, ee))
            if dev_switch:   # ... and leaves the global CPU stream alone, on dense and on sparse steps
                assert torch.equal(state, torch.get_rng_state())
                best = blocks[1].attn.processor.last_best_mask_idx
                assert bool((best == -1).all()) == (t > 900.0)
        cls.device_switch = True
        torch.testing.assert_close(outs[0][0].float(), outs[1][0].float(), atol=1e-2, rtol=1e-2)
        torch.testing.assert_close(outs[0][1].float(), outs[1][1].float(), atol=1e-2, rtol=1e-2)


def test_hunyuan_svg_processor_prescaled_q_equals_plain_path():
    """prescale_q = True (opt-in since round 4, flex_attention's PRESCALE_QK trade-off: the fused prologue folds sm_scale * log2(e) into
    its last rounding of q and the kernels run their pre-scaled forms) against the default (plain q, scale applied inside the kernel): the same processor
    output up to the rounding of q — sparse step, dense warm-up step and the device-switched path, double- and single-stream block."""
    from svg.models import _core
    from svg.models.hyvideo.inference import replace_hyvideo_attention

    torch.manual_seed(1)
    heads, hd = 4, 128
    dim = heads * hd
    blocks = [Block(Attention(dim, heads, added_kv=True, dtype=DT), "attn"), Block(Attention(dim, heads, dtype=DT), "attn")]
    tr = Transformer(blocks[:1], "transformer_blocks")
    tr.single_transformer_blocks = torch.nn.ModuleList(blocks[1:])
    pipe = Pipe(tr)
    tr.cuda()
    L = 21
    cls = replace_hyvideo_attention(pipe, 160, 320, 17, L, first_layers_fp=0, first_times_fp=900.0, pattern="SVG",
                                    num_sampled_rows=32, sparsity=0.45)
    assert not cls.prescale_q      # the default is the reference's formulation
    ctx, F_, P_ = cls.context_length, cls.num_frame, cls.frame_size
    cls.sample_mse_max_row = F_ * P_
    V = F_ * P_
    hidden = (torch.randn(1, V, dim) * 0.3).to(DT).cuda()
    enc = (torch.randn(1, ctx, dim) * 0.3).to(DT).cuda()
    amask = torch.zeros(1, V + ctx, dtype=torch.bool)
    amask[:, : V + L] = True
    rope = rope_tables(V, hd)
    try:
        for blk in blocks:
            for ts in (torch.tensor([100.0]), torch.tensor([950.0]), torch.tensor([100.0]).cuda(), torch.tensor([950.0]).cuda()):
                outs = []
                for pre in (True, False):
                    cls.prescale_q = pre
                    torch.manual_seed(7)
                    _core.reseed_switch_generator(7)
                    with torch.no_grad():
                        outs.append(blk.attn(hidden, encoder_hidden_states=enc, attention_mask=amask.cuda(), image_rotary_emb=rope,
                                             timestep=ts))
                for a, b in zip(*outs):
                    torch.testing.assert_close(a.float(), b.float(), atol=2e-2, rtol=2e-2)
                    e = ((a.float() - b.float()).norm() / b.float().norm()).item()
                    assert e < 6e-3, e
    finally:
        cls.prescale_q = False


def _clustered(H, N, D, modes, gen):
    centers = torch.randn(H, modes, D, generator=gen) * 2.0
    lab = torch.randint(0, modes, (H, N), generator=gen)
    return torch.gather(centers, 1, lab[..., None].expand(-1, -1, D)) + 0.4 * torch.randn(H, N, D, generator=gen)


def test_kmeans_clustering_two_streams_equals_one_stream():
    """`_core.kmeans_clustering` runs the q-side Lloyd loop on a side stream beside the k side (KMEANS_TWO_STREAMS, the default).
    Several layers and steps — the init call of every layer (random points drawn from the same torch seed), then two warm-started calls on
    drifting data, with allocator churn in between (tensors freed on the main stream while the side stream's outputs are still in use) —
    give bit-identical labels, centroids, sizes, iteration counts and sorted indices with the switch on and off.
    Memory: the q-side outputs come from the side stream's allocator pool and are record_stream'ed to the main stream, so their blocks
    return to the pool only after the main stream's pending work — at Wan 2.1 720p (40 heads) that is 12 MB of labels + 12 MB of sorted
    indices + 3 MB of centroids per layer-call held a little longer, not a growing footprint (asserted below on the allocator's counters)."""
    from svg.models import _core

    H, N, D, QC, KC = 4, 6000, 128, 24, 40
    gen = torch.Generator().manual_seed(5)
    base_q, base_k = _clustered(H, N, D, 16, gen), _clustered(H, N, D, 16, gen)

    def run(two_streams):
        old = _core.KMEANS_TWO_STREAMS
        _core.KMEANS_TWO_STREAMS = two_streams
        try:
            store = _core.CentroidStore()
            outs = []
            g2 = torch.Generator().manual_seed(9)
            for step in range(3):
                drift = 0.05 * step
                for layer in range(3):
                    q = (base_q + drift * torch.randn(H, N, D, generator=g2) + 0.01 * layer)[None].to(DT).cuda()
                    k = (base_k + drift * torch.randn(H, N, D, generator=g2) - 0.01 * layer)[None].to(DT).cuda()
                    torch.manual_seed(100 + layer)                     # the init call's random points (svg/kmeans_utils.py:706-709)
                    junk = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")   # allocator churn on the main stream
                    (ql, qc, qs, qit, qidx), (kl, kc, ks, kit, kidx) = _core.kmeans_clustering(store, layer, q, k, QC, KC, 5, 2)
                    del junk
                    outs.append([t.clone() if isinstance(t, torch.Tensor) else torch.tensor(t) for t in (ql, qc, qs, qit, qidx, kl, kc, ks, kit, kidx)])
            torch.cuda.synchronize()
            return outs
        finally:
            _core.KMEANS_TWO_STREAMS = old

    a = run(True)
    mem_after_first = torch.cuda.memory_reserved()
    b = run(False)
    a2 = run(True)
    assert len(a) == len(b) == 9
    for call, (x, y, z) in enumerate(zip(a, b, a2)):
        for i, (t1, t2, t3) in enumerate(zip(x, y, z)):
            assert torch.equal(t1.cpu(), t2.cpu()), (call, i)
            assert torch.equal(t1.cpu(), t3.cpu()), (call, i)
    assert torch.cuda.memory_reserved() <= mem_after_first + (256 << 20), "repeated two-stream calls must not grow the reserved memory"


@pytest.mark.parametrize("model", ["hy", "wan"])
def test_svg2_core_against_oracle_and_dense(model):
    """SVG2 sparse branch: (a) top_p = 1 keeps every block -> must equal dense attention (permutation invariance);
    (b) top_p < 1: equals oracle attention under the element mask built from the kernels' labels and block map, which are themselves
    checked against the oracle's k-means loop and exact-mode block map."""
    from svg import _native as nat
    from svg.kmeans_utils import batch_kmeans_Euclid, identify_dynamic_map
    from svg.models import _core

    gen = torch.Generator().manual_seed(3)
    H, D, F_, P_ = 3, 128, 6, 150
    ctx, L = (256, 40) if model == "hy" else (0, 0)
    V = F_ * P_
    S = V + ctx
    q = _clustered(H, S, D, 12, gen)[None].to(DT).cuda()
    k = _clustered(H, S, D, 20, gen)[None].to(DT).cuda()
    v = torch.randn(1, H, S, D, generator=gen).to(DT).cuda()
    geo = _core.Geometry(ctx, F_, P_)
    QC, KC = 10, 24
    # (a) everything kept
    store = _core.CentroidStore()
    torch.manual_seed(11)
    o_all = _core.svg2_sparse_attention(q, k, v, geo, store, 0, QC, KC, top_p=1.0, min_kc_ratio=1.0, iter_init=3, iter_step=1,
                                        prompt_length=L)
    dense_prm = O.dense_band_params(S, V + L) if ctx else O.dense_band_params(S)
    ref_dense = O.masked_attention(q.cpu(), k.cpu(), v.cpu(), O.band_mask(S, **dense_prm))
    torch.testing.assert_close(o_all.float().cpu(), ref_dense, atol=1e-2, rtol=1e-2)
    # (b) warm-started step with top-p selection, re-derived from the kernels' labels / map
    qc0, kc0 = store.q[0].clone(), store.k[0].clone()
    o = _core.svg2_sparse_attention(q, k, v, geo, store, 0, QC, KC, top_p=0.6, min_kc_ratio=0.1, iter_init=3, iter_step=2,
                                    prompt_length=L)
    ql, qc, qs, _ = batch_kmeans_Euclid(q[0, :, :V].contiguous(), QC, max_iters=2, init_centroids=qc0)
    kl, kc, ks, _ = batch_kmeans_Euclid(k[0, :, :V].contiguous(), KC, max_iters=2, init_centroids=kc0)
    dmap = identify_dynamic_map(qc[None], kc[None], qs[None], ks[None], 0.6, 0.1)[0].cpu()
    assert 0.05 < dmap.float().mean() < 0.95
    ql, kl = ql.cpu(), kl.cpu()
    # the clustering half against the ORACLE (not only against the kernels themselves): labels of the 2-iteration warm-started
    # k-means equal the oracle's loop up to rounding-level near-ties, cluster sizes follow the labels, and the block map equals
    # the oracle's exact mode bit for bit on the kernels' own centroids (tests/test_gpu_kernels.py holds the per-kernel forms)
    for x, init, lab, cent, sizes, K in ((q, qc0, ql, qc, qs, QC), (k, kc0, kl, kc, ks, KC)):
        xv = x[0, :, :V].cpu()
        rl, rc, rcnt, rit = O.batch_kmeans_euclid(xv, K, max_iters=2, init_centroids=init.cpu())
        assert rit == 2 and (lab != rl).float().mean() < 2e-2
        assert torch.equal(sizes.cpu(), torch.stack([torch.bincount(lab[h], minlength=K) for h in range(H)]).to(torch.int32))
        assert (cent.float().cpu() - rc.float()).abs().mean() < 5e-3
    ref_map = O.identify_dynamic_map(qc[None].cpu(), kc[None].cpu(), qs[None].cpu(), ks[None].cpu(), 0.6, 0.1, exact=True)[0]
    assert torch.equal(dmap.bool(), ref_map)
    for h in range(H):
        em = torch.zeros(S, S, dtype=torch.bool)
        em[:V, :V] = dmap[h][ql[h]][:, kl[h]]
        if ctx:
            em[:V, V:V + L] = True       # everything real sees the prompt
            em[V:V + L, : V + L] = True  # the prompt sees everything real
            em[V + L:, V + L:] = True    # unused prompt tokens see only themselves
        ref = O.masked_attention(q[0, h].cpu(), k[0, h].cpu(), v[0, h].cpu(), em)
        torch.testing.assert_close(o[0, h].float().cpu(), ref, atol=1e-2, rtol=1e-2)


def test_wan_and_cog_svg_processors_run_and_match():
    from svg.models.cog.attention import CogVideoX_SparseAttn_Processor2_0 as CogP
    from svg.models.cog.utils import generate_temporal_head_mask_mod as cog_mm
    from svg.models.wan.attention import WanAttn_SVGAttn_Processor2_0 as WanP
    from svg.models.wan.utils import generate_temporal_head_mask_mod as wan_mm

    torch.manual_seed(1)
    heads, hd = 2, 64
    dim = heads * hd
    # ---- Cog: text first, LayerNorm qk-norm, returns (hidden, encoder)
    ctx, F_, P_ = 26, 4, 180
    CogP.context_length, CogP.num_frame, CogP.frame_size = ctx, F_, P_
    CogP.first_layers_fp, CogP.first_times_fp, CogP.num_sampled_rows = 0.0, 0.2, 16
    CogP.block_mask = cog_mm(ctx, F_, P_, mul=1.5)
    attn = Attention(dim, heads, qk_norm="layer", dtype=DT).cuda()
    attn.set_processor(CogP(0))
    hidden = (torch.randn(2, F_ * P_, dim) * 0.3).to(DT).cuda()
    enc = (torch.randn(2, ctx, dim) * 0.3).to(DT).cuda()
    with torch.no_grad():
        h, e = attn(hidden, encoder_hidden_states=enc, image_rotary_emb=None, timestep=torch.tensor([100.0]))
        best = attn.processor.last_best_mask_idx.cpu()
        best = torch.where(torch.isnan(best.float()), torch.zeros_like(best), best)
        a = attn.cpu().float()
        x = torch.cat([enc, hidden], 1).float().cpu()
        q, k, v = (m(x).view(2, -1, heads, hd).transpose(1, 2) for m in (a.to_q, a.to_k, a.to_v))
        q, k = a.norm_q(q), a.norm_k(k)
        q, k, v = (t.to(DT) for t in (q, k, v))
        S = ctx + F_ * P_
        qp, kp, vp = (O.head_placement(t, best, ctx, F_, P_, text_first=True) for t in (q, k, v))
        o = O.head_placement(O.masked_attention(qp, kp, vp, O.band_mask(S, *CogP.block_mask.as_tuple())), best, ctx, F_, P_,
                             text_first=True, inverse=True)
        ref = a.to_out[0](o.transpose(1, 2).reshape(2, -1, dim))
    torch.testing.assert_close(h.float().cpu(), ref[:, ctx:], atol=3e-2, rtol=3e-2)
    torch.testing.assert_close(e.float().cpu(), ref[:, :ctx], atol=3e-2, rtol=3e-2)
    # ---- Wan: no text, qk-norm across heads, complex rope as (real, imag)
    F_, P_ = 5, 160
    WanP.context_length, WanP.num_frame, WanP.frame_size = 0, F_, P_
    WanP.first_layers_fp, WanP.first_times_fp, WanP.num_sampled_rows, WanP.sample_mse_max_row = 0, 900.0, 16, 400
    WanP.block_mask = wan_mm(0, 0, F_, P_, mul=1.2)
    attn = Attention(dim, heads, qk_norm="rms", across_heads=True, dtype=DT).cuda()
    attn.set_processor(WanP(0))
    S = F_ * P_
    hidden = (torch.randn(1, S, dim) * 0.3).to(DT).cuda()
    ang = torch.rand(S, hd // 2) * 6.28
    rope = (ang.cos().cuda(), ang.sin().cuda())
    with torch.no_grad():
        out = attn(hidden, rotary_emb=rope, timestep=torch.tensor([100.0]))
        best = attn.processor.last_best_mask_idx.cpu()
        a = attn.cpu().float()
        x = hidden.float().cpu()
        q, k, v = a.to_q(x), a.to_k(x), a.to_v(x)
        q, k = a.norm_q(q), a.norm_k(k)
        q, k, v = (t.unflatten(2, (heads, -1)).transpose(1, 2) for t in (q, k, v))
        fr = torch.complex(ang.cos().double(), ang.sin().double())[None, None]
        rot = lambda t: torch.view_as_real(torch.view_as_complex(t.double().unflatten(3, (-1, 2))) * fr).flatten(3, 4).float()
        q, k = rot(q), rot(k)
        q, k, v = (t.to(DT) for t in (q, k, v))
        qp, kp, vp = (O.head_placement(t, best, 0, F_, P_) for t in (q, k, v))
        o = O.head_placement(O.masked_attention(qp, kp, vp, O.band_mask(S, *WanP.block_mask.as_tuple())), best, 0, F_, P_,
                             inverse=True)
        ref = a.to_out[0](o.transpose(1, 2).flatten(2, 3))
    torch.testing.assert_close(out.float().cpu(), ref, atol=3e-2, rtol=3e-2)


@pytest.mark.parametrize("hd,complex_table", [(128, False), (64, True)])
def test_wan_fused_prologue_equals_the_three_steps(hd, complex_table):
    """WanAttn_SVGAttn_Processor2_0: `fused_prologue` (qk_norm + transpose + rotary_emb in one kernel, svg_rmsnorm_rope_transpose) on and off give
    the same processor output bit for bit — dense and sparse steps, rotary table as the (real, imag) pair of the patched model forward or as
    diffusers' complex tensor — and the cross attention (head views instead of contiguous copies) equals the copying path."""
    from svg.models.wan.attention import WanAttn_SVGAttn_Processor2_0 as WanP
    from svg.models.wan.utils import generate_temporal_head_mask_mod as wan_mm

    torch.manual_seed(8)
    heads, F_, P_ = 3, 5, 160
    dim, S = heads * hd, F_ * P_
    WanP.context_length, WanP.num_frame, WanP.frame_size = 0, F_, P_
    WanP.first_layers_fp, WanP.first_times_fp, WanP.num_sampled_rows, WanP.sample_mse_max_row = 0, 900.0, 16, 400
    WanP.block_mask = wan_mm(0, 0, F_, P_, mul=1.2)
    attn = Attention(dim, heads, qk_norm="rms", across_heads=True, dtype=DT).cuda()
    with torch.no_grad():
        attn.norm_q.weight.mul_(1.3).add_(0.1 * torch.randn_like(attn.norm_q.weight))
        attn.norm_k.weight.mul_(0.8).add_(0.1 * torch.randn_like(attn.norm_k.weight))
    if complex_table:      # norm weights kept in fp32 beside a 16-bit model: both paths use them in fp32
        attn.norm_q.float(), attn.norm_k.float()
    attn.set_processor(WanP(0))
    hidden = (torch.randn(1, S, dim) * 0.3).to(DT).cuda()
    enc = (torch.randn(1, 37, dim) * 0.3).to(DT).cuda()
    ang = torch.rand(S, hd // 2) * 6.28
    rope = torch.complex(ang.cos(), ang.sin())[None, None].cuda() if complex_table else (ang.cos().cuda(), ang.sin().cuda())
    outs = {}
    try:
        for fused in (True, False):
            WanP.fused_prologue = fused
            with torch.no_grad():
                torch.manual_seed(3)      # (the sampled rows of the online profiler)
                sparse = attn(hidden, rotary_emb=rope, timestep=torch.tensor([100.0]))
                dense = attn(hidden, rotary_emb=rope, timestep=torch.tensor([950.0]))
                cross = attn(hidden, encoder_hidden_states=enc)
            outs[fused] = (sparse.clone(), dense.clone(), cross.clone())
    finally:
        WanP.fused_prologue = True
    for a, b in zip(outs[True], outs[False]):
        assert torch.isfinite(a.float()).all() and torch.equal(a, b)


@pytest.mark.parametrize("hd", [64, 128])
def test_wan_and_cog_prescaled_q_equals_plain_path(hd):
    """Wan / Cog SVG1 processors: prescale_q = True (opt-in — the HIP RoPE pass folds the softmax scale into its rounding of q, pre-scaled
    kernels downstream, at head_dim 64 and 128) against the default prescale_q = False: the same processor output up to the rounding of q, on
    sparse and dense steps, host- and device-switched.  Cross attention (Wan) never pre-scales: its q feeds torch SDPA."""
    from svg.models import _core
    from svg.models.cog.attention import CogVideoX_SparseAttn_Processor2_0 as CogP
    from svg.models.cog.utils import generate_temporal_head_mask_mod as cog_mm
    from svg.models.wan.attention import WanAttn_SVGAttn_Processor2_0 as WanP
    from svg.models.wan.utils import generate_temporal_head_mask_mod as wan_mm

    torch.manual_seed(4)
    heads = 2
    dim = heads * hd

    def both(cls, run):
        outs = []
        try:
            for pre in (True, False):
                cls.prescale_q = pre
                torch.manual_seed(9)
                _core.reseed_switch_generator(9)
                with torch.no_grad():
                    outs.append(run())
        finally:
            cls.prescale_q = False
        for a, b in zip(*outs):
            torch.testing.assert_close(a.float(), b.float(), atol=2e-2, rtol=2e-2)
            assert ((a.float() - b.float()).norm() / b.float().norm()).item() < 6e-3
        return outs[0]

    # ---- Wan (self attention with complex RoPE; then a cross attention call, which must not touch q)
    F_, P_ = 5, 160
    S = F_ * P_
    WanP.context_length, WanP.num_frame, WanP.frame_size = 0, F_, P_
    WanP.first_layers_fp, WanP.first_times_fp, WanP.num_sampled_rows, WanP.sample_mse_max_row = 0, 900.0, 16, 400
    WanP.block_mask = wan_mm(0, 0, F_, P_, mul=1.2)
    attn = Attention(dim, heads, qk_norm="rms", across_heads=True, dtype=DT).cuda()
    attn.set_processor(WanP(0))
    hidden = (torch.randn(1, S, dim) * 0.3).to(DT).cuda()
    ang = torch.rand(S, hd // 2) * 6.28
    rope = (ang.cos().cuda(), ang.sin().cuda())
    for ts in (torch.tensor([100.0]), torch.tensor([950.0]), torch.tensor([100.0]).cuda(), torch.tensor([950.0]).cuda()):
        both(WanP, lambda: (attn(hidden, rotary_emb=rope, timestep=ts),))
    enc = (torch.randn(1, 33, dim) * 0.3).to(DT).cuda()
    xo = both(WanP, lambda: (attn(hidden, encoder_hidden_states=enc),))
    assert torch.isfinite(xo[0].float()).all()
    # ---- Cog (text first, LayerNorm qk-norm, cos / sin RoPE on the video tokens only)
    ctx, F_, P_ = 26, 4, 180
    CogP.context_length, CogP.num_frame, CogP.frame_size = ctx, F_, P_
    CogP.first_layers_fp, CogP.first_times_fp, CogP.num_sampled_rows = 0.0, 0.2, 16
    CogP.block_mask = cog_mm(ctx, F_, P_, mul=1.5)
    cattn = Attention(dim, heads, qk_norm="layer", dtype=DT).cuda()
    cattn.set_processor(CogP(0))
    ch = (torch.randn(2, F_ * P_, dim) * 0.3).to(DT).cuda()
    ce = (torch.randn(2, ctx, dim) * 0.3).to(DT).cuda()
    crope = tuple(t.cuda() for t in rope_tables(F_ * P_, hd))
    for ts in (torch.tensor([100.0]), torch.tensor([950.0]), torch.tensor([100.0]).cuda()):
        both(CogP, lambda: cattn(ch, encoder_hidden_states=ce, image_rotary_emb=crope, timestep=ts))


def test_wan_i2v_image_cross_attention_branch_and_fp8():
    """Wan 2.1 I2V (BASELINE.json configs[4] names it): blocks carry `add_k_proj` / `add_v_proj` / `norm_added_k`; in the cross
    attention the first 257 encoder tokens are CLIP image tokens that get their own small dense attention whose output is added to
    the text cross attention (ref: svg/models/wan/attention.py:174-188) — checked against a torch restatement; the self attention of
    the same block runs the sparse path with the package's fp8 opt-in (`set_attention_dtype("fp8")`; head_dim 128) and stays within the
    e4m3 tolerance of its 16-bit result."""
    import torch.nn.functional as F
    from standins import RMSNorm

    from svg.models import _core
    from svg.models.wan.attention import WanAttn_SVGAttn_Processor2_0 as WanP
    from svg.models.wan.utils import generate_temporal_head_mask_mod as wan_mm

    torch.manual_seed(2)
    heads, hd = 2, 128
    dim = heads * hd
    F_, P_ = 5, 160
    S = F_ * P_
    WanP.context_length, WanP.num_frame, WanP.frame_size = 0, F_, P_
    WanP.first_layers_fp, WanP.first_times_fp, WanP.num_sampled_rows, WanP.sample_mse_max_row = 0, 900.0, 16, 400
    WanP.block_mask = wan_mm(0, 0, F_, P_, mul=1.2)
    attn = Attention(dim, heads, qk_norm="rms", across_heads=True, dtype=DT)
    attn.add_k_proj, attn.add_v_proj, attn.norm_added_k = torch.nn.Linear(dim, dim), torch.nn.Linear(dim, dim), RMSNorm(dim)
    attn.to(DT).cuda()
    attn.set_processor(WanP(0))
    hidden = (torch.randn(1, S, dim) * 0.3).to(DT).cuda()
    n_txt = 40
    enc = (torch.randn(1, 257 + n_txt, dim) * 0.3).to(DT).cuda()
    with torch.no_grad():
        out = attn(hidden, encoder_hidden_states=enc)          # cross attention: timestep None, no rotary embedding
        a = attn.cpu().float()
        x, e = hidden.float().cpu(), enc.float().cpu()
        e_img, e_txt = e[:, :257], e[:, 257:]
        split = lambda t: t.unflatten(2, (heads, -1)).transpose(1, 2)   # noqa: E731
        q = split(a.norm_q(a.to_q(x)))
        k, v = split(a.norm_k(a.to_k(e_txt))), split(a.to_v(e_txt))
        k_img, v_img = split(a.norm_added_k(a.add_k_proj(e_img))), split(a.add_v_proj(e_img))
        o = F.scaled_dot_product_attention(q, k, v) + F.scaled_dot_product_attention(q, k_img, v_img)
        ref = a.to_out[0](o.transpose(1, 2).flatten(2, 3))
        attn.to(DT).cuda()
    torch.testing.assert_close(out.float().cpu(), ref, atol=3e-2, rtol=3e-2)
    # self attention of the same I2V block: sparse step, 16-bit and fp8 kernels
    ang = torch.rand(S, hd // 2) * 6.28
    rope = (ang.cos().cuda(), ang.sin().cuda())
    outs = {}
    try:
        for name in ("bf16", "fp8"):
            _core.set_attention_dtype(name)
            torch.manual_seed(5)
            with torch.no_grad():
                outs[name] = attn(hidden, rotary_emb=rope, timestep=torch.tensor([100.0])).float()
    finally:
        _core.set_attention_dtype("bf16")
    assert torch.isfinite(outs["fp8"]).all()
    e8 = ((outs["fp8"] - outs["bf16"]).norm() / outs["bf16"].norm()).item()
    assert 1e-4 < e8 < 0.1, e8      # e4m3 QK^T / PV really ran (not bit-equal) and stays inside its tolerance


def test_cosmos_svg_processor_matches_torch():
    """Cosmos (4th model family): per-head RMS qk-norm after the head split, half-split RoPE, no text; sparse core = Wan's."""
    from svg.models.cosmos.attention import Cosmos_SVG_AttnProcessor2_0 as CosP, apply_rotary_emb_half
    from svg.models.cosmos.inference import replace_cosmos_attention
    from svg.models.cosmos.utils import generate_temporal_head_mask_mod as mm

    torch.manual_seed(4)
    heads, hd = 2, 128
    dim = heads * hd
    F_, P_ = 5, 160
    CosP.context_length, CosP.num_frame, CosP.frame_size = 0, F_, P_
    CosP.first_layers_fp, CosP.first_times_fp, CosP.num_sampled_rows, CosP.sample_mse_max_row = 0, 900.0, 16, 400
    CosP.block_mask = mm(0, 0, F_, P_, mul=1.2)
    attn = Attention(dim, heads, qk_norm="rms", dtype=DT).cuda()      # per-head RMSNorm(hd)
    attn.set_processor(CosP(0))
    S = F_ * P_
    hidden = (torch.randn(1, S, dim) * 0.3).to(DT).cuda()
    ang = torch.rand(S, hd // 2) * 6.28
    cos, sin = torch.cat([ang.cos(), ang.cos()], -1), torch.cat([ang.sin(), ang.sin()], -1)
    with torch.no_grad():
        out = attn(hidden, image_rotary_emb=(cos.cuda(), sin.cuda()), timestep=torch.tensor([100.0]))
        best = attn.processor.last_best_mask_idx.cpu()
        a = attn.cpu().float()
        x = hidden.float().cpu()
        q, k, v = (m(x).unflatten(2, (heads, -1)).transpose(1, 2) for m in (a.to_q, a.to_k, a.to_v))
        q, k = a.norm_q(q), a.norm_k(k)
        q, k = apply_rotary_emb_half(q, (cos, sin)), apply_rotary_emb_half(k, (cos, sin))
        q, k, v = (t.to(DT) for t in (q, k, v))
        qp, kp, vp = (O.head_placement(t, best, 0, F_, P_) for t in (q, k, v))
        o = O.head_placement(O.masked_attention(qp, kp, vp, O.band_mask(S, *CosP.block_mask.as_tuple())), best, 0, F_, P_,
                             inverse=True)
        ref = a.to_out[0](o.transpose(1, 2).reshape(1, -1, dim))
    torch.testing.assert_close(out.float().cpu(), ref, atol=3e-2, rtol=3e-2)
    # dense warm-up branch and the cross-attention branch stay finite
    attn.cuda().to(DT)
    with torch.no_grad():
        d = attn(hidden, image_rotary_emb=(cos.cuda(), sin.cuda()), timestep=torch.tensor([950.0]))
        c = attn(hidden, encoder_hidden_states=hidden[:, :77], timestep=None)
    assert torch.isfinite(d.float()).all() and torch.isfinite(c.float()).all()
    assert callable(replace_cosmos_attention)
