"""The HIP kernels against fixtures PRODUCED BY THE REFERENCE'S OWN TRITON KERNELS (tests/golden/triton_golden.npz: the reference's
`@triton.jit` sources executed with Triton's interpreter through the reference's wrappers, tests/golden/make_golden_triton.py).
tests/test_triton_golden.py checks the oracle against the same fixtures on the CPU; here the product path is compared with them
directly, through the C ABI — no oracle in between except where a tolerance has to be derived."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import svg_oracle as O

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden" / "triton_golden.npz"


@pytest.fixture(scope="module")
def nat():
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "sparse-videogen_amd"))
    from svg import _native
    _native.load()
    return _native


@pytest.fixture(scope="module")
def g():
    return np.load(G)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.mark.parametrize("tag,text_first", [("pl_hy", False), ("pl_wan", False), ("pl_cog", True)])
def test_head_placement_equals_the_references_triton_kernels(nat, g, tag, text_first):
    """svg_head_placement == {hunyuan,wan,}_sparse_head_placement_kernel / *_hidden_states_placement_kernel, bit for bit"""
    ctx, F_, P_ = (int(x) for x in g[tag + "_geo"])
    best = T(g[tag + "_best"]).cuda()
    names = ("q", "k", "v") if tag == "pl_hy" else ("q",)
    srcs = [T(g[f"{tag}_{n}"]).cuda() for n in names]
    dsts = [torch.zeros_like(s) for s in srcs]
    nat.head_placement(srcs, dsts, best, ctx, F_, P_, text_first, False)
    for n, d in zip(names, dsts):
        assert torch.equal(d.cpu(), T(g[f"{tag}_{n}o"]))
    back = [torch.zeros_like(s) for s in srcs]
    nat.head_placement(dsts, back, best, ctx, F_, P_, text_first, True)
    for s, b in zip(srcs, back):
        assert torch.equal(s, b)


def test_permutation_equals_the_references_triton_kernels(nat, g):
    """svg_argsort_labels (stable) + svg_permute_rows / svg_inverse_permute_rows == _permute_kernel / _inverse_permute_kernel"""
    x, labels, sidx, xp = T(g["pm_x"]), T(g["pm_labels"]), T(g["pm_sidx"]), T(g["pm_xp"])
    B, H, S, D = x.shape
    idx, counts = nat.argsort_labels(labels.reshape(B * H, S).cuda().contiguous(), 9)
    assert torch.equal(idx.cpu().reshape(B, H, S), sidx)
    assert torch.equal(counts.cpu().long(), torch.stack([torch.bincount(l, minlength=9) for l in labels.reshape(B * H, S).long()]))
    y = nat.permute_rows(x.reshape(B * H, S, D).cuda().contiguous(), idx)
    assert torch.equal(y.cpu().reshape(B, H, S, D), xp)
    assert torch.equal(nat.permute_rows(y, idx, inverse=True).cpu().reshape(B, H, S, D), x)


@pytest.mark.parametrize("sfx", ["h", "f"])
def test_modulate_equals_the_references_triton_kernels(nat, g, sfx):
    """svg_modulate_shift_forward / svg_modulate_gate_residual_forward == _modulate_shift_fwd_fused / _modulate_gate_residual_fwd_fused
    (fp32 arithmetic, one rounding: at most one ulp of the output type apart — contraction of the multiply-add)"""
    t = {n: T(g[f"gl_{sfx}_{n}"]) for n in ("x", "scale", "shift", "gate", "att", "ln_n", "ms", "gr")}
    dt = t["x"].dtype
    ms = nat.modulate_shift_forward(t["ln_n"].cuda(), t["scale"].cuda(), t["shift"].cuda(), dt).cpu()
    gr = nat.modulate_gate_residual_forward(t["x"].cuda(), t["att"].cuda(), t["gate"].cuda(), dt).cpu()
    for mine, ref in ((ms, t["ms"]), (gr, t["gr"])):
        ulp = torch.finfo(dt).eps * ref.float().abs().clamp_min(1e-3)
        assert ((mine.float() - ref.float()).abs() <= ulp).all()
        assert (mine == ref).float().mean().item() > 0.99


def test_variable_block_attention_equals_the_references_triton_kernel(nat, g):
    """svg_varblock_attention == _dynamic_block_sparse_fwd_kernel (fp16, head_dim 64, ragged clusters, EMPTY clusters on both sides)"""
    q, k, v, o = (T(g[f"vb_a_{n}"])[0] for n in ("q", "k", "v", "o"))       # [H, S, D]
    dmap, qc, kc = T(g["vb_a_map"])[0], T(g["vb_a_qc"])[0], T(g["vb_a_kc"])[0]
    assert (qc == 0).any() and (kc == 0).any()
    for variant in (-1, 0, 1, 3):
        got = nat.varblock_attention(q.cuda(), k.cuda(), v.cuda(), dmap.cuda(), qc.cuda().contiguous(), kc.cuda().contiguous(),
                                     variant=variant).cpu()
        torch.testing.assert_close(got.float(), o.float(), atol=2e-3, rtol=2e-3)
        e = ((got.float() - o.float()).norm() / o.float().norm()).item()
        assert e < 1e-3, (variant, e)      # (two fp16 kernels with different summation orders)


@pytest.mark.parametrize("tag", ["as_c", "as_d"])
def test_kmeans_assignment_against_the_references_triton_kernel(nat, g, tag):
    """svg_kmeans_iter's labels vs _euclid_assign_kernel in fp16.  The Triton kernel reduces the centroid norms IN fp16
    (tests/test_triton_golden.py), the HIP kernel in fp32: the labels must agree except on points whose two candidate distances are
    closer than that rounding noise — and must agree EXACTLY with the oracle's fp32-norm statement wherever the gap is not a tie."""
    x, c, ids = T(g[tag + "_x"]), T(g[tag + "_c"]), T(g[tag + "_ids"]).long()
    B, N, D = x.shape
    K = c.shape[1]
    buf = nat.KmeansBuffers(B, N, K, D, "cuda")
    c_out = torch.empty_like(c).cuda()
    nat.kmeans_iter(x.cuda(), None, c.cuda(), c_out, buf)
    lab = buf.labels.cpu().long()
    d = O.kmeans_distances(x, O.kmeans_xsq(x), c)
    mine = d.argmin(-1)
    # vs the oracle (same norms): equal unless the two distances are within fp32 rounding of each other
    dif = lab != mine
    assert ((d.gather(2, lab[..., None]) - d.gather(2, mine[..., None]))[..., 0][dif].abs() <= 1e-3 * d.abs().max()).all()
    assert dif.float().mean().item() < 0.01
    # duplicated centroids: the lower index wins
    assert not (lab == 5).any()
    # vs the reference's kernel: inside the fp16 noise of ITS norms
    p = c * c
    acc = torch.zeros(c.shape[:-1], dtype=c.dtype)
    for j in range(D):
        acc = acc + p[..., j]
    noise = (acc.float() - O.kmeans_csq(c)).abs().max().item()
    bad = lab != ids
    gap = (d.gather(2, ids[..., None]) - d.gather(2, lab[..., None]))[..., 0][bad]
    assert bad.float().mean().item() < 0.2 and (gap.abs() <= 2 * noise + 1e-3).all(), (int(bad.sum()), float(gap.abs().max()), noise)
    # and the update half on the kernel's own labels: the new centroids are the fp32 means of the assigned points, cast
    cent, cnt = O.kmeans_update(x, lab, c)
    assert torch.equal(buf.counts.cpu(), cnt)
    ulp = torch.finfo(x.dtype).eps * cent.float().abs().clamp_min(2.0 ** -14)
    assert ((c_out.cpu().float() - cent.float()).abs() <= ulp).all()


@pytest.mark.parametrize("tag", ["sap_hy", "sap_wan"])
def test_sap_processors_equal_the_references_processors(nat, g, tag):
    """The product's SAP processors (`attention_core_logic`: HIP k-means, block map, fused permutation + variable-block attention)
    against the OUTPUT of the reference's processors run as they are on the reference's Triton kernels (fixture; the flashinfer kernel
    replaced by the reference's own Triton attention kernel).  Same warm-start centroids, two k-means iterations, top-p 0.8."""
    from svg.models.hyvideo.attention import Hunyuan_SAPAttn_Processor2_0
    from svg.models.wan.attention import WanAttn_SAPAttn_Processor

    H, D, F_, P_, ctx, L, QC, KC = (int(x) for x in g[tag + "_geo"])
    q, k, v = (T(g[f"{tag}_{n}"]).cuda() for n in ("q", "k", "v"))
    o = T(g[tag + "_o"])
    cls = Hunyuan_SAPAttn_Processor2_0 if tag == "sap_hy" else WanAttn_SAPAttn_Processor
    names = ("context_length", "num_frame", "frame_size", "num_q_centroids", "num_k_centroids", "top_p_kmeans", "min_kc_ratio",
             "kmeans_iter_init", "kmeans_iter_step", "first_layers_fp", "first_times_fp", "prompt_length")
    saved = {n: getattr(cls, n) for n in names if hasattr(cls, n)}
    try:
        for n, val in zip(names, (ctx, F_, P_, QC, KC, 0.8, 0.1, 0, 2, 0, 1.0, L)):
            setattr(cls, n, val)
        proc = cls(0)
        if tag == "sap_hy":
            cls.reset_state()
        store = proc.centroid_store
        store.q[0], store.k[0] = T(g[tag + "_init_q"]).cuda(), T(g[tag + "_init_k"]).cuda()
        ts = torch.tensor([0.5], device="cuda")
        if tag == "sap_hy":
            out = proc.attention_core_logic(q.clone(), k.clone(), v.clone(), ts, 0, None)
        else:
            out = proc.attention_core_logic(q.clone(), k.clone(), v.clone(), ts)
        torch.cuda.synchronize()
        assert out.shape == o.shape
        e = ((out.float().cpu() - o.float()).norm() / o.float().norm()).item()
        assert e < 2e-3, e
        torch.testing.assert_close(out.float().cpu(), o.float(), atol=4e-3, rtol=4e-3)
        # the k-means state the processors leave behind: the reference's, to one fp16 ulp
        for mine, ref in ((store.q[0], T(g[tag + "_cq"])), (store.k[0], T(g[tag + "_ck"]))):
            mine = mine.reshape(ref.shape).float().cpu()
            assert ((mine - ref.float()).abs() <= torch.finfo(torch.float16).eps * ref.float().abs().clamp_min(2.0 ** -14)).all()
    finally:
        for n in names:
            if n in saved:
                setattr(cls, n, saved[n])
            elif n in cls.__dict__:
                delattr(cls, n)
        if tag == "sap_hy":
            cls.reset_state()


@pytest.mark.parametrize("tag", ["svg1", "svg1_wan"])
@pytest.mark.parametrize("device_switch", [False, True])
def test_svg1_processor_equals_the_references_processor(nat, g, tag, device_switch):
    """The product's Hunyuan / Wan SVG1 processors' attention_core_logic (HIP online profiler, block-sparse band attention with the fused
    layout transformation) against the OUTPUT of the reference's processors run as they are (their sample_mse, Triton placement kernels,
    torch flex_attention under their BlockMask).  The heads are built so that the profiler's choice is unambiguous (MSE ratio > 20x between
    the two masks): the decisions must be the reference's, the output the reference's to 16-bit accuracy."""
    H, D, F_, P_, ctx, L = (int(x) for x in g[tag + "_geo"])
    mul = float(g[tag + "_mul"])
    q, k, v = (T(g[f"{tag}_{n}"]).cuda() for n in "qkv")              # fp16
    best, o = T(g[tag + "_best"]), T(g[tag + "_o"]).float()
    if tag == "svg1":
        from svg.models.hyvideo.attention import Hunyuan_SVGAttn_Processor2_0 as cls
        from svg.models.hyvideo.utils import generate_temporal_head_mask_mod

        mask = generate_temporal_head_mask_mod(ctx, L, F_, P_, mul=mul)
    else:
        from svg.models.wan.attention import WanAttn_SVGAttn_Processor2_0 as cls
        from svg.models.wan.utils import generate_temporal_head_mask_mod

        mask = generate_temporal_head_mask_mod(0, 0, F_, P_, mul=mul)
    names = ("context_length", "num_frame", "frame_size", "prompt_length", "num_sampled_rows", "sample_mse_max_row", "first_layers_fp",
             "first_times_fp", "block_mask", "device_switch")
    saved = {n: getattr(cls, n) for n in names if hasattr(cls, n)}
    try:
        for n, val in zip(names, (ctx, F_, P_, L, 32, F_ * P_, 0, 1.0, mask, device_switch)):
            setattr(cls, n, val)
        proc = cls(0)
        ts = torch.tensor([0.5], device="cuda" if device_switch else "cpu")
        if tag == "svg1":
            out = proc.attention_core_logic(q.clone(), k.clone(), v.clone(), ts, 0, None)
        else:
            out = proc.attention_core_logic(q.clone(), k.clone(), v.clone(), ts)
        torch.cuda.synchronize()
        assert torch.equal(proc.last_best_mask_idx.cpu().reshape(best.shape).long(), best.long())
        e = ((out.float().cpu() - o).norm() / o.norm()).item()
        assert e < 3e-3, e
        torch.testing.assert_close(out.float().cpu(), o, atol=6e-3, rtol=6e-3)
    finally:
        for n in names:
            if n in saved:
                setattr(cls, n, saved[n])
            elif n in cls.__dict__:
                delattr(cls, n)


def test_cog_profiler_nan_quirk_and_attention_equal_the_references_processor(nat, g):
    """CogVideoX (text first).  The reference's processor drew a TEXT row; its temporal profiling mask admits no key on text rows, the
    softmax of that row is NaN, the mask's MSE is NaN and torch.argmin picks it for every head (tests/test_triton_golden.py).  With the
    rows the reference drew, svg_sample_mse reproduces that — NaN under the temporal mask, the spatial MSE equal to the reference's —,
    and svg_band_attention with the resulting decisions (text-first placement fused in) gives the reference processor's output."""
    from svg.models.cog.utils import generate_temporal_head_mask_mod, profile_desc

    tag = "svg1_cog"
    H, D, F_, P_, ctx, L = (int(x) for x in g[tag + "_geo"])
    mul = float(g[tag + "_mul"])
    q, k, v = (T(g[f"{tag}_{n}"]).cuda() for n in "qkv")
    best, o, rows, ref_mse = T(g[tag + "_best"]), T(g[tag + "_o"]).float(), T(g[tag + "_rows"]).long(), T(g[tag + "_mse"])
    mse = nat.sample_mse(q[0], k[0], v[0], rows.cuda(), profile_desc(ctx, F_, P_, emulate_bf16=False)).cpu()   # [2, H], fp32 arithmetic like the fixture
    assert torch.isnan(mse[1]).all() and torch.isnan(ref_mse[1]).all()
    torch.testing.assert_close(mse[0], ref_mse[0, 0], rtol=3e-2, atol=1e-6)     # fp16 inputs / bf16-emulating profiler vs the reference's fp32 run
    mine_best = mse.argmin(0)[None]
    assert torch.equal(mine_best, best.long())
    out = nat.band_attention(q, k, v, generate_temporal_head_mask_mod(ctx, F_, P_, mul=mul), head_perm_flag=mine_best.cuda(), vid0=ctx,
                             num_frame=F_, frame_size=P_).float().cpu()
    e = ((out - o).norm() / o.norm()).item()
    assert e < 3e-3, e


@pytest.mark.parametrize("sfx", ["h", "f"])
def test_rmsnorm_equals_the_references_triton_kernel(nat, g, sfx):
    """svg_rmsnorm_forward (and the reference-named shim svg.kernels.triton.rmsnorm.triton_rmsnorm_forward, which the Wan processors call
    for their q / k normalisation like the reference's do) == _rms_norm_fwd_fused: fp32 x * rstd * w, ONE rounding"""
    from svg.kernels.triton.rmsnorm import triton_rmsnorm_forward

    x, w, ref = T(g[f"gl_{sfx}_x"]), T(g[f"gl_{sfx}_w"]), T(g[f"gl_{sfx}_rms"])
    dt = x.dtype
    y = triton_rmsnorm_forward(x.cuda(), w.to(dt).cuda(), 1e-6).cpu().reshape(ref.shape)
    assert y.dtype == dt
    ulp = torch.finfo(dt).eps * ref.float().abs().clamp_min(1e-3)
    assert ((y.float() - ref.float()).abs() <= (1 if dt == torch.float16 else 8) * ulp).all()
    if dt == torch.float16:
        assert (y == ref).float().mean().item() > 0.99
        # and it is NOT diffusers' two-rounding form everywhere (what the module's forward would have computed)
        xf = x.float()
        two = ((xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)).to(dt) * w.to(dt)).reshape(ref.shape)
        assert (two != ref).any()


def test_wan_processor_call_equals_the_references_call(nat, g):
    """The product's WanAttn_SVGAttn_Processor2_0.__call__ on a duck-typed attention module (fp16 weights and inputs, everything on the
    HIP path: RMSNorm across heads, fused RoPE with the softmax scale folded into q, online profiler, band attention with fused layout
    transformation) against the OUTPUT of the reference's processor `__call__` run as it is in fp32 on the same (fp16-representable)
    weights and inputs: same profiler decisions, output equal to 16-bit accuracy."""
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from standins import Attention
    from svg.models.wan.attention import WanAttn_SVGAttn_Processor2_0 as cls
    from svg.models.wan.utils import generate_temporal_head_mask_mod

    heads, hd, F_, P_ = (int(x) for x in g["call_wan_geo"])
    mul, best = float(g["call_wan_mul"]), T(g["call_wan_best"])
    dim, S = heads * hd, F_ * P_
    dt = torch.float16
    attn = Attention(dim, heads, qk_norm="rms", across_heads=True, dtype=dt)
    with torch.no_grad():
        for lin in (attn.to_q, attn.to_k):
            lin.weight.copy_(torch.eye(dim))
            lin.bias.zero_()
        attn.to_v.weight.copy_(T(g["call_wan_wv"])), attn.to_v.bias.copy_(T(g["call_wan_bv"]))
        attn.to_out[0].weight.copy_(T(g["call_wan_wo"])), attn.to_out[0].bias.copy_(T(g["call_wan_bo"]))
        attn.norm_q.weight.copy_(T(g["call_wan_nq"])), attn.norm_k.weight.copy_(T(g["call_wan_nk"]))
    attn.cuda()
    names = ("context_length", "num_frame", "frame_size", "num_sampled_rows", "sample_mse_max_row", "first_layers_fp", "first_times_fp",
             "block_mask")
    saved = {n: getattr(cls, n) for n in names}
    try:
        for n, val in zip(names, (0, F_, P_, 32, S, 0, 1.0, generate_temporal_head_mask_mod(0, 0, F_, P_, mul=mul))):
            setattr(cls, n, val)
        attn.set_processor(cls(0))
        ang = T(g["call_wan_rope_ang"]).float().cuda()
        with torch.no_grad():
            out = attn(T(g["call_wan_hidden"]).cuda(), rotary_emb=(ang.cos(), ang.sin()), timestep=torch.tensor([0.5]))
        torch.cuda.synchronize()
        assert torch.equal(attn.processor.last_best_mask_idx.cpu().reshape(best.shape).long(), best.long())
        ref = T(g["call_wan_o"]).float()
        e = ((out.float().cpu() - ref).norm() / ref.norm()).item()
        assert e < 5e-3, e
        torch.testing.assert_close(out.float().cpu(), ref, atol=2e-2, rtol=2e-2)
    finally:
        for n, val in saved.items():
            setattr(cls, n, val)
