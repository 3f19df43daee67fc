"""The product's processors and k-means halves against what the EXECUTED reference produced (tests/golden/triton_golden.npz,
tests/golden/make_golden_triton.py: the reference's processors' `__call__`, Wan block forward and Triton k-means kernels run in the build
container): Hunyuan double / single stream, CogVideoX (both profiler outcomes), Cosmos, Wan cross attention + I2V image branch, Wan block
forward (fast and torch branch), euclid_assign_triton / triton_centroid_update_sorted_euclid under the reference's names.
Written at the end of round 3 (opt-in then, tests/test_gpu_experimental.py); first run on a GPU in round 4 — all green
(profiles/r04a_pytest_parked.txt) — and part of the default `-m gpu` run since."""
from pathlib import Path

import pytest
import torch

from oracle import svg_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "sparse-videogen_amd"))
    from svg import _native
    _native.load()
    return _native


@pytest.mark.parametrize("tag", ["as_c", "as_d"])
def test_euclid_assign_under_the_references_name(nat, tag):
    """svg.kmeans_utils.euclid_assign_triton (svg_kmeans_assign) against the fixture the reference's Triton kernel produced — the same
    check as tests/test_gpu_triton_golden.py makes on svg_kmeans_iter's labels"""
    import numpy as np

    from svg.kmeans_utils import euclid_assign_triton

    g = np.load(Path(__file__).resolve().parent / "golden" / "triton_golden.npz")
    x, c, ids = (torch.from_numpy(g[tag + s]) for s in ("_x", "_c", "_ids"))
    lab = euclid_assign_triton(x.cuda(), c.cuda(), None).cpu()
    assert lab.dtype == torch.int64
    d = O.kmeans_distances(x, O.kmeans_xsq(x), c)
    mine = d.argmin(-1)
    dif = lab != mine
    assert dif.float().mean().item() < 0.01
    assert ((d.gather(2, lab[..., None]) - d.gather(2, mine[..., None]))[..., 0][dif].abs() <= 1e-3 * d.abs().max()).all()
    assert (lab != ids.long()).float().mean().item() < 0.2          # the Triton kernel's fp16 norms re-label near-ties
    buf = nat.KmeansBuffers(*x.shape[:2], c.shape[1], x.shape[2], "cuda")
    nat.kmeans_iter(x.cuda(), None, c.cuda(), torch.empty_like(c).cuda(), buf)
    assert torch.equal(buf.labels.cpu().long(), lab), "the half is the whole iteration's assignment"


def test_centroid_update_under_the_references_name(nat):
    """svg.kmeans_utils.triton_centroid_update_sorted_euclid (svg_kmeans_update) against the fixture the reference's kernel produced"""
    import numpy as np

    from svg.kmeans_utils import triton_centroid_update_euclid, triton_centroid_update_sorted_euclid

    g = np.load(Path(__file__).resolve().parent / "golden" / "triton_golden.npz")
    x, ids, old = (torch.from_numpy(g["up_b" + s]) for s in ("_x", "_ids", "_old"))       # fp16, D = 128, K = 40, one empty cluster
    ref, cnt = torch.from_numpy(g["up_b_cent"]), torch.from_numpy(g["up_b_cnt"])
    cent, counts = triton_centroid_update_sorted_euclid(x.cuda(), ids.long().cuda(), old.cuda())
    assert torch.equal(counts.cpu(), cnt) and cent.dtype == x.dtype
    ulp = torch.finfo(x.dtype).eps * ref.float().abs().clamp_min(2.0 ** -14)
    assert ((cent.float().cpu() - ref.float()).abs() <= ulp).all()
    empty = cnt == 0
    assert empty.any() and torch.equal(cent.cpu()[empty], old[empty])
    assert torch.equal(triton_centroid_update_euclid(x.cuda(), ids.long().cuda(), old.cuda()), cent)


@pytest.mark.parametrize("tag", ["call_hyd", "call_hys"])
def test_hunyuan_processor_call_equals_the_references_call(nat, tag):
    """The product's Hunyuan_SVGAttn_Processor2_0.__call__ on a duck-typed attention module
    (fp16 weights and inputs, HIP path) against the OUTPUT of the reference's processor `__call__` executed in fp32 on the same
    fp16-representable weights and inputs (tests/golden/make_golden_triton.py section 12): double-stream block (text stream with its own
    projections / norms / output projection) and single-stream block (concatenated sequence, no output projection).  Same profiler
    decisions, both outputs equal to 16-bit accuracy."""
    import sys

    import numpy as np
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from standins import Attention
    from svg.models.hyvideo.attention import Hunyuan_SVGAttn_Processor2_0 as cls
    from svg.models.hyvideo.utils import generate_temporal_head_mask_mod

    g = np.load(Path(__file__).resolve().parent / "golden" / "triton_golden.npz")

    def T(name):
        return torch.from_numpy(g[f"{tag}_{name}"])

    single = tag == "call_hys"
    heads, hd, F_, P_, ctx, L = (int(x) for x in g[tag + "_geo"])
    mul, best = float(g[tag + "_mul"]), T("best")
    dim, V = heads * hd, F_ * P_
    S = V + ctx
    dt = torch.float16
    attn = Attention(dim, heads, qk_norm="rms", added_kv=not single, dtype=dt)
    with torch.no_grad():
        for lin in (attn.to_q, attn.to_k):
            lin.weight.copy_(torch.eye(dim))
            lin.bias.zero_()
        attn.to_v.weight.copy_(T("wv")), attn.to_v.bias.copy_(T("bv"))
        attn.norm_q.weight.copy_(T("nq")), attn.norm_k.weight.copy_(T("nk"))
        if single:
            attn.to_out = None
        else:
            attn.to_out[0].weight.copy_(T("wo")), attn.to_out[0].bias.copy_(T("bo"))
            attn.norm_added_q.weight.copy_(T("naq")), attn.norm_added_k.weight.copy_(T("nak"))
            for n, lin in (("aq", attn.add_q_proj), ("ak", attn.add_k_proj), ("av", attn.add_v_proj), ("ao", attn.to_add_out)):
                lin.weight.copy_(T("w" + n)), lin.bias.copy_(T("b" + n))
    attn.cuda()
    names = ("context_length", "num_frame", "frame_size", "num_sampled_rows", "sample_mse_max_row", "prompt_length", "first_layers_fp",
             "first_times_fp", "block_mask")
    saved = {n: getattr(cls, n) for n in names}
    try:
        for n, val in zip(names, (ctx, F_, P_, 32, V, L, 0, 1.0, generate_temporal_head_mask_mod(ctx, L, F_, P_, mul=mul))):
            setattr(cls, n, val)
        attn.set_processor(cls(0))
        ang = T("rope_ang").float().cuda()
        rope = (ang.cos().repeat_interleave(2, -1), ang.sin().repeat_interleave(2, -1))
        amask = torch.zeros(S, dtype=torch.bool, device="cuda")
        amask[:V + L] = True
        with torch.no_grad():
            o_h, o_e = attn(T("hidden").cuda(), encoder_hidden_states=T("enc").cuda(), attention_mask=amask, image_rotary_emb=rope,
                            timestep=torch.tensor([0.5]))
        torch.cuda.synchronize()
        assert torch.equal(attn.processor.last_best_mask_idx.cpu().reshape(best.shape).long(), best.long())
        for got, name in ((o_h, "o_h"), (o_e, "o_e")):
            ref = T(name).float()
            e = ((got.float().cpu() - ref).norm() / ref.norm()).item()
            assert e < 5e-3, (name, e)
            torch.testing.assert_close(got.float().cpu(), ref, atol=2e-2, rtol=2e-2)
    finally:
        for n, val in saved.items():
            setattr(cls, n, val)


@pytest.mark.parametrize("which", ["v", "t"])
def test_cog_processor_call_equals_the_references_call(nat, which):
    """The product's CogVideoX_SparseAttn_Processor2_0.__call__
    (fp16, HIP path: LayerNorm over head_dim, RoPE on the video rows with the softmax scale folded in, profiler, band attention with
    fused layout transformation) against the reference's processor `__call__` executed in fp32 (make_golden_triton.py section 13).  The
    profiler draws its rows from the CPU generator like the reference, so seeding the same way profiles the same rows: `v` = video rows
    only (the heads' structure decides), `t` = a text row among them (NaN -> every head temporal; reference quirk, reproduced)."""
    import sys

    import numpy as np
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from standins import Attention
    from svg.models.cog.attention import CogVideoX_SparseAttn_Processor2_0 as cls
    from svg.models.cog.utils import generate_temporal_head_mask_mod

    g = np.load(Path(__file__).resolve().parent / "golden" / "triton_golden.npz")

    def T(name):
        return torch.from_numpy(g["call_cog_" + name])

    heads, hd, F_, P_, ctx = (int(x) for x in g["call_cog_geo"])
    mul, best = float(g["call_cog_mul"]), T(which + "_best")
    dim = heads * hd
    dt = torch.float16
    attn = Attention(dim, heads, qk_norm="layer", dtype=dt)
    with torch.no_grad():
        for lin in (attn.to_q, attn.to_k):
            lin.weight.copy_(torch.eye(dim))
            lin.bias.zero_()
        attn.to_v.weight.copy_(T("wv")), attn.to_v.bias.copy_(T("bv"))
        attn.to_out[0].weight.copy_(T("wo")), attn.to_out[0].bias.copy_(T("bo"))
        attn.norm_q.weight.copy_(T("nq")), attn.norm_q.bias.copy_(T("nqb"))
        attn.norm_k.weight.copy_(T("nk")), attn.norm_k.bias.copy_(T("nkb"))
    attn.cuda()
    names = ("context_length", "num_frame", "frame_size", "num_sampled_rows", "first_layers_fp", "first_times_fp", "block_mask")
    saved = {n: getattr(cls, n) for n in names}
    try:
        for n, val in zip(names, (ctx, F_, P_, 32, 0, 0.0, generate_temporal_head_mask_mod(ctx, F_, P_, mul=mul))):
            setattr(cls, n, val)
        attn.set_processor(cls(0))
        ang = T("rope_ang").float().cuda()
        rope = (ang.cos().repeat_interleave(2, -1), ang.sin().repeat_interleave(2, -1))
        torch.manual_seed(int(g[f"call_cog_{which}_seed"]))
        with torch.no_grad():
            o_h, o_e = attn(T("hidden").cuda(), encoder_hidden_states=T("enc").cuda(), image_rotary_emb=rope, timestep=torch.tensor([0.5]))
        torch.cuda.synchronize()
        assert torch.equal(attn.processor.last_best_mask_idx.cpu().reshape(best.shape).long(), best.long())
        for got, name in ((o_h, "o_h"), (o_e, "o_e")):
            ref = T(f"{which}_{name}").float()
            e = ((got.float().cpu() - ref).norm() / ref.norm()).item()
            assert e < 5e-3, (name, e)
            torch.testing.assert_close(got.float().cpu(), ref, atol=2e-2, rtol=2e-2)
    finally:
        for n, val in saved.items():
            setattr(cls, n, val)


@pytest.mark.parametrize("branch", ["fast", "torch"])
def test_wan_block_forward_equals_the_references_block(nat, branch):
    """The product's wan_block_forward (HIP glue: fused LayerNorm + modulate, gate-residual; fp16) against
    the reference's WanTransformerBlock_Sparse.forward executed in fp32 (make_golden_triton.py section 14) on the same fp16-representable
    block: `fast` = the reference on its Triton kernels — the product with svg.kernels.triton.layernorm.REFERENCE_PADDING on reproduces its
    padded variance; `torch` = the reference's fall-back (FP32LayerNorm), the product's default.  Hidden size 192 with a row mean: the two
    references differ by up to 0.6, so each switch position can only match its own."""
    import types

    import numpy as np
    from svg.kernels.triton import layernorm as ln_mod
    from svg.models.wan.custom_models import wan_block_forward

    g = np.load(Path(__file__).resolve().parent / "golden" / "triton_golden.npz")
    dt = torch.float16
    t = {n[4:]: torch.from_numpy(g[n]).cuda() for n in g.files if n.startswith("blk_")}
    C = t["hidden"].shape[-1]

    def lin(name, x):
        return torch.nn.functional.linear(x, t[name + "_w"].to(dt), t[name + "_b"].to(dt))

    n1, n3 = (torch.nn.LayerNorm(C, eps=1e-6, elementwise_affine=False) for _ in range(2))
    n2 = torch.nn.LayerNorm(C, eps=1e-6, elementwise_affine=True).cuda()
    with torch.no_grad():
        n2.weight.copy_(t["n2w"]), n2.bias.copy_(t["n2b"])
    blk = types.SimpleNamespace(
        scale_shift_table=t["table"].float(), norm1=n1, norm2=n2, norm3=n3,
        attn1=lambda hidden_states, rotary_emb=None, timestep=None: lin("attn1", torch.roll(hidden_states, 1, 1)),
        attn2=lambda hidden_states, encoder_hidden_states=None: lin("attn2", hidden_states) + encoder_hidden_states.mean(1, keepdim=True),
        ffn=lambda x: torch.tanh(lin("ffn", x)))
    saved = ln_mod.REFERENCE_PADDING
    try:
        ln_mod.REFERENCE_PADDING = branch == "fast"
        with torch.no_grad():
            out = wan_block_forward(blk, t["hidden"].to(dt), t["enc"].to(dt), t["temb"].float(), None, timestep=0)
    finally:
        ln_mod.REFERENCE_PADDING = saved
    torch.cuda.synchronize()
    ref, other = t[branch + "_out"].float(), t[("torch" if branch == "fast" else "fast") + "_out"].float()
    e = ((out.float() - ref).norm() / ref.norm()).item()
    e_other = ((out.float() - other).norm() / other.norm()).item()
    assert e < 5e-3 and e_other > 5 * e, (e, e_other)
    torch.testing.assert_close(out.float(), ref, atol=3e-2, rtol=2e-2)


def test_cosmos_processor_call_equals_the_references_call(nat):
    """The product's Cosmos_SVG_AttnProcessor2_0.__call__ (fp16: per-head RMSNorm on the HIP path, half-split
    RoPE, the Wan sparse core) and the same processor as cross attention against the reference's executed `__call__`
    (make_golden_triton.py section 15)."""
    import sys

    import numpy as np
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from standins import Attention
    from svg.models.cosmos.attention import Cosmos_SVG_AttnProcessor2_0 as cls
    from svg.models.wan.utils import generate_temporal_head_mask_mod

    g = np.load(Path(__file__).resolve().parent / "golden" / "triton_golden.npz")

    def T(name):
        return torch.from_numpy(g["call_cos_" + name])

    heads, hd, F_, P_ = (int(x) for x in g["call_cos_geo"])
    mul, best = float(g["call_cos_mul"]), T("best")
    dim, S = heads * hd, F_ * P_
    dt = torch.float16
    attn = Attention(dim, heads, qk_norm="rms", dtype=dt)
    with torch.no_grad():
        for lin in (attn.to_q, attn.to_k):
            lin.weight.copy_(torch.eye(dim))
            lin.bias.zero_()
        attn.to_v.weight.copy_(T("wv")), attn.to_v.bias.copy_(T("bv"))
        attn.to_out[0].weight.copy_(T("wo")), attn.to_out[0].bias.copy_(T("bo"))
        attn.norm_q.weight.copy_(T("nq")), attn.norm_k.weight.copy_(T("nk"))
    attn.cuda()
    names = ("context_length", "num_frame", "frame_size", "num_sampled_rows", "sample_mse_max_row", "first_layers_fp", "first_times_fp",
             "block_mask")
    saved = {n: getattr(cls, n) for n in names}
    try:
        for n, val in zip(names, (0, F_, P_, 32, S, 0, 1.0, generate_temporal_head_mask_mod(0, 0, F_, P_, mul=mul))):
            setattr(cls, n, val)
        attn.set_processor(cls(0))
        ang = T("rope_ang").float().cuda()
        rope = (torch.cat([ang.cos()] * 2, -1), torch.cat([ang.sin()] * 2, -1))
        with torch.no_grad():
            out = attn(T("hidden").cuda(), image_rotary_emb=rope, timestep=torch.tensor([0.5]))
            oc = attn(T("hidden").cuda(), encoder_hidden_states=T("enc").cuda())
        torch.cuda.synchronize()
        assert torch.equal(attn.processor.last_best_mask_idx.cpu().reshape(best.shape).long(), best.long())
        for got, name in ((out, "o"), (oc, "o_cross")):
            ref = T(name).float()
            e = ((got.float().cpu() - ref).norm() / ref.norm()).item()
            assert e < 5e-3, (name, e)
            torch.testing.assert_close(got.float().cpu(), ref, atol=2e-2, rtol=2e-2)
    finally:
        for n, val in saved.items():
            setattr(cls, n, val)


@pytest.mark.parametrize("tag", ["xwan_t2v", "xwan_i2v"])
def test_wan_cross_attention_and_i2v_equal_the_references_call(nat, tag):
    """The product's Wan processor as cross attention (fp16 on the GPU: RMSNorm across heads through
    svg_rmsnorm_forward, torch SDPA), text only and with the I2V image branch, against the reference's executed call
    (make_golden_triton.py section 16)."""
    import sys

    import numpy as np
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from standins import RMSNorm, Attention
    from svg.models.wan.attention import WanAttn_SVGAttn_Processor2_0 as cls

    g = np.load(Path(__file__).resolve().parent / "golden" / "triton_golden.npz")
    t = {n[len(tag) + 1:]: torch.from_numpy(g[n]) for n in g.files if n.startswith(tag + "_") and not n.endswith("_geo")}
    heads, hd = (int(x) for x in g[tag + "_geo"])
    i2v, dim, dt = tag == "xwan_i2v", heads * hd, torch.float16
    attn = Attention(dim, heads, qk_norm="rms", across_heads=True, added_kv=i2v, dtype=dt)
    mods = [("q", attn.to_q), ("k", attn.to_k), ("v", attn.to_v), ("o", attn.to_out[0])]
    if i2v:
        attn.norm_added_k = RMSNorm(dim).to(dt)
        mods += [("ak", attn.add_k_proj), ("av", attn.add_v_proj)]
    with torch.no_grad():
        for n, m in mods:
            m.weight.copy_(t["w" + n]), m.bias.copy_(t["b" + n])
        attn.norm_q.weight.copy_(t["nq"]), attn.norm_k.weight.copy_(t["nk"])
        if i2v:
            attn.norm_added_k.weight.copy_(t["nak"])
    attn.cuda()
    with torch.no_grad():
        got = cls(0)(attn, t["hidden"].cuda(), encoder_hidden_states=t["enc"].cuda())
    torch.cuda.synchronize()
    ref = t["o"].float()
    e = ((got.float().cpu() - ref).norm() / ref.norm()).item()
    assert e < 5e-3, e
    torch.testing.assert_close(got.float().cpu(), ref, atol=2e-2, rtol=2e-2)
