"""BASELINE.json configs[0]: CogVideoX, 49 frames 480p geometry (cfg=2, H=48, D=64, text 226 first, F=13, P=1350,
S=17776), ONE denoise step through a stand-in 2-block stack on CPU with dense torch SDPA — the reference's own CPU-runnable
case.  It checks the plumbing (install hook, class-level config, timestep hook, processor protocol, dense branch) and that
the sparse branch refuses CPU tensors instead of silently falling back."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import svg_oracle as O
from standins import Attention, Block, Pipe, Transformer


def make_cog(heads=48, head_dim=64, layers=2, dtype=torch.bfloat16):
    torch.manual_seed(0)
    blocks = [Block(Attention(heads * head_dim, heads, qk_norm="layer", dtype=dtype), "attn1") for _ in range(layers)]
    return Pipe(Transformer(blocks, "transformer_blocks"))


def torch_reference_block(attn, hidden, enc):
    """CogVideoX attention block in plain torch (text first, LayerNorm QK, no rope) for the dense branch."""
    x = torch.cat([enc, hidden], dim=1)
    B = x.shape[0]
    q, k, v = attn.to_q(x), attn.to_k(x), attn.to_v(x)
    hd = q.shape[-1] // attn.heads
    q, k, v = (t.view(B, -1, attn.heads, hd).transpose(1, 2) for t in (q, k, v))
    q, k = attn.norm_q(q), attn.norm_k(k)
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.transpose(1, 2).reshape(B, -1, attn.heads * hd)
    o = attn.to_out[0](o)
    return o[:, enc.shape[1]:], o[:, : enc.shape[1]]


def test_cog_config1_dense_step_on_cpu():
    from svg.models.cog.attention import CogVideoX_SparseAttn_Processor2_0
    from svg.models.cog.inference import replace_cog_attention

    torch.set_num_threads(8)
    pipe = make_cog()
    cls = replace_cog_attention(pipe, "v1", num_sampled_rows=32, sparsity=0.25, first_layers_fp=0.025, first_times_fp=0.2)
    assert cls is CogVideoX_SparseAttn_Processor2_0
    assert (cls.context_length, cls.num_frame, cls.frame_size) == (226, 13, 1350)
    assert cls.block_mask.band == math.floor(cls.block_mask.band / 128) * 128 and cls.block_mask.rowfull_hi == 226
    cfg, S_text, S_vid, dim = 2, 226, 13 * 1350, 48 * 64
    torch.manual_seed(1)
    hidden = torch.randn(cfg, S_vid, dim).to(torch.bfloat16) * 0.1
    enc = torch.randn(cfg, S_text, dim).to(torch.bfloat16) * 0.1
    t_dense = torch.tensor([999.0])  # > 1000 * (1 - 0.2): warm-up step -> dense branch (ref cog/attention.py:173-176)
    with torch.no_grad():
        out_h, out_e = pipe.transformer(hidden, encoder_hidden_states=enc, timestep=t_dense)
        # plain-torch statement of the same two blocks
        h, e = hidden, enc
        for b in pipe.transformer.transformer_blocks:
            dh, de = torch_reference_block(b.attn1, h, e)
            h, e = h + dh, e + de
    torch.testing.assert_close(out_h.float(), h.float(), atol=2e-2, rtol=2e-2)
    torch.testing.assert_close(out_e.float(), e.float(), atol=2e-2, rtol=2e-2)
    # a sparse step on CPU tensors must fail loudly: the sparse path exists only as HIP kernels
    cls.first_layers_fp = 0.0
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU fallback"):
        pipe.transformer(hidden, encoder_hidden_states=enc, timestep=torch.tensor([100.0]))


def test_timestep_hook_and_explicit_keyword():
    from svg.models import context
    from svg.models.cog.inference import replace_cog_attention

    pipe = make_cog(heads=2, head_dim=64, layers=1, dtype=torch.float32)
    replace_cog_attention(pipe, "v1", 32, 0.25, 0.0, 0.2)
    seen = []
    proc = pipe.transformer.transformer_blocks[0].attn1.processor
    orig = proc.attention_core_logic
    proc.attention_core_logic = lambda q, k, v, t: (seen.append(t), orig(q, k, v, torch.tensor([1000.0])))[1]
    S_vid = 13 * 1350
    with torch.no_grad():
        pipe.transformer(torch.zeros(1, S_vid, 128), encoder_hidden_states=torch.zeros(1, 226, 128), timestep=torch.tensor([7.0]))
        assert seen[-1].item() == 7.0 and context.current_timestep() is None
        attn = pipe.transformer.transformer_blocks[0].attn1
        attn(torch.zeros(1, S_vid, 128), encoder_hidden_states=torch.zeros(1, 226, 128), timestep=torch.tensor([5.0]))
        assert seen[-1].item() == 5.0


def test_hunyuan_and_wan_install_hooks_set_class_config():
    from svg.models.hyvideo.attention import Hunyuan_SAPAttn_Processor2_0, Hunyuan_SVGAttn_Processor2_0
    from svg.models.hyvideo.inference import replace_hyvideo_attention
    from svg.models.wan.inference import replace_wan_attention

    blocks = [Block(Attention(256, 2, added_kv=(i < 1)), "attn") for i in range(3)]
    tr = Transformer(blocks[:1], "transformer_blocks")
    tr.single_transformer_blocks = torch.nn.ModuleList(blocks[1:])
    pipe = Pipe(tr)
    cls = replace_hyvideo_attention(pipe, 720, 1280, 129, 64, first_layers_fp=1, first_times_fp=900.0, pattern="SVG",
                                    num_sampled_rows=64, sparsity=0.25)
    assert cls is Hunyuan_SVGAttn_Processor2_0
    assert (cls.context_length, cls.num_frame, cls.frame_size, cls.prompt_length) == (256, 33, 3600, 64)
    assert cls.block_mask.band == 15616 and cls.block_mask.real_len == 33 * 3600 + 64
    assert [b.attn.processor.layer_idx for b in blocks] == [0, 1, 2]
    cls = replace_hyvideo_attention(pipe, 720, 1280, 129, 64, 1, 900.0, pattern="SAP", num_q_centroids=400,
                                    num_k_centroids=1000, top_p_kmeans=0.9, min_kc_ratio=0.1, kmeans_iter_init=50,
                                    kmeans_iter_step=2, zero_step_kmeans_init=True)
    assert cls is Hunyuan_SAPAttn_Processor2_0 and cls.num_k_centroids == 1000 and cls.kmeans_iter_step == 2

    class WanCfg:
        patch_size = (1, 2, 2)

    wblocks = [Block(Attention(256, 2, across_heads=True), "attn1") for _ in range(2)]
    wtr = Transformer(wblocks, "blocks")
    wtr.config = WanCfg()
    wpipe = Pipe(wtr)
    wpipe.vae_scale_factor_temporal, wpipe.vae_scale_factor_spatial = 4, 8
    wcls = replace_wan_attention(wpipe, 720, 1280, 81, first_layers_fp=1, first_times_fp=800.0, pattern="SVG", sparsity=0.3)
    assert (wcls.context_length, wcls.num_frame, wcls.frame_size) == (0, 21, 3600)
    assert wcls.block_mask.band == 12417 and wcls.block_mask.colfull_hi == 3600


def test_density_log_is_deferred_and_flushable(tmp_path):
    """density records are written off the critical path (no per-layer .item()); same fields as the reference's
    JSON lines (hyvideo/attention.py:786-802)."""
    import json

    from svg.models import _core

    path = tmp_path / "density.jsonl"
    _core.DENSITY_LOG.push(str(path), {"timestep": 981.0, "layer": 7}, torch.tensor([[0.25, 0.5, 0.75]]))
    _core.DENSITY_LOG.push(str(path), {"timestep": 981.0, "layer": 8}, torch.tensor([[1.0, 0.0, 0.5]]))
    _core.flush_density_log()
    lines = [json.loads(l) for l in path.read_text().splitlines()]
    assert [l["layer"] for l in lines] == [7, 8]
    assert list(lines[0]) == ["timestep", "layer", "avg_density", "density"]
    assert lines[0]["avg_density"] == pytest.approx(0.5) and lines[1]["density"] == [[1.0, 0.0, 0.5]]
    assert not _core.DENSITY_LOG.pending


def test_switch_generator_follows_seeding():
    """The device-switched SVG1 path draws its profiler rows from a private CPU generator: it must restart whenever the process is
    re-seeded — seed_everything (also with the SAME seed: a second video), a changed torch.manual_seed, the install hooks — and must
    never consume the global CPU stream."""
    from svg.models import _core
    from svg.utils.seed import seed_everything

    def draw():
        return torch.randint(0, 10000, (16,), generator=_core._switch_generator())

    seed_everything(7)
    state = torch.get_rng_state()
    a = draw()
    assert torch.equal(state, torch.get_rng_state())
    b = draw()
    assert not torch.equal(a, b)
    seed_everything(7)                       # same seed again: same rows as the first video
    assert torch.equal(draw(), a) and torch.equal(draw(), b)
    torch.manual_seed(8)                     # a different global seed is picked up without any hook
    c = draw()
    assert not torch.equal(c, a)
    _core.reseed_switch_generator()          # what replace_*_attention calls
    assert torch.equal(draw(), c)


def test_token_reorder_helpers_match_the_reference():
    """`*_token_reorder_to_token_major / _frame_major` of svg/models/{hyvideo,wan,cog,cosmos}/placement.py: in place on every head, text
    last (first for CogVideoX); the reference's own torch helpers are run beside them when /root/reference is there (this container)."""
    import importlib
    import sys
    import types
    from pathlib import Path

    F_, P_, ctx = 3, 5, 4
    for model, prefix, text_first in (("hyvideo", "hunyuan_", False), ("wan", "wan_", False), ("cog", "", True), ("cosmos", "cosmos_", False)):
        mod = importlib.import_module(f"svg.models.{model}.placement")
        to_tok, to_frm = getattr(mod, prefix + "token_reorder_to_token_major"), getattr(mod, prefix + "token_reorder_to_frame_major")
        x = torch.arange(2 * 3 * (F_ * P_ + ctx) * 2, dtype=torch.float32).reshape(2, 3, F_ * P_ + ctx, 2)
        y = to_tok(x.clone(), ctx, F_ * P_, F_, P_)
        ones = torch.ones(2, 3, dtype=torch.int32)
        assert torch.equal(y, O.head_placement(x, ones, ctx, F_, P_, text_first=text_first))
        assert torch.equal(to_frm(y.clone(), ctx, F_ * P_, F_, P_), x)
        t = x.clone()
        assert to_tok(t, ctx, F_ * P_, F_, P_) is t                      # in place, returns its argument
        x0 = x[:, :, : F_ * P_].contiguous() if not text_first else x[:, :, ctx:].contiguous()
        assert torch.equal(to_frm(to_tok(x0.clone(), 0, F_ * P_, F_, P_), 0, F_ * P_, F_, P_), x0)     # fix_len == 0 works here
    ref_root = Path("/root/reference")
    if ref_root.exists():       # the reference's own helpers (torch only; its module imports triton, which this image has)
        spec = importlib.util.spec_from_file_location("_ref_cog_placement", ref_root / "svg/models/cog/placement.py")
        ref = importlib.util.module_from_spec(spec)
        try:
            spec.loader.exec_module(ref)
        except Exception:
            ref = None
        if ref is not None:
            from svg.models.cog import placement as mine

            x = torch.randn(1, 2, F_ * P_ + ctx, 4)
            assert torch.equal(mine.token_reorder_to_token_major(x.clone(), ctx, F_ * P_, F_, P_),
                               ref.token_reorder_to_token_major(x.clone(), ctx, F_ * P_, F_, P_))
            assert torch.equal(mine.token_reorder_to_frame_major(x.clone(), ctx, F_ * P_, F_, P_),
                               ref.token_reorder_to_frame_major(x.clone(), ctx, F_ * P_, F_, P_))


def test_wan_bsr_backend_helpers_layout(monkeypatch):
    """svg/models/wan/utils.py flashinfer_sparse_attn_forward / prepare_flashinfer_attention (the uniform-block alternative backend of the
    reference's Wan processors): the [cfg, H, S, D] <-> [S, cfg * H, D] plumbing around the BSR op, checked on the CPU with the op
    replaced by the oracle under the same block mask (the op itself is tested on the GPU, tests/test_gpu_bsr.py)."""
    import numpy as np

    from svg.kernels.ops import attention_ops_wan as W
    from svg.models.cosmos import utils as cosmos_u
    from svg.models.wan import attention as wan_attn
    from svg.models.wan import utils as wan_u

    F_, P_, mul, cfg, H, D = 3, 8, 1.2, 2, 2, 16
    S = F_ * P_
    meta = wan_attn.prepare_flashinfer_attention(cfg, H, D, torch.float32, "cpu", 0, 0, F_, P_, diag_width=mul, multiplier=mul)
    indptr, cols, (bs, _) = meta
    assert bs == W.get_factor(F_, P_) == 8 and cosmos_u.gen_temporal_mask is wan_u.gen_temporal_mask
    blk = torch.from_numpy(W.ref_gen_temporal_mask(F_, P_, mul) != -1)
    mask = blk.repeat_interleave(bs, 0).repeat_interleave(bs, 1)

    def fake_op(q, k, v, metadata):          # [S, heads, D] like the op
        assert metadata.temporal_mask_metadata is meta and q.shape == (S, cfg * H, D)
        return O.masked_attention(q.permute(1, 0, 2), k.permute(1, 0, 2), v.permute(1, 0, 2), mask).permute(1, 0, 2)

    monkeypatch.setattr(W, "wan_sparse_attn_forward", fake_op)
    g = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(cfg, H, S, D, generator=g) for _ in range(3))
    o = wan_u.flashinfer_sparse_attn_forward(q, k, v, meta)
    ref = O.masked_attention(q, k, v, mask)
    torch.testing.assert_close(o, ref, atol=1e-6, rtol=1e-6)
    # BSR arrays: row pointer + active column indices in row-major order (+ the reference's 256 padding zeros)
    n_active = int(blk.sum())
    assert indptr.tolist() == [0] + np.cumsum(blk.sum(1).numpy()).tolist() and cols.numel() == n_active + 256
    assert cols[:n_active].tolist() == blk.nonzero()[:, 1].tolist()


# ---- the install hooks against what the reference's hooks computed (tests/golden/make_golden_install.py executes them) ----------------
def _install_golden():
    import json
    from pathlib import Path
    return json.loads((Path(__file__).resolve().parent / "golden" / "install_golden.json").read_text())


def _mask_fields(m):
    return {n: int(getattr(m, n)) for n in ("real_len", "band", "colfull_lo", "colfull_hi", "rowfull_lo", "rowfull_hi")}


_CONFIG_KEYS = ("context_length", "num_frame", "frame_size", "prompt_length", "num_sampled_rows", "sample_mse_max_row", "first_layers_fp",
                "first_times_fp", "sparsity", "version", "num_q_centroids", "num_k_centroids", "top_p_kmeans", "min_kc_ratio",
                "kmeans_iter_init", "kmeans_iter_step", "zero_step_kmeans_init")


_SVG_ONLY = ("num_sampled_rows", "sample_mse_max_row", "sparsity")          # what only the SVG branch of a hook sets: on the SAP class (a
_SAP_ONLY = _CONFIG_KEYS[10:]                                               # subclass) they are whatever an earlier SVG install left behind


def _check_config(cls, ref_cfg, tag):
    checked = 0
    skip = _SVG_ONLY if tag.endswith("_sap") else _SAP_ONLY
    for k in _CONFIG_KEYS:
        if k in ref_cfg and hasattr(cls, k) and k not in skip:
            assert getattr(cls, k) == ref_cfg[k], (tag, k, getattr(cls, k), ref_cfg[k])
            checked += 1
    assert checked >= 6, (tag, checked)


@pytest.mark.parametrize("tag", ["hy_720p_svg", "hy_480p_svg", "hy_odd_svg", "hy_720p_sap"])
def test_hunyuan_install_hook_equals_the_references(tag):
    """replace_hyvideo_attention with the arguments of the reference's scripts: every scalar class attribute the reference's hook set,
    the band mask that follows from the multiplier IT handed to prepare_flexattention, and the processor class / layer index per block."""
    from svg.models.hyvideo.inference import replace_hyvideo_attention, replace_hyvideo_flashattention

    gold = _install_golden()
    ref = gold[tag]
    n = len(ref["blocks"])
    blocks = [Block(Attention(256, 2, added_kv=(i < 3)), "attn") for i in range(n)]
    tr = Transformer(blocks[:3], "transformer_blocks")
    tr.single_transformer_blocks = torch.nn.ModuleList(blocks[3:])
    call = dict(ref["call"])
    cls = replace_hyvideo_attention(Pipe(tr), call.pop("height"), call.pop("width"), call.pop("num_frames"), call.pop("prompt_length"),
                                    call.pop("first_layers_fp"), call.pop("first_times_fp"), **call)
    _check_config(cls, ref["config"], tag)
    assert [[type(b.attn.processor).__name__, b.attn.processor.layer_idx] for b in blocks] == [b[:2] for b in ref["blocks"]]
    if ref["call"]["pattern"] == "SVG":
        (flex,) = ref["recorded"]["prepare_flexattention"]
        ctx, L, F_, P_ = flex["args"][5:9]
        assert flex["kwargs"]["diag_width"] == flex["kwargs"]["multiplier"]
        assert _mask_fields(cls.block_mask) == O.hy_band_params(F_ * P_ + ctx, ctx, L, F_, P_, flex["kwargs"]["multiplier"])
        # the profiling masks were asked for with the same geometry
        assert ref["recorded"]["get_attention_mask"][0][1:5] == [cls.sample_mse_max_row, cls.context_length, cls.num_frame, cls.frame_size]
    if tag == "hy_720p_svg":
        replace_hyvideo_flashattention(Pipe(tr))
        dense = gold["hy_dense"]["blocks"]
        assert [[type(b.attn.processor).__name__, b.attn.processor.layer_idx] for b in blocks[:len(dense)]] == [[d[0], i] for i, d in enumerate(dense)]


@pytest.mark.parametrize("tag", ["wan_720p_svg", "wan_480p_svg", "wan_odd_svg", "wan_720p_sap"])
def test_wan_install_hook_equals_the_references(tag):
    from svg.models.wan.inference import replace_wan_attention

    ref = _install_golden()[tag]

    class WanCfg:
        patch_size = (1, 2, 2)

    blocks = [Block(Attention(256, 2, across_heads=True), "attn1") for _ in ref["blocks"]]
    tr = Transformer(blocks, "blocks")
    tr.config, tr.num_attention_heads, tr.attention_head_dim = WanCfg(), 40, 128
    pipe = Pipe(tr)
    pipe.vae_scale_factor_temporal, pipe.vae_scale_factor_spatial = 4, 8
    call = dict(ref["call"])
    cls = replace_wan_attention(pipe, call.pop("height"), call.pop("width"), call.pop("num_frames"), call.pop("first_layers_fp"),
                                call.pop("first_times_fp"), **call)
    _check_config(cls, ref["config"], tag)
    got = [[type(b.attn1.processor).__name__, b.attn1.processor.layer_idx, getattr(b.attn1.processor, "num_layers", None)] for b in blocks]
    assert got == ref["blocks"]
    if ref["call"]["pattern"] == "SVG":
        (flex,) = ref["recorded"]["prepare_flexattention"]
        F_, P_, diag, mult = flex["args"][7:11]
        assert diag == mult and flex["args"][5:7] == [0, 0]
        assert _mask_fields(cls.block_mask) == O.wan_band_params(F_ * P_, F_, P_, mult)


@pytest.mark.parametrize("tag", ["cog_v1", "cog_v15"])
def test_cog_install_hook_equals_the_references(tag):
    from svg.models.cog.inference import replace_cog_attention

    ref = _install_golden()[tag]
    pipe = make_cog(heads=2, head_dim=64, layers=len(ref["blocks"]))
    c = ref["call"]
    cls = replace_cog_attention(pipe, c["version"], c["num_sampled_rows"], c["sparsity"], c["first_layers_fp"], c["first_times_fp"])
    _check_config(cls, ref["config"], tag)
    got = [[type(b.attn1.processor).__name__, b.attn1.processor.layer_idx, getattr(b.attn1.processor, "num_layers", None)]
           for b in pipe.transformer.transformer_blocks]
    assert got == ref["blocks"]
    (flex,) = ref["recorded"]["prepare_flexattention"]
    ctx, F_, P_, diag, mult = flex["args"][5:10]
    assert diag == mult
    assert _mask_fields(cls.block_mask) == O.cog_band_params(F_ * P_ + ctx, ctx, F_, P_, mult)


@pytest.mark.parametrize("tag", ["cosmos_svg", "cosmos_sap"])
def test_cosmos_install_hook_equals_the_references(tag):
    from svg.models.cosmos.inference import replace_cosmos_attention

    ref = _install_golden()[tag]

    class Cfg:
        patch_size = (1, 2, 2)

    blocks = [Block(Attention(256, 2), "attn1") for _ in ref["blocks"]]
    tr = Transformer(blocks, "transformer_blocks")
    tr.config, tr.num_attention_heads, tr.attention_head_dim = Cfg(), 40, 128
    pipe = Pipe(tr)
    pipe.vae_scale_factor_temporal, pipe.vae_scale_factor_spatial = 8, 8
    call = dict(ref["call"])
    cls = replace_cosmos_attention(pipe, call.pop("height"), call.pop("width"), call.pop("num_frames"), call.pop("first_layers_fp"),
                                   call.pop("first_times_fp"), **call)
    _check_config(cls, ref["config"], tag)
    got = [[type(b.attn1.processor).__name__, b.attn1.processor.layer_idx, getattr(b.attn1.processor, "num_layers", None)] for b in blocks]
    assert got == ref["blocks"]
    if ref["call"]["pattern"] == "SVG":
        (flex,) = ref["recorded"]["prepare_flexattention"]
        F_, P_, diag, mult = flex["args"][7:11]
        assert diag == mult
        assert _mask_fields(cls.block_mask) == O.wan_band_params(F_ * P_, F_, P_, mult)     # (cosmos/utils.py is wan/utils.py)


def test_wan_install_hook_flashinfer_backend_equals_the_references():
    """attention_backend="flashinfer": the reference prepares BSR metadata instead of a flex BlockMask (wan/inference.py:92-117); same
    class configuration, same per-block processors, and the product's metadata is the uniform-block statement of the same mask."""
    from svg.models.wan.inference import replace_wan_attention

    ref = _install_golden()["wan_720p_svg_flashinfer"]

    class WanCfg:
        patch_size = (1, 2, 2)

    blocks = [Block(Attention(256, 2, across_heads=True), "attn1") for _ in ref["blocks"]]
    tr = Transformer(blocks, "blocks")
    tr.config, tr.num_attention_heads, tr.attention_head_dim = WanCfg(), 40, 128
    pipe = Pipe(tr)
    pipe.vae_scale_factor_temporal, pipe.vae_scale_factor_spatial = 4, 8
    call = dict(ref["call"])
    cls = replace_wan_attention(pipe, call.pop("height"), call.pop("width"), call.pop("num_frames"), call.pop("first_layers_fp"),
                                call.pop("first_times_fp"), **call)
    _check_config(cls, ref["config"], "wan_720p_svg_flashinfer")
    got = [[type(b.attn1.processor).__name__, b.attn1.processor.layer_idx, getattr(b.attn1.processor, "num_layers", None)] for b in blocks]
    assert got == ref["blocks"]
    (fi,) = ref["recorded"]["prepare_flashinfer_attention"]
    assert fi[5:9] == [0, 0, cls.num_frame, cls.frame_size] and fi[9] == fi[10]
    assert cls.temporal_mask_metadata is not None
