#!/usr/bin/env python3
"""Pins the class-level configuration the reference's INSTALL HOOKS compute — `replace_hyvideo_attention`, `replace_wan_attention`,
`replace_cog_attention`, `replace_cosmos_attention` (svg/models/{hyvideo,wan,cog,cosmos}/inference.py) — by EXECUTING them in the build container on duck-typed pipelines at
the geometries of the reference's own scripts (scripts/{hyvideo,wan,cog}/*.sh) and a few odd ones.  Three names inside each hook's module
are replaced by recorders, because what they do is out of reach here and not what is being pinned: `get_attention_mask` (a [10000, S]
fp32 tensor built through a 56 GB host tensor at 720p), `prepare_flexattention` (compiles flex_attention on "cuda") and
`replace_sparse_forward` (patches diffusers classes).  What is recorded: every scalar class attribute the hook sets (geometry, profiler
settings, warm-up thresholds, k-means settings), the arguments it hands to `prepare_flexattention` (the band multiplier derived from the
sparsity) and to `get_attention_mask`, and which processor class / layer index lands on which block.

    python tests/golden/make_golden_install.py     ->  tests/golden/install_golden.json  (checked by tests/test_processors_cpu.py against
                                                       the product's hooks of the same names)"""
import json
import sys
import types
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_golden as MG  # noqa: E402


class Attn:
    def __init__(self):
        self.processor = types.SimpleNamespace()      # (the CogVideoX hook first writes layer_idx onto the processor it finds)

    def set_processor(self, p):
        self.processor = p


def hy_pipe(n_double, n_single):
    tr = types.SimpleNamespace(transformer_blocks=[types.SimpleNamespace(attn=Attn()) for _ in range(n_double)],
                               single_transformer_blocks=[types.SimpleNamespace(attn=Attn()) for _ in range(n_single)])
    return types.SimpleNamespace(transformer=tr)


def wan_pipe(n_blocks, heads=40, head_dim=128):
    tr = types.SimpleNamespace(blocks=[types.SimpleNamespace(attn1=Attn(), attn2=Attn()) for _ in range(n_blocks)],
                               config=types.SimpleNamespace(patch_size=(1, 2, 2)), num_attention_heads=heads, attention_head_dim=head_dim)
    return types.SimpleNamespace(transformer=tr, vae_scale_factor_temporal=4, vae_scale_factor_spatial=8, device="cpu")


def cog_pipe(n_blocks):
    blocks = [types.SimpleNamespace(attn1=Attn()) for _ in range(n_blocks)]
    tr = types.SimpleNamespace(transformer_blocks=blocks, named_modules=lambda: [(f"transformer_blocks.{i}.attn1", b.attn1) for i, b in enumerate(blocks)])
    return types.SimpleNamespace(transformer=tr)


SCALARS = (int, float, bool, str, type(None))


def class_config(cls):
    out = {}
    for k in dir(cls):
        if k.startswith("_"):
            continue
        v = getattr(cls, k)
        if isinstance(v, torch.Tensor) and v.numel() == 1:
            v = v.item()
        if isinstance(v, SCALARS):
            out[k] = v
    return out


def main():
    MG.install_stubs()
    MG._stub("diffusers.models.normalization", RMSNorm=type("RMSNorm", (), {}), FP32LayerNorm=type("FP32LayerNorm", (), {}))
    MG._stub("diffusers.models.modeling_outputs", Transformer2DModelOutput=object)
    MG._stub("diffusers.models.transformers")
    for name, classes in (("transformer_wan", ("WanTransformer3DModel", "WanTransformerBlock")),
                          ("transformer_hunyuan_video", ("HunyuanVideoTransformer3DModel", "HunyuanVideoTransformerBlock", "HunyuanVideoSingleTransformerBlock")),
                          ("cogvideox_transformer_3d", ("CogVideoXTransformer3DModel", "CogVideoXBlock")),
                          ("transformer_cosmos", ("CosmosTransformer3DModel", "CosmosTransformerBlock", "Transformer2DModelOutput"))):
        MG._stub("diffusers.models.transformers." + name, **{c: type(c, (), {}) for c in classes})
    MG._stub("diffusers.utils", USE_PEFT_BACKEND=False, scale_lora_layers=lambda *a, **k: None, unscale_lora_layers=lambda *a, **k: None,
             is_torch_version=lambda *a, **k: True, export_to_video=None, load_image=None, is_torchvision_available=lambda: False, logging=types.SimpleNamespace(get_logger=lambda *a, **k: types.SimpleNamespace(warning=print, info=print)))
    if "matplotlib" not in sys.modules:
        try:
            import matplotlib  # noqa: F401
        except ImportError:
            MG._stub("matplotlib", pyplot=types.SimpleNamespace())
            MG._stub("matplotlib.pyplot")
    sys.path.insert(0, MG.REF)
    torch.Tensor.cuda = lambda self, *a, **k: self
    import importlib

    mods = {}
    for m in ("hyvideo", "wan", "cog", "cosmos"):
        try:
            mods[m] = importlib.import_module(f"svg.models.{m}.inference")
        except Exception as e:      # a diffusers name the stubs above do not cover
            print(f"cannot import svg.models.{m}.inference here: {type(e).__name__}: {e}")
            raise

    calls = {}

    def recorders(mod, tag):
        rec = calls.setdefault(tag, {"get_attention_mask": [], "prepare_flexattention": [], "replace_sparse_forward": 0})

        def gam(*a, **k):
            rec["get_attention_mask"].append([x if isinstance(x, SCALARS) else str(x) for x in a] + [f"{kk}={vv}" for kk, vv in k.items()])
            return None

        def pfa(*a, **k):
            rec["prepare_flexattention"].append({"args": [x if isinstance(x, SCALARS) else str(x) for x in a], "kwargs": {kk: (vv if isinstance(vv, SCALARS) else str(vv)) for kk, vv in k.items()}})
            return types.SimpleNamespace(to_string=lambda **k: "BLOCK_MASK")

        def rsf(*a, **k):
            rec["replace_sparse_forward"] += 1

        mod.get_attention_mask, mod.prepare_flexattention, mod.replace_sparse_forward = gam, pfa, rsf
        if hasattr(mod, "prepare_flashinfer_attention"):
            rec["prepare_flashinfer_attention"] = []
            mod.prepare_flashinfer_attention = lambda *a, **k: (rec["prepare_flashinfer_attention"].append([x if isinstance(x, SCALARS) else str(x) for x in a]), ("indptr", "indices", "bsz"))[1]
            mod.visualize_sparse_bsr = lambda *a, **k: "(bsr picture)"
        return rec

    out = {}

    def blocks_of(pipe, kind):
        tr = pipe.transformer
        if kind == "hy":
            mods_ = [b.attn for b in tr.transformer_blocks] + [b.attn for b in tr.single_transformer_blocks]
        elif kind == "wan":
            mods_ = [b.attn1 for b in tr.blocks]
        else:
            mods_ = [b.attn1 for b in tr.transformer_blocks]
        return [[type(a.processor).__name__, getattr(a.processor, "layer_idx", None), getattr(a.processor, "num_layers", None)] for a in mods_]

    # ---- HunyuanVideo: scripts/hyvideo/*_svg.sh / *_sap.sh (720p and 480p, 129 frames) and an odd geometry ----
    hy = mods["hyvideo"]
    for tag, (h, w, nf, L, lfp, tfp, kw) in {
        "hy_720p_svg": (720, 1280, 129, 77, 1, 901.0, dict(pattern="SVG", num_sampled_rows=64, sample_mse_max_row=10000, sparsity=0.25)),
        "hy_480p_svg": (480, 848, 129, 23, 0, 1001, dict(pattern="SVG", num_sampled_rows=32, sparsity=0.3)),
        "hy_odd_svg": (544, 960, 61, 256, 2, 500.0, dict(pattern="SVG", sparsity=0.15, sample_mse_max_row=5000)),
        "hy_720p_sap": (720, 1280, 129, 77, 1, 901.0, dict(pattern="SAP", num_q_centroids=400, num_k_centroids=1000, top_p_kmeans=0.9, min_kc_ratio=0.1,
                                                           kmeans_iter_init=50, kmeans_iter_step=2, zero_step_kmeans_init=True)),
    }.items():
        rec = recorders(hy, tag)
        pipe = hy_pipe(3, 4)
        hy.replace_hyvideo_attention(pipe, h, w, nf, L, lfp, tfp, **kw)
        cls = hy.Hunyuan_SVGAttn_Processor2_0 if kw["pattern"] == "SVG" else hy.Hunyuan_SAPAttn_Processor2_0
        out[tag] = {"call": dict(height=h, width=w, num_frames=nf, prompt_length=L, first_layers_fp=lfp, first_times_fp=tfp, **kw),
                    "config": class_config(cls), "blocks": blocks_of(pipe, "hy"), "recorded": rec}
    pipe = hy_pipe(2, 2)
    hy.replace_hyvideo_flashattention(pipe)
    out["hy_dense"] = {"blocks": blocks_of(pipe, "hy")}

    # ---- Wan 2.1: scripts/wan/*_svg.sh / *_sap.sh (720p T2V 81 frames, 480p I2V) ----
    wan = mods["wan"]
    for tag, (h, w, nf, lfp, tfp, kw) in {
        "wan_720p_svg": (720, 1280, 81, 1, 901.0, dict(pattern="SVG", num_sampled_rows=64, sample_mse_max_row=10000, sparsity=0.3)),
        "wan_480p_svg": (480, 832, 81, 0, 1001, dict(pattern="SVG", sparsity=0.25)),
        "wan_odd_svg": (512, 768, 33, 3, 400.0, dict(pattern="SVG", sparsity=0.4, num_sampled_rows=16, sample_mse_max_row=2000)),
        "wan_720p_sap": (720, 1280, 81, 1, 901.0, dict(pattern="SAP", num_q_centroids=300, num_k_centroids=1000, top_p_kmeans=0.9, min_kc_ratio=0.1,
                                                       kmeans_iter_init=50, kmeans_iter_step=2, zero_step_kmeans_init=True)),
    }.items():
        rec = recorders(wan, tag)
        pipe = wan_pipe(5)
        wan.replace_wan_attention(pipe, h, w, nf, lfp, tfp, **kw)
        cls = wan.WanAttn_SVGAttn_Processor2_0 if kw["pattern"] == "SVG" else wan.WanAttn_SAPAttn_Processor
        out[tag] = {"call": dict(height=h, width=w, num_frames=nf, first_layers_fp=lfp, first_times_fp=tfp, **kw),
                    "config": class_config(cls), "blocks": blocks_of(pipe, "wan"), "recorded": rec}

    rec = recorders(wan, "wan_720p_svg_flashinfer")
    pipe = wan_pipe(3)
    kw = dict(pattern="SVG", attention_backend="flashinfer", num_sampled_rows=64, sparsity=0.3)
    wan.replace_wan_attention(pipe, 720, 1280, 81, 1, 901.0, **kw)
    out["wan_720p_svg_flashinfer"] = {"call": dict(height=720, width=1280, num_frames=81, first_layers_fp=1, first_times_fp=901.0, **kw),
                                      "config": class_config(wan.WanAttn_SVGAttn_Processor2_0), "blocks": blocks_of(pipe, "wan"), "recorded": rec}

    # ---- Cosmos: scripts/cosmos/cosmos_t2v_{svg,sap}.sh (704 x 1280, 121 frames) ----
    cos = mods["cosmos"]
    for tag, (h, w, nf, lfp, tfp, kw) in {
        "cosmos_svg": (704, 1280, 121, 1, 700.0, dict(pattern="SVG", num_sampled_rows=64, sparsity=0.25)),
        "cosmos_sap": (704, 1280, 121, 1, 700.0, dict(pattern="SAP", num_q_centroids=200, num_k_centroids=800, top_p_kmeans=0.9, min_kc_ratio=0.1,
                                                      kmeans_iter_init=50, kmeans_iter_step=2, zero_step_kmeans_init=True)),
    }.items():
        rec = recorders(cos, tag)
        blocks = [types.SimpleNamespace(attn1=Attn(), attn2=Attn()) for _ in range(4)]
        tr = types.SimpleNamespace(transformer_blocks=blocks, config=types.SimpleNamespace(patch_size=(1, 2, 2)), num_attention_heads=40, attention_head_dim=128)
        pipe = types.SimpleNamespace(transformer=tr, vae_scale_factor_temporal=8, vae_scale_factor_spatial=8, device="cpu")
        cos.replace_cosmos_attention(pipe, h, w, nf, lfp, tfp, **kw)
        cls = cos.Cosmos_SVG_AttnProcessor2_0 if kw["pattern"] == "SVG" else cos.Cosmos_SAPAttn_Processor
        out[tag] = {"call": dict(height=h, width=w, num_frames=nf, first_layers_fp=lfp, first_times_fp=tfp, **kw), "config": class_config(cls),
                    "blocks": [[type(b.attn1.processor).__name__, getattr(b.attn1.processor, "layer_idx", None), getattr(b.attn1.processor, "num_layers", None)] for b in blocks],
                    "recorded": rec}

    # ---- CogVideoX: scripts/cog/cog_inference.sh (v1 and v1.5) ----
    cog = mods["cog"]
    cog.Attention = Attn          # `isinstance(m, Attention)` over named_modules() (cog/inference.py:66-69)
    for tag, (ver, rows, sp, lfp, tfp) in {"cog_v1": ("v1", 32, 0.25, 0.025, 0.075), "cog_v15": ("v1.5", 64, 0.3, 0.0, 0.2)}.items():
        rec = recorders(cog, tag)
        pipe = cog_pipe(4)
        cog.replace_cog_attention(pipe, ver, rows, sp, lfp, tfp)
        out[tag] = {"call": dict(version=ver, num_sampled_rows=rows, sparsity=sp, first_layers_fp=lfp, first_times_fp=tfp),
                    "config": class_config(cog.CogVideoX_SparseAttn_Processor2_0), "blocks": blocks_of(pipe, "cog"), "recorded": rec}

    p = HERE / "install_golden.json"
    p.write_text(json.dumps(out, indent=1, sort_keys=True))
    for tag, d in out.items():
        cfg = d.get("config", {})
        pf = d.get("recorded", {}).get("prepare_flexattention", [])
        print(tag, {k: cfg[k] for k in ("context_length", "num_frame", "frame_size", "first_layers_fp", "first_times_fp") if k in cfg},
              "flex args:", (pf[0]["args"][5:] + list(pf[0]["kwargs"].values())) if pf else None, "blocks:", d["blocks"][:2], "...")
    print(f"wrote {p} ({p.stat().st_size / 1024:.0f} KB)")


if __name__ == "__main__":
    main()
