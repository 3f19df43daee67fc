#!/usr/bin/env python3
"""Measured denoise step (BASELINE.json configs[3] at N = 1): a synthetic HunyuanVideo 720p / 129-frame transformer forward —
20 double-stream + 40 single-stream blocks (the layer counts of the upstream checkpoint; hidden 3072, 24 heads x 128,
MLP 12288) with random-initialised bf16 weights — run layer by layer on one MI355X with every tensor at its real shape
(118800 video + 256 text tokens).  The block structure follows the forward the reference patches
(ref: svg/models/hyvideo/custom_models.py:134-256, double blocks :16-131 and the single-block processor): per block

    LayerNorm + modulate  ->  q / k / v projections  ->  QK RMSNorm + RoPE + head-major transpose  ->  attention
      -> output projection, gate * x + residual  ->  LayerNorm + modulate  ->  MLP (GELU-tanh)  ->  gate * x + residual

GEMMs are torch.mm (hipBLASLt); norm / modulate / gate-residual, the fused QK-norm + RoPE + transpose and the attention are this
repo's HIP kernels (libsvgattn).  Attention per layer exactly as the SVG processors run it (hyvideo/attention.py:491-524): dense for
the first `first_layers_fp` * 60 = 1 layer, sparse (online profiler + band attention with fused head placement) for the other 59;
a warm-up step (the first `first_times_fp` * 50 = 5 of 50 steps, scripts/hyvideo/hyvideo_t2v_720p_svg.sh:4-7) is dense in all layers.

Not modelled (outside the transformer blocks, < 1 % of the FLOPs): patch embedding, time / text embedders, the final layer, the
scheduler update, text encoder and VAE.  Simplification: the text stream uses the image stream's QK-norm weights (one fused kernel
call over the concatenated sequence); modulation vectors are random constants instead of Linear(SiLU(vec)).

    python bench_step.py [--steps K] [--warmup W] [--layers-double 20] [--layers-single 40]
prints one JSON line: seconds per sparse / dense denoise step, the attention share, denoise steps per second for a sparse step and
averaged over a 50-step video (5 dense + 45 sparse steps), GEMM TFLOP/s.
"""
from __future__ import annotations

import argparse
import json
import math
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))

import torch  # noqa: E402

HID, HEADS, HD, MLP = 3072, 24, 128, 12288
F_, P_, CTX, L = 33, 3600, 256, 64
V = F_ * P_
S = V + CTX


def _w(out_f, in_f, dev, gen):
    return (torch.randn(out_f, in_f, device=dev, dtype=torch.float32, generator=gen) * (1.0 / math.sqrt(in_f))).to(torch.bfloat16)


class Stack:
    def __init__(self, n_double, n_single, dev):
        g = torch.Generator(device=dev).manual_seed(0)
        self.dev = dev

        def stream():
            return dict(wq=_w(HID, HID, dev, g), wk=_w(HID, HID, dev, g), wv=_w(HID, HID, dev, g), wo=_w(HID, HID, dev, g),
                        w1=_w(MLP, HID, dev, g), w2=_w(HID, MLP, dev, g),
                        mod=[torch.randn(HID, device=dev, generator=g) * 0.1 for _ in range(6)])   # shift1 scale1 gate1 shift2 scale2 gate2

        self.double = [dict(img=stream(), txt=stream()) for _ in range(n_double)]
        self.single = [dict(wq=_w(HID, HID, dev, g), wk=_w(HID, HID, dev, g), wv=_w(HID, HID, dev, g), wm=_w(MLP, HID, dev, g),
                            w2a=_w(HID, HID, dev, g), w2b=_w(HID, MLP, dev, g),
                            mod=[torch.randn(HID, device=dev, generator=g) * 0.1 for _ in range(3)]) for _ in range(n_single)]
        self.qn = torch.ones(HD, device=dev, dtype=torch.bfloat16)
        self.kn = torch.ones(HD, device=dev, dtype=torch.bfloat16)
        pos = torch.arange(V, device=dev, dtype=torch.float32)[:, None]
        inv = torch.exp(-torch.arange(0, HD, 2, device=dev, dtype=torch.float32) / HD * math.log(10000.0))[None]
        ang = torch.cat([pos * inv, pos * inv], dim=1)                       # [V, HD]
        self.cos, self.sin = ang.cos().contiguous(), ang.sin().contiguous()
        self.gemm_flops = 0.0


def run_step(st: Stack, img, txt, sparse_step: bool, first_layers_fp: int, attn_events, nat, core, geo, mask, prof):
    """one transformer forward; attention launches are bracketed by events collected in attn_events"""
    q_buf, k_buf, v_buf = (torch.empty(1, S, HID, device=st.dev, dtype=torch.bfloat16) for _ in range(3))
    layer = 0

    def proj(x_img, x_txt, wi, wt):
        for buf, key in ((q_buf, "wq"), (k_buf, "wk"), (v_buf, "wv")):
            torch.mm(x_img, wi[key].t(), out=buf[0, :V])
            torch.mm(x_txt, wt[key].t(), out=buf[0, V:])
        st.gemm_flops += 2.0 * S * HID * HID * 3

    def attention():
        nonlocal layer
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        # (like Hunyuan_SVGAttn_Processor2_0: the fused prologue hands the attention core a q that carries the softmax scale)
        q, k = nat.qk_norm_rope_transpose(q_buf, k_buf, HEADS, HEADS, 1, st.qn, None, st.kn, None, 1e-6, 1, st.cos, st.sin, 0, V,
                                          q_scale=nat.softmax_q_scale(HD))
        v, _ = nat.qk_norm_rope_transpose(v_buf, None, HEADS, 0)
        if sparse_step and layer >= first_layers_fp:
            o, _ = core.svg1_sparse_attention(q, k, v, geo, mask, prof, 64, min(10000, V), q_prescaled=True)
        else:
            o = core.dense_attention(q, k, v, valid_len=V + L, q_prescaled=True)
        o = o.transpose(1, 2).reshape(S, HID)      # head-major -> token-major for the output projection (one copy)
        e1.record()
        attn_events.append((e0, e1))
        layer += 1
        return o

    def mlp(x, w1, w2):
        h = torch.nn.functional.gelu(torch.mm(x, w1.t()), approximate="tanh")
        st.gemm_flops += 2.0 * x.shape[0] * HID * MLP * 2
        return torch.mm(h, w2.t())

    for blk in st.double:
        bi, bt = blk["img"], blk["txt"]
        xi = nat.layernorm_modulate_forward(img, scale=bi["mod"][1], shift=bi["mod"][0], eps=1e-6)
        xt = nat.layernorm_modulate_forward(txt, scale=bt["mod"][1], shift=bt["mod"][0], eps=1e-6)
        proj(xi, xt, bi, bt)
        o = attention()
        img = nat.modulate_gate_residual_forward(img, torch.mm(o[:V], bi["wo"].t()), bi["mod"][2], out_dtype=torch.bfloat16)
        txt = nat.modulate_gate_residual_forward(txt, torch.mm(o[V:], bt["wo"].t()), bt["mod"][2], out_dtype=torch.bfloat16)
        st.gemm_flops += 2.0 * S * HID * HID
        xi = nat.layernorm_modulate_forward(img, scale=bi["mod"][4], shift=bi["mod"][3], eps=1e-6)
        xt = nat.layernorm_modulate_forward(txt, scale=bt["mod"][4], shift=bt["mod"][3], eps=1e-6)
        img = nat.modulate_gate_residual_forward(img, mlp(xi, bi["w1"], bi["w2"]), bi["mod"][5], out_dtype=torch.bfloat16)
        txt = nat.modulate_gate_residual_forward(txt, mlp(xt, bt["w1"], bt["w2"]), bt["mod"][5], out_dtype=torch.bfloat16)
    x = torch.cat([img, txt], dim=0)
    for blk in st.single:
        xm = nat.layernorm_modulate_forward(x, scale=blk["mod"][1], shift=blk["mod"][0], eps=1e-6)
        proj(xm[:V], xm[V:], blk, blk)
        h = torch.nn.functional.gelu(torch.mm(xm, blk["wm"].t()), approximate="tanh")
        o = attention()
        out = torch.mm(o, blk["w2a"].t())
        out.addmm_(h, blk["w2b"].t())                # linear2 over cat([attn, mlp]) without materialising the concatenation
        st.gemm_flops += 2.0 * S * HID * (MLP + HID + MLP)
        x = nat.modulate_gate_residual_forward(x, out, blk["mod"][2], out_dtype=torch.bfloat16)
    return x


def measure(steps: int = 1, warmup: int = 1, n_double: int = 20, n_single: int = 40):
    from svg import _native as nat
    from svg.models import _core as core
    from svg.models.hyvideo.utils import sparsity_to_width

    nat.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    st = Stack(n_double, n_single, dev)
    n_layers = n_double + n_single
    first_layers_fp = math.floor(0.03 * n_layers)                 # scripts/hyvideo/hyvideo_t2v_720p_svg.sh:5, hyvideo_t2v_inference.py:95
    width = sparsity_to_width(0.25, CTX, F_, P_)
    tf = math.floor(width * P_ / 128) * 128
    mask = nat.BandMask(real_len=V + L, band=tf, colfull_lo=V, colfull_hi=V + L, rowfull_lo=V, rowfull_hi=V + L)
    geo = core.Geometry(CTX, F_, P_)
    bb = int((P_ * 1.5) // 128)
    prof = nat.ProfileDesc(0, F_, P_, 1)
    prof.variant[0] = nat.ProfileVariant(0, 0, V, bb, 0, V, S)
    prof.variant[1] = nat.ProfileVariant(1, 0, V, bb, 0, V, S)
    g = torch.Generator(device=dev).manual_seed(1)
    img = (torch.randn(V, HID, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    txt = (torch.randn(CTX, HID, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    res = {}
    for kind, sparse_step in (("sparse", True), ("dense", False), ("sparse_fp8", True)):
        core.set_attention_dtype("fp8" if kind == "sparse_fp8" else "bf16")   # fp8: e4m3 QK^T / PV in the 59 sparse layers
        times, attn_ms = [], []
        for it in range(warmup + steps):
            ev = []
            st.gemm_flops = 0.0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            out = run_step(st, img, txt, sparse_step, first_layers_fp, ev, nat, core, geo, mask, prof)
            e1.record()
            torch.cuda.synchronize()
            assert torch.isfinite(out.float()).all()
            if it >= warmup:
                times.append(e0.elapsed_time(e1))
                attn_ms.append(sum(a.elapsed_time(b) for a, b in ev))
        order = sorted(range(len(times)), key=lambda i: times[i])
        mid = order[len(order) // 2]            # the median step (and ITS attention share)
        t, a = times[mid], attn_ms[mid]
        res[kind] = {"ms": round(t, 2), "ms_all_steps": [round(x, 1) for x in times], "attention_ms": round(a, 2), "attention_share": round(a / t, 4),
                     "gemm_and_glue_ms": round(t - a, 2), "gemm_tflop": round(st.gemm_flops / 1e12, 1),
                     "gemm_tflops_lower_bound": round(st.gemm_flops / ((t - a) * 1e-3) / 1e12, 1)}
    core.set_attention_dtype("bf16")
    ts, td = res["sparse"]["ms"] * 1e-3, res["dense"]["ms"] * 1e-3
    video = (5 * td + 45 * ts) / 50
    return {
        "metric": "denoise_step_hy720p",
        "workload": f"synthetic HunyuanVideo 720p/129f transformer forward: {n_double} double + {n_single} single blocks, hidden {HID}, "
                    f"{HEADS} x {HD} heads, MLP {MLP}, S = {S} ({V} video + {CTX} text tokens, prompt {L}), bf16, random weights; "
                    f"sparse step = {first_layers_fp} dense + {n_layers - first_layers_fp} SVG1 layers (sparsity 0.25, band {tf})",
        "steps": steps, "warmup": warmup,
        "sparse_step": res["sparse"], "dense_step": res["dense"], "sparse_step_fp8_attention": res["sparse_fp8"],
        "denoise_steps_per_s_fp8_attention": round(1e3 / res["sparse_fp8"]["ms"], 4),
        "denoise_steps_per_s": round(1.0 / ts, 4),
        "denoise_steps_per_s_dense": round(1.0 / td, 4),
        "denoise_steps_per_s_video_average": round(1.0 / video, 4),
        "speedup_sparse_vs_dense_step": round(td / ts, 3),
        "not_modelled": "patch / time / text embedders, final layer, scheduler, text encoder, VAE",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--layers-double", type=int, default=20)
    ap.add_argument("--layers-single", type=int, default=40)
    a = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("bench_step.py needs an MI355X (no GPU visible); the HIP path has no CPU fallback")
    print(json.dumps(measure(a.steps, a.warmup, a.layers_double, a.layers_single)))


if __name__ == "__main__":
    main()
