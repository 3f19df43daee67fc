#!/usr/bin/env python3
"""Measured denoise step (BASELINE.json configs[3]): a synthetic HunyuanVideo 720p / 129-frame transformer forward — 20
double-stream + 40 single-stream blocks (the layer counts of the upstream checkpoint; hidden 3072, 24 heads x 128,
MLP 12288) with random-initialised bf16 weights — run layer by layer with every tensor at its real shape (118800 video + 256 text
tokens), on 1 GPU or token-sharded over the N GPUs of one node.  The block structure follows the forward the reference patches
(ref: svg/models/hyvideo/custom_models.py:134-256, double blocks :16-131 and the single-block processor): per block

    LayerNorm + modulate  ->  q / k / v projections  ->  QK RMSNorm + RoPE + head-major transpose  ->  attention
      -> output projection, gate * x + residual  ->  LayerNorm + modulate  ->  MLP (GELU-tanh)  ->  gate * x + residual

GEMMs are torch.mm (hipBLASLt); norm / modulate / gate-residual, the fused QK-norm + RoPE + transpose and the attention are this
repo's HIP kernels (libsvgattn).  Attention per layer exactly as the SVG processors run it (hyvideo/attention.py:491-524): dense for
the first `first_layers_fp` * 60 = 1 layer, sparse (online profiler + band attention with fused head placement) for the other 59;
a warm-up step (the first `first_times_fp` * 50 = 5 of 50 steps, scripts/hyvideo/hyvideo_t2v_720p_svg.sh:4-7) is dense in all layers.

N > 1 (the shape of the hooks: svg/models/wan_orig/distributed/xdit_context_parallel.py:120,129): hidden states live
TOKEN-sharded in units of 128 tokens (`svg.distributed.token_range(unit=StepGeo.unit)`: 14976 / 14848 tokens per rank at N = 8, largest /
mean = 1.006 — whole frames would be 5 / 4 frames, 1.21; the text tokens are simply the tail of the last rank's range), so norms, projections, the
fused prologue, output projection, MLP and glue run on the local tokens only; around the attention of every layer the q, k, v of the
local tokens are exchanged for this rank's heads over the full sequence (`tokens_to_heads`: 3 x all_to_all_single, every peer to every
peer directly over xGMI), the head-sharded SVG1 attention runs unchanged, and `heads_to_tokens` brings the output back to token
shards for `to_out`.  One all-gather of the final hidden states per step ("the RCCL all-gather of latents").  Every staging buffer
of the exchanges is allocated once (`svg.distributed.ExchangeBuffers`); `rccl_bytes_per_step` = bytes this rank RECEIVES from peers.

Not modelled (outside the transformer blocks, < 1 % of the FLOPs): patch embedding, time / text embedders, the final layer, the
scheduler update, text encoder and VAE.  Simplification: the text stream uses the image stream's QK-norm weights (one fused kernel
call over the concatenated sequence); modulation vectors are random constants instead of Linear(SiLU(vec)).

    python bench_step.py [--steps K] [--warmup W] [--layers-double 20] [--layers-single 40]           (one GPU)
    python -m torch.distributed.run --nproc-per-node N ... bench_step.py --gpus N                     (one rank per GPU, RCCL)
prints one JSON line: seconds per sparse / dense denoise step, the attention share, denoise steps per second for a sparse step and
averaged over a 50-step video (5 dense + 45 sparse steps), GEMM TFLOP/s, and for N > 1 the exchange volume.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
from dataclasses import dataclass
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))

import torch  # noqa: E402


@dataclass(frozen=True)
class StepGeo:
    """geometry of the synthetic stack (default: HunyuanVideo 720p / 129 frames)"""
    F: int = 33
    P: int = 3600
    ctx: int = 256
    L: int = 64
    hid: int = 3072
    heads: int = 24
    hd: int = 128
    mlp: int = 12288
    unit: int = 128      # granularity of the token shards at N > 1 (svg.distributed.token_range): S = 119056 over 8 ranks -> 14976 / 14848 (+16) tokens

    @property
    def V(self):
        return self.F * self.P

    @property
    def S(self):
        return self.V + self.ctx


HY720P = StepGeo()
HID, HEADS, HD, MLP = HY720P.hid, HY720P.heads, HY720P.hd, HY720P.mlp   # (names other tools import)
F_, P_, CTX, L = HY720P.F, HY720P.P, HY720P.ctx, HY720P.L
V, S = HY720P.V, HY720P.S


def _w(out_f, in_f, dev, gen, dtype):
    return (torch.randn(out_f, in_f, device=dev, dtype=torch.float32, generator=gen) * (1.0 / math.sqrt(in_f))).to(dtype)


class Stack:
    """random weights of the synthetic stack (every rank builds the same ones: same seed)"""

    def __init__(self, n_double, n_single, dev, geo: StepGeo = HY720P, dtype=torch.bfloat16):
        g = torch.Generator(device=dev).manual_seed(0)
        self.dev, self.geo, self.dtype = dev, geo, dtype
        hid, mlp, hd = geo.hid, geo.mlp, geo.hd

        def stream():
            return dict(wq=_w(hid, hid, dev, g, dtype), wk=_w(hid, hid, dev, g, dtype), wv=_w(hid, hid, dev, g, dtype),
                        wo=_w(hid, hid, dev, g, dtype), w1=_w(mlp, hid, dev, g, dtype), w2=_w(hid, mlp, dev, g, dtype),
                        mod=[torch.randn(hid, device=dev, generator=g) * 0.1 for _ in range(6)])   # shift1 scale1 gate1 shift2 scale2 gate2

        self.double = [dict(img=stream(), txt=stream()) for _ in range(n_double)]
        self.single = [dict(wq=_w(hid, hid, dev, g, dtype), wk=_w(hid, hid, dev, g, dtype), wv=_w(hid, hid, dev, g, dtype),
                            wm=_w(mlp, hid, dev, g, dtype), w2a=_w(hid, hid, dev, g, dtype), w2b=_w(hid, mlp, dev, g, dtype),
                            mod=[torch.randn(hid, device=dev, generator=g) * 0.1 for _ in range(3)]) for _ in range(n_single)]
        self.qn = torch.ones(hd, device=dev, dtype=dtype)
        self.kn = torch.ones(hd, device=dev, dtype=dtype)
        pos = torch.arange(geo.V, device=dev, dtype=torch.float32)[:, None]
        inv = torch.exp(-torch.arange(0, hd, 2, device=dev, dtype=torch.float32) / hd * math.log(10000.0))[None]
        ang = torch.cat([pos * inv, pos * inv], dim=1)                       # [V, hd]
        self.cos, self.sin = ang.cos().contiguous(), ang.sin().contiguous()
        self.gemm_flops = 0.0


class HipOps:
    """The product ops of a step: libsvgattn kernels through svg._native / svg.models._core (GPU only).  tests/ supplies a torch
    statement of the same interface to check the sharding logic of run_step on CPU (gloo) — never used by the benchmark."""

    def __init__(self, geo: StepGeo, first_layers_fp: int):
        from svg import _native as nat
        from svg.models import _core as core
        from svg.models.hyvideo.utils import sparsity_to_width

        nat.load()
        self.nat, self.core, self.geo, self.first_layers_fp = nat, core, geo, first_layers_fp
        width = sparsity_to_width(0.25, geo.ctx, geo.F, geo.P)
        self.band = math.floor(width * geo.P / 128) * 128
        Vv, Ll = geo.V, geo.L
        self.mask = nat.BandMask(real_len=Vv + Ll, band=self.band, colfull_lo=Vv, colfull_hi=Vv + Ll, rowfull_lo=Vv, rowfull_hi=Vv + Ll)
        self.cgeo = core.Geometry(geo.ctx, geo.F, geo.P)
        bb = int((geo.P * 1.5) // 128)
        self.prof = nat.ProfileDesc(0, geo.F, geo.P, 1)
        self.prof.variant[0] = nat.ProfileVariant(0, 0, Vv, bb, 0, Vv, geo.S)
        self.prof.variant[1] = nat.ProfileVariant(1, 0, Vv, bb, 0, Vv, geo.S)
        # like Hunyuan_SVGAttn_Processor2_0 (prescale_q False by default since round 4: the reference's formulation — scale applied to the
        # fp32 scores; SVG_STEP_PRESCALE=1: the opt-in pre-scaled path, q leaves the prologue carrying the softmax scale)
        self.prescale = bool(int(os.environ.get("SVG_STEP_PRESCALE", "0") or 0)) and geo.hd == 128
        self.q_scale = nat.softmax_q_scale(geo.hd) if self.prescale else 1.0

    def ln_mod(self, x, scale, shift):
        return self.nat.layernorm_modulate_forward(x, scale=scale, shift=shift, eps=1e-6)

    def gate_res(self, res, x, gate):
        return self.nat.modulate_gate_residual_forward(res, x, gate, out_dtype=res.dtype)

    def prologue(self, st, q_buf, k_buf, v_buf, pos0, n_rot):
        """projection outputs [1, S_r, hid] of the tokens at positions pos0 .. -> head-major q, k, v [1, H, S_r, hd]; the first n_rot
        tokens are video tokens at positions pos0 + i (rotated), the rest text"""
        g = self.geo
        q, k = self.nat.qk_norm_rope_transpose(q_buf, k_buf, g.heads, g.heads, 1, st.qn, None, st.kn, None, 1e-6, 1 if n_rot else 0,
                                               st.cos[pos0:pos0 + n_rot].contiguous() if n_rot else None,
                                               st.sin[pos0:pos0 + n_rot].contiguous() if n_rot else None, 0, n_rot, q_scale=self.q_scale)
        v, _ = self.nat.qk_norm_rope_transpose(v_buf, None, g.heads, 0)
        return q, k, v

    def attention(self, q, k, v, sparse: bool):
        """q, k, v [1, H_local, S, hd] -> o [1, H_local, S, hd]"""
        g = self.geo
        if sparse:
            return self.core.svg1_sparse_attention(q, k, v, self.cgeo, self.mask, self.prof, 64, min(10000, g.V), q_prescaled=self.prescale)[0]
        return self.core.dense_attention(q, k, v, valid_len=g.V + g.L, q_prescaled=self.prescale)

    def gelu(self, x):
        return torch.nn.functional.gelu(x, approximate="tanh")


class Sharding:
    """token / head sharding of one rank (world = 1: everything local, no exchange)"""

    def __init__(self, geo: StepGeo, rank: int, world: int, dev, dtype, group=None, host_staged: bool = False):
        from svg import distributed as D

        self.D, self.geo, self.rank, self.world, self.group = D, geo, rank, world, group
        self.a, self.b = D.token_range(geo.S, rank, world, unit=geo.unit) if world > 1 else (0, geo.S)
        self.nv = max(0, min(self.b, geo.V) - self.a)     # local video tokens: positions a .. a + nv
        self.nt = (self.b - self.a) - self.nv             # local text tokens: text positions t0 .. t0 + nt
        self.t0 = max(self.a, geo.V) - geo.V
        self.ranges = [D.token_range(geo.S, r, world, unit=geo.unit) for r in range(world)] if world > 1 else [(0, geo.S)]
        self.host_staged = host_staged                    # gloo smoke run on one GPU: the exchanges go through host memory
        self.buf = None
        if world > 1:
            assert geo.heads % world == 0, f"{geo.heads} heads over {world} ranks"
            self.buf = D.ExchangeBuffers(geo.heads, geo.S, geo.hd, dtype, torch.device("cpu") if host_staged else dev, group, unit=geo.unit)
        self.dev = dev

    def to_heads(self, x, which):   # [H, S_r, hd] -> [H_local, S, hd]
        if self.host_staged:
            return self.buf.tokens_to_heads(x.cpu(), which).to(self.dev)
        return self.buf.tokens_to_heads(x, which)

    def to_tokens(self, o):         # [H_local, S, hd] -> [H, S_r, hd]
        if self.host_staged:
            return self.buf.heads_to_tokens(o.cpu()).to(self.dev)
        return self.buf.heads_to_tokens(o)

    def gather_tokens(self, x):
        """the per-step all-gather of the final hidden states [S_r, hid] -> [S, hid] (ragged token shards: padded to the largest)"""
        import torch.distributed as dist

        tr = [self.D.token_range(self.geo.S, r, self.world, unit=self.geo.unit) for r in range(self.world)]
        mx = max(b - a for a, b in tr)
        src = x.cpu() if self.host_staged else x
        pad = torch.zeros((mx, x.shape[1]), dtype=x.dtype, device=src.device)
        pad[: x.shape[0]] = src
        out = torch.empty((self.world * mx, x.shape[1]), dtype=x.dtype, device=src.device)
        dist.all_gather_into_tensor(out, pad, group=self.group)
        full = torch.cat([out[r * mx: r * mx + (b - a)] for r, (a, b) in enumerate(tr)], dim=0)
        return full.to(self.dev), (self.world - 1) * mx * x.shape[1] * x.element_size()


def run_step(st: Stack, img, txt, sparse_step: bool, first_layers_fp: int, attn_events, ops, sh: Sharding = None, events=True):
    """one transformer forward on this rank's token shard (img [nv, hid], txt [nt, hid]); attention (+ its exchanges) is bracketed by
    events collected in attn_events.  Returns the final hidden states of the local tokens [S_r, hid]."""
    geo = st.geo
    hid, mlp_dim = geo.hid, geo.mlp
    nv, nt = img.shape[0], txt.shape[0]
    Sr = nv + nt
    pos0 = sh.a if sh is not None else 0
    sharded = sh is not None and sh.world > 1
    q_buf, k_buf, v_buf = (torch.empty(1, Sr, hid, device=img.device, dtype=img.dtype) for _ in range(3))
    layer = 0

    def proj(x_img, x_txt, wi, wt):
        for buf, key in ((q_buf, "wq"), (k_buf, "wk"), (v_buf, "wv")):
            if nv:
                torch.mm(x_img, wi[key].t(), out=buf[0, :nv])
            if nt:
                torch.mm(x_txt, wt[key].t(), out=buf[0, nv:])
        st.gemm_flops += 2.0 * Sr * hid * hid * 3

    def attention():
        nonlocal layer
        e0 = e1 = None
        if events:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        q, k, v = ops.prologue(st, q_buf, k_buf, v_buf, pos0, nv)           # [1, H, S_r, hd]
        if sharded:
            q, k, v = (sh.to_heads(x[0], i)[None] for i, x in enumerate((q, k, v)))   # [1, H_local, S, hd]
        o = ops.attention(q, k, v, sparse_step and layer >= first_layers_fp)
        if sharded:
            o = sh.to_tokens(o[0])[None]                                     # [1, H, S_r, hd]
        o = o.transpose(1, 2).reshape(Sr, hid)      # head-major -> token-major for the output projection (one copy)
        if events:
            e1.record()
            attn_events.append((e0, e1))
        layer += 1
        return o

    def mlp(x, w1, w2):
        h = ops.gelu(torch.mm(x, w1.t()))
        st.gemm_flops += 2.0 * x.shape[0] * hid * mlp_dim * 2
        return torch.mm(h, w2.t())

    def stream_in(x, m, i_scale, i_shift):
        return ops.ln_mod(x, m[i_scale], m[i_shift]) if x.shape[0] else x

    for blk in st.double:
        bi, bt = blk["img"], blk["txt"]
        xi, xt = stream_in(img, bi["mod"], 1, 0), stream_in(txt, bt["mod"], 1, 0)
        proj(xi, xt, bi, bt)
        o = attention()
        if nv:
            img = ops.gate_res(img, torch.mm(o[:nv], bi["wo"].t()), bi["mod"][2])
            img = ops.gate_res(img, mlp(stream_in(img, bi["mod"], 4, 3), bi["w1"], bi["w2"]), bi["mod"][5])
        if nt:
            txt = ops.gate_res(txt, torch.mm(o[nv:], bt["wo"].t()), bt["mod"][2])
            txt = ops.gate_res(txt, mlp(stream_in(txt, bt["mod"], 4, 3), bt["w1"], bt["w2"]), bt["mod"][5])
        st.gemm_flops += 2.0 * Sr * hid * hid
    x = torch.cat([img, txt], dim=0)
    for blk in st.single:
        xm = ops.ln_mod(x, blk["mod"][1], blk["mod"][0])
        proj(xm[:nv], xm[nv:], blk, blk)
        h = ops.gelu(torch.mm(xm, blk["wm"].t()))
        o = attention()
        out = torch.mm(o, blk["w2a"].t())
        out.addmm_(h, blk["w2b"].t())                # linear2 over cat([attn, mlp]) without materialising the concatenation
        st.gemm_flops += 2.0 * Sr * hid * (mlp_dim + hid + mlp_dim)
        x = ops.gate_res(x, out, blk["mod"][2])
    return x


def measure(steps: int = 1, warmup: int = 1, n_double: int = 20, n_single: int = 40, geo: StepGeo = HY720P, rank: int = 0, world: int = 1,
            group=None, kinds=("sparse", "dense", "sparse_fp8"), host_staged: bool = False):
    from svg.models import _core as core

    dev = torch.device("cuda", torch.cuda.current_device())
    st = Stack(n_double, n_single, dev, geo)
    n_layers = n_double + n_single
    first_layers_fp = math.floor(0.03 * n_layers)                 # scripts/hyvideo/hyvideo_t2v_720p_svg.sh:5, hyvideo_t2v_inference.py:95
    ops = HipOps(geo, first_layers_fp)
    sh = Sharding(geo, rank, world, dev, torch.bfloat16, group, host_staged)
    g = torch.Generator(device=dev).manual_seed(1)
    img_all = (torch.randn(geo.V, geo.hid, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    txt_all = (torch.randn(geo.ctx, geo.hid, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    img = img_all[sh.a: sh.a + sh.nv].contiguous()
    txt = txt_all[sh.t0: sh.t0 + sh.nt].contiguous()
    del img_all, txt_all
    res = {}
    for kind in kinds:
        sparse_step = kind != "dense"
        core.set_attention_dtype("fp8" if kind == "sparse_fp8" else "bf16")   # fp8: e4m3 QK^T / PV in the 59 sparse layers
        times, attn_ms, bytes_step = [], [], 0
        for it in range(warmup + steps):
            ev = []
            st.gemm_flops = 0.0
            if sh.buf is not None:
                sh.buf.bytes_in = sh.buf.bytes_out = 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            if world > 1:
                import torch.distributed as dist

                dist.barrier(group)
            e0.record()
            out = run_step(st, img, txt, sparse_step, first_layers_fp, ev, ops, sh)
            gathered = 0
            if world > 1:
                out, gathered = sh.gather_tokens(out)          # every rank ends the step with all hidden states
            e1.record()
            torch.cuda.synchronize()
            assert torch.isfinite(out.float()).all()
            if it >= warmup:
                times.append(e0.elapsed_time(e1))
                attn_ms.append(sum(a.elapsed_time(b) for a, b in ev))
                if sh.buf is not None:
                    bytes_step = sh.buf.bytes_in + sh.buf.bytes_out + gathered
        if world > 1:   # the step is as slow as its slowest rank
            tt = torch.tensor(times, dtype=torch.float64, device="cpu" if host_staged else dev)
            import torch.distributed as dist

            dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=group)
            times = tt.tolist()
        order = sorted(range(len(times)), key=lambda i: times[i])
        mid = order[len(order) // 2]            # the median step (and ITS attention share)
        t, a = times[mid], attn_ms[mid]
        gf = st.gemm_flops
        res[kind] = {"ms": round(t, 2), "ms_all_steps": [round(x, 1) for x in times], "attention_ms": round(a, 2),
                     "attention_share": round(a / t, 4), "gemm_and_glue_ms": round(t - a, 2), "gemm_tflop_this_rank": round(gf / 1e12, 1),
                     "gemm_tflops_lower_bound_this_rank": round(gf / max((t - a) * 1e-3, 1e-9) / 1e12, 1)}
        if world > 1:
            res[kind]["attention_ms_includes"] = "prologue + 3 x tokens_to_heads + attention + heads_to_tokens + transpose copy"
            res[kind]["rccl_bytes_received_per_step_this_rank"] = int(bytes_step)
    core.set_attention_dtype("bf16")
    out = {
        "metric": "denoise_step_hy720p" if geo == HY720P else "denoise_step_custom",
        "workload": f"synthetic HunyuanVideo-style transformer forward: {n_double} double + {n_single} single blocks, hidden {geo.hid}, "
                    f"{geo.heads} x {geo.hd} heads, MLP {geo.mlp}, S = {geo.S} ({geo.V} video + {geo.ctx} text tokens, prompt {geo.L}), bf16, "
                    f"random weights; sparse step = {first_layers_fp} dense + {n_layers - first_layers_fp} SVG1 layers (sparsity 0.25, band "
                    f"{ops.band}); " + ("q pre-scaled by the fused prologue (SVG_STEP_PRESCALE=1)" if ops.prescale else "plain q, softmax scale on the fp32 scores (the reference's formulation)"),
        "steps": steps, "warmup": warmup, "n_gpus": world,
        "not_modelled": "patch / time / text embedders, final layer, scheduler, text encoder, VAE",
    }
    if world > 1:
        out["parallelism"] = (f"tokens/{world} (units of {geo.unit} tokens, largest / mean shard {max(b - a for a, b in sh.ranges) * world / geo.S:.4f}) for norms / GEMMs / prologue / glue, heads/{world} for the "
                              f"attention; per layer 3 x all_to_all in + 1 x all_to_all out, one all-gather of the hidden states per step; "
                              f"this rank: tokens [{sh.a}, {sh.b})")
        out["exchange_backend"] = "gloo through host memory (SVG_BENCH_SMOKE: control-flow run on one GPU, not a measurement)" if host_staged else "RCCL"
    if "sparse" in res:
        out["sparse_step"] = res["sparse"]
        out["denoise_steps_per_s"] = round(1e3 / res["sparse"]["ms"], 4)
    if "dense" in res:
        out["dense_step"] = res["dense"]
        out["denoise_steps_per_s_dense"] = round(1e3 / res["dense"]["ms"], 4)
    if "sparse_fp8" in res:
        out["sparse_step_fp8_attention"] = res["sparse_fp8"]
        out["denoise_steps_per_s_fp8_attention"] = round(1e3 / res["sparse_fp8"]["ms"], 4)
    if "sparse" in res and "dense" in res:
        ts, td = res["sparse"]["ms"] * 1e-3, res["dense"]["ms"] * 1e-3
        out["denoise_steps_per_s_video_average"] = round(50.0 / (5 * td + 45 * ts), 4)
        out["speedup_sparse_vs_dense_step"] = round(td / ts, 3)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--layers-double", type=int, default=20)
    ap.add_argument("--layers-single", type=int, default=40)
    ap.add_argument("--tiny", action="store_true", help="a small geometry (F = 5, P = 600; hidden 512, 4 heads) for control-flow tests")
    a = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("bench_step.py needs an MI355X (no GPU visible); the HIP path has no CPU fallback")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    smoke = bool(os.environ.get("SVG_BENCH_SMOKE"))   # all ranks on cuda:0 over gloo: control flow only
    if smoke:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if smoke:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    geo = StepGeo(F=5, P=600, ctx=256, L=64, hid=512, heads=4, hd=128, mlp=1024) if a.tiny else HY720P
    kinds = ("sparse", "dense", "sparse_fp8") if world == 1 else ("sparse", "dense")
    out = measure(a.steps, a.warmup, a.layers_double, a.layers_single, geo, rank, world, kinds=kinds, host_staged=smoke and world > 1)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
