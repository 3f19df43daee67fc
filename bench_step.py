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
prints one JSON line: seconds per sparse / dense denoise step, the attention share, denoise steps per second, GEMM TFLOP/s, and for
N > 1 the exchange volume.  `--model wan720p`: the Wan 2.1 14B stack with SVG2 / SAP attention (BASELINE.json configs[2]), below.
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


TUNED_GEMMS = ROOT / "sparse-videogen_amd" / "tuning" / "tunableop_mi355x.csv"
_TUNED = {"loaded": None}


def enable_tuned_gemms():
    """Solution selection for four GEMM shapes of the HunyuanVideo stack where torch's default hipBLASLt pick loses 4 - 27 % on MI355X
    (sparse-videogen_amd/tuning/README.md): a TunableOp results file, read with tuning OFF — shapes without an entry keep the default pick, and
    TunableOp rejects the file on any other library stack (its validators).  SVG_STEP_TUNABLEOP=0: skip."""
    if _TUNED["loaded"] is not None:
        return _TUNED["loaded"]
    ok = False
    if os.environ.get("SVG_STEP_TUNABLEOP", "1") != "0" and TUNED_GEMMS.exists() and hasattr(torch.cuda, "tunable"):
        try:
            torch.cuda.tunable.enable(True)
            torch.cuda.tunable.tuning_enable(False)
            torch.cuda.tunable.record_untuned_enable(False)
            ok = bool(torch.cuda.tunable.read_file(str(TUNED_GEMMS)))
            if not ok:
                torch.cuda.tunable.enable(False)
        except Exception:  # noqa: BLE001
            ok = False
    _TUNED["loaded"] = ok
    return ok


def gemm_backend_info():
    """which BLAS the GEMMs of the step run on, as far as torch tells (the kernels' names are in the committed rocprofv3 trace of a step)"""
    info = {"library": "unknown"}
    info["tunableop_file_loaded"] = bool(_TUNED["loaded"])
    if _TUNED["loaded"]:
        try:
            info["tunableop_solutions"] = {r[1]: r[2] for r in torch.cuda.tunable.get_results()}
        except Exception:  # noqa: BLE001
            pass
    try:
        info["library"] = str(torch.backends.cuda.preferred_blas_library()).split(".")[-1]
    except Exception:  # noqa: BLE001
        pass
    info["TORCH_BLAS_PREFER_HIPBLASLT"] = os.environ.get("TORCH_BLAS_PREFER_HIPBLASLT")
    info["hip"] = getattr(torch.version, "hip", None)
    return info


def gemm_shapes_tflops(shapes, reps=20, dtype=torch.bfloat16):
    """per-shape rate of the step's GEMMs, each alone on the GPU: [(M, N, K)] -> {"MxNxK": TFLOP/s} (y[M, N] = x[M, K] @ w[N, K]^T)"""
    out = {}
    dev = torch.device("cuda", torch.cuda.current_device())
    for M, N, K in shapes:
        x = torch.randn(M, K, device=dev, dtype=dtype)
        w = torch.randn(N, K, device=dev, dtype=dtype)
        y = torch.empty(M, N, device=dev, dtype=dtype)
        for _ in range(5):       # (the first launches of a shape after idle time run at a lower clock: 1085 vs 1290 TFLOP/s measured)
            torch.mm(x, w.t(), out=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            torch.mm(x, w.t(), out=y)
        e1.record()
        torch.cuda.synchronize()
        out[f"{M}x{N}x{K}"] = round(2.0 * M * N * K * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12, 1)
        del x, w, y
    return out


def _w(out_f, in_f, dev, gen, dtype):
    return (torch.randn(out_f, in_f, device=dev, dtype=torch.float32, generator=gen) * (1.0 / math.sqrt(in_f))).to(dtype)


class Sections:
    """HIP-event brackets summed per name: where a step's time goes (`step_breakdown`).  Nested sections are allowed; the caller keeps the
    names disjoint."""

    def __init__(self, on=True):
        self.on, self.ev = on and torch.cuda.is_available(), {}

    class _Ctx:
        def __init__(self, owner, name):
            self.o, self.n = owner, name

        def __enter__(self):
            if self.o.on:
                self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                self.e0.record()

        def __exit__(self, *exc):
            if self.o.on:
                self.e1.record()
                self.o.ev.setdefault(self.n, []).append((self.e0, self.e1))
            return False

    def __call__(self, name):
        return Sections._Ctx(self, name)

    def totals(self):
        return {n: sum(a.elapsed_time(b) for a, b in v) for n, v in self.ev.items()}


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
        core.TOKEN_MAJOR_IO = bool(int(os.environ.get("SVG_STEP_TOKEN_MAJOR_IO", "1") or 1))   # 0: the copies of the reference's processors (A/B)
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
        if self.v_in_place:     # the attention kernels read v where the projection wrote it (svg_attn_layout_t): a view, no transpose copy
            return q, k, v_buf.unflatten(2, (g.heads, -1)).transpose(1, 2)
        v, _ = self.nat.qk_norm_rope_transpose(v_buf, None, g.heads, 0)
        return q, k, v

    v_in_place = False

    def set_sharded(self, sharded: bool):
        """one GPU, head_dim 128: v stays in the projection layout and o comes back token-major (svg.models._core.TOKEN_MAJOR_IO); the
        head-sharded step exchanges contiguous head slices and keeps the copies"""
        self.v_in_place = (not sharded) and self.core.TOKEN_MAJOR_IO and self.geo.hd == 128 and not self.prescale

    def attention(self, q, k, v, sparse: bool):
        """q, k, v [1, H_local, S, hd] -> o [1, H_local, S, hd]"""
        g = self.geo
        if sparse:
            return self.core.svg1_sparse_attention(q, k, v, self.cgeo, self.mask, self.prof, 64, min(10000, g.V), q_prescaled=self.prescale)[0]
        return self.core.dense_attention(q, k, v, valid_len=g.V + g.L, q_prescaled=self.prescale)

    def gelu(self, x):
        return torch.nn.functional.gelu(x, approximate="tanh")

    def mm_gelu(self, x, w):
        """gelu_tanh(x @ w^T) with the activation in the GEMM's epilogue (hipBLASLt through torch._addmm_activation; the model's Linear has a
        bias, here zero) — checked once against the two-pass form; None: not available on this build, the caller runs GEMM + GELU pass"""
        if not hasattr(self, "_fused_gelu"):
            self._zero_bias = {}
            try:
                xs = x[:256]
                zb = torch.zeros(w.shape[0], device=x.device, dtype=x.dtype)
                a = torch._addmm_activation(zb, xs, w.t(), use_gelu=True)
                r = torch.nn.functional.gelu(torch.mm(xs, w.t()), approximate="tanh")
                self._fused_gelu = bool(torch.allclose(a.float(), r.float(), atol=3e-2, rtol=3e-2))
            except Exception:  # noqa: BLE001
                self._fused_gelu = False
        if not self._fused_gelu:
            return None
        zb = self._zero_bias.get(w.shape[0])
        if zb is None:
            zb = self._zero_bias[w.shape[0]] = torch.zeros(w.shape[0], device=x.device, dtype=x.dtype)
        return torch._addmm_activation(zb, x, w.t(), use_gelu=True)


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


def run_step(st: Stack, img, txt, sparse_step: bool, first_layers_fp: int, attn_events, ops, sh: Sharding = None, events=True, sec=None):
    """one transformer forward on this rank's token shard (img [nv, hid], txt [nt, hid]); attention (+ its exchanges) is bracketed by
    events collected in attn_events; `sec` (Sections) brackets the stages for `step_breakdown`.  Returns the final hidden states of the
    local tokens [S_r, hid]."""
    sec = sec if sec is not None else Sections(False)
    geo = st.geo
    hid, mlp_dim = geo.hid, geo.mlp
    nv, nt = img.shape[0], txt.shape[0]
    Sr = nv + nt
    pos0 = sh.a if sh is not None else 0
    sharded = sh is not None and sh.world > 1
    q_buf, k_buf, v_buf = (torch.empty(1, Sr, hid, device=img.device, dtype=img.dtype) for _ in range(3))
    layer = 0
    if hasattr(ops, "set_sharded"):
        ops.set_sharded(sharded)

    def proj(x_img, x_txt, wi, wt):
        with sec("gemm"):
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
        with sec("prologue"):
            q, k, v = ops.prologue(st, q_buf, k_buf, v_buf, pos0, nv)           # [1, H, S_r, hd]
        with sec("exchange"):
            if sharded:
                q, k, v = (sh.to_heads(x[0], i)[None] for i, x in enumerate((q, k, v)))   # [1, H_local, S, hd]
        with sec("attention"):
            o = ops.attention(q, k, v, sparse_step and layer >= first_layers_fp)
        with sec("exchange"):
            if sharded:
                o = sh.to_tokens(o[0])[None]                                     # [1, H, S_r, hd]
        with sec("prologue"):
            o = o.transpose(1, 2).reshape(Sr, hid)      # head-major -> token-major for the output projection (one copy; a view when the
                                                        # attention wrote o token-major: svg.models._core.TOKEN_MAJOR_IO)
        if events:
            e1.record()
            attn_events.append((e0, e1))
        layer += 1
        return o

    def mm(a, b):
        with sec("gemm"):
            return torch.mm(a, b)

    def up_gelu(x, w):
        """MLP up-projection + GELU(tanh): the activation in the GEMM's epilogue where the op table offers it (HipOps.mm_gelu), else GEMM + a pass"""
        if hasattr(ops, "mm_gelu"):
            with sec("gemm"):
                h = ops.mm_gelu(x, w)
            if h is not None:
                return h
        h = mm(x, w.t())
        with sec("elementwise"):
            return ops.gelu(h)

    def mlp(x, w1, w2):
        h = up_gelu(x, w1)
        st.gemm_flops += 2.0 * x.shape[0] * hid * mlp_dim * 2
        return mm(h, w2.t())

    def stream_in(x, m, i_scale, i_shift):
        with sec("glue"):
            return ops.ln_mod(x, m[i_scale], m[i_shift]) if x.shape[0] else x

    def gate_res(res, x, gate):
        with sec("glue"):
            return ops.gate_res(res, x, gate)

    for blk in st.double:
        bi, bt = blk["img"], blk["txt"]
        xi, xt = stream_in(img, bi["mod"], 1, 0), stream_in(txt, bt["mod"], 1, 0)
        proj(xi, xt, bi, bt)
        o = attention()
        if nv:
            img = gate_res(img, mm(o[:nv], bi["wo"].t()), bi["mod"][2])
            img = gate_res(img, mlp(stream_in(img, bi["mod"], 4, 3), bi["w1"], bi["w2"]), bi["mod"][5])
        if nt:
            txt = gate_res(txt, mm(o[nv:], bt["wo"].t()), bt["mod"][2])
            txt = gate_res(txt, mlp(stream_in(txt, bt["mod"], 4, 3), bt["w1"], bt["w2"]), bt["mod"][5])
        st.gemm_flops += 2.0 * Sr * hid * hid
    with sec("elementwise"):
        x = torch.cat([img, txt], dim=0)
    for blk in st.single:
        xm = stream_in(x, blk["mod"], 1, 0)
        proj(xm[:nv], xm[nv:], blk, blk)
        h = up_gelu(xm, blk["wm"])
        o = attention()
        with sec("gemm"):
            out = torch.mm(o, blk["w2a"].t())
            out.addmm_(h, blk["w2b"].t())                # linear2 over cat([attn, mlp]) without materialising the concatenation
        st.gemm_flops += 2.0 * Sr * hid * (mlp_dim + hid + mlp_dim)
        x = gate_res(x, out, blk["mod"][2])
    return x


def measure(steps: int = 1, warmup: int = 1, n_double: int = 20, n_single: int = 40, geo: StepGeo = HY720P, rank: int = 0, world: int = 1,
            group=None, kinds=("sparse", "dense", "sparse_fp8"), host_staged: bool = False):
    from svg.models import _core as core

    enable_tuned_gemms()
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
        times, attn_ms, bytes_step, breakdowns = [], [], 0, []
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
            sec = Sections()
            out = run_step(st, img, txt, sparse_step, first_layers_fp, ev, ops, sh, sec=sec)
            gathered = 0
            if world > 1:
                out, gathered = sh.gather_tokens(out)          # every rank ends the step with all hidden states
            e1.record()
            torch.cuda.synchronize()
            assert torch.isfinite(out.float()).all()
            if it >= warmup:
                times.append(e0.elapsed_time(e1))
                attn_ms.append(sum(a.elapsed_time(b) for a, b in ev))
                breakdowns.append(sec.totals())
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
        brk = {k_: round(v_, 2) for k_, v_ in sorted(breakdowns[mid].items())}
        brk["other"] = round(t - sum(breakdowns[mid].values()), 2)
        res[kind] = {"ms": round(t, 2), "ms_all_steps": [round(x, 1) for x in times], "attention_ms": round(a, 2),
                     "attention_share": round(a / t, 4), "gemm_and_glue_ms": round(t - a, 2), "gemm_tflop_this_rank": round(gf / 1e12, 1),
                     "gemm_tflops_lower_bound_this_rank": round(gf / max((t - a) * 1e-3, 1e-9) / 1e12, 1),
                     # where the step goes: attention = profiler + band kernel (or the dense kernel); prologue = fused QK-norm + RoPE + transpose
                     # (one GPU, `token_major_io`: v stays in the projection layout and o is written token-major — no V transpose, no output
                     # transpose copy; otherwise both are in this section); glue = LayerNorm + modulate / gate-residual (libsvgattn); gemm = torch.mm (+ GELU in
                     # the epilogue when `gelu_in_gemm_epilogue`); elementwise = what is left in torch (GELU pass if not fused, cat)
                     "step_breakdown_ms": brk,
                     "gemm_tflops_this_rank": round(gf / max(breakdowns[mid].get("gemm", 0.0) * 1e-3, 1e-9) / 1e12, 1)}
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
        "gelu_in_gemm_epilogue": bool(getattr(ops, "_fused_gelu", False)),
        "token_major_io": bool(getattr(ops, "v_in_place", False)),     # v read in place, o written token-major (svg_attn_layout_t)
        "gemm_backend": gemm_backend_info(),
    }
    if world > 1:
        out["parallelism"] = (f"tokens/{world} (units of {geo.unit} tokens, largest / mean shard {max(b - a for a, b in sh.ranges) * world / geo.S:.4f}) for norms / GEMMs / prologue / glue, heads/{world} for the "
                              f"attention; per layer 3 x all_to_all in + 1 x all_to_all out, one all-gather of the hidden states per step; "
                              f"this rank: tokens [{sh.a}, {sh.b})")
        out["exchange_backend"] = "gloo through host memory (SVG_BENCH_SMOKE: control-flow run on one GPU, not a measurement)" if host_staged else "RCCL"
    if world == 1 and geo == HY720P:
        del st
        torch.cuda.empty_cache()
        out["gemm_shapes_tflops"] = gemm_shapes_tflops([(geo.V, geo.hid, geo.hid), (geo.V, geo.mlp, geo.hid), (geo.V, geo.hid, geo.mlp),
                                                        (geo.S, geo.hid, geo.hid), (geo.S, geo.mlp, geo.hid), (geo.S, geo.hid, geo.mlp)])
    if "sparse" in res:
        out["sparse_step"] = res["sparse"]
        out["denoise_steps_per_s"] = round(1e3 / res["sparse"]["ms"], 4)
    if "dense" in res:
        out["dense_step"] = res["dense"]
        out["denoise_steps_per_s_dense"] = round(1e3 / res["dense"]["ms"], 4)
    if "sparse_fp8" in res:   # one block, labelled: the fp8 path was measured and closed (README; BASELINE configs[4])
        out["fp8_attention"] = {"status": "closed: not usable (8.8 % rel. L2 against the 16-bit kernel on SVG2, profiles/r05l_fp8_int8_precision_study.txt)"
                                          " - a measurement, not an option",
                                "sparse_step": res["sparse_fp8"], "denoise_steps_per_s": round(1e3 / res["sparse_fp8"]["ms"], 4)}
    if "sparse" in res and "dense" in res:
        out["speedup_sparse_vs_dense_step"] = round(res["dense"]["ms"] / res["sparse"]["ms"], 3)
    return out


# =====================================================================================================================
# Wan 2.1 T2V 14B, 720p / 81 frames, SVG2 (SAP) — BASELINE.json configs[2]: the metric's "denoise steps/sec" half on the SVG2 config
# =====================================================================================================================
@dataclass(frozen=True)
class WanGeo:
    """geometry of the synthetic Wan 2.1 14B stack (the upstream checkpoint: 40 blocks, dim 5120, 40 x 128 heads, FFN 13824, 512 text
    tokens; 720p / 81 frames -> 21 x 45 x 80 latent tokens)"""
    F: int = 21
    P: int = 3600
    hid: int = 5120
    heads: int = 40
    hd: int = 128
    ffn: int = 13824
    text: int = 512
    layers: int = 40
    qc: int = 300          # scripts/wan/wan_t2v_720p_sap.sh:14-19
    kc: int = 1000
    top_p: float = 0.9
    min_kc_ratio: float = 0.1
    iter_init: int = 50
    iter_step: int = 2
    unit: int = 128
    ctx: int = 0           # no text tokens in the self-attention sequence (Sharding reads it)

    @property
    def V(self):
        return self.F * self.P

    @property
    def S(self):
        return self.V


WAN720P = WanGeo()


class WanStack:
    """random weights of the synthetic Wan stack (same seed on every rank).  Every Linear of the model has a bias."""

    def __init__(self, dev, geo: WanGeo = WAN720P, dtype=torch.bfloat16, layers=None):
        g = torch.Generator(device=dev).manual_seed(0)
        self.dev, self.geo, self.dtype = dev, geo, dtype
        hid, ffn = geo.hid, geo.ffn
        n = geo.layers if layers is None else layers

        def lin(o, i):
            return _w(o, i, dev, g, dtype), (torch.randn(o, device=dev, generator=g) * 0.02).to(dtype)

        def vec(scale=0.1, base=0.0):
            return torch.randn(hid, device=dev, generator=g) * scale + base

        self.blocks = []
        for _ in range(n):
            b = {}
            for name in ("q", "k", "v", "o", "cq", "ck", "cv", "co"):
                b["w" + name], b["b" + name] = lin(hid, hid)
            b["w1"], b["b1"] = lin(ffn, hid)
            b["w2"], b["b2"] = lin(hid, ffn)
            # RMSNorm across all heads.  Self attention: weights around 1.5, the element scale of bench_svg2.py's clustered q / k (centroid-level
            # logits of a few units: the top-p map is selective, as on video latents; weights around 1 give logits ~N(0, 1) and a dense map)
            b["nq"], b["nk"] = (vec(0.05, 1.5).to(dtype) for _ in range(2))
            b["cnq"], b["cnk"] = (vec(0.05, 1.0).to(dtype) for _ in range(2))
            b["n2w"], b["n2b"] = vec(0.05, 1.0), vec(0.02)                                           # norm2: FP32LayerNorm with affine
            b["mod"] = [vec() for _ in range(6)]     # shift_msa scale_msa gate_msa c_shift c_scale c_gate (scale_shift_table + temb)
            self.blocks.append(b)
        # complex rotary table, as the patched model forward hands it over: fp32 (real, imag) [S, hd / 2].  Default: angle 0 (real 1, imag 0) — the
        # RoPE kernel does the same work, and the synthetic mixture structure of the hidden states survives into q / k; with random projection
        # weights there is no learned structure that real rotations would preserve, they would only scramble the mixture (dense block maps).
        # SVG_STEP_WAN_ROPE=real: positions x 10000^(-2i/d).
        pos = torch.arange(geo.V, device=dev, dtype=torch.float32)[:, None]
        inv = torch.exp(-torch.arange(0, geo.hd, 2, device=dev, dtype=torch.float32) / geo.hd * math.log(10000.0))[None]
        self.rope = os.environ.get("SVG_STEP_WAN_ROPE", "zero")
        ang = pos * inv * (1.0 if self.rope == "real" else 0.0)
        self.rot_real, self.rot_imag = ang.cos().contiguous(), ang.sin().contiguous()
        self.text = (torch.randn(geo.text, hid, device=dev, generator=g) * 0.5).to(dtype)            # encoder_hidden_states (replicated)
        self.gemm_flops = 0.0


class WanHipOps:
    """The product ops of a Wan block: libsvgattn kernels through svg._native / svg.models._core (GPU only); tests/step_ops_torch.py
    holds a torch statement of the same interface for the CPU sharding test."""

    def __init__(self, geo: WanGeo, first_layers_fp: int):
        from svg import _native as nat
        from svg.models import _core as core

        nat.load()
        self.nat, self.core, self.geo, self.first_layers_fp = nat, core, geo, first_layers_fp
        core.TOKEN_MAJOR_IO = bool(int(os.environ.get("SVG_STEP_TOKEN_MAJOR_IO", "1") or 1))   # 0: the copies of the reference's processors (A/B)
        self.cgeo = core.Geometry(0, geo.F, geo.P)
        self.store = core.CentroidStore()       # per-layer centroids: 50 iterations on a layer's first sparse call, 2 warm-started ones after
        self.logging_file = None                # set: the sparse layers append their block-map densities (the processors' logging_file)

    def ln_mod(self, x, scale, shift):          # norm1 / norm3 (no affine) + modulate — custom_models.py:44-58
        return self.nat.layernorm_modulate_forward(x, scale=scale, shift=shift, eps=1e-6)

    def ln_affine(self, x, w, b):               # norm2 (affine, no modulation) — custom_models.py:71-80
        return self.nat.layernorm_modulate_forward(x, weight=w, bias=b, eps=1e-6)

    def gate_res(self, res, x, gate):           # custom_models.py:61-69, 101-108
        return self.nat.modulate_gate_residual_forward(res, x, gate, out_dtype=res.dtype)

    def rms(self, x, w):                        # RMSNorm across all heads, before the head split — wan/attention.py:105-120
        return self.nat.rmsnorm_forward(x, w, 1e-6)

    def prologue(self, blk, st, q_buf, k_buf, v_buf, pos0, n):
        """projection outputs [1, S_r, hid] -> head-major q, k, v [1, H, S_r, hd]: RMSNorm across all heads + complex RoPE at positions
        pos0 .. + transpose in ONE pass (svg_rmsnorm_rope_transpose — what WanAttn_*Processor.get_fused_prologue runs)"""
        vin = None if self.v_in_place else v_buf
        q, k, v = self.nat.rmsnorm_rope_transpose(q_buf, k_buf, vin, self.geo.heads, blk["nq"], blk["nk"], 1e-6, 2,
                                                  st.rot_real[pos0:pos0 + n].contiguous(), st.rot_imag[pos0:pos0 + n].contiguous(), 0, n)
        if self.v_in_place:     # the attention kernels read v where the projection wrote it (svg_attn_layout_t): a view, no transpose copy
            v = v_buf.unflatten(2, (self.geo.heads, -1)).transpose(1, 2)
        return q, k, v

    v_in_place = False

    def set_sharded(self, sharded: bool):
        """one GPU, head_dim 128: v stays in the projection layout and o comes back token-major (svg.models._core.TOKEN_MAJOR_IO); the
        head-sharded step exchanges contiguous head slices and keeps the copies"""
        self.v_in_place = (not sharded) and self.core.TOKEN_MAJOR_IO and self.geo.hd == 128

    def self_attention(self, q, k, v, layer: int, sparse: bool, head_shard=None):
        g = self.geo
        if sparse:
            return self.core.svg2_sparse_attention(q, k, v, self.cgeo, self.store, layer, g.qc, g.kc, g.top_p, g.min_kc_ratio, g.iter_init,
                                                   g.iter_step, logging_file=self.logging_file, _head_shard=head_shard)
        return self.core.dense_attention(q, k, v)

    def cross_attention(self, q, k, v):
        """q [1, S_r, hid], k / v [1, text, hid] -> [1, S_r, hid]: torch SDPA on head views, what the processor runs for
        `encoder_hidden_states is not None` (wan/attention.py:198-201)"""
        H = self.geo.heads
        qh, kh, vh = (x.unflatten(2, (H, -1)).transpose(1, 2) for x in (q, k, v))
        o = torch.nn.functional.scaled_dot_product_attention(qh, kh, vh, dropout_p=0.0, is_causal=False)
        return o.transpose(1, 2).flatten(2, 3)

    def linear_gelu(self, x, w, b):
        """FFN up-projection + GELU(tanh): the activation rides the GEMM epilogue where hipBLASLt offers it (torch._addmm_activation),
        checked once against the unfused form"""
        if not hasattr(self, "_fused_gelu"):
            try:
                xs = x[:256]
                a = torch._addmm_activation(b, xs, w.t(), use_gelu=True)
                r = torch.nn.functional.gelu(torch.addmm(b, xs, w.t()), approximate="tanh")
                self._fused_gelu = bool(torch.allclose(a.float(), r.float(), atol=3e-2, rtol=3e-2))
            except Exception:  # noqa: BLE001
                self._fused_gelu = False
        if self._fused_gelu:
            return torch._addmm_activation(b, x, w.t(), use_gelu=True)
        return torch.nn.functional.gelu(torch.addmm(b, x, w.t()), approximate="tanh")


def run_step_wan(st: WanStack, x, sparse_step: bool, first_layers_fp: int, ops, sh: Sharding = None, sec: Sections = None):
    """one forward of the Wan block stack on this rank's token shard x [S_r, hid] (ref: WanTransformerBlock_Sparse.forward,
    svg/models/wan/custom_models.py:37-111, and the processors' __call__, svg/models/wan/attention.py:150-209).  `sec` brackets the
    stages for the step breakdown."""
    geo = st.geo
    hid, H = geo.hid, geo.heads
    Sr = x.shape[0]
    pos0 = sh.a if sh is not None else 0
    sharded = sh is not None and sh.world > 1
    sec = sec if sec is not None else Sections(False)
    head_shard = None
    if sharded:
        Hl = H // sh.world
        head_shard = (sh.rank * Hl, (sh.rank + 1) * Hl, H)
    lin = torch.addmm
    text = st.text
    if hasattr(ops, "set_sharded"):
        ops.set_sharded(sharded)

    for layer, b in enumerate(st.blocks):
        m = b["mod"]
        # 1. self attention
        with sec("glue"):
            xn = ops.ln_mod(x, m[1], m[0])
        with sec("gemm"):
            q_buf, k_buf, v_buf = lin(b["bq"], xn, b["wq"].t()), lin(b["bk"], xn, b["wk"].t()), lin(b["bv"], xn, b["wv"].t())
            st.gemm_flops += 2.0 * Sr * hid * hid * 3
        with sec("prologue"):
            q, k, v = ops.prologue(b, st, q_buf[None], k_buf[None], v_buf[None], pos0, Sr)
        with sec("exchange"):
            if sharded:
                q, k, v = (sh.to_heads(t[0], i)[None] for i, t in enumerate((q, k, v)))
        with sec("self_attention"):
            o = ops.self_attention(q, k, v, layer, sparse_step and layer >= first_layers_fp, head_shard)
        with sec("exchange"):
            if sharded:
                o = sh.to_tokens(o[0])[None]
        with sec("prologue"):
            o = o.transpose(1, 2).reshape(Sr, hid)       # head-major -> token-major for the output projection (one copy; a view when the
                                                         # attention wrote o token-major: svg.models._core.TOKEN_MAJOR_IO)
        with sec("gemm"):
            o = lin(b["bo"], o, b["wo"].t())
            st.gemm_flops += 2.0 * Sr * hid * hid
        with sec("glue"):
            x = ops.gate_res(x, o, m[2])
            # 2. cross attention over the text tokens (dense, 512 keys)
            xn = ops.ln_affine(x, b["n2w"], b["n2b"])
        with sec("gemm"):
            cq = lin(b["bcq"], xn, b["wcq"].t())
            ck, cv = lin(b["bck"], text, b["wck"].t()), lin(b["bcv"], text, b["wcv"].t())
            st.gemm_flops += 2.0 * (Sr + 2 * geo.text) * hid * hid
        with sec("prologue"):
            cq, ck = ops.rms(cq, b["cnq"]), ops.rms(ck, b["cnk"])
        with sec("cross_attention"):
            co = ops.cross_attention(cq[None], ck[None], cv[None])[0]
        with sec("gemm"):
            if co.is_contiguous():
                x = torch.addmm(x, co, b["wco"].t()).add_(b["bco"])      # `hidden_states + attn_output` in the GEMM's epilogue (beta = 1)
            else:
                x = x + lin(b["bco"], co.contiguous(), b["wco"].t())
            st.gemm_flops += 2.0 * Sr * hid * hid
        # 3. feed-forward
        with sec("glue"):
            xn = ops.ln_mod(x, m[4], m[3])
        with sec("gemm"):
            h = ops.linear_gelu(xn, b["w1"], b["b1"])
            f = lin(b["b2"], h, b["w2"].t())
            del h
            st.gemm_flops += 2.0 * Sr * hid * geo.ffn * 2
        with sec("glue"):
            x = ops.gate_res(x, f, m[5])
    return x


def measure_wan(steps: int = 2, warmup: int = 1, geo: WanGeo = WAN720P, rank: int = 0, world: int = 1, group=None,
                kinds=("sparse", "dense"), host_staged: bool = False, layers=None):
    """Wan 2.1 720p SVG2 denoise step: 1 dense + 39 SAP layers (first_layers_fp = 0.03, wan_t2v_720p_sap.sh:4-5); the FIRST sparse step
    of a video runs the 50-iteration k-means init in every layer (reported as `first_sparse_step`), later steps 2 warm-started
    iterations.  Dense comparator: all 40 layers dense (the first first_times_fp = 0.2 * 50 = 10 steps of a video)."""
    enable_tuned_gemms()
    dev = torch.device("cuda", torch.cuda.current_device())
    st = WanStack(dev, geo, layers=layers)
    n_layers = len(st.blocks)
    first_layers_fp = math.floor(0.03 * n_layers)                # wan_t2v_inference.py:89
    ops = WanHipOps(geo, first_layers_fp)
    sh = Sharding(geo, rank, world, dev, torch.bfloat16, group, host_staged)
    if world > 1:      # the k-means stopping rule is a maximum over ALL heads: all-reduced over the head shards (svg.distributed.all_reduce_max_)
        from svg import distributed as svg_dist

        svg_dist.enable(group)
    # hidden states: a 64-mode Gaussian mixture (centres x 1.5, spread 0.35: bench_svg2.py's statistics) — iid hidden states give iid q / k,
    # whose k-means clusters all look alike, and top-p then keeps ~every key block (measured: sparse step slower than dense)
    g = torch.Generator(device=dev).manual_seed(1)
    centers = torch.randn(64, geo.hid, device=dev, generator=g) * 1.5
    lab = torch.randint(0, 64, (geo.V,), device=dev, generator=g)
    x0 = torch.empty(sh.b - sh.a, geo.hid, device=dev, dtype=torch.bfloat16)
    for c0 in range(0, geo.V, 8192):       # (generated in chunks for every rank alike: the same stream of random numbers at any world size)
        c1 = min(geo.V, c0 + 8192)
        chunk = centers[lab[c0:c1]] + 0.35 * torch.randn(c1 - c0, geo.hid, device=dev, generator=g)
        lo, hi = max(c0, sh.a), min(c1, sh.b)
        if lo < hi:
            x0[lo - sh.a: hi - sh.a] = chunk[lo - c0: hi - c0].to(torch.bfloat16)
    del centers, lab, chunk
    res, first_sparse_ms = {}, None
    for kind in kinds:
        sparse_step = kind == "sparse"
        times, parts = [], []
        n_warm = warmup + (1 if sparse_step else 0)              # + the step that initialises the centroids
        for it in range(n_warm + steps):
            st.gemm_flops = 0.0
            sec = Sections()
            if sh.buf is not None:
                sh.buf.bytes_in = sh.buf.bytes_out = 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            if world > 1:
                import torch.distributed as dist

                dist.barrier(group)
            e0.record()
            out = run_step_wan(st, x0, sparse_step, first_layers_fp, ops, sh, sec)
            gathered = 0
            if world > 1:
                out, gathered = sh.gather_tokens(out)
            e1.record()
            torch.cuda.synchronize()
            assert torch.isfinite(out.float()).all()
            t = e0.elapsed_time(e1)
            if sparse_step and it == 0:
                first_sparse_ms = t
            if it >= n_warm:
                times.append(t)
                parts.append(sec.totals())
        if world > 1:
            import torch.distributed as dist

            tt = torch.tensor(times, dtype=torch.float64, device="cpu" if host_staged else dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=group)
            times = tt.tolist()
        order = sorted(range(len(times)), key=lambda i: times[i])
        mid = order[len(order) // 2]
        t, p = times[mid], parts[mid]
        gf = st.gemm_flops
        brk = {k_: round(v_, 2) for k_, v_ in sorted(p.items())}
        brk["other"] = round(t - sum(p.values()), 2)
        res[kind] = {"ms": round(t, 2), "ms_all_steps": [round(x_, 1) for x_ in times], "step_breakdown_ms": brk,
                     "attention_share": round((p.get("self_attention", 0.0)) / t, 4),
                     "gemm_tflop_this_rank": round(gf / 1e12, 1),
                     "gemm_tflops_this_rank": round(gf / max(p.get("gemm", 0.0) * 1e-3, 1e-9) / 1e12, 1)}
        if world > 1 and sh.buf is not None:
            res[kind]["rccl_bytes_received_per_step_this_rank"] = int(sh.buf.bytes_in + sh.buf.bytes_out + gathered)
        if sparse_step and world == 1:
            # what sparsity the timed steps ran at: one more (untimed) sparse step with the processors' density log switched on
            import tempfile

            with tempfile.TemporaryDirectory() as td:
                ops.logging_file = os.path.join(td, "density.jsonl")
                run_step_wan(st, x0, True, first_layers_fp, ops, sh, Sections(False))
                torch.cuda.synchronize()
                ops.core.flush_density_log()
                ops.logging_file = None
                dens = [json.loads(l)["avg_density"] for l in open(os.path.join(td, "density.jsonl")) if l.strip()]
            if dens:
                res[kind]["block_map_density"] = {"mean": round(sum(dens) / len(dens), 4), "min": round(min(dens), 4), "max": round(max(dens), 4),
                                                  "layers": len(dens)}
    out = {
        "metric": "denoise_step_wan720p_svg2" if geo == WAN720P and layers is None else "denoise_step_wan_custom",
        "workload": f"synthetic Wan 2.1 14B-style transformer forward: {n_layers} blocks (self attention, cross attention over {geo.text} text "
                    f"tokens, FFN {geo.ffn}), hidden {geo.hid}, {geo.heads} x {geo.hd} heads, S = {geo.S} video tokens, bf16, random weights; "
                    f"sparse step = {first_layers_fp} dense + {n_layers - first_layers_fp} SVG2 layers (QC {geo.qc}, KC {geo.kc}, top-p {geo.top_p}, "
                    f"min_kc_ratio {geo.min_kc_ratio}, {geo.iter_step} warm-started k-means iterations per layer and step)",
        "steps": steps, "warmup": warmup, "n_gpus": world,
        "step_breakdown_sections": "glue = LayerNorm + modulate / gate-residual kernels (libsvgattn); prologue = RMSNorm across heads, complex "
                                   "RoPE + head-major transpose of q and k (v and o: in the projection layout when `token_major_io`, else a "
                                   "transpose copy each), RMSNorm of the cross-attention q / k; self_attention = k-means + block map + variable-block "
                                   "attention (sparse layers) or the dense kernel; cross_attention = torch SDPA over the text tokens; gemm = "
                                   "torch.addmm (hipBLASLt), GELU in the epilogue where available; exchange = RCCL all-to-alls (N > 1)",
        "gelu_in_gemm_epilogue": bool(getattr(ops, "_fused_gelu", False)),
        "token_major_io": bool(getattr(ops, "v_in_place", False)),     # v read in place, o written token-major (svg_attn_layout_t)
        "not_modelled": "patch / time / text embedders, final layer, scheduler, text encoder, VAE",
        "data": "synthetic: hidden states = 64-mode Gaussian mixture (centres x 1.5, spread 0.35, bench_svg2.py's statistics), QK-norm weights ~1.5, "
                + ("rotary table = real positions" if st.rope == "real" else "rotary table at angle 0 (the RoPE kernel runs unchanged; random projection "
                   "weights carry no structure that real rotations would preserve)") + "; the block-map density the steps ran at is in "
                "sparse_step.block_map_density (iid hidden states give density ~1: a sparse step slower than the dense one)",
        "gemm_backend": gemm_backend_info(),
    }
    if world > 1:
        out["parallelism"] = (f"tokens/{world} (units of {geo.unit}) for norms / GEMMs / prologue / cross attention / glue, heads/{world} for the "
                              f"self attention (k-means, map and attention per head; the stopping rule's shift all-reduced); this rank: tokens [{sh.a}, {sh.b})")
        out["exchange_backend"] = "gloo through host memory (SVG_BENCH_SMOKE: control flow only)" if host_staged else "RCCL"
    if world == 1 and geo == WAN720P and layers is None:
        del st
        torch.cuda.empty_cache()
        out["gemm_shapes_tflops"] = gemm_shapes_tflops([(geo.V, geo.hid, geo.hid), (geo.V, geo.ffn, geo.hid), (geo.V, geo.hid, geo.ffn)])
    if "sparse" in res:
        out["sparse_step"] = res["sparse"]
        out["denoise_steps_per_s"] = round(1e3 / res["sparse"]["ms"], 4)
        out["first_sparse_step_ms_with_kmeans_init"] = round(first_sparse_ms, 1) if first_sparse_ms is not None else None
    if "dense" in res:
        out["dense_step"] = res["dense"]
        out["denoise_steps_per_s_dense"] = round(1e3 / res["dense"]["ms"], 4)
    if "sparse" in res and "dense" in res:
        out["speedup_sparse_vs_dense_step"] = round(res["dense"]["ms"] / res["sparse"]["ms"], 3)
    if world > 1:
        svg_dist.disable()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--layers-double", type=int, default=20)
    ap.add_argument("--layers-single", type=int, default=40)
    ap.add_argument("--tiny", action="store_true", help="a small geometry (F = 5, P = 600; hidden 512, 4 heads) for control-flow tests")
    ap.add_argument("--model", default="hy720p", choices=["hy720p", "wan720p"],
                    help="hy720p: HunyuanVideo SVG1 stack (BASELINE configs[3]); wan720p: Wan 2.1 14B SVG2 / SAP stack (configs[2])")
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
    if a.model == "wan720p":
        wgeo = WanGeo(F=5, P=600, hid=512, heads=4, hd=128, ffn=1024, text=64, layers=3, qc=20, kc=40) if a.tiny else WAN720P
        out = measure_wan(a.steps, a.warmup, wgeo, rank, world, host_staged=smoke and world > 1)
    else:
        geo = StepGeo(F=5, P=600, ctx=256, L=64, hid=512, heads=4, hd=128, mlp=1024) if a.tiny else HY720P
        kinds = ("sparse", "dense", "sparse_fp8") if world == 1 else ("sparse", "dense")
        out = measure(a.steps, a.warmup, a.layers_double, a.layers_single, geo, rank, world, kinds=kinds, host_staged=smoke and world > 1)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
