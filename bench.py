#!/usr/bin/env python3
"""Headline benchmark: SVG1 block-sparse attention of one HunyuanVideo 720p / 129-frame layer-call on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload hy720p|hy480p|tiny] [--no-cpu] [--no-dense]

Workload (BASELINE.json configs[1]): q, k, v = randn([1, 24, 119056, 128], bf16) (F=33, P=3600, ctx=256, prompt 64),
sparsity 0.25 -> band 15616 tokens, half of the heads temporal (fused layout transformation inside the kernel).
One *step* = one sparse layer-call = online profiler (sample_mse, 64 rows) + band attention with fused placement, i.e.
exactly what Hunyuan_SVGAttn_Processor2_0.attention_core_logic does per layer in the sparse branch
(ref: svg/models/hyvideo/attention.py:507-524).  Inputs are resident in HBM before the timed region.

metric  = attention TFLOP/s, algorithmic: 4 * D * H * (#unmasked (q,k) pairs) per layer-call (42.78 TFLOP
          at L=64) divided by wall time per step; `attention_only_steps_per_s` (60 such layer-calls, nothing else of a
          denoise step — no projections, norms, MLPs) is reported beside it.
roofline: bound = MFMA (dense bf16 peak 2.5 PFLOP/s); `achieved` = algorithmic FLOPs of the dominant kernel
          (band_attn_m16_kernel since round 4) / its mean launch duration measured with HIP events on the launch stream.
cpu_baseline: BASELINE.md §3 — the reference's CPU-capable dense path torch SDPA (ref: svg/models/wan/attention.py:279-281,
          svg/models/hyvideo_orig/modules/attenion.py:488-491) on all host cores, bf16, ONE head at the full sequence length
          (median of 3), or the longest sequence that fits the time bound; plus flex_attention eager on the CPU with the
          reference's mask_mod for the sparse semantics on a reduced geometry — see `sample`.
denoise_step_hy720p: BASELINE.json configs[3] at N = 1 — a MEASURED denoise step of a synthetic 60-block HunyuanVideo stack
          (hipBLASLt GEMMs + this repo's glue / prologue / attention kernels), sparse and dense (bench_step.measure); --no-step skips it.
svg2_wan720p: BASELINE.json configs[2] (SVG2 / SAP layer-call of Wan 2.1 720p) measured in the same process after the
          headline workload (bench_svg2.measure); --no-svg2 skips it.
N > 1   : heads are independent units; rank r owns heads r::N of the same layer-call (strong scaling), no data-path
          collective inside the attention.  A step = the exchanges either side of it + the attention: q, k, v start
          TOKEN-sharded (whole frames per rank, all heads — how the token-wise layers before the attention leave them) and
          are turned into head shards by three all_to_all_single (inbound, svg.distributed.tokens_to_heads); the output is
          all-gathered (the exchange the next op, `to_out`, needs) in row segments overlapped with the launch.  RCCL.
          The overlap uses one-wave waiter kernels with a deadline; if one times out during warm-up the bench falls back
          to one launch per chunk of heads and says so in `exchange`.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
# what every fp8 block of the line says about itself (README: configs[4] was measured and closed in round 5)
FP8_STATUS = ("closed: not usable (8.8 % rel. L2 against the 16-bit kernel on SVG2; no e4m3 / int8 form inside 3 % with a projected gain, "
              "profiles/r05l_fp8_int8_precision_study.txt) - a measurement, not an option")
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import warnings  # noqa: E402

import torch  # noqa: E402

warnings.filterwarnings("ignore", message="flex_attention called without torch.compile")

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_FP8_TFLOPS = 5000.0   # dense MFMA fp8 (same guide)

WORKLOADS = {
    # name: (H, D, F, P, ctx, prompt_len, sparsity)
    "hy720p": (24, 128, 33, 3600, 256, 64, 0.25),
    "hy480p": (24, 128, 33, 1350, 256, 64, 0.25),
    "tiny": (4, 128, 5, 600, 256, 64, 0.4),
}


def allowed_pairs_hy(V: int, ctx: int, L: int, tf: int) -> int:
    """#unmasked (q,k) pairs of the Hunyuan mask: band over video + text rows/cols + pad block."""
    tf = min(tf, V)
    band = V * (2 * tf - 1) - tf * (tf - 1)
    return band + 2 * V * L + L * L + (ctx - L) ** 2


def band_pairs(m, S: int) -> int:
    """#allowed (q, k) pairs of a BandMask (the kernels' interval form, BandPolicy::row_intervals) — algorithmic FLOPs = 4 D H pairs"""
    import numpy as np

    q = np.arange(S, dtype=np.int64)
    real = m.real_len
    rq = q < real
    rowf = (q >= m.rowfull_lo) & (q < m.rowfull_hi)
    lo = np.where(rq, np.where(rowf, 0, np.maximum(q - m.band + 1, 0)), real)
    hi = np.where(rq, np.where(rowf, real, np.minimum(q + m.band, real)), S)
    alen = np.maximum(hi - lo, 0)
    ch = min(m.colfull_hi, real)
    b0, b1 = m.colfull_lo, max(ch, m.colfull_lo)
    use_b = rq & ~rowf
    inter = np.maximum(np.minimum(hi, b1) - np.maximum(lo, b0), 0)
    blen = np.where(use_b, (b1 - b0) - inter, 0)
    return int((alen + blen).sum())


BAND_KERNELS = {0: "band_attn_m16_kernel<bf16>", 8: "band_attn_m16_kernel<bf16>", 3: "band_attn_w4_kernel<bf16,128>", 2: "band_attn_pp2_kernel<bf16,128>",
                1: "band_attn_kernel<bf16,128,4>"}


def _pmc_traffic(workload: str, kernel: str):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction +
    WRITE_SIZE, tools/gpu_pmc.sh).  PMC collection needs its own rocprofv3 runs, so bench.py reports the value of the newest
    committed profile of THIS kernel (profiles/r*_pmc_traffic.json) for the headline workload, with its file name as
    `traffic_source`; (null, null) otherwise."""
    if workload != "hy720p":
        return None, None
    for p in sorted((ROOT / "profiles").glob("r*_pmc_traffic*.json"), reverse=True):
        try:
            d = json.loads(p.read_text())
            if d.get("kernel", "band_attn_pp2_kernel<bf16,128>").split("<")[0] == kernel.split("<")[0]:
                return float(d["traffic_bytes_per_launch"]), f"profiles/{p.name} (committed rocprofv3 --pmc passes, not collected in this run)"
        except Exception:  # noqa: BLE001
            continue
    return None, None


def cpu_baseline(H: int, D: int, S: int, budget_s: float = 36.0):
    """BASELINE.md §3.  Dense leg: torch SDPA bf16 on all host cores, ONE head at the full sequence length S, median of 3 — if
    a probe at S/8 predicts more than budget_s for that, the longest power-of-two fraction of S that fits is timed instead and
    `sample` says so.  Sparse leg: flex_attention, eager, on the CPU with the reference's Hunyuan mask_mod (the reference's sparse
    semantics on its only CPU-capable path) on a reduced geometry — eager flex materialises the [S, S] scores."""
    import statistics
    import torch.nn.functional as F

    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)

    def sdpa_s(s, reps):
        q, k, v = (torch.randn(1, 1, s, D, dtype=torch.bfloat16, generator=g) for _ in range(3))
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            F.scaled_dot_product_attention(q, k, v)
            ts.append(time.perf_counter() - t0)
        return ts

    probe_s = max(2048, S // 8)
    sdpa_s(probe_s, 1)                       # page in / thread pool
    t_probe = min(sdpa_s(probe_s, 2))
    s_run, frac = S, 1
    while t_probe * (s_run / probe_s) ** 2 * 3 > budget_s and s_run > probe_s:
        s_run //= 2
        frac *= 2
    ts = sdpa_s(s_run, 3)
    t_med = statistics.median(ts)
    tflops = 4.0 * s_run * s_run * D / t_med / 1e12
    full_head = 4.0 * S * S * D / 1e12
    out = {
        "value": round(tflops, 4),
        "unit": "TFLOP/s",
        "cores": cores,
        "kind": "reference",
        "sample": f"torch SDPA dense bf16 (reference CPU path svg/models/wan/attention.py:279-281), 1 head of {H}, S={s_run}"
                  f"{'' if frac == 1 else f' (= S/{frac}: the full S={S} would exceed the {budget_s:.0f} s bound)'}, D={D}, median of 3: "
                  f"{t_med:.2f} s; one full head = {full_head:.2f} TFLOP, the {H}-head layer-call = {full_head * H:.1f} TFLOP dense "
                  f"= {full_head * H / tflops:.0f} s at this rate",
        "seconds": [round(t, 3) for t in ts],
    }
    try:
        from torch.nn.attention.flex_attention import create_block_mask, flex_attention

        from svg.models.hyvideo.utils import sparsity_to_width

        F_, P_, ctx, L, Hs = 9, 512, 256, 64, 2
        Vs = F_ * P_
        Ss = Vs + ctx
        mul = sparsity_to_width(0.25, ctx, F_, P_)
        two_frame = math.floor(mul * P_ / 128) * 128
        real = Vs + L

        def mask_mod(b, h, qi, ki):   # the predicate of generate_temporal_head_mask_mod, ref: svg/models/hyvideo/utils.py:20-44
            both_real = (ki < real) & (qi < real)
            both_pad = (ki >= real) & (qi >= real)
            band = torch.abs(qi - ki) < two_frame
            text = ((ki >= Vs) & (ki < real)) | ((qi >= Vs) & (qi < real))
            return (both_real & (band | text)) | both_pad

        q, k, v = (torch.randn(1, Hs, Ss, D, dtype=torch.bfloat16, generator=g) for _ in range(3))
        bm = create_block_mask(mask_mod, None, None, Ss, Ss, device="cpu", _compile=False)
        flex_attention(q, k, v, block_mask=bm)
        t0 = time.perf_counter()
        flex_attention(q, k, v, block_mask=bm)
        tf = time.perf_counter() - t0
        tfw = two_frame
        pairs = allowed_pairs_hy(Vs, ctx, L, tfw)
        out["sparse_flex_cpu"] = {
            "what": f"flex_attention eager on CPU, reference mask_mod (svg/models/hyvideo/utils.py:20-44), H={Hs} F={F_} "
                    f"P={P_} ctx={ctx} prompt={L} S={Ss} band={tfw}",
            "seconds": round(tf, 3),
            "tflops_algorithmic": round(4.0 * D * Hs * pairs / tf / 1e12, 4),
            "tflops_dense_equivalent": round(4.0 * D * Hs * Ss * Ss / tf / 1e12, 4),
        }
    except Exception as e:  # noqa: BLE001
        out["sparse_flex_cpu"] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="hy720p", choices=sorted(WORKLOADS))
    ap.add_argument("--variant", type=int, default=0, help="svg_band_attention schedule (include/svg_attn.h): 0 default (= 8 at head_dim 128), "
                    "1 lock-step 4 waves, 2 two-phase ping-pong on 32x32x16 MFMAs, 3 one wave per SIMD, 8 two-phase on 16x16x32 MFMAs")
    ap.add_argument("--chunks", type=int, default=1, help="N = 1 only: split the launch into this many head chunks on two streams")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp8"], help="fp8: e4m3 QK^T / PV (svg_band_attention_fp8: quantise + "
                    "placement pre-pass and attention kernel, both inside the timed step; BASELINE.json configs[4]); N = 1 only")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-dense", action="store_true")
    ap.add_argument("--no-svg2", action="store_true", help="skip the SVG2 (BASELINE.json configs[2]) extras block")
    ap.add_argument("--no-hbm", action="store_true", help="skip the HBM-bound kernels block (bench_hbm.measure)")
    ap.add_argument("--no-step", action="store_true", help="skip the measured 60-layer denoise step (bench_step.measure)")
    ap.add_argument("--no-profiler", action="store_true", help="time the attention kernel only")
    ap.add_argument("--no-ab", action="store_true", help="skip the same-box A/B block (frozen reference schedule and the other schedules timed beside the default)")
    ap.add_argument("--prescaled", action="store_true", help="opt-in pre-scaled path (flex_attention's PRESCALE_QK trade-off; the processors' "
                    "prescale_q = True): q carries sm_scale * log2(e) — what the fused prologue hands the attention core then "
                    "(svg_qk_norm_rope_transpose_qscale; here multiplied and rounded once before the timed region) — and the step runs "
                    "svg_sample_mse / svg_band_attention_prescaled on it.  Default since round 4: plain q, the softmax scale applied to the fp32 "
                    "scores inside the kernel (the reference's formulation, svg/models/hyvideo/attention.py:401-403)")
    ap.add_argument("--heads", default="alt", choices=["alt", "spatial", "temporal"], help="best_mask_idx pattern")
    ap.add_argument("--extras", default="auto", choices=["auto", "small"], help="auto: the SVG2 / denoise-step extras blocks at full size beside the "
                    "hy720p headline workload and nowhere else; small: the same blocks on reduced geometries beside ANY workload (contract tests)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no GPU visible); the HIP path has no CPU fallback")
    # (SVG_BENCH_SMOKE=1: all ranks on cuda:0 over gloo — a control-flow smoke test of the N > 1 path on a one-GPU box, not a measurement)
    smoke = bool(os.environ.get("SVG_BENCH_SMOKE"))
    if smoke:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        if smoke:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from svg import _native as nat
    from svg.models.hyvideo.utils import sparsity_to_width

    nat.load()
    H, D, F_, P_, ctx, L, sparsity = WORKLOADS[a.workload]
    V = F_ * P_
    S = V + ctx
    width = sparsity_to_width(sparsity, ctx, F_, P_)
    tf = math.floor(width * P_ / 128) * 128
    mask = nat.BandMask(real_len=V + L, band=tf, colfull_lo=V, colfull_hi=V + L, rowfull_lo=V, rowfull_hi=V + L)
    pairs = allowed_pairs_hy(V, ctx, L, tf)
    flops_call = 4.0 * D * H * pairs
    dense_flops = 4.0 * D * H * S * S

    # heads of this rank (strong scaling: the layer-call is split by heads).  The heads are taken in super-groups of
    # world * n consecutive heads, n per rank, so that the all-gather of one local chunk of n heads lands as ONE contiguous,
    # naturally ordered slice of the full [H, S, D] output — no reordering copy — and the gather of chunk c overlaps the
    # attention of chunk c + 1 ("overlappable").
    from svg.distributed import chunked_head_layout, gather_chunk, token_range, tokens_to_heads

    # (one head per chunk: with one launch + completion counters a chunk costs a waiter and an all-gather call, not a kernel
    #  launch, and the exposed gather at the end of the step is that of ONE head instead of a third of the rank's heads)
    n_chunks, n_per, my_heads = chunked_head_layout(H, rank, world, max_chunks=24)
    if world == 1 and a.chunks > 1:   # exercise the chunked two-stream launch path on one GPU (no collective)
        assert H % a.chunks == 0
        n_chunks, n_per = a.chunks, H // a.chunks
    Hl = len(my_heads)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)

    def head_rows(h: int, which: int, lo: int = 0, hi: int = S) -> "torch.Tensor":
        """rows [lo, hi) of head h of q (0) / k (1) / v (2): seeded by the GLOBAL head index, so that every world size sees the same
        tensors and the gathered output of an N-rank run can be held against the 1-rank run bit for bit (`output_checksum` below)"""
        gh = torch.Generator(device=dev).manual_seed(7919 * (3 * h + which) + 1234)
        return torch.randn(S, D, device=dev, dtype=torch.bfloat16, generator=gh)[lo:hi]
    if world > 1:
        # token-sharded inputs: ALL heads (ordered owner by owner, as the fused prologue would write them) of this rank's tokens
        # (shards in units of 128 tokens like bench_step.StepGeo.unit: largest / mean 1.006 at N = 8 where whole frames give 1.21)
        TOKEN_UNIT = 128 if S >= 8 * 128 * world else 8
        head_lists = [chunked_head_layout(H, r, world, max_chunks=24)[2] for r in range(world)]
        ta, tb = token_range(S, rank, world, unit=TOKEN_UNIT)
        owner_order = [h for hl in head_lists for h in hl]     # position p of a token shard holds head owner_order[p]
        q_tok, k_tok, v_tok = (torch.stack([head_rows(h, w, ta, tb) for h in owner_order]) for w in range(3))
        q, k, v = (torch.empty(1, Hl, S, D, device=dev, dtype=torch.bfloat16) for _ in range(3))
        seen = torch.ones(1, device=dev)
        dist.all_reduce(seen)                      # every rank of the communicator answered
        rccl_ranks_seen = int(seen.item())
    else:
        q, k, v = (torch.stack([head_rows(h, w) for h in my_heads])[None] for w in range(3))
    o = torch.empty_like(q)
    pat = {"alt": lambda h: h % 2, "spatial": lambda h: 0, "temporal": lambda h: 1}[a.heads]
    best = torch.tensor([[pat(h) for h in my_heads]], device=dev, dtype=torch.int64)  # default: alternate spatial / temporal
    rows = torch.randint(0, min(10000, V), (64,), device=dev)
    bb = int((P_ * 1.5) // 128)
    prof = nat.ProfileDesc(0, F_, P_, 1)
    prof.variant[0] = nat.ProfileVariant(0, 0, V, bb, 0, V, S)
    prof.variant[1] = nat.ProfileVariant(1, 0, V, bb, 0, V, S)
    full = torch.empty(H, S, D, device=dev, dtype=torch.bfloat16) if world > 1 else None   # every rank ends with all heads

    ev_a0, ev_a1 = [], []
    # N > 1: ONE launch over this rank's heads that counts completions per head (svg_band_attention_notify; the dispatch is
    # head-major, so heads finish in order); for every chunk of heads a one-wave kernel on a side stream returns once the chunk's
    # counters are full, and the chunk's all-gather is enqueued behind it — the exchange of chunk c runs while the launch is
    # still working on chunks c + 1 ...  (Chunked launches instead — SVG_BENCH_CHUNK_LAUNCHES=1, kept for A/B — cost
    # 5.7 ms per rank at N = 8 against 4.9 ms for the single launch: every launch ends with a partly idle round.)
    chunk_launches = bool(os.environ.get("SVG_BENCH_CHUNK_LAUNCHES"))
    mode = {"chunk_launches": chunk_launches, "fallback": False}
    WAIT_MS = int(os.environ.get("SVG_BENCH_WAITER_TIMEOUT_MS", "5000"))   # a step is tens of ms; a waiter that sits this long is stuck
    timed_out = torch.zeros(1, device=dev, dtype=torch.int32)
    stuck = 1 if os.environ.get("SVG_BENCH_TEST_STUCK_WAITER") else 0   # test hook: waiters wait for a count that never comes
    side = [torch.cuda.Stream(device=dev) for _ in range(2)] if n_chunks > 1 else None
    # every head is cut into row segments with their own counters: what stays exposed at the end of a step is the gather of the
    # last segment of the last head (a quarter of a head per rank)
    nseg, row_bounds, seg_targets = nat.band_notify_layout(S, mask, 4)
    done = nat.notify_counters(Hl, nseg, dev) if n_chunks > 1 else None   # + BH words of library scratch behind the counters

    def gather_segment(c: int, sg: int):
        """all-gather rows [row_bounds[sg], row_bounds[sg + 1]) of local chunk c (n_per heads per rank) into the natural [H, S, D]
        layout: one contiguous all_gather_into_tensor (the call the head-granular path uses) into a staging buffer, then `world`
        strided copies on the same side stream (the received blocks of different ranks are S * D apart in `full`)."""
        a0, a1 = row_bounds[sg], row_bounds[sg + 1]
        src = o[0, c * n_per:(c + 1) * n_per, a0:a1].contiguous()
        tmp = torch.empty((world,) + tuple(src.shape), device=dev, dtype=src.dtype)
        dist.all_gather_into_tensor(tmp.view(world * n_per, a1 - a0, D), src, async_op=True).wait()   # the stream waits, not the host
        h0 = c * world * n_per
        full[h0:h0 + world * n_per, a0:a1].view(world, n_per, a1 - a0, D).copy_(tmp)

    fp8 = a.dtype == "fp8"
    if fp8:
        assert world == 1, "--dtype fp8 is a single-GPU line"
        a.no_step = True
    q_att = q
    a.prescaled = a.prescaled and not fp8 and a.variant == 0 and D == 128
    if a.prescaled and world == 1:
        q_att = (q.float() * nat.softmax_q_scale(D)).to(q.dtype)   # outside the timed region: the prologue's job (svg_qk_norm_rope* q_scale)
    if a.prescaled and world > 1:   # token shards as the token-sharded prologue would write them: pre-scaled q
        q_tok = (q_tok.float() * nat.softmax_q_scale(D)).to(q_tok.dtype)
    LN2 = math.log(2.0)
    ev_p0, ev_p1 = [], []
    ev_q0, ev_q1 = [], []

    def step(timed: bool):
        if world > 1:   # inbound exchange: token shards -> this rank's heads over the full sequence
            for x_tok, x in ((q_tok, q), (k_tok, k), (v_tok, v)):
                if smoke:   # gloo: stage through host memory (the smoke run checks control flow and placement, not speed)
                    x[0].copy_(tokens_to_heads(x_tok.cpu(), S, unit=TOKEN_UNIT, head_lists=head_lists, presorted=True))
                else:
                    tokens_to_heads(x_tok, S, unit=TOKEN_UNIT, head_lists=head_lists, presorted=True, out=x[0])
        if not a.no_profiler:
            if timed:
                ev_q0.append(torch.cuda.Event(enable_timing=True))
                ev_q0[-1].record()
            mse = nat.sample_mse(q_att[0], k[0], v[0], rows, prof, sm_scale=LN2 if a.prescaled else None)
            if timed:
                ev_q1.append(torch.cuda.Event(enable_timing=True))
                ev_q1[-1].record()
            _ = mse.argmin(0)  # best_mask_idx (kept on device; the bench uses the fixed alternating pattern)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        works = []
        main = torch.cuda.current_stream()
        kw = dict(vid0=0, num_frame=F_, frame_size=P_, variant=a.variant, q_prescaled=a.prescaled)
        if fp8:
            kw8 = dict(vid0=0, num_frame=F_, frame_size=P_, head_perm_flag=best, out=o)
            nat.band_attention_fp8(q, k, v, mask, stage=1, **kw8)      # absolute maxima + quantise / placement / V transpose
            p1 = torch.cuda.Event(enable_timing=True)
            p1.record()
            nat.band_attention_fp8(q, k, v, mask, stage=2, **kw8)      # attention on e4m3 operands
            if timed:
                ev_p0.append(e0)
                ev_p1.append(p1)
        elif side is None:
            nat.band_attention(q_att, k, v, mask, head_perm_flag=best, out=o, **kw)
        elif mode["chunk_launches"]:
            for c in range(n_chunks):
                sl = slice(c * n_per, (c + 1) * n_per)
                st = side[c % 2]
                st.wait_stream(main)   # inputs and the profiler's result are produced on the main stream
                with torch.cuda.stream(st):
                    nat.band_attention(q_att[:, sl], k[:, sl], v[:, sl], mask, head_perm_flag=best[:, sl].contiguous(), out=o[:, sl], **kw)
                    if world > 1:
                        works.append(gather_chunk(full, o[0, sl], c, n_per, world))
        else:
            done.zero_()
            zeroed = torch.cuda.Event()
            zeroed.record()
            nat.band_attention(q, k, v, mask, head_perm_flag=best, out=o, done=done, done_nseg=nseg, **kw)
            cnt = done[:Hl * nseg].view(Hl, nseg)
            i = 0
            for c in range(n_chunks):
                for sg in range(nseg):
                    st = side[i % 2]
                    i += 1
                    st.wait_event(zeroed)          # NOT the launch itself: the waiter runs beside it
                    with torch.cuda.stream(st):
                        for h in range(c * n_per, (c + 1) * n_per):   # (views of the live counters, never copies)
                            nat.wait_counters(cnt[h, sg:sg + 1], seg_targets[sg] + stuck, timeout_ms=WAIT_MS, timed_out=timed_out)
                        if world > 1:   # RCCL all-gather of this row segment as soon as it is complete on every head of the chunk
                            gather_segment(c, sg)
        if side:
            for st in side:
                main.wait_stream(st)
        e1.record()
        if timed:
            ev_a0.append(e0)
            ev_a1.append(e1)
        for w in works:
            w.wait()

    for _ in range(a.warmup):
        step(False)
    if side is not None and not mode["chunk_launches"]:
        # watchdog: did any waiter of the warm-up give up?  (every rank must take the same path: MAX over ranks.)  With no warm-up
        # steps one untimed step is run for this check.
        if a.warmup == 0:
            step(False)
        torch.cuda.synchronize()
        flag = timed_out.clone().float()
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if flag.item() > 0:
            mode["chunk_launches"] = mode["fallback"] = True
            timed_out.zero_()
            step(False)          # warm the fallback path

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # sustained shader clock of the timed steps: a one-wave probe kernel beside them (s_memtime against the 100 MHz counter)
    # (N = 1 without side streams only: the probe occupies a hardware queue for the whole span, and a process has few of them — with
    #  the waiter / exchange streams of the N > 1 path a launch could be queued BEHIND the probe that waits for it)
    clock = None
    if rank == 0 and world == 1 and side is None:
        try:
            clock = nat.ClockProbe(dev)
        except Exception:  # noqa: BLE001
            clock = None
    if clock:
        # rehearsal: one untimed step beside a short-lived probe.  Should the runtime ever put the probe and the main stream on the same
        # hardware queue, the step would sit behind the probe until its deadline — seen here as a step that takes about the deadline
        # instead of milliseconds — and the probe is then left out of the timed region (clock: null) instead of poisoning it.
        torch.cuda.synchronize()
        clock.start(max_ms=1500)
        tr0 = time.perf_counter()
        step(False)
        torch.cuda.current_stream().synchronize()
        rehearsal_s = time.perf_counter() - tr0
        clock.arm_stop()
        clock.result()
        if rehearsal_s > 1.0:
            clock = None
    barrier()
    if clock:
        clock.start(max_ms=20000)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    if clock:
        clock.arm_stop()   # (stream-ordered behind the last step: the barrier below must not wait for a probe that waits for it)
    barrier()
    dt = time.perf_counter() - t0
    clock_info = None
    if clock:
        mhz = clock.result()
        clock_info = {"sclk_mhz_timed_steps": mhz, "frac_of_2400": round(mhz / 2400.0, 4) if mhz else None,
                      "how": "svg_debug_clock_probe: shader-clock ticks / 100 MHz ticks of one sleeping wave while the timed steps ran"}
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    if smoke and world > 1:
        # One more step with o and `full` poisoned: a gather that runs ahead of the attention kernel, or a segment that lands in the
        # wrong place, leaves NaNs / other heads' rows behind.  Checked against a plain all_gather of the finished outputs.
        o.fill_(float("nan"))
        full.fill_(float("nan"))
        step(False)
        torch.cuda.synchronize()
        # inbound: this rank's q must be its heads' rows of every rank's token shard
        shards = [None] * world
        dist.all_gather_object(shards, (ta, tb, q_tok.cpu()))
        order = [h for hl in head_lists for h in hl]
        pos = torch.tensor([order.index(h) for h in my_heads])
        for a_, b_, xt in shards:
            assert torch.equal(q[0][:, a_:b_].cpu(), xt[pos]), "inbound exchange put a token shard in the wrong place"
        outs = [torch.empty_like(o[0]) for _ in range(world)]
        dist.all_gather(outs, o[0].contiguous())
        assert not torch.isnan(o.float()).any(), "attention output incomplete"
        for r in range(world):
            idx = torch.tensor(chunked_head_layout(H, r, world, max_chunks=24)[2], device=dev)
            assert torch.equal(full.index_select(0, idx), outs[r]), f"exchange of rank {r}'s heads is wrong or ran ahead of the kernel"
    ms_step = dt / a.steps * 1e3
    attn_ms = sum(x.elapsed_time(y) for x, y in zip(ev_a0, ev_a1)) / len(ev_a0)
    # 64-bit checksum of the WHOLE layer-call output (all H heads: `o` at N = 1, the gathered `full` at N > 1), order-dependent over
    # heads: equal across world sizes and across the overlapped / fallback exchange paths, or something is wrong (tools/scale_run.sh,
    # tests/test_gpu_bench_contract.py)
    torch.cuda.synchronize()
    whole = full if world > 1 else o[0]
    acc = 0
    for h in range(whole.shape[0]):
        bits = whole[h].contiguous().view(torch.int16).to(torch.int64)
        acc = (acc * 1000003 + int((bits * (torch.arange(bits.shape[-1], device=dev) + 1)).sum().item()) + 31 * int(bits.sum().item())) % (1 << 64)
    output_checksum = f"{acc:016x}"

    out = None
    if rank == 0:
        value = flops_call / (ms_step * 1e-3) / 1e12
        kernel_name = "band_attn_m16q_kernel<bf16>" if a.prescaled else BAND_KERNELS.get(a.variant, f"band_attn variant {a.variant}")
        peak = PEAK_BF16_TFLOPS
        if fp8:
            kernel_name, peak = "band_attn_f8_kernel<bf16>", PEAK_FP8_TFLOPS
            pre_ms = sum(x.elapsed_time(y) for x, y in zip(ev_p0, ev_p1)) / len(ev_p0)
            attn_ms -= pre_ms                      # roofline: the attention kernel alone; the pre-pass is reported beside it
            kern_tf = (flops_call * Hl / H) / (attn_ms * 1e-3) / 1e12
        traffic, traffic_src = _pmc_traffic(a.workload, kernel_name)
        kern_tf = (flops_call * Hl / H) / (attn_ms * 1e-3) / 1e12
        out = {
            "metric": "attn_tflops_svg1_block_sparse",
            "value": round(value, 2),
            "unit": "TFLOP/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(ms_step, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": a.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"HunyuanVideo T2V {a.workload} SVG1 sparse layer-call: sample_mse(64 rows) + band attention with "
                            f"fused head placement, cfg=1 H={H} D={D} F={F_} P={P_} ctx={ctx} prompt={L} S={S} band={tf} "
                            f"density={pairs / S / S:.4f}",
                "parallelism": f"heads/{world}" if world > 1 else "single",
                "variant": a.variant,
                "heads": a.heads,
            },
            "algorithmic_tflop_per_step": round(flops_call / 1e12, 3),
            "output_checksum": output_checksum,    # all heads of the last step's output; the same at every N (inputs are seeded per global head)
            # 60 sparse attention layer-calls per second and NOTHING else of a denoise step (no projections / norms / MLPs / VAE)
            "attention_only_steps_per_s": round(1.0 / (60 * ms_step * 1e-3), 4),
            "exchange": None if world == 1 else {
                "rccl_ranks_seen": rccl_ranks_seen,
                "inbound": "3 x all_to_all_single (token shards [H, S/N, D] -> head shards [H/N, S, D]), inside the timed step",
                "inbound_bytes_received_per_rank": int(3 * Hl * (S - (tb - ta)) * D * 2),
                "outbound": "all_gather_into_tensor per row segment of a head chunk, behind completion counters of the single launch"
                            if not mode["chunk_launches"] else "one launch + all-gather per chunk of heads on two streams",
                "outbound_bytes_received_per_rank": int((H - Hl) * S * D * 2),
                "waiter_timeout_ms": WAIT_MS,
                "fallback_to_chunk_launches": mode["fallback"],
                "waiter_timeouts_in_timed_steps": int(timed_out.item()),
            },
            "roofline": {
                "bound": "mfma",
                "kernel": kernel_name,
                "achieved": round(kern_tf, 2),
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": round(kern_tf / peak, 4),
                "kernel_ms": round(attn_ms, 3),
                "traffic": traffic,                           # HBM bytes per launch (PMC)
                "traffic_unit": "B/launch",
                "traffic_source": traffic_src,
                "algorithmic_bytes": 4.0 * H * S * D * 2,     # q, k, v read once + o written once
            },
            "clock": clock_info,   # sustained shader clock during the timed steps (the 2.5 PFLOP/s peak is quoted at 2.4 GHz)
        }
        if ev_q0:
            # the layer-call's other kernel, HBM-bound: K and V of every head streamed once for the 64 sampled rows.  The same
            # K / V stream by LDS-DMA with no arithmetic takes 0.309 ms at this geometry (4.7 TB/s, profiles/r05zj_profiler_bound_ablations.txt)
            prof_ms = sum(x.elapsed_time(y) for x, y in zip(ev_q0, ev_q1)) / len(ev_q0)
            kv_bytes = 2.0 * Hl * S * D * 2
            out["online_profiler"] = {
                "kernels": "profile16_kernel + profile_combine_kernel + profile_finalize_kernel (svg_sample_mse)",
                "ms": round(prof_ms, 4), "algorithmic_bytes": kv_bytes, "GBs": round(kv_bytes / prof_ms / 1e6, 1),
                "frac_of_8TBs": round(kv_bytes / prof_ms / 1e6 / 8000.0, 4), "bound": "hbm",
                "reference": "svg/models/hyvideo/attention.py:376-399 (sample_mse), svg/models/hyvideo/utils.py:47-93 (the two masks)",
            }
        if a.prescaled:
            out["config"]["q_prescaled"] = "--prescaled: q carries sm_scale * log2(e), as the fused prologue writes it under prescale_q = True (opt-in; default: plain q)"
            ob = nat.band_attention(q[:, :2].contiguous(), k[:, :2].contiguous(), v[:, :2].contiguous(), mask,
                                    head_perm_flag=best[:, :2].contiguous(), vid0=0, num_frame=F_, frame_size=P_).float()
            out["prescaled_rel_l2_vs_default_kernel"] = round(((o[:, :2].float() - ob).norm() / ob.norm()).item(), 6)
            del ob

    if rank == 0 and world == 1 and not fp8 and not a.no_ab and a.workload == "hy720p" and a.variant == 0 and a.chunks == 1:
        # In-run calibration: the other schedules of the library — among them the FROZEN reference schedule (variant 6: the two-phase
        # body with the round-1 softmax and operand fetch) — timed beside the default in this process, on this box, same inputs:
        # `default_ms / frozen_ms` shows a schedule gain whatever clock the box settles at.  Attention kernel only, HIP events,
        # one warm-up + mean of 3 launches each.
        def time_attn(fn, n=3):
            fn()
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
            evs[0].record()
            for i in range(n):
                fn()
                evs[i + 1].record()
            torch.cuda.synchronize()
            return evs[0].elapsed_time(evs[n]) / n

        pk = dict(head_perm_flag=best, vid0=0, num_frame=F_, frame_size=P_, out=o)
        ab = {}
        try:
            qs = q_att if a.prescaled else (q.float() * nat.softmax_q_scale(D)).to(q.dtype)
            plain = lambda: nat.band_attention(q, k, v, mask, **pk)                          # noqa: E731  (the default path of the processors)
            pre = lambda: nat.band_attention(qs, k, v, mask, q_prescaled=True, **pk)        # noqa: E731  (opt-in, prescale_q = True)
            ab["default_ms"] = round(time_attn(plain), 3)
            ab["frozen_ms"] = round(time_attn(lambda: nat.band_attention(q, k, v, mask, variant=6, **pk)), 3)
            ab["prescaled_q_ms"] = round(time_attn(pre), 3)
            ab["variant_2_mfma_32x32x16_ms"] = round(time_attn(lambda: nat.band_attention(q, k, v, mask, variant=2, **pk)), 3)
            ab["variant_3_ms"] = round(time_attn(lambda: nat.band_attention(q, k, v, mask, variant=3, **pk)), 3)
            ab["variant_1_ms"] = round(time_attn(lambda: nat.band_attention(q, k, v, mask, variant=1, **pk), n=2), 3)
            ab["default_ms_again"] = round(time_attn(plain), 3)   # (drift over the block)
            ab["default_over_frozen"] = round(ab["default_ms"] / ab["frozen_ms"], 4)
            ab["prescaled_q_over_frozen"] = round(ab["prescaled_q_ms"] / ab["frozen_ms"], 4)
            ab["what"] = ("attention kernel ms, same process / box / inputs: default = svg_band_attention variant 0 on the plain q (= 8: two-phase body on "
                          "16x16x32 MFMAs, max-free softmax, carried operands, one barrier per tile; softmax scale on the fp32 scores — the reference's "
                          "formulation), variant_2 = the same schedule on 32x32x16 MFMAs (the default until round 3), "
                          "prescaled_q = svg_band_attention_prescaled (opt-in: score accumulators started at minus the reference, q pre-scaled by the "
                          "prologue — flex_attention's PRESCALE_QK trade-off), frozen = variant 6 (two-phase body with the round-1 softmax and operand "
                          "fetch, plain q), 3 = one wave per SIMD, 1 = lock-step 4 waves")
            del qs
        except Exception as e:  # noqa: BLE001
            ab["error"] = f"{type(e).__name__}: {str(e)[:300]}"
        out["same_box_ab"] = ab

    if rank == 0 and fp8:
        out["fp8"] = {
            "what": "e4m3 q, k, v (x 448 / amax per head) and probabilities (x 2^8), fp32 softmax and accumulation, bf16 output; inputs "
                    "and output are the bf16 tensors of the bf16 line",
            "prepass_ms": round(pre_ms, 3),
            "prepass": "per-head amax + quantise with fused head placement and per-tile V transpose (reads 3 x, writes 1.5 x one bf16 tensor)",
            "accuracy": "tests/test_gpu_fp8.py: rel. L2 5 % vs the fp32 oracle on randn inputs (2.2 % vs the oracle on dequantised inputs)",
        }
        # accuracy on THIS workload against the bf16 kernel (two heads: a frame-major and a token-major one)
        ob = nat.band_attention(q[:, :2].contiguous(), k[:, :2].contiguous(), v[:, :2].contiguous(), mask,
                                head_perm_flag=best[:, :2].contiguous(), vid0=0, num_frame=F_, frame_size=P_).float()
        df = o[:, :2].float() - ob
        out["fp8"]["rel_l2_vs_bf16_kernel_this_workload"] = round((df.norm() / ob.norm()).item(), 5)
        out["fp8"]["psnr_db_vs_bf16_kernel_this_workload"] = round((20.0 * torch.log10(ob.abs().max() / df.pow(2).mean().sqrt())).item(), 2)
        del ob, df

    if world == 1 and not fp8 and a.workload == "hy720p" and D == 128:
        # BASELINE.json configs[4] beside the bf16 headline: the same layer-call with e4m3 QK^T / PV (python bench.py --dtype fp8 is
        # the full line).  Attention only (pre-pass + kernel), HIP events, mean of 3 after one warm-up.
        try:
            kw8 = dict(vid0=0, num_frame=F_, frame_size=P_, head_perm_flag=best, out=o)
            ref16 = o[:, :2].clone()
            nat.band_attention_fp8(q, k, v, mask, **kw8)
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
            for i in range(3):
                evs[2 * i].record()
                nat.band_attention_fp8(q, k, v, mask, stage=1, **kw8)
                evs[2 * i + 1].record()
                nat.band_attention_fp8(q, k, v, mask, stage=2, **kw8)
            evs[6].record()
            torch.cuda.synchronize()
            pre = sum(evs[2 * i].elapsed_time(evs[2 * i + 1]) for i in range(3)) / 3
            tot = evs[0].elapsed_time(evs[6]) / 3
            out["fp8_hy720p"] = {
                "status": FP8_STATUS,
                "what": "the headline layer-call's attention with e4m3 q, k, v, probabilities (svg_band_attention_fp8); bf16 in / out",
                "prepass_ms": round(pre, 3), "kernel_ms": round(tot - pre, 3), "attention_ms": round(tot, 3),
                "kernel_tflops": round(flops_call / ((tot - pre) * 1e-3) / 1e12, 1),
                "kernel_frac_of_5pflops_fp8": round(flops_call / ((tot - pre) * 1e-3) / 1e12 / PEAK_FP8_TFLOPS, 4),
                "speedup_vs_bf16_kernel": round(attn_ms / tot, 3),
                "rel_l2_vs_bf16_kernel": round(((o[:, :2].float() - ref16.float()).norm() / ref16.float().norm()).item(), 5),
                "psnr_db_vs_bf16_kernel": round((20.0 * torch.log10(ref16.float().abs().max()
                                                                     / (o[:, :2].float() - ref16.float()).pow(2).mean().sqrt())).item(), 2),
            }
            del ref16
        except Exception as e:  # noqa: BLE001
            out["fp8_hy720p"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}

    if world == 1 and not fp8 and a.workload == "hy720p" and not a.no_ab and a.variant == 0:
        # the same kernel on the reference's other SVG1 geometries (production masks of svg/models/{wan,cog}/utils.py, alternating
        # spatial / temporal heads, the processors' default path — plain q — unless --prescaled): Wan 2.1 720p and CogVideoX-v1.5 (head_dim 64)
        try:
            from svg.models.cog import utils as cog_u
            from svg.models.wan import utils as wan_u

            others = {}
            for name, BHo, Do, Fo, Po, ctxo, text_first, mk in (
                    ("wan21_720p_svg1", 40, 128, 21, 3600, 0, False,
                     lambda: wan_u.generate_temporal_head_mask_mod(0, 0, 21, 3600, mul=sparsity_to_width(0.30, 0, 21, 3600))),
                    ("cogvideox15_768p_svg1", 96, 64, 11, 4080, 226, True,
                     lambda: cog_u.generate_temporal_head_mask_mod(226, 11, 4080, mul=sparsity_to_width(0.25, 226, 11, 4080)))):
                So = Fo * Po + ctxo
                mo = mk()
                go = torch.Generator(device=dev).manual_seed(77)
                qo_, ko_, vo_ = (torch.randn(1, BHo, So, Do, device=dev, dtype=torch.bfloat16, generator=go) for _ in range(3))
                if a.prescaled:
                    qo_ = (qo_.float() * nat.softmax_q_scale(Do)).to(torch.bfloat16)
                oo_ = torch.empty_like(qo_)
                besto = torch.tensor([[h % 2 for h in range(BHo)]], device=dev, dtype=torch.int64)
                call = lambda: nat.band_attention(qo_, ko_, vo_, mo, head_perm_flag=besto, vid0=ctxo if text_first else 0,  # noqa: E731
                                                  num_frame=Fo, frame_size=Po, out=oo_, q_prescaled=bool(a.prescaled))
                call()
                evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                evs[0].record()
                for i in range(3):
                    call()
                    evs[i + 1].record()
                torch.cuda.synchronize()
                ms_o = evs[0].elapsed_time(evs[3]) / 3
                fl = 4.0 * Do * BHo * band_pairs(mo, So)
                others[name] = {"H": BHo, "D": Do, "S": So, "density": round(fl / (4.0 * Do * BHo * So * So), 4), "kernel_ms": round(ms_o, 3),
                                "tflops": round(fl / (ms_o * 1e-3) / 1e12, 1), "frac_of_2500": round(fl / (ms_o * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)}
                del qo_, ko_, vo_, oo_
            out["svg1_other_models"] = others
        except Exception as e:  # noqa: BLE001
            out["svg1_other_models"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        torch.cuda.empty_cache()

    # ---- extras on rank 0 at N = 1: dense comparator on the same GPU, CPU baseline ----
    if world == 1 and not a.no_dense:
        dmask = nat.BandMask(real_len=V + L, band=S + 1, colfull_lo=0, colfull_hi=0, rowfull_lo=0, rowfull_hi=0)
        dense_call = (lambda: nat.band_attention_fp8(q, k, v, dmask, out=o)) if fp8 else \
            (lambda: nat.band_attention(q, k, v, dmask, variant=a.variant, out=o))
        dense_call()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        dense_call()
        e1.record()
        torch.cuda.synchronize()
        dms = e0.elapsed_time(e1)
        dense_pairs = (V + L) ** 2 + (ctx - L) ** 2
        out["dense_same_gpu"] = {
            "kernel": "our dense mode (cu_seqlens [0, V+L, S])" + (" fp8, pre-pass included" if fp8 else ""),
            "ms": round(dms, 3),
            "tflops": round(4.0 * D * H * dense_pairs / (dms * 1e-3) / 1e12, 2),
            "speedup_sparse_vs_dense": round(dms / ms_step, 3),
        }
        try:
            import torch.nn.functional as Fn

            from torch.nn.attention import SDPBackend, sdpa_kernel

            qs, ks, vs = q[:, :4].contiguous(), k[:, :4].contiguous(), v[:, :4].contiguous()
            # flash backend only: the math fallback would materialise a [4, S, S] score tensor (>100 GB)
            with sdpa_kernel([SDPBackend.FLASH_ATTENTION]):
                Fn.scaled_dot_product_attention(qs, ks, vs)
                torch.cuda.synchronize()
                e0.record()
                Fn.scaled_dot_product_attention(qs, ks, vs)
                e1.record()
                torch.cuda.synchronize()
            sms = e0.elapsed_time(e1) * (H / 4)
            out["dense_same_gpu"]["torch_sdpa_ms_scaled_from_4_heads"] = round(sms, 3)
            out["dense_same_gpu"]["speedup_sparse_vs_torch_sdpa"] = round(sms / ms_step, 3)
        except Exception as e:  # noqa: BLE001
            out["dense_same_gpu"]["torch_sdpa_error"] = str(e)[:200]
    extras_full = a.workload == "hy720p" and a.extras == "auto"
    extras_small = a.extras == "small"
    failed_extras = []          # an extras block that raises is reported in the line AND fails the run (exit code 3 after the line is printed)
    if world == 1 and not a.no_svg2 and (extras_full or extras_small):
        # BASELINE.json configs[2]: one SVG2 / SAP layer-call of Wan 2.1 720p (k-means, block map, variable-block attention)
        del q, k, v, o
        torch.cuda.empty_cache()
        svg2_wl, svg2_steps = ("wan720p", 3) if extras_full else ("small", 1)
        for key, f8 in (("svg2_wan720p", fp8),) + ((("svg2_wan720p_fp8", True),) if not fp8 else ()):   # fp8: BASELINE.json configs[4]
            try:
                import bench_svg2

                if os.environ.get("SVG_BENCH_TEST_BREAK_EXTRAS"):    # test hook (tests/test_gpu_bench_contract.py): the failure path below
                    raise RuntimeError("SVG_BENCH_TEST_BREAK_EXTRAS")
                out[key] = bench_svg2.measure(svg2_wl, steps=svg2_steps, warmup=1, fp8=f8)
                if f8 and key.endswith("_fp8"):
                    out[key]["status"] = FP8_STATUS
            except Exception as e:  # noqa: BLE001
                out[key] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
                failed_extras.append(key)
        if extras_full and not fp8:
            # the SAP layer-call of HunyuanVideo 720p (QC 400, KC 1000 + the prompt / unused-prompt pseudo clusters of dynamic_map_post_processing,
            # ref svg/models/hyvideo/attention.py:657-702, scripts/hyvideo/hyvideo_t2v_720p_sap.sh:12-17)
            try:
                out["svg2_hy720p"] = bench_svg2.measure("hy720p", steps=2, warmup=1)
            except Exception as e:  # noqa: BLE001
                out["svg2_hy720p"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
                failed_extras.append("svg2_hy720p")
            torch.cuda.empty_cache()
    if world == 1 and not a.no_step and (extras_full or extras_small):
        # BASELINE.json configs[3] at N = 1: a measured denoise step of the synthetic 60-block HunyuanVideo stack (GEMMs + glue +
        # attention), sparse and dense; `denoise_steps_per_s` here is measured, unlike attention_only_steps_per_s above
        torch.cuda.empty_cache()
        try:
            import bench_step

            if extras_full:
                out["denoise_step_hy720p"] = bench_step.measure(steps=3, warmup=1)
            else:
                out["denoise_step_hy720p"] = bench_step.measure(steps=1, warmup=0, n_double=1, n_single=1)
        except Exception as e:  # noqa: BLE001
            out["denoise_step_hy720p"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
            failed_extras.append("denoise_step_hy720p")
        torch.cuda.empty_cache()
        # BASELINE.json configs[2], the metric's second half: a measured denoise step of the synthetic 40-block Wan 2.1 14B stack with SVG2 / SAP
        # attention (block glue, RMSNorm-across-heads + complex-RoPE prologue, warm-started two-stream k-means, block map, variable-block
        # attention together), sparse and dense (bench_step.measure_wan)
        try:
            import bench_step

            if extras_full:
                out["denoise_step_wan720p_svg2"] = bench_step.measure_wan(steps=2, warmup=1)
            else:
                out["denoise_step_wan720p_svg2"] = bench_step.measure_wan(
                    steps=1, warmup=0, geo=bench_step.WanGeo(F=5, P=600, hid=512, heads=4, hd=128, ffn=1024, text=64, layers=2, qc=20, kc=40))
        except Exception as e:  # noqa: BLE001
            out["denoise_step_wan720p_svg2"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
            failed_extras.append("denoise_step_wan720p_svg2")
        torch.cuda.empty_cache()
    if world == 1 and not a.no_hbm and (extras_full or extras_small):
        # the HBM-bound rows of SURVEY §8(d) (placement, inverse placement, SVG2 gather / scatter, label sort, prologue, block glue) at
        # production sizes: ms, GB/s, fraction of the 8 TB/s spec and of the 6.29 TB/s measured copy (bench_hbm.py)
        try:
            import bench_hbm

            out["hbm_kernels"] = bench_hbm.measure("full" if extras_full else "small", reps=5 if extras_full else 2)
        except Exception as e:  # noqa: BLE001
            out["hbm_kernels"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
            failed_extras.append("hbm_kernels")
        torch.cuda.empty_cache()
    if world > 1 and not a.no_step and not fp8:
        # BASELINE.json configs[3] at N > 1: the denoise step token-sharded over the ranks (bench_step.py: tokens/N for norms, GEMMs,
        # prologue and glue, heads/N for the attention, all-to-all exchanges either side of it, one all-gather of the hidden states per
        # step).  Every rank takes part; rank 0 reports (the step time is the MAX over ranks).
        torch.cuda.empty_cache()
        try:
            import bench_step

            if a.workload == "hy720p":
                sd = bench_step.measure(steps=2, warmup=1, rank=rank, world=world, kinds=("sparse",), host_staged=smoke)
            else:   # reduced workloads (tests, smoke runs): a 1 + 1 block stack on a small geometry
                sd = bench_step.measure(steps=1, warmup=1, n_double=1, n_single=1, rank=rank, world=world, kinds=("sparse",), host_staged=smoke,
                                        geo=bench_step.StepGeo(F=F_, P=P_, ctx=ctx, L=L, hid=H * D, heads=H, hd=D, mlp=2 * H * D))
            if rank == 0:
                out["denoise_step_hy720p"] = sd
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                out["denoise_step_hy720p"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
            raise
        # BASELINE.json configs[2] at N > 1: the Wan 2.1 SVG2 step, tokens/N + heads/N (40 heads: N in 2, 4, 8), k-means stopping rule all-reduced
        torch.cuda.empty_cache()
        try:
            if a.workload == "hy720p":
                sw = bench_step.measure_wan(steps=2, warmup=1, rank=rank, world=world, kinds=("sparse",), host_staged=smoke)
            else:
                sw = bench_step.measure_wan(steps=1, warmup=0, rank=rank, world=world, kinds=("sparse",), host_staged=smoke,
                                            geo=bench_step.WanGeo(F=5, P=600, hid=512, heads=4, hd=128, ffn=1024, text=64, layers=2, qc=20, kc=40))
            if rank == 0:
                out["denoise_step_wan720p_svg2"] = sw
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                out["denoise_step_wan720p_svg2"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
            raise
    if world == 1 and not a.no_cpu:
        out["cpu_baseline"] = cpu_baseline(H, D, S)
    elif rank == 0:
        out["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if failed_extras:
        print(f"bench.py: extras block(s) failed: {failed_extras} (their 'error' strings are in the line above)", file=sys.stderr)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
