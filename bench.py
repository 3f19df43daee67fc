#!/usr/bin/env python3
"""Headline benchmark: SVG1 block-sparse attention of one HunyuanVideo 720p / 129-frame layer-call on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload hy720p|hy480p|tiny] [--no-cpu] [--no-dense]

Workload (BASELINE.json configs[1]): q, k, v = randn([1, 24, 119056, 128], bf16) (F=33, P=3600, ctx=256, prompt 64),
sparsity 0.25 -> band 15616 tokens, half of the heads temporal (fused layout transformation inside the kernel).
One *step* = one sparse layer-call = online profiler (sample_mse, 64 rows) + band attention with fused placement, i.e.
exactly what Hunyuan_SVGAttn_Processor2_0.attention_core_logic does per layer in the sparse branch
(ref: svg/models/hyvideo/attention.py:507-524).  Inputs are resident in HBM before the timed region.

metric  = attention TFLOP/s, algorithmic: 4 * D * H * (#unmasked (q,k) pairs) per layer-call (42.78 TFLOP
          at L=64) divided by wall time per step; `denoise_steps_per_s` (60 layer-calls per denoise step, attention only)
          is reported beside it.
roofline: bound = MFMA (dense bf16 peak 2.5 PFLOP/s); `achieved` = algorithmic FLOPs of the dominant kernel
          (band_attn_kernel) / its mean launch duration measured with HIP events on the launch stream.
cpu_baseline: the reference's CPU-capable dense path torch SDPA (ref: svg/models/wan/attention.py:279-281) on the host
          cores, bf16, on a bounded sample (one head, shortened sequence), scaled — see `sample`.
N > 1   : heads are independent units; rank r owns heads r::N of the same layer-call (strong scaling), no data-path
          collective during attention, one all-gather of the attention output per step (the exchange the next op,
          `to_out`, needs) over RCCL.
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
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16, /opt/skills/guides/MI355X_MICROARCH.md

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


def _pmc_traffic(workload: str):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction +
    WRITE_SIZE, tools/gpu_pmc.sh).  PMC collection needs its own rocprofv3 runs, so bench.py reports the value of the
    committed profile (profiles/r01_pmc_traffic.json) for the headline workload and null otherwise."""
    p = ROOT / "profiles" / "r01_pmc_traffic.json"
    if workload != "hy720p" or not p.exists():
        return None
    try:
        return float(json.loads(p.read_text())["traffic_bytes_per_launch"])
    except Exception:  # noqa: BLE001
        return None


def cpu_baseline(H: int, D: int, S: int, seconds_target: float = 15.0):
    """torch SDPA (dense, bf16) on the host cores: one head, sequence shortened so that it runs ~10-30 s."""
    import torch.nn.functional as F

    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    s = 8192
    q, k, v = (torch.randn(1, 1, s, D, dtype=torch.bfloat16) for _ in range(3))
    F.scaled_dot_product_attention(q, k, v)  # warm
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < seconds_target and reps < 64:
        F.scaled_dot_product_attention(q, k, v)
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    tflops = 4.0 * s * s * D / dt / 1e12
    return {
        "value": round(tflops, 4),
        "unit": "TFLOP/s",
        "cores": cores,
        "kind": "reference",
        "sample": f"torch SDPA dense bf16, 1 head, S={s}, D={D}, {reps} reps (reference CPU path "
                  f"svg/models/wan/attention.py:279-281); a full {H}-head S={S} layer-call is "
                  f"{4.0 * S * S * D * H / 1e12:.1f} TFLOP dense = {4.0 * S * S * D * H / 1e12 / tflops:.0f} s at this rate",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="hy720p", choices=sorted(WORKLOADS))
    ap.add_argument("--variant", type=int, default=0, help="0: 8 waves/WG, 1: 4 waves/WG")
    ap.add_argument("--chunks", type=int, default=1, help="N = 1 only: split the launch into this many head chunks on two streams")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-dense", action="store_true")
    ap.add_argument("--no-profiler", action="store_true", help="time the attention kernel only")
    ap.add_argument("--heads", default="alt", choices=["alt", "spatial", "temporal"], help="best_mask_idx pattern")
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
    from svg.distributed import chunked_head_layout, gather_chunk

    # (one head per chunk: with one launch + completion counters a chunk costs a waiter and an all-gather call, not a kernel
    #  launch, and the exposed gather at the end of the step is that of ONE head instead of a third of the rank's heads)
    n_chunks, n_per, my_heads = chunked_head_layout(H, rank, world, max_chunks=24)
    if world == 1 and a.chunks > 1:   # exercise the chunked two-stream launch path on one GPU (no collective)
        assert H % a.chunks == 0
        n_chunks, n_per = a.chunks, H // a.chunks
    Hl = len(my_heads)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    q, k, v = (torch.randn(1, Hl, S, D, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3))
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

    def step(timed: bool):
        if not a.no_profiler:
            mse = nat.sample_mse(q[0], k[0], v[0], rows, prof)
            _ = mse.argmin(0)  # best_mask_idx (kept on device; the bench uses the fixed alternating pattern)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        works = []
        main = torch.cuda.current_stream()
        kw = dict(vid0=0, num_frame=F_, frame_size=P_, variant=a.variant)
        if side is None:
            nat.band_attention(q, k, v, mask, head_perm_flag=best, out=o, **kw)
        elif chunk_launches:
            for c in range(n_chunks):
                sl = slice(c * n_per, (c + 1) * n_per)
                st = side[c % 2]
                st.wait_stream(main)   # inputs and the profiler's result are produced on the main stream
                with torch.cuda.stream(st):
                    nat.band_attention(q[:, sl], k[:, sl], v[:, sl], mask, head_perm_flag=best[:, sl].contiguous(), out=o[:, sl], **kw)
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
                            nat.wait_counters(cnt[h, sg:sg + 1], seg_targets[sg])
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

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    if smoke and world > 1:   # the gathered output must hold this rank's heads as they are after the launch (an early gather would not)
        mine_idx = torch.tensor(my_heads, device=dev)
        assert torch.equal(full.index_select(0, mine_idx), o[0]), "all-gather ran ahead of the attention kernel"
    ms_step = dt / a.steps * 1e3
    attn_ms = sum(x.elapsed_time(y) for x, y in zip(ev_a0, ev_a1)) / len(ev_a0)

    out = None
    if rank == 0:
        value = flops_call / (ms_step * 1e-3) / 1e12
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
            "dtype": "bf16",
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
            "denoise_steps_per_s": round(1.0 / (60 * ms_step * 1e-3), 4),
            "roofline": {
                "bound": "mfma",
                "kernel": "band_attn_pp2_kernel<bf16,128>" if a.variant in (0, 128) else f"band_attn variant {a.variant}",
                "achieved": round(kern_tf, 2),
                "peak": PEAK_BF16_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(kern_tf / PEAK_BF16_TFLOPS, 4),
                "kernel_ms": round(attn_ms, 3),
                "traffic": _pmc_traffic(a.workload),          # HBM bytes per launch (PMC), profiles/r01_pmc_traffic.json
                "traffic_unit": "B/launch",
                "algorithmic_bytes": 4.0 * H * S * D * 2,     # q, k, v read once + o written once
            },
        }

    # ---- extras on rank 0 at N = 1: dense comparator on the same GPU, CPU baseline ----
    if world == 1 and not a.no_dense:
        dmask = nat.BandMask(real_len=V + L, band=S + 1, colfull_lo=0, colfull_hi=0, rowfull_lo=0, rowfull_hi=0)
        nat.band_attention(q, k, v, dmask, variant=a.variant, out=o)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        nat.band_attention(q, k, v, dmask, variant=a.variant, out=o)
        e1.record()
        torch.cuda.synchronize()
        dms = e0.elapsed_time(e1)
        dense_pairs = (V + L) ** 2 + (ctx - L) ** 2
        out["dense_same_gpu"] = {
            "kernel": "our dense mode (cu_seqlens [0, V+L, S])",
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
    if world == 1 and not a.no_cpu:
        out["cpu_baseline"] = cpu_baseline(H, D, S)
    elif rank == 0:
        out["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
