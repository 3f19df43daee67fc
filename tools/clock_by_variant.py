#!/usr/bin/env python3
"""Sustained shader clock and cycle count of every schedule of the SVG1 band kernel on the headline workload (HunyuanVideo 720p,
24 heads, alternating spatial / temporal masks).  The chip is power-limited under the 16-bit attention kernels, so
milliseconds mix two things: how many cycles a schedule needs, and which clock the power management grants it.  This separates
them: per variant, N back-to-back launches beside svg_debug_clock_probe (s_memtime shader ticks over 100 MHz wall ticks) ->
  ms per launch, sustained MHz, Mcycles per launch (= ms x MHz), frac at the granted clock (= FLOPs / (cycles x dense peak per cycle)),
  frac of the nominal peak (= the roofline figure).
Usage: python tools/clock_by_variant.py [launches per variant, default 8]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch  # noqa: E402
from svg import _native as nat  # noqa: E402
from svg.models.hyvideo import utils as hy  # noqa: E402
from svg.models.hyvideo.utils import sparsity_to_width  # noqa: E402
from svg.models.cog import utils as cog  # noqa: E402
from svg1_models import pairs  # noqa: E402

PEAK_FLOP_PER_CYCLE = 2.5e15 / 2.4e9     # dense bf16 MFMA peak of the chip at its nominal 2.4 GHz
PEAK_F8_PER_CYCLE = 5.0e15 / 2.4e9


def run(n, dev, name, BH, D, F_, P_, ctx, vid0, mask, only=None):
    S = F_ * P_ + ctx
    dmask = nat.BandMask(real_len=mask.real_len, band=S + 1, colfull_lo=0, colfull_hi=0, rowfull_lo=0, rowfull_hi=0)
    g = torch.Generator(device=dev).manual_seed(0)
    q, k, v = (torch.randn(1, BH, S, D, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3))
    qs = (q.float() * nat.softmax_q_scale(D)).to(q.dtype)
    o = torch.empty_like(q)
    best = torch.tensor([[h % 2 for h in range(BH)]], device=dev, dtype=torch.int64)
    pk = dict(head_perm_flag=best, vid0=vid0, num_frame=F_, frame_size=P_, out=o)
    fl = 4.0 * D * BH * pairs(mask, S)
    dfl = 4.0 * D * BH * pairs(dmask, S)
    cases = [
        ("two-phase on 16x16x32 MFMAs, plain q (default at head_dim 128; at head_dim 64 the default is the two-phase 32x32x16 body, four waves per SIMD)", fl, PEAK_FLOP_PER_CYCLE,
         lambda: nat.band_attention(q, k, v, mask, **pk)),
        ("two-phase on 32x32x16 MFMAs, plain q (variant 2)", fl, PEAK_FLOP_PER_CYCLE, lambda: nat.band_attention(q, k, v, mask, variant=2, **pk)),
        ("pre-scaled q, opt-in (PRE form of the default body)", fl, PEAK_FLOP_PER_CYCLE, lambda: nat.band_attention(qs, k, v, mask, q_prescaled=True, **pk)),
        ("frozen round-1 body (variant 6)", fl, PEAK_FLOP_PER_CYCLE, lambda: nat.band_attention(q, k, v, mask, variant=6, **pk)),
        ("one wave per SIMD, 64 rows (variant 3)", fl, PEAK_FLOP_PER_CYCLE, lambda: nat.band_attention(q, k, v, mask, variant=3, **pk)),
        ("lock-step 4 x 32 rows (variant 1)", fl, PEAK_FLOP_PER_CYCLE, lambda: nat.band_attention(q, k, v, mask, variant=1, **pk)),
        ("dense mode, default body, plain q", dfl, PEAK_FLOP_PER_CYCLE, lambda: nat.band_attention(q, k, v, dmask, out=o)),
        ("fp8 two-phase (attention stage only)", fl, PEAK_F8_PER_CYCLE, None),
    ]
    probe = nat.ClockProbe(dev)
    print(f"\n{name}: S = {S}, {BH} heads, head_dim {D}")
    print("| schedule | ms / launch | sustained MHz | Mcycles / launch | frac at the granted clock | frac of nominal peak |")
    print("|---|---|---|---|---|---|")
    for i, (name, flops, peak_cyc, fn) in enumerate(cases):
        if only is not None and i not in only:
            continue
        reps = n if flops == fl else max(2, n // 4)
        if fn is None:   # (the fp8 case)
            nat.band_attention_fp8(q, k, v, mask, stage=1, **pk)     # pre-pass once (the library caches the workspace per stream)

            def fn():
                return nat.band_attention_fp8(q, k, v, mask, stage=2, **pk)
        fn()
        fn()
        torch.cuda.synchronize()
        probe.start(max_ms=20000)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        probe.arm_stop()
        e1.synchronize()
        mhz = probe.result()
        ms = e0.elapsed_time(e1) / reps
        if not mhz:
            print(f"| {name} | {ms:.3f} | n/a | | | {flops / (ms * 1e-3) / (peak_cyc * 2.4e9):.4f} |", flush=True)
            continue
        cyc = ms * 1e-3 * mhz * 1e6
        print(f"| {name} | {ms:.3f} | {mhz:.0f} | {cyc / 1e6:.2f} | {flops / (cyc * peak_cyc):.4f} | "
              f"{flops / (ms * 1e-3) / (peak_cyc * 2.4e9):.4f} |", flush=True)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda", 0)
    F_, P_, ctx = 33, 3600, 256
    run(n, dev, "HunyuanVideo 720p 129f", 24, 128, F_, P_, ctx, 0,
        hy.generate_temporal_head_mask_mod(ctx, 64, F_, P_, mul=sparsity_to_width(0.25, ctx, F_, P_)))
    torch.cuda.empty_cache()
    F_, P_, ctx = 11, 4080, 226
    run(n, dev, "CogVideoX-v1.5 768p 81f", 96, 64, F_, P_, ctx, ctx,
        cog.generate_temporal_head_mask_mod(ctx, F_, P_, mul=sparsity_to_width(0.25, ctx, F_, P_)), only=(0, 2, 4))


if __name__ == "__main__":
    main()
