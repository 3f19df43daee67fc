#!/usr/bin/env python3
"""Same-process A/B on the headline workload: the two-phase band kernel on 32x32x16 MFMAs (variant 2, plain q; and its pre-scaled form)
against the same schedule on 16x16x32 MFMAs (variant 8, csrc/attn_m16.h) and its pre-scaled form (svg_band_attention_prescaled): ms per launch, sustained clock, Mcycles, and the outputs
against each other.  usage: python tools/ab_m16.py [launches per leg, default 4] [rounds, default 2]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
import torch  # noqa: E402
from svg import _native as nat  # noqa: E402
from svg.models.hyvideo import utils as hy  # noqa: E402
from svg.models.hyvideo.utils import sparsity_to_width  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = torch.device("cuda", 0)
    F_, P_, ctx, BH, D = 33, 3600, 256, 24, 128
    S = F_ * P_ + ctx
    mask = hy.generate_temporal_head_mask_mod(ctx, 64, F_, P_, mul=sparsity_to_width(0.25, ctx, F_, P_))
    g = torch.Generator(device=dev).manual_seed(0)
    q, k, v = (torch.randn(1, BH, S, D, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3))
    qs = (q.float() * nat.softmax_q_scale(D)).to(q.dtype)
    best = torch.tensor([[h % 2 for h in range(BH)]], device=dev, dtype=torch.int64)
    pk = dict(head_perm_flag=best, vid0=0, num_frame=F_, frame_size=P_)
    legs = [("32x32x16 plain q (variant 2)", lambda o: nat.band_attention(q, k, v, mask, variant=2, out=o, **pk)),
            ("16x16x32 plain q (variant 8)", lambda o: nat.band_attention(q, k, v, mask, variant=8, out=o, **pk)),
            ("16x16x32 pre-scaled q (opt-in)", lambda o: nat.band_attention(qs, k, v, mask, q_prescaled=True, out=o, **pk))]
    outs = {}
    probe = nat.ClockProbe(dev)
    for rnd in range(rounds):
        for name, fn in legs:
            o = torch.empty_like(q)
            fn(o)
            torch.cuda.synchronize()
            probe.start(max_ms=20000)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn(o)
            e1.record()
            probe.arm_stop()
            e1.synchronize()
            mhz = probe.result()
            ms = e0.elapsed_time(e1) / n
            print(f"round {rnd} {name:42s}: {ms:7.3f} ms / launch, sustained {mhz} MHz, {ms * 1e-3 * (mhz or 0):7.2f} Mcycles", flush=True)
            outs[name] = o
    a, b = outs[legs[0][0]].float(), outs[legs[1][0]].float()
    print(f"rel L2 between variant 2 and variant 8: {((a - b).norm() / a.norm()).item():.3e}, max abs {(a - b).abs().max().item():.3e}; "
          f"finite: {bool(torch.isfinite(b).all())}", flush=True)


if __name__ == "__main__":
    main()
