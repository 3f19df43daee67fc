#!/usr/bin/env python3
"""Same-process A/B of the cyclic sweep start of the band kernel (csrc/band_policy.h init(): SVG_BAND_ROTATE, read per launch) on the
headline workload: ms per launch with the switch off / on (alternating), the two outputs against each other, and the sustained clock.
usage: python tools/ab_rotate.py [launches per leg, default 4]"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
import torch  # noqa: E402
from svg import _native as nat  # noqa: E402
from svg.models.hyvideo import utils as hy  # noqa: E402
from svg.models.hyvideo.utils import sparsity_to_width  # noqa: E402


ROTS = tuple(os.environ.get("SVG_AB_ROTS", "0,1,2,3").split(","))   # 0: off; n: cyclic start, n - 1 key tiles of stagger between members


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda", 0)
    F_, P_, ctx, BH, D = 33, 3600, 256, 24, 128
    S = F_ * P_ + ctx
    mask = hy.generate_temporal_head_mask_mod(ctx, 64, F_, P_, mul=sparsity_to_width(0.25, ctx, F_, P_))
    g = torch.Generator(device=dev).manual_seed(0)
    q, k, v = (torch.randn(1, BH, S, D, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3))
    qs = (q.float() * nat.softmax_q_scale(D)).to(q.dtype)
    best = torch.tensor([[h % 2 for h in range(BH)]], device=dev, dtype=torch.int64)
    outs = {}
    probe = nat.ClockProbe(dev)
    for name, fn in (("plain q (default)", lambda o: nat.band_attention(q, k, v, mask, head_perm_flag=best, vid0=0, num_frame=F_, frame_size=P_, out=o)),
                     ("pre-scaled q", lambda o: nat.band_attention(qs, k, v, mask, q_prescaled=True, head_perm_flag=best, vid0=0, num_frame=F_, frame_size=P_, out=o))):
        for rnd in range(2):
            for rot in ROTS:
                os.environ["SVG_BAND_ROTATE"] = rot
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
                print(f"{name:18s} rotate {rot} round {rnd}: {e0.elapsed_time(e1) / n:7.3f} ms / launch, sustained {mhz} MHz", flush=True)
                outs[(name, rot)] = o
        a, b = outs[(name, ROTS[0])].float(), outs[(name, ROTS[-1])].float()
        print(f"{name:18s} rel L2 between the two sweep orders: {((a - b).norm() / a.norm()).item():.3e}, max abs {(a - b).abs().max().item():.3e}", flush=True)
    os.environ.pop("SVG_BAND_ROTATE", None)


if __name__ == "__main__":
    main()
