#!/usr/bin/env python3
"""A/B of the experiment configurations of the 16x16x32 body (a -DSVG_M16_EXPERIMENTS build: SVG_M16_CFG = prio * 2 + onebar, read per
launch): ms per launch and sustained clock on the headline workload.  usage: SVG_ATTN_LIB=.../libsvgattn_m16x.so python tools/ab_m16_cfg.py [cfgs]"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
import torch  # noqa: E402
from svg import _native as nat  # noqa: E402
from svg.models.hyvideo import utils as hy  # noqa: E402
from svg.models.hyvideo.utils import sparsity_to_width  # noqa: E402

cfgs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["2", "0", "4", "3", "1", "5"]
names = {"0": "no priority, 2 barriers", "1": "no priority, 1 barrier", "2": "priority in M, 2 barriers (first version)", "3": "priority in M, 1 barrier",
         "4": "priority in N, 2 barriers", "5": "priority in N, 1 barrier"}
dev = torch.device("cuda", 0)
F_, P_, ctx, BH, D = 33, 3600, 256, 24, 128
S = F_ * P_ + ctx
mask = hy.generate_temporal_head_mask_mod(ctx, 64, F_, P_, mul=sparsity_to_width(0.25, ctx, F_, P_))
g = torch.Generator(device=dev).manual_seed(0)
q, k, v = (torch.randn(1, BH, S, D, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3))
best = torch.tensor([[h % 2 for h in range(BH)]], device=dev, dtype=torch.int64)
o = torch.empty_like(q)
ref = nat.band_attention(q, k, v, mask, variant=2, head_perm_flag=best, vid0=0, num_frame=F_, frame_size=P_).float()
probe = nat.ClockProbe(dev)
for rnd in range(2):
    for c in cfgs:
        os.environ["SVG_M16_CFG"] = c
        fn = lambda: nat.band_attention(q, k, v, mask, variant=8, head_perm_flag=best, vid0=0, num_frame=F_, frame_size=P_, out=o)  # noqa: E731
        fn()
        torch.cuda.synchronize()
        probe.start(max_ms=20000)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            fn()
        e1.record()
        probe.arm_stop()
        e1.synchronize()
        mhz = probe.result()
        ms = e0.elapsed_time(e1) / 4
        err = ((o.float() - ref).norm() / ref.norm()).item()
        print(f"round {rnd} cfg {c} ({names.get(c, '?'):42s}): {ms:7.3f} ms, {mhz} MHz, {ms * 1e-3 * (mhz or 0):6.2f} Mcycles, rel L2 vs variant 2 {err:.2e}", flush=True)
