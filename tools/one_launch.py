#!/usr/bin/env python3
"""One warm-up + one launch of the band kernel on the headline workload (for rocprofv3 --pmc passes of a single setting):
python tools/one_launch.py [plain|pre] [variant]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
import torch  # noqa: E402
from svg import _native as nat  # noqa: E402
from svg.models.hyvideo import utils as hy  # noqa: E402
from svg.models.hyvideo.utils import sparsity_to_width  # noqa: E402

pre = len(sys.argv) > 1 and sys.argv[1] == "pre"
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda", 0)
F_, P_, ctx, BH, D = 33, 3600, 256, 24, 128
S = F_ * P_ + ctx
mask = hy.generate_temporal_head_mask_mod(ctx, 64, F_, P_, mul=sparsity_to_width(0.25, ctx, F_, P_))
g = torch.Generator(device=dev).manual_seed(0)
q, k, v = (torch.randn(1, BH, S, D, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3))
if pre:
    q = (q.float() * nat.softmax_q_scale(D)).to(q.dtype)
best = torch.tensor([[h % 2 for h in range(BH)]], device=dev, dtype=torch.int64)
o = torch.empty_like(q)
for _ in range(2):
    nat.band_attention(q, k, v, mask, q_prescaled=pre, variant=variant, head_perm_flag=best, vid0=0, num_frame=F_, frame_size=P_, out=o)
torch.cuda.synchronize()
print("ok", float(o.float().abs().mean()))
