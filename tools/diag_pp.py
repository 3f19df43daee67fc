import sys, torch
sys.path.insert(0, "sparse-videogen_amd"); sys.path.insert(0, ".")
from svg import _native as nat
from oracle import svg_oracle as O
torch.manual_seed(0)
dev = torch.device("cuda", 0)
VAR = int(sys.argv[1]) if len(sys.argv) > 1 else 3   # schedule under test (include/svg_attn.h: 1 lock-step, 2 ping-pong, 3 one wave per SIMD)
for (S, D, dt) in [(256, 128, torch.bfloat16), (320, 128, torch.bfloat16), (790, 128, torch.bfloat16), (790, 64, torch.float16), (2048, 128, torch.bfloat16)]:
    q, k, v = (torch.randn(1, 2, S, D).to(dt) for _ in range(3))
    m = nat.BandMask(**O.dense_band_params(S))
    o0 = nat.band_attention(q.to(dev), k.to(dev), v.to(dev), m, variant=1).float().cpu()
    o1 = nat.band_attention(q.to(dev), k.to(dev), v.to(dev), m, variant=VAR).float().cpu()
    ref = O.masked_attention(q, k, v, None).float()
    e0 = (o0 - ref).abs().amax(dim=(0, 1, 3)); e1 = (o1 - ref).abs().amax(dim=(0, 1, 3))
    print(S, D, dt, "v1 max", float(e0.max()), "var max", float(e1.max()), "nan", int(torch.isnan(o1).sum()))
    bad = (e1 > 0.02).nonzero().flatten()
    if len(bad): print("  bad rows:", bad[:20].tolist(), "...", len(bad))
S, D, dt = 256, 128, torch.bfloat16
q, k, v = (torch.randn(1, 1, S, D).to(dt) for _ in range(3))
m = nat.BandMask(**O.dense_band_params(S))
o1 = nat.band_attention(q.to(dev), k.to(dev), v.to(dev), m, variant=VAR).float().cpu()[0, 0]
nz = torch.isnan(o1).nonzero()
print("nan count", len(nz), "rows", sorted(set(nz[:, 0].tolist()))[:40], "cols", sorted(set(nz[:, 1].tolist()))[:40])
ref = O.masked_attention(q, k, v, None).float()[0, 0]
err = (o1 - ref).abs(); err[torch.isnan(err)] = 0
print("max err non-nan", float(err.max()))
