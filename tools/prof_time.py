import sys, torch
sys.path.insert(0, "sparse-videogen_amd")
from svg import _native as nat
H, D, F_, P_, ctx = 24, 128, 33, 3600, 256
V = F_ * P_; S = V + ctx
dev = torch.device("cuda", 0)
q, k, v = (torch.randn(1, H, S, D, device=dev, dtype=torch.bfloat16) for _ in range(3))
rows = torch.randint(0, 10000, (64,), device=dev)
bb = int((P_ * 1.5) // 128)
for emu in (1, 0):
    for which in ("both", "spatial2", "temporal2"):
        prof = nat.ProfileDesc(0, F_, P_, emu)
        a = nat.ProfileVariant(0, 0, V, bb, 0, V, S)
        b = nat.ProfileVariant(1, 0, V, bb, 0, V, S)
        prof.variant[0] = a if which != "temporal2" else b
        prof.variant[1] = b if which != "spatial2" else a
        nat.sample_mse(q[0], k[0], v[0], rows, prof); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): nat.sample_mse(q[0], k[0], v[0], rows, prof)
        e1.record(); torch.cuda.synchronize()
        print(f"emulate={emu} masks={which}: {e0.elapsed_time(e1)/5:.3f} ms")
