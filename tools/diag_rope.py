import sys, torch
sys.path.insert(0, "sparse-videogen_amd"); sys.path.insert(0, ".")
from svg import _native as nat
from oracle import svg_oracle as O
torch.manual_seed(0)
for kind, dt in (("cossin", torch.bfloat16), ("complex", torch.float16), ("complex", torch.bfloat16)):
    bsz, H, S, D, L = 1, 16, 151, 64, 15
    q = torch.randn(bsz, H, S, D).to(dt); k = torch.randn(bsz, H, S, D).to(dt)
    cols = D // 2 if kind == "complex" else D
    a, b = torch.randn(S - L, cols), torch.randn(S - L, cols)
    gq, gk = q.cuda(), k.cuda()
    fn = {"cossin": nat.apply_qk_rope_inplace_cossin, "complex": nat.apply_qk_rope_inplace_cossin_complex}[kind]
    fn(gq, gk, a.cuda(), b.cuda(), L)
    rq, rk = O.apply_qk_rope(q, k, a, b, L, kind)
    print("k mismatch frac", (gk.cpu() != rk).float().mean().item())
    d = (gq.cpu() != rq)
    print(kind, dt, "mismatch frac", d.float().mean().item(), "max abs", (gq.cpu().float() - rq.float()).abs().max().item())
    idx = d.nonzero()[:5]
    for i in idx.tolist():
        print("  at", i, gq.cpu()[tuple(i)].item(), rq[tuple(i)].item(), "orig", q[tuple(i)].item())
