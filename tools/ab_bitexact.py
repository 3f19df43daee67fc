#!/usr/bin/env python3
"""Bit-exactness + timing A/B of the fp8 and the pre-scaled-q attention kernels of two builds of the library, in ONE process on the same inputs:
    python tools/ab_bitexact.py sparse-videogen_amd/lib/libsvgattn_prev.so [sparse-videogen_amd/lib/libsvgattn.so] [--kmeans]
Both libraries are loaded side by side (svg._native is re-pointed between the calls); for every case the outputs are compared with
torch.equal — two builds that differ only in instruction selection / scheduling must agree bit for bit — and the kernel times are
printed (A, B, A again: the third column shows the drift of the box).
  pre  : svg_band_attention_prescaled (the headline kernel) on the headline workload, head_dim 64 / fp16 / spiky rows, the switch entry
  band : svg_band_attention_fp8 on the headline workload (HunyuanVideo 720p, 24 heads, alternating masks; stage 2 timed alone)
  vb   : svg_varblock_attention_fp8 on the production-size SVG2 cases of tests/test_gpu_fullsize_svg2.py (fused permutation)"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from svg import _native as nat  # noqa: E402
from svg.models.hyvideo import utils as hy  # noqa: E402
from svg.models.hyvideo.utils import sparsity_to_width  # noqa: E402


def checksum(o):
    w = o.contiguous().view(torch.int16).to(torch.int64)
    return int(w.sum().item()), int((w * w).sum().item())


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


LIBS = {}


def use(tag):
    """point svg._native at library `tag` (both stay loaded: ctypes handles of different files are independent)"""
    import ctypes

    nat._lib = None
    import os

    os.environ["SVG_ATTN_LIB"] = LIBS[tag]
    nat.load()
    assert isinstance(nat._lib, ctypes.CDLL)


def ab(name, fn, n):
    """fn() -> output tensor; runs under A, B, A"""
    res = {}
    for tag in ("A", "B", "A2"):
        use(tag[0])
        ms = timed(fn, n)
        o = fn().clone()
        torch.cuda.synchronize()
        res[tag] = (ms, o)
    same = torch.equal(res["A"][1], res["B"][1]) and torch.equal(res["A"][1], res["A2"][1])
    print(f"{name}: A {res['A'][0]:.3f} ms  B {res['B'][0]:.3f} ms  A again {res['A2'][0]:.3f} ms   B/A {res['B'][0] / (0.5 * (res['A'][0] + res['A2'][0])):.4f}   "
          f"bit-identical: {same}   checksum {checksum(res['B'][1])}", flush=True)
    return same


def main():
    dev = torch.device("cuda", 0)
    LIBS["A"] = str(Path([a for a in sys.argv[1:] if not a.startswith("--")][0]).resolve())
    paths = [a for a in sys.argv[1:] if not a.startswith("--")]
    LIBS["B"] = str(Path(paths[1]).resolve()) if len(paths) > 1 else str(ROOT / "sparse-videogen_amd" / "lib" / "libsvgattn.so")
    print("A =", LIBS["A"], "\nB =", LIBS["B"])
    ok = True
    BH, D, F_, P_, ctx = 24, 128, 33, 3600, 256
    S = F_ * P_ + ctx
    mask = hy.generate_temporal_head_mask_mod(ctx, 64, F_, P_, mul=sparsity_to_width(0.25, ctx, F_, P_))
    g = torch.Generator(device=dev).manual_seed(0)
    q, k, v = (torch.randn(1, BH, S, D, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3))
    o = torch.empty_like(q)
    best = torch.tensor([[h % 2 for h in range(BH)]], device=dev, dtype=torch.int64)
    pk = dict(head_perm_flag=best, vid0=0, num_frame=F_, frame_size=P_, out=o)
    qs = (q.float() * nat.softmax_q_scale(D)).to(q.dtype)
    ok &= ab("pre-scaled band hy720p (24 heads, bf16)", lambda: nat.band_attention(qs, k, v, mask, q_prescaled=True, **pk), 8)
    dmask = nat.BandMask(real_len=mask.real_len, band=S + 1, colfull_lo=0, colfull_hi=0, rowfull_lo=0, rowfull_hi=0)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    o2 = torch.empty_like(q[:, :6])
    for fv in (0, 1):
        flag.fill_(fv)
        ok &= ab(f"pre-scaled switch entry, flag {fv} (6 heads)",
                 lambda: nat.band_attention_switch(qs[:, :6].contiguous(), k[:, :6].contiguous(), v[:, :6].contiguous(), mask, dmask, flag,
                                                   head_perm_flag=best[:, :6].contiguous(), vid0=0, num_frame=F_, frame_size=P_, out=o2,
                                                   q_prescaled=True), 2)
    del o2
    for dt, D2 in ((torch.float16, 128), (torch.bfloat16, 64), (torch.float16, 64)):
        g2 = torch.Generator(device=dev).manual_seed(5)
        q2, k2, v2 = (torch.randn(1, 6, 9000, D2, device=dev, dtype=dt, generator=g2) for _ in range(3))
        q2[0, 2, 4321] *= 25          # a spiky row: the exact path of the softmax, reference rewritten
        k2[0, 3, 100:140] *= 6
        q2s = (q2.float() * nat.softmax_q_scale(D2)).to(dt)
        m3 = nat.BandMask(real_len=8990, band=1536, colfull_lo=8900, colfull_hi=8990, rowfull_lo=8900, rowfull_hi=8990)
        ok &= ab(f"pre-scaled band small {str(dt)[6:]} D={D2}, spiky, ragged", lambda: nat.band_attention(q2s, k2, v2, m3, q_prescaled=True), 3)
    del qs
    ws = {}

    def band_full():
        # (the workspace of the pre-pass is per library instance of the cache key: fill it under the library that reads it)
        if id(nat._lib) not in ws:
            nat.clear_workspace_cache()
            nat.band_attention_fp8(q, k, v, mask, stage=1, **pk)
            ws.clear()
            ws[id(nat._lib)] = True
        return nat.band_attention_fp8(q, k, v, mask, stage=2, **pk)

    ok &= ab("band hy720p (24 heads, stage 2)", band_full, 8)
    # fp16 inputs, a spiky row (exact path of the softmax) and a ragged length
    qh, kh, vh = (x[:, :4, :5000].contiguous().to(torch.float16) for x in (q, k, v))
    qh[0, 1, 777] *= 30
    m2 = nat.BandMask(real_len=4990, band=1280, colfull_lo=4900, colfull_hi=4990, rowfull_lo=4900, rowfull_hi=4990)
    ok &= ab("band small, f16, spiky row, ragged", lambda: nat.band_attention_fp8(qh, kh, vh, m2), 3)
    del q, k, v, o
    torch.cuda.empty_cache()
    from tests.test_gpu_fullsize_svg2 import build_case

    for name in ("wan720p", "hy720p"):
        c = build_case(name)
        fn = lambda: nat.varblock_attention(c["q"], c["k"], c["v"], c["dmap"], c["q_sizes"], c["k_sizes"], q_row_idx=c["qidx"],  # noqa: E731
                                            kv_row_idx=c["kidx"], fp8=True)
        ok &= ab(f"varblock {name} ({c['H']} heads, fused permutation)", fn, 5)
        del c
        torch.cuda.empty_cache()


    if "--kmeans" in sys.argv:
        # flash-kmeans loop at the Wan 720p geometry (8 of the 40 heads): labels, counts and centroids of two builds must be identical
        # (e.g. B = `SVG_EXTRA_HIPCC_FLAGS=-DSVG_KMEANS_V2=1 python sparse-videogen_amd/build.py --tag km2` -> lib/libsvgattn_km2.so)
        from tests.test_gpu_fullsize_svg2 import clustered

        gen = torch.Generator(device="cuda").manual_seed(3)
        for N, K in ((75600, 1000), (75600, 300)):
            x = clustered(8, N, 128, 64, gen)
            init = x[:, :K].contiguous()

            def loop():
                lab, cent, cnt, n_it, _ = nat.kmeans_loop(x, None, init, 2, 1e-4)
                return torch.cat([lab.reshape(-1).float(), cnt.reshape(-1).float(), cent.reshape(-1).float()])

            ok &= ab(f"k-means loop N={N} K={K} (8 heads, 2 iterations)", loop, 5)
    print("ALL BIT-IDENTICAL" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
