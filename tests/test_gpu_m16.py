"""The two-phase band kernel on v_mfma_f32_16x16x32 (csrc/attn_m16.h, svg_band_attention variant 8; head_dim 128) against the fp32
oracle: the SVG1 masks of the three models and the dense modes, both 16-bit types, the rare softmax paths (late spikes, rows far below
zero), random members of the mask family, the fused head placement, and the other schedule on the same inputs."""
import random

import pytest
import torch

from oracle import svg_oracle as O
from test_gpu_kernels import _band_case, check_attn, dev

pytestmark = pytest.mark.gpu
M16 = 8


@pytest.fixture(scope="module")
def nat():
    from svg import _native

    _native.load()
    return _native


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("model", ["hy", "wan", "cog", "dense", "dense2"])
def test_m16_band_attention(nat, model, dtype):
    torch.manual_seed(2)
    F_, P_, ctx, L, mul = 5, 150, 40, 11, 2.3
    S, prm, mask, _ = _band_case(model, F_, P_, ctx, L, mul)
    H, D = 3, 128
    q, k, v = (torch.randn(1, H, S, D).to(dtype) for _ in range(3))
    o = nat.band_attention(dev(q), dev(k), dev(v), nat.BandMask(**prm), variant=M16)
    check_attn(o, O.masked_attention(q, k, v, mask), dtype)
    o2 = nat.band_attention(dev(q), dev(k), dev(v), nat.BandMask(**prm), variant=2)
    e = ((o.float() - o2.float()).norm() / o2.float().norm()).item()
    assert e < (4e-3 if dtype == torch.bfloat16 else 6e-4), e     # two roundings of the same fp32 result apart


@pytest.mark.parametrize("spike", [30.0, 120.0, 400.0])
def test_m16_score_spikes(nat, spike):
    torch.manual_seed(11)
    S, D, H = 1500, 128, 2
    q, k, v = (torch.randn(1, H, S, D) for _ in range(3))
    scale = 1.0 / D ** 0.5
    for (qi, ki) in [(5, 900), (300, 1340), (301, 70), (1400, 1499), (1401, 3), (17, 18), (31, 1200)]:
        for h in range(H):
            qd = q[0, h, qi]
            k[0, h, ki] = qd / qd.norm() ** 2 * (spike / scale)
    q, k, v = (x.to(torch.bfloat16) for x in (q, k, v))
    o = nat.band_attention(dev(q), dev(k), dev(v), nat.BandMask(**O.dense_band_params(S)), variant=M16)
    assert torch.isfinite(o.float()).all()
    check_attn(o, O.masked_attention(q, k, v, None), torch.bfloat16)


def test_m16_all_scores_very_negative(nat):
    torch.manual_seed(13)
    S, D, H = 1200, 128, 2
    u = torch.randn(D)
    u = u / u.norm()
    a = (150.0 * D ** 0.5) ** 0.5
    q = (-a * u + 0.5 * torch.randn(1, H, S, D)).to(torch.bfloat16)
    k = (a * u + 0.5 * torch.randn(1, H, S, D)).to(torch.bfloat16)
    v = torch.randn(1, H, S, D).to(torch.bfloat16)
    for prm in (O.dense_band_params(S), dict(real_len=S, band=200, colfull_lo=0, colfull_hi=0, rowfull_lo=0, rowfull_hi=0)):
        o = nat.band_attention(dev(q), dev(k), dev(v), nat.BandMask(**prm), variant=M16)
        ref = O.masked_attention(q, k, v, O.band_mask(S, **prm))
        assert torch.isfinite(o.float()).all() and o.float().abs().max() > 0
        torch.testing.assert_close(o.float().cpu(), ref, atol=6e-2, rtol=6e-2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("seed", range(12))
def test_m16_random_mask_family(nat, seed, dtype):
    rng = random.Random(1000 + seed)
    S = rng.choice([300, 513, 777, 1024, 1301])
    real = rng.choice([S, S, rng.randint(1, S), max(1, S - rng.randint(0, 300))])
    band = rng.choice([0, 1, rng.randint(2, 200), rng.randint(100, S), S + 1])
    lo = min(S, rng.choice([0, 256, rng.randint(0, S - 1)]))
    cf = (lo, min(S, lo + rng.choice([0, 1, 64, rng.randint(1, 300)])))
    lo = min(S, rng.choice([0, 256, 512, rng.randint(0, S - 1)]))
    rf = (lo, min(S, lo + rng.choice([0, 1, 30, 256, rng.randint(1, 400)])))
    prm = dict(real_len=real, band=band, colfull_lo=cf[0], colfull_hi=cf[1], rowfull_lo=rf[0], rowfull_hi=rf[1])
    torch.manual_seed(seed)
    H, D = 2, 128
    q, k, v = (torch.randn(1, H, S, D).to(dtype) for _ in range(3))
    o = nat.band_attention(dev(q), dev(k), dev(v), nat.BandMask(**prm), variant=M16)
    check_attn(o, O.masked_attention(q, k, v, O.band_mask(S, **prm)), dtype)


@pytest.mark.parametrize("model", ["hy", "cog"])
def test_m16_fused_placement(nat, model):
    """head_perm_flag: token-major heads read K / V rows and write O rows through the index map inside the kernel == placement ->
    attention -> inverse placement of the oracle"""
    torch.manual_seed(3)
    F_, P_, ctx, L, mul, D, H = 6, 130, 24, 7, 1.6, 128, 4
    S, prm, mask, vid0 = _band_case(model, F_, P_, ctx, L, mul)
    q, k, v = (torch.randn(1, H, S, D).to(torch.bfloat16) for _ in range(3))
    best = torch.tensor([[0, 1, 1, 0]])
    o = nat.band_attention(dev(q), dev(k), dev(v), nat.BandMask(**prm), head_perm_flag=dev(best), vid0=vid0, num_frame=F_, frame_size=P_,
                           variant=M16)
    tf = model == "cog"
    qp, kp, vp = (O.head_placement(x, best, ctx, F_, P_, text_first=tf) for x in (q, k, v))
    ref = O.head_placement(O.masked_attention(qp, kp, vp, mask), best, ctx, F_, P_, inverse=True, text_first=tf)
    check_attn(o, ref, torch.bfloat16)


# ---- the same body behind the variable-block policy (svg_varblock_attention variant 8; SVG2) ----
_VB = [(hq, hkv, S, MB, NB, dens, dt) for (hq, hkv) in [(1, 1), (4, 4), (4, 1), (16, 4)] for (S, MB, NB) in [(256, 10, 50), (256, 20, 100)]
       for dens in (0.2, 0.9) for dt in (torch.bfloat16, torch.float16)] + \
      [(4, 4, 4096, 20, 100, 0.7, torch.bfloat16), (16, 4, 4096, 10, 50, 0.2, torch.float16), (1, 1, 8192, 10, 100, 0.7, torch.float16),
       (4, 1, 8192, 20, 100, 0.5, torch.bfloat16)]


@pytest.mark.parametrize("hq,hkv,S,MB,NB,density,dtype", _VB)
def test_m16_varblock_attention(nat, hq, hkv, S, MB, NB, density, dtype):
    from test_gpu_kernels import random_partition_batch, rel_l2

    D = 128
    gen = torch.Generator().manual_seed(hq * 1000 + S + MB)
    rsz = random_partition_batch(S, MB, hkv, gen)
    csz = random_partition_batch(S, NB, hkv, gen)
    bmap = torch.rand(hkv, MB, NB, generator=gen) > density
    q = torch.randn(hq, S, D, generator=gen).to(dtype)
    k = torch.randn(hkv, S, D, generator=gen).to(dtype)
    v = torch.randn(hkv, S, D, generator=gen).to(dtype)
    o = nat.varblock_attention(dev(q), dev(k), dev(v), dev(bmap), dev(rsz), dev(csz), variant=M16).float().cpu()
    g = hq // hkv
    for h in range(hkv):
        em = O.block_mask_to_element_mask(bmap[h], rsz[h], csz[h])
        ref = O.masked_attention(q[h * g:(h + 1) * g], k[h:h + 1], v[h:h + 1], em)
        torch.testing.assert_close(o[h * g:(h + 1) * g], ref, atol=1e-2, rtol=1e-2)
        assert rel_l2(o[h * g:(h + 1) * g], ref) <= (3e-3 if dtype == torch.bfloat16 else 1e-3)


def test_m16_varblock_fused_gather(nat):
    """q_row_idx / kv_row_idx (the fused permutation of SVG2): == the same call on the 32x32x16 body (variant 9)"""
    from test_gpu_kernels import random_partition_batch

    torch.manual_seed(5)
    H, S, D, MB, NB = 4, 3000, 128, 12, 40
    gen = torch.Generator().manual_seed(77)
    rsz, csz = random_partition_batch(S, MB, H, gen), random_partition_batch(S, NB, H, gen)
    bmap = torch.rand(H, MB, NB, generator=gen) > 0.5
    q, k, v = (torch.randn(H, S, D, generator=gen).to(torch.bfloat16) for _ in range(3))
    qi = torch.stack([torch.randperm(S, generator=gen) for _ in range(H)]).to(torch.int32)
    ki = torch.stack([torch.randperm(S, generator=gen) for _ in range(H)]).to(torch.int32)
    kw = dict(q_row_idx=dev(qi), kv_row_idx=dev(ki))
    o8 = nat.varblock_attention(dev(q), dev(k), dev(v), dev(bmap), dev(rsz), dev(csz), variant=M16, **kw)
    o3 = nat.varblock_attention(dev(q), dev(k), dev(v), dev(bmap), dev(rsz), dev(csz), variant=9, **kw)   # 9: the 32x32x16 body
    e = ((o8.float() - o3.float()).norm() / o3.float().norm()).item()
    assert torch.isfinite(o8.float()).all() and e < 4e-3, e
