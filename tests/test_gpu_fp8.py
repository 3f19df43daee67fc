"""fp8 (e4m3) QK^T / PV attention (svg_band_attention_fp8, BASELINE.json configs[4]).  The reference has no fp8 path
(README.md:117), so the statements are about distance: to the fp32 oracle on the original inputs (what a user sees), to the fp32
oracle on the DEQUANTISED inputs (isolates the kernel: only the e4m3 probabilities and the accumulation order are left), and to the
16-bit kernel.  Tolerances are the measured fp8 error with head-room, written next to each assert; masks, placement and edge handling
must be exactly those of the 16-bit path (same policy code), which the structured cases below would expose as O(1) errors."""
import pytest
import torch

from oracle import svg_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    from svg import _native

    _native.load()
    assert torch.cuda.is_available()
    return _native


def rel_l2(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp(min=1e-20)).item()


def dequant(x):
    """per-head e4m3 quantisation of the pre-pass (x * 448 / amax(head)) and back, on the CPU"""
    amax = x.float().abs().amax(dim=(-2, -1), keepdim=True).clamp(min=1e-30)
    s = 448.0 / amax
    return (x.float() * s).to(torch.float8_e4m3fn).float() / s


def dequant_q(q, k):
    """q as the SVG1 pre-pass quantises it: it carries the softmax scale, q8 = q * (scale_log2 / sk) * 2^-e with the power of two e
    that puts the head's largest |q8| into (224, 448] (csrc/attention_f8.hip: f8_quantize_kernel)"""
    import math

    D = q.shape[-1]
    scale_log2 = (1.0 / math.sqrt(D)) * 1.4426950408889634
    aq = q.float().abs().amax(dim=(-2, -1), keepdim=True).clamp(min=1e-30)
    sk = 448.0 / k.float().abs().amax(dim=(-2, -1), keepdim=True).clamp(min=1e-30)
    ideal = scale_log2 / sk
    e = torch.ceil(torch.log2(aq * ideal / 448.0))
    mq = ideal * torch.exp2(-e)
    return (q.float() * mq).to(torch.float8_e4m3fn).float() / mq


def _case(model, F_, P_, ctx, L, mul):
    V = F_ * P_
    if model == "hy":
        S = V + ctx
        return S, O.hy_band_params(S, ctx, L, F_, P_, mul), O.hy_mask(S, ctx, L, F_, P_, mul), 0
    if model == "wan":
        return V, O.wan_band_params(V, F_, P_, mul), O.wan_mask(V, F_, P_, mul), 0
    if model == "dense":
        S = V + ctx
        return S, O.dense_band_params(S), torch.ones(S, S, dtype=torch.bool), 0
    raise ValueError(model)


@pytest.mark.parametrize("model", ["hy", "wan", "dense"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fp8_band_attention_vs_oracle(nat, model, dtype):
    torch.manual_seed(4)
    F_, P_, ctx, L, mul, D, H = 5, 150, 40, 11, 2.3, 128, 3
    S, prm, mask, _ = _case(model, F_, P_, ctx, L, mul)
    q, k, v = (torch.randn(1, H, S, D).to(dtype) for _ in range(3))
    o = nat.band_attention_fp8(q.cuda(), k.cuda(), v.cuda(), nat.BandMask(**prm)).float().cpu()
    o16 = nat.band_attention(q.cuda(), k.cuda(), v.cuda(), nat.BandMask(**prm)).float().cpu()
    ref = O.masked_attention(q, k, v, mask)
    ref_dq = O.masked_attention(dequant_q(q, k), dequant(k), dequant(v), mask)
    e_ref, e_dq, e_16 = rel_l2(o, ref), rel_l2(o, ref_dq), rel_l2(o, o16)
    print(f"[fp8 {model} {dtype}] rel L2 vs fp32 oracle {e_ref:.4f}, vs oracle on dequantised inputs {e_dq:.4f}, vs 16-bit kernel {e_16:.4f}; "
          f"max abs {float((o - ref).abs().max()):.4f} (output rms {float(ref.pow(2).mean().sqrt()):.4f})")
    assert torch.isfinite(o).all()
    assert e_dq < 4e-2      # e4m3 probabilities (3 mantissa bits): measured ~2e-2
    assert e_ref < 8e-2     # + e4m3 q, k, v: measured ~4e-2


def test_fp8_fused_placement_and_rows(nat):
    """temporal heads: the pre-pass applies the placement, the kernel the inverse placement — compared with the 16-bit kernel's fused
    path head by head (same mask code, so any indexing error is O(1)); v = 1 -> every output element is 1 up to fp8 rounding of 1."""
    torch.manual_seed(6)
    F_, P_, ctx, L, mul, D, H = 6, 170, 40, 11, 2.3, 128, 4
    S, prm, mask, _ = _case("hy", F_, P_, ctx, L, mul)
    q, k, v = (torch.randn(1, H, S, D).to(torch.bfloat16).cuda() for _ in range(3))
    best = torch.tensor([[0, 1, 1, 0]]).cuda()
    kw = dict(head_perm_flag=best, vid0=0, num_frame=F_, frame_size=P_)
    o8 = nat.band_attention_fp8(q, k, v, nat.BandMask(**prm), **kw).float()
    o16 = nat.band_attention(q, k, v, nat.BandMask(**prm), **kw).float()
    for h in range(H):
        e = rel_l2(o8[0, h], o16[0, h])
        assert e < 8e-2, (h, e)
    ones = torch.ones_like(v)
    o1 = nat.band_attention_fp8(q, k, ones, nat.BandMask(**prm), **kw).float()
    assert (o1 - 1).abs().max().item() <= 2 ** -6


def test_fp8_outliers_and_scales(nat):
    """per-head scales: one head 50x larger than the others, one with a single large outlier — the outputs stay within the fp8
    tolerance of the oracle HEAD BY HEAD (a shared scale would flush the small head to zero)."""
    torch.manual_seed(8)
    S, D, H = 1100, 128, 3
    q, k, v = (torch.randn(1, H, S, D) for _ in range(3))
    q[0, 1] *= 5.0
    k[0, 1] *= 0.2
    v[0, 1] *= 50.0
    v[0, 2, 77, 5] = 300.0
    q, k, v = (x.to(torch.bfloat16) for x in (q, k, v))
    prm = O.dense_band_params(S)
    o = nat.band_attention_fp8(q.cuda(), k.cuda(), v.cuda(), nat.BandMask(**prm)).float().cpu()
    ref = O.masked_attention(q, k, v, None)
    for h in range(H):
        e = rel_l2(o[0, h], ref[0, h])
        print(f"[fp8 scales] head {h}: rel L2 {e:.4f}")
        assert e < (0.25 if h == 2 else 8e-2), (h, e)    # head 2: one outlier costs the whole head 7 bits of range


def test_fp8_rejects_unsupported(nat):
    q = torch.randn(1, 2, 300, 64).to(torch.bfloat16).cuda()
    with pytest.raises(RuntimeError):
        nat.band_attention_fp8(q, q, q, nat.BandMask(**O.dense_band_params(300)))


def _random_partition(seq_len, num_blocks, bsz, gen):
    sizes = torch.empty((bsz, num_blocks), dtype=torch.int32)
    for i in range(bsz):
        cut = torch.sort(torch.randperm(seq_len - 1, generator=gen)[: num_blocks - 1] + 1).values
        sizes[i] = torch.diff(torch.cat((torch.tensor([0]), cut, torch.tensor([seq_len]))))
    return sizes


@pytest.mark.parametrize("hq,hkv,S,MB,NB,density,dtype", [
    (4, 4, 4096, 20, 100, 0.7, torch.bfloat16), (4, 1, 2048, 10, 50, 0.5, torch.float16), (2, 2, 256, 10, 50, 0.2, torch.bfloat16),
    (16, 4, 4096, 10, 50, 0.9, torch.bfloat16), (1, 1, 8192, 20, 100, 0.7, torch.bfloat16),
])
def test_fp8_varblock_attention(nat, hq, hkv, S, MB, NB, density, dtype):
    """svg_varblock_attention_fp8 (SVG2 with e4m3 QK^T / PV, gathered rows, ds_read_b64_tr_b8 V^T) on the parameter family of
    svg/kernels/test/test_sparse_attn_dyn_blk_wan.py:74-133: vs the fp32 oracle under the expanded block mask, vs the oracle on
    dequantised inputs and vs the 16-bit kernel; ragged row / column blocks, GQA, empty blocks."""
    D = 128
    gen = torch.Generator().manual_seed(hq * 1000 + S + MB)
    rsz = _random_partition(S, MB, hkv, gen)
    csz = _random_partition(S, NB, hkv, gen)
    bmap = torch.rand(hkv, MB, NB, generator=gen) > density
    q = torch.randn(hq, S, D, generator=gen).to(dtype)
    k = torch.randn(hkv, S, D, generator=gen).to(dtype)
    v = torch.randn(hkv, S, D, generator=gen).to(dtype)
    args = (q.cuda(), k.cuda(), v.cuda(), bmap.cuda(), rsz.cuda(), csz.cuda())
    o = nat.varblock_attention(*args, fp8=True).float().cpu()
    o16 = nat.varblock_attention(*args).float().cpu()
    assert torch.isfinite(o).all()
    gq = hq // hkv
    worst = 0.0
    for h in range(hkv):
        em = O.block_mask_to_element_mask(bmap[h], rsz[h], csz[h])
        sl = slice(h * gq, (h + 1) * gq)
        ref = O.masked_attention(q[sl], k[h:h + 1], v[h:h + 1], em)
        ref_dq = O.masked_attention(torch.cat([dequant_q(q[i:i + 1], k[h:h + 1]) for i in range(sl.start, sl.stop)]), dequant(k[h:h + 1]),
                                    dequant(v[h:h + 1]), em)
        e_ref, e_dq = rel_l2(o[sl], ref), rel_l2(o[sl], ref_dq)
        worst = max(worst, e_ref)
        assert e_dq < 4e-2 and e_ref < 8e-2, (h, e_dq, e_ref)
        rows_without_keys = ~em.any(dim=1)
        assert (o[sl][:, rows_without_keys] == 0).all()        # block-rows with no active key block give zeros like the 16-bit path
    print(f"[fp8 varblock hq={hq} hkv={hkv} S={S}] worst rel L2 vs fp32 oracle {worst:.4f}, vs 16-bit kernel {rel_l2(o, o16):.4f}")


def test_fp8_varblock_fused_permutation(nat):
    """row-index gather / scatter (the fused token permutation of SVG2) through the fp8 kernel: same statement as the 16-bit test —
    attention on un-permuted tensors with index arrays equals attention on permuted tensors followed by the inverse permutation."""
    torch.manual_seed(12)
    H, S, D, MB, NB = 3, 2048, 128, 12, 40
    gen = torch.Generator().manual_seed(5)
    rsz, csz = _random_partition(S, MB, H, gen), _random_partition(S, NB, H, gen)
    bmap = torch.rand(H, MB, NB, generator=gen) > 0.5
    q, k, v = (torch.randn(H, S, D, generator=gen).to(torch.bfloat16).cuda() for _ in range(3))
    qi = torch.stack([torch.randperm(S, generator=gen) for _ in range(H)]).to(torch.int32).cuda()
    ki = torch.stack([torch.randperm(S, generator=gen) for _ in range(H)]).to(torch.int32).cuda()
    fused = nat.varblock_attention(q, k, v, bmap.cuda(), rsz.cuda(), csz.cuda(), q_row_idx=qi, kv_row_idx=ki, fp8=True)
    qp, kp, vp = nat.permute_rows(q, qi), nat.permute_rows(k, ki), nat.permute_rows(v, ki)
    op = nat.varblock_attention(qp, kp, vp, bmap.cuda(), rsz.cuda(), csz.cuda(), fp8=True)
    mat = nat.permute_rows(op, qi, inverse=True)
    assert torch.equal(fused, mat)        # per-head scales do not depend on the row order: bit-identical


def test_processor_core_fp8_switch(nat):
    """svg.models._core.set_attention_dtype("fp8"): the SVG1 and SVG2 attention cores the processors call run the e4m3 kernels (head
    dim 128) and stay within the fp8 distance of their 16-bit results; "bf16" restores the default bit for bit."""
    from svg.models import _core

    H, D, F_, P_, ctx, L = 3, 128, 4, 200, 40, 11
    V = F_ * P_
    S = V + ctx
    g = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(1, H, S, D, generator=g).to(torch.bfloat16).cuda() for _ in range(3))
    geo = _core.Geometry(ctx, F_, P_)
    mask = nat.BandMask(real_len=V + L, band=256, colfull_lo=V, colfull_hi=V + L, rowfull_lo=V, rowfull_hi=V + L)
    prof = nat.ProfileDesc(0, F_, P_, 1)
    prof.variant[0] = nat.ProfileVariant(0, 0, V, 2, 0, V, S)
    prof.variant[1] = nat.ProfileVariant(1, 0, V, 2, 0, V, S)

    def svg1():
        torch.manual_seed(11)
        return _core.svg1_sparse_attention(q, k, v, geo, mask, prof, 32, V)

    def svg2():
        torch.manual_seed(5)
        torch.cuda.manual_seed(5)
        return _core.svg2_sparse_attention(q[:, :, :V].contiguous(), k[:, :, :V].contiguous(), v[:, :, :V].contiguous(),
                                           _core.Geometry(0, F_, P_), _core.CentroidStore(), 0, 10, 25, 0.9, 0.1, 4, 2)

    assert _core.attention_dtype() == "bf16"
    o1, b1 = svg1()
    o2 = svg2()
    try:
        _core.set_attention_dtype("fp8")
        o1f, b1f = svg1()
        o2f = svg2()
    finally:
        _core.set_attention_dtype("bf16")
    assert torch.equal(b1, b1f)                                  # the profiler stays 16-bit
    assert 1e-3 < rel_l2(o1f, o1) < 8e-2 and 1e-3 < rel_l2(o2f, o2) < 8e-2
    o1b, _ = svg1()
    assert torch.equal(o1b, o1)


@pytest.mark.parametrize("seed", range(8))
def test_fp8_band_random_mask_family(nat, seed):
    """Random members of the svg_band_mask_t family (full rows / columns anywhere, real_len < S, bands 0 .. S + 1, fused placement on
    some heads) through the fp8 kernel against the 16-bit kernel on the same arguments: the mask / tile / placement logic is shared
    policy code, so a wrong tile range or predicate shows up as an O(1) difference on some head, far above the fp8 distance."""
    import random

    rng = random.Random(2000 + seed)
    F_, P_ = rng.choice([(4, 160), (5, 130), (3, 333)])
    ctx = rng.choice([0, 40, 77])
    V = F_ * P_
    S = V + ctx
    real = rng.choice([S, S, V + rng.randint(0, ctx) if ctx else S])
    band = rng.choice([0, 1, rng.randint(2, 200), rng.randint(100, S), S + 1])
    lo = min(S, rng.choice([0, 256, rng.randint(0, S - 1)]))
    cf = (lo, min(S, lo + rng.choice([0, 1, 64, rng.randint(1, 300)])))
    lo = min(S, rng.choice([0, 256, 512, rng.randint(0, S - 1)]))
    rf = (lo, min(S, lo + rng.choice([0, 1, 30, 256, rng.randint(1, 400)])))
    prm = dict(real_len=real, band=band, colfull_lo=cf[0], colfull_hi=cf[1], rowfull_lo=rf[0], rowfull_hi=rf[1])
    torch.manual_seed(seed)
    H = 3
    q, k, v = (torch.randn(1, H, S, 128).to(torch.bfloat16).cuda() for _ in range(3))
    best = torch.tensor([[rng.randint(0, 1) for _ in range(H)]]).cuda()
    kw = dict(head_perm_flag=best, vid0=0, num_frame=F_, frame_size=P_)
    o8 = nat.band_attention_fp8(q, k, v, nat.BandMask(**prm), **kw).float()
    o16 = nat.band_attention(q, k, v, nat.BandMask(**prm), **kw).float()
    assert torch.isfinite(o8).all()
    for h in range(H):
        ref = o16[0, h]
        rows = ref.abs().sum(-1) > 0                    # rows without any allowed key are zero in both
        assert torch.equal(o8[0, h][~rows], ref[~rows])
        if rows.any():
            e = rel_l2(o8[0, h][rows], ref[rows])
            assert e < 0.1, (prm, h, e)


@pytest.mark.parametrize("spike", [30.0, 120.0, 400.0])
def test_fp8_score_spikes(nat, spike):
    """The rare softmax path of the fp8 bodies: one key row aligned with a few query rows so that their score jumps by `spike`
    (natural-log units) over everything before it, in a LATE key tile — the probability sum check fails (also with a non-finite
    sum), the wave takes the exact path and rescales O and l.  Outputs stay finite and within the fp8 distance of the oracle."""
    torch.manual_seed(11)
    S, D, H = 1500, 128, 2
    q, k, v = (torch.randn(1, H, S, D) for _ in range(3))
    scale = 1.0 / D ** 0.5
    for (qi, ki) in [(5, 900), (300, 1340), (301, 70), (1400, 1499), (1401, 3)]:
        for h in range(H):
            qd = q[0, h, qi]
            k[0, h, ki] = qd / qd.norm() ** 2 * (spike / scale)
    q, k, v = (x.to(torch.bfloat16) for x in (q, k, v))
    o = nat.band_attention_fp8(q.cuda(), k.cuda(), v.cuda(), nat.BandMask(**O.dense_band_params(S))).float().cpu()
    ref = O.masked_attention(q, k, v, None)
    assert torch.isfinite(o).all()
    # the spiked k rows make the head's k scale coarse for all other keys (per-head scale): bound the spiked rows and the rest apart
    rows = torch.tensor([5, 300, 301, 1400, 1401])
    e_spiked = rel_l2(o[0, :, rows], ref[0, :, rows])
    print(f"[fp8 spike {spike}] rel L2 on the spiked rows {e_spiked:.4f}, overall {rel_l2(o, ref):.4f}")
    assert e_spiked < 0.1


def test_fp8_all_scores_very_negative(nat):
    """see test_band_attention_all_scores_very_negative: the fp8 bodies start from the same pseudo-reference"""
    torch.manual_seed(13)
    S, D, H = 1200, 128, 2
    u = torch.randn(D)
    u = u / u.norm()
    a = (150.0 * D ** 0.5) ** 0.5
    q = (-a * u + 0.5 * torch.randn(1, H, S, D)).to(torch.bfloat16)
    k = (a * u + 0.5 * torch.randn(1, H, S, D)).to(torch.bfloat16)
    v = torch.randn(1, H, S, D).to(torch.bfloat16)
    # dense, and a band whose rows start on tiles that are fully masked for some waves of the row tile
    for prm in (O.dense_band_params(S), dict(real_len=S, band=200, colfull_lo=0, colfull_hi=0, rowfull_lo=0, rowfull_hi=0)):
        o = nat.band_attention_fp8(q.cuda(), k.cuda(), v.cuda(), nat.BandMask(**prm)).float().cpu()
        o16 = nat.band_attention(q.cuda(), k.cuda(), v.cuda(), nat.BandMask(**prm)).float().cpu()
        assert torch.isfinite(o).all() and o.abs().max() > 0
        # e4m3 q and k at score magnitude 150: the softmax is much noisier than the 16-bit one; the statement here is that the rows
        # are normalised averages of v (|o| bounded by max |v|, non-zero), not zeros from an underflowed reference
        assert o.abs().max() <= v.float().abs().max() * 1.01 and (o.abs().sum(-1) > 0).all()
        assert rel_l2(o, o16) < 1.0
