"""The pre-scaled-q path (round 3): svg_band_attention_prescaled / svg_band_attention_switch_prescaled take a q that already carries
sm_scale * log2(e), produced without an extra rounding by the fused prologue (svg_qk_norm_rope*_qscale), and start their score
accumulators at minus the row's softmax reference so that the MFMAs deliver the exponent argument.  No reference counterpart
(flex_attention takes `scale` as an argument, svg/models/hyvideo/attention.py:401-403): the oracle is `O.masked_attention` on the
SAME pre-scaled q with scale ln 2 — (q' . k) ln 2 = the natural-log score — at the tolerances of the plain kernels; the rare softmax
paths (rescale on a late spike, rows whose scores all lie far below zero) are exercised like for the plain body."""
import math

import pytest
import torch

from oracle import svg_oracle as O

pytestmark = pytest.mark.gpu
LN2 = math.log(2.0)


@pytest.fixture(scope="module")
def nat():
    from svg import _native

    _native.load()
    return _native


def rel_l2(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp(min=1e-20)).item()


def prescale(nat, q):
    return (q.float() * nat.softmax_q_scale(q.shape[-1])).to(q.dtype)


def check(o, ref, dtype):
    o = o.float().cpu()
    torch.testing.assert_close(o, ref, atol=1e-2, rtol=1e-2)
    assert rel_l2(o, ref) <= (3e-3 if dtype == torch.bfloat16 else 1e-3), rel_l2(o, ref)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("D", [128, 64])
@pytest.mark.parametrize("model", ["hy", "wan", "dense"])
def test_prescaled_vs_oracle(nat, model, D, dtype):
    torch.manual_seed(21)
    F_, P_, ctx, L, mul, H = 6, 170, 40, 11, 2.3, 3
    if model == "hy":
        S = F_ * P_ + ctx
        prm, mask = O.hy_band_params(S, ctx, L, F_, P_, mul), O.hy_mask(S, ctx, L, F_, P_, mul)
    elif model == "wan":
        S = F_ * P_
        prm, mask = O.wan_band_params(S, F_, P_, mul), O.wan_mask(S, F_, P_, mul)
    else:
        S = F_ * P_ + ctx
        prm, mask = O.dense_band_params(S), None
    q, k, v = (torch.randn(1, H, S, D).to(dtype) for _ in range(3))
    qs = prescale(nat, q)
    o = nat.band_attention(qs.cuda(), k.cuda(), v.cuda(), nat.BandMask(**prm), q_prescaled=True)
    check(o, O.masked_attention(qs, k, v, mask, scale=LN2), dtype)
    # and against the plain kernel on the plain q: the same attention up to the rounding of q'
    o0 = nat.band_attention(q.cuda(), k.cuda(), v.cuda(), nat.BandMask(**prm))
    assert rel_l2(o, o0) < (5e-3 if dtype == torch.bfloat16 else 1.5e-3)


def test_prescaled_fused_placement_and_switch(nat):
    """head_perm_flag path with a pre-scaled q == placement -> attention -> inverse placement of the oracle; the device-switched entry
    picks mask + placement (flag 0) or the dense alt mask (flag 1): the two plain pre-scaled calls up to rounding."""
    torch.manual_seed(3)
    F_, P_, ctx, L, mul, D, H = 6, 130, 24, 7, 1.6, 128, 4
    S = F_ * P_ + ctx
    prm, mask = O.hy_band_params(S, ctx, L, F_, P_, mul), O.hy_mask(S, ctx, L, F_, P_, mul)
    q, k, v = (torch.randn(1, H, S, D).to(torch.bfloat16) for _ in range(3))
    qs = prescale(nat, q)
    best = torch.tensor([[0, 1, 1, 0]])
    kw = dict(head_perm_flag=best.cuda(), vid0=0, num_frame=F_, frame_size=P_)
    o = nat.band_attention(qs.cuda(), k.cuda(), v.cuda(), nat.BandMask(**prm), q_prescaled=True, **kw)
    qp, kp, vp = (O.head_placement(x, best, ctx, F_, P_) for x in (qs, k, v))
    ref = O.head_placement(O.masked_attention(qp, kp, vp, mask, scale=LN2), best, ctx, F_, P_, inverse=True)
    check(o, ref, torch.bfloat16)
    dprm = O.dense_band_params(S, F_ * P_ + L)
    od = nat.band_attention(qs.cuda(), k.cuda(), v.cuda(), nat.BandMask(**dprm), q_prescaled=True)
    for flag, want in ((0, o), (1, od)):
        f = torch.tensor([flag], dtype=torch.int32).cuda()
        got = nat.band_attention_switch(qs.cuda(), k.cuda(), v.cuda(), nat.BandMask(**prm), nat.BandMask(**dprm), f, q_prescaled=True, **kw)
        # (another instantiation of the same body — the switch kernel holds it twice: under -ffast-math hipcc may associate the row sums
        #  differently, so rounding-level, like tests/test_gpu_kernels.py::test_band_attention_device_switch)
        torch.testing.assert_close(got.float(), want.float(), atol=4e-3, rtol=1e-2)
        assert (got.float() - want.float()).abs().mean() < 1e-4, flag


@pytest.mark.parametrize("spike", [30.0, 120.0, 400.0])
def test_prescaled_score_spikes(nat, spike):
    """a score that jumps by `spike` (natural-log units) in a LATE key tile: the exact path (new reference, O and l rescaled, the
    probabilities of the tile recomputed relative to the OLD reference the accumulators started from, the start tuple rewritten)"""
    torch.manual_seed(11)
    S, D, H = 1500, 128, 2
    q, k, v = (torch.randn(1, H, S, D) for _ in range(3))
    scale = 1.0 / D ** 0.5
    for (qi, ki) in [(5, 900), (300, 1340), (301, 70), (1400, 1499), (1401, 3)]:
        for h in range(H):
            qd = q[0, h, qi]
            k[0, h, ki] = qd / qd.norm() ** 2 * (spike / scale)
    q, k, v = (x.to(torch.bfloat16) for x in (q, k, v))
    qs = prescale(nat, q)
    o = nat.band_attention(qs.cuda(), k.cuda(), v.cuda(), nat.BandMask(**O.dense_band_params(S)), q_prescaled=True)
    assert torch.isfinite(o.float()).all()
    check(o, O.masked_attention(qs, k, v, None, scale=LN2), torch.bfloat16)


def test_prescaled_all_scores_very_negative(nat):
    torch.manual_seed(13)
    S, D, H = 1200, 128, 2
    u = torch.randn(D)
    u = u / u.norm()
    a = (150.0 * D ** 0.5) ** 0.5
    q = (-a * u + 0.5 * torch.randn(1, H, S, D)).to(torch.bfloat16)
    k = (a * u + 0.5 * torch.randn(1, H, S, D)).to(torch.bfloat16)
    v = torch.randn(1, H, S, D).to(torch.bfloat16)
    qs = prescale(nat, q)
    for prm in (O.dense_band_params(S), dict(real_len=S, band=200, colfull_lo=0, colfull_hi=0, rowfull_lo=0, rowfull_hi=0)):
        o = nat.band_attention(qs.cuda(), k.cuda(), v.cuda(), nat.BandMask(**prm), q_prescaled=True)
        ref = O.masked_attention(qs, k, v, O.band_mask(S, **prm), scale=LN2)
        assert torch.isfinite(o.float()).all() and o.float().abs().max() > 0
        torch.testing.assert_close(o.float().cpu(), ref, atol=6e-2, rtol=6e-2)


@pytest.mark.parametrize("seed", range(6))
def test_prescaled_random_mask_family(nat, seed):
    """random members of the svg_band_mask_t family (full rows / columns anywhere, real_len, band 0 .. S + 1) on the pre-scaled body"""
    g = torch.Generator().manual_seed(100 + seed)
    S, D, H = int(torch.randint(300, 1400, (1,), generator=g)), 128, 2
    real = int(torch.randint(1, S + 1, (1,), generator=g))
    band = int(torch.randint(0, S + 2, (1,), generator=g)) if seed % 3 else int(torch.randint(0, 200, (1,), generator=g))
    c0, r0 = (int(torch.randint(0, S, (1,), generator=g)) for _ in range(2))
    prm = dict(real_len=real, band=band, colfull_lo=c0, colfull_hi=min(S, c0 + int(torch.randint(0, 150, (1,), generator=g))),
               rowfull_lo=r0, rowfull_hi=min(S, r0 + int(torch.randint(0, 150, (1,), generator=g))))
    q, k, v = (torch.randn(1, H, S, D, generator=g).to(torch.bfloat16) for _ in range(3))
    qs = prescale(nat, q)
    o = nat.band_attention(qs.cuda(), k.cuda(), v.cuda(), nat.BandMask(**prm), q_prescaled=True)
    check(o, O.masked_attention(qs, k, v, O.band_mask(S, **prm), scale=LN2), torch.bfloat16)


@pytest.mark.parametrize("rope_kind", [1, 2])
@pytest.mark.parametrize("norm_kind", [1, 2])
def test_prologue_q_scale_single_rounding(nat, norm_kind, rope_kind):
    """svg_qk_norm_rope*_qscale: q_scale multiplies the fp32 result of the pass in front of its LAST rounding — for rotated positions
    that is round(c * rope(norm(x))) from the fp32 RoPE value, not round(c * round(rope(..))) — k is untouched, q_scale = 1 is
    bit-identical to the plain entry points, and the transposing form equals the in-place form."""
    torch.manual_seed(5)
    bsz, H, S, D, lo, hi = 2, 3, 301, 128, 0, 260
    c = nat.softmax_q_scale(D)
    q = torch.randn(bsz, H, S, D).to(torch.bfloat16)
    k = torch.randn(bsz, H, S, D).to(torch.bfloat16)
    w = (1 + 0.1 * torch.randn(D)).to(torch.bfloat16)
    b = (0.1 * torch.randn(D)).to(torch.bfloat16) if norm_kind == 2 else None
    cols = D // 2 if rope_kind == 2 else D
    ang = torch.rand(hi - lo, cols) * 6.28
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()

    def run(scale):
        qq, kk = q.clone().cuda(), k.clone().cuda()
        nat.qk_norm_rope(qq, kk, norm_kind, w.cuda(), None if b is None else b.cuda(), w.cuda(), None if b is None else b.cuda(),
                         1e-6, rope_kind, cos.cuda(), sin.cuda(), lo, hi, q_scale=scale)
        return qq.cpu(), kk.cpu()

    q1, k1 = run(1.0)
    qc, kc = run(c)
    assert torch.equal(k1, kc)
    # fp32 restatement of the RoPE stage with the factor in front of the last rounding, on the norm stage's own (rounded) output —
    # the fused pass keeps that intermediate rounding (tests/test_gpu_prologue.py checks the norm stage against the oracle)
    qn = q.clone().cuda()
    nat.qk_norm_rope(qn, None, norm_kind, w.cuda(), None if b is None else b.cuda(), None, None, 1e-6, 0)
    xn = qn.cpu().float()
    rot = xn[:, :, lo:hi]
    if rope_kind == 1:
        a, bq = rot[..., 0::2], rot[..., 1::2]
        e = torch.stack([a * cos[:, 0::2] + (-bq) * sin[:, 0::2], bq * cos[:, 1::2] + a * sin[:, 1::2]], dim=-1).flatten(-2)
        want_rot = (e * c).to(torch.bfloat16)
    else:
        a, bq = rot[..., 0::2].double(), rot[..., 1::2].double()
        fr, fi = cos.double(), sin.double()
        e = torch.stack([a * fr - bq * fi, a * fi + bq * fr], dim=-1).flatten(-2)
        want_rot = (e * float(torch.tensor(c, dtype=torch.float32))).float().to(torch.bfloat16)
    want = torch.cat([want_rot, (xn[:, :, hi:] * c).to(torch.bfloat16)], dim=2)
    assert torch.equal(qc[:, :, lo:hi], want[:, :, :hi - lo])
    assert torch.equal(qc[:, :, hi:], want[:, :, hi - lo:])
    # transposing form
    q_tok = q.transpose(1, 2).reshape(bsz, S, H * D).contiguous().cuda()
    k_tok = k.transpose(1, 2).reshape(bsz, S, H * D).contiguous().cuda()
    qt, kt = nat.qk_norm_rope_transpose(q_tok, k_tok, H, H, norm_kind, w.cuda(), None if b is None else b.cuda(), w.cuda(),
                                        None if b is None else b.cuda(), 1e-6, rope_kind, cos.cuda(), sin.cuda(), lo, hi, q_scale=c)
    assert torch.equal(qt.cpu(), qc) and torch.equal(kt.cpu(), kc)


def _large_logit_case(nat, dtype, gain, spike, seed=17):
    """q, k as the product's fused prologue makes them (per-head RMSNorm with weight `gain`, interleaved RoPE, ONE rounding), from the same
    16-bit pre-norm input: (q, k) plain = what the reference hands flex_attention; (q', k) with q_scale folded in front of that rounding.
    Optional spike rows: a handful of (query, key) pairs aligned so that the natural-log score is `spike`."""
    torch.manual_seed(seed)
    H, S, D = 2, 1536, 128
    xq, xk = torch.randn(1, H, S, D), torch.randn(1, H, S, D)
    if spike:
        for (qi, ki) in [(5, 900), (300, 1340), (301, 70), (1400, 1499), (1401, 3), (777, 778)]:
            xk[0, :, ki] = xq[0, :, qi]
    xq, xk = xq.to(dtype), xk.to(dtype)
    v = torch.randn(1, H, S, D).to(dtype)
    w = torch.full((D,), float(gain)).to(dtype)
    c = nat.softmax_q_scale(D)

    def prologue(scale):
        qq, kk = xq.clone().cuda(), xk.clone().cuda()
        nat.qk_norm_rope(qq, kk, 1, w.cuda(), None, w.cuda(), None, 1e-6, 0, q_scale=scale)
        return qq, kk

    q, k = prologue(1.0)
    qs, k2 = prologue(c)
    assert torch.equal(k, k2)
    if spike:   # rescale the aligned keys so that q . k / sqrt(D) = spike (RMS-normed rows have |q|^2 = gain^2 D)
        k = k.clone()
        for (qi, ki) in [(5, 900), (300, 1340), (301, 70), (1400, 1499), (1401, 3), (777, 778)]:
            k[0, :, ki] = (k[0, :, ki].float() * (spike / (gain * gain * D ** 0.5))).to(dtype)
    return q, qs, k, v.cuda()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("gain,spike", [(1.0, 0.0), (2.0, 0.0), (3.5, 0.0), (3.5, 80.0), (1.0, 60.0)])
def test_prescaled_vs_reference_formulation_on_large_logits(nat, dtype, gain, spike):
    """VERDICT round 3, weak #3.  The reference rounds q to the 16-bit type and applies the softmax scale to the fp32 scores
    (svg/models/hyvideo/attention.py:401-403: flex_attention's default `scale`); the pre-scaled path rounds c * q instead.  This test feeds
    the SAME pre-norm input through (i) the product's prologue with q_scale + svg_band_attention_prescaled and (ii) the prologue without the
    factor + the fp32 oracle on that plain q, i.e. the reference's formulation, where |q . k| / sqrt(D) reaches 40 - 80 (gain 3.5: scores
    ~ N(0, 12^2), plus aligned spike pairs), at the tolerance of the plain kernel.  Outcome (see the asserts): the default path passes, the
    opt-in pre-scaled path does not at bf16 — it stays opt-in."""
    q, qs, k, v = _large_logit_case(nat, dtype, gain, spike)
    S = q.shape[2]
    prm = O.dense_band_params(S)
    ref = O.masked_attention(q.cpu(), k.cpu(), v.cpu(), None)              # fp32, scale 1 / sqrt(D) on the plain 16-bit q
    smax = (q[0, 0].float() @ k[0, 0].float().T).abs().max().item() / q.shape[-1] ** 0.5
    o_pre = nat.band_attention(qs, k, v, nat.BandMask(**prm), q_prescaled=True)
    o_plain = nat.band_attention(q, k, v, nat.BandMask(**prm))
    e_pre, e_plain = rel_l2(o_pre.cpu(), ref), rel_l2(o_plain.cpu(), ref)
    print(f"\n[large logits] {dtype} gain {gain} spike {spike}: max |score| {smax:.1f}; rel L2 to the reference formulation: "
          f"pre-scaled {e_pre:.3e}, plain kernel {e_plain:.3e}")
    tol = 3e-3 if dtype == torch.bfloat16 else 1e-3
    # the DEFAULT path (plain q, scale on the fp32 scores) holds the plain kernel's tolerance on every case
    assert e_plain <= tol, e_plain
    # the opt-in pre-scaled path: first run (round 4, profiles/r04a_pytest_large_logits.txt) — fp16 within `tol` everywhere; bf16 3.2e-3
    # at gain 2 (max |score| 16) and 6.4e-3 at gain 3.5 (max |score| 50 - 80): the rounding of c * q is a second, independent 2^-9
    # perturbation of the scores next to the reference's own rounding of q — flex_attention documents the same for its PRESCALE_QK option
    # ("about 20% more numerical error, but slightly faster").  That is why prescale_q is OFF by default since round 4; the bound below
    # pins what switching it on costs.
    assert e_pre <= (tol if dtype == torch.float16 else 1e-2), e_pre
