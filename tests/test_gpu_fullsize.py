"""Full-size (HunyuanVideo 720p / 129 frames, S = 119056) checks of the headline kernel through size-independent
properties and spot rows against the oracle (a dense CPU oracle of the whole problem would be 174 TFLOP)."""
import math

import pytest
import torch

from oracle import svg_oracle as O

pytestmark = pytest.mark.gpu

H, D, F_, P_, CTX, L = 2, 128, 33, 3600, 256, 64
V = F_ * P_
S = V + CTX


@pytest.fixture(scope="module")
def setup():
    from svg import _native as nat

    nat.load()
    torch.manual_seed(0)
    g = torch.Generator(device="cuda").manual_seed(7)
    q, k, v = (torch.randn(1, H, S, D, device="cuda", dtype=torch.bfloat16, generator=g) for _ in range(3))
    width = O.sparsity_to_width(0.25, CTX, F_, P_)
    prm = O.hy_band_params(S, CTX, L, F_, P_, width)
    assert prm["band"] == 15616
    return nat, q, k, v, prm


def _oracle_rows(q, k, v, rows, prm, temporal):
    """Attention output of selected *logical* rows of one head, fp32, under the band mask in logical order."""
    qh, kh, vh = q.float().cpu(), k.float().cpu(), v.float().cpu()
    if temporal:  # logical order = token-major
        def perm(x):
            return torch.cat([x[:V].reshape(F_, P_, D).transpose(0, 1).reshape(V, D), x[V:]])
        qh, kh, vh = perm(qh), perm(kh), perm(vh)
    kk = torch.arange(S)[None, :]
    qq = torch.tensor(rows)[:, None]
    real, band = prm["real_len"], prm["band"]
    rq, rk = qq < real, kk < real
    m = (rq & rk & (((qq - kk).abs() < band) | ((kk >= prm["colfull_lo"]) & (kk < prm["colfull_hi"])) |
                    ((qq >= prm["rowfull_lo"]) & (qq < prm["rowfull_hi"])))) | (~rq & ~rk)
    return O.masked_attention(qh[rows], kh, vh, m)


def test_fullsize_spot_rows_and_fused_equals_materialised(setup):
    nat, q, k, v, prm = setup
    best = torch.tensor([[0, 1]], device="cuda")
    mask = nat.BandMask(**prm)
    o = nat.band_attention(q, k, v, mask, head_perm_flag=best, vid0=0, num_frame=F_, frame_size=P_)
    # (1) fused layout transformation is bit-identical to placement -> attention -> inverse placement
    qp, kp, vp = (torch.empty_like(x) for x in (q, k, v))
    nat.head_placement([q, k, v], [qp, kp, vp], best, CTX, F_, P_, False, False)
    om = nat.band_attention(qp, kp, vp, mask)
    oi = torch.empty_like(om)
    nat.head_placement([om], [oi], best, CTX, F_, P_, False, True)
    assert torch.equal(oi, o)
    # (2) spot rows (band interior, both band edges, sequence ends, prompt rows, pad rows) against the oracle
    rows = [0, 1, 63, 64, 255, 256, 15615, 15616, 15617, 60000, 60001, V - 15616, V - 1, V, V + 1, V + L - 1, V + L, S - 1]
    rows += torch.randint(0, S, (30,), generator=torch.Generator().manual_seed(1)).tolist()
    for h, temporal in ((0, False), (1, True)):
        ref = _oracle_rows(q[0, h], k[0, h], v[0, h], rows, prm, temporal)
        got = om[0, h].float().cpu()[rows]  # om is in logical order for both heads
        torch.testing.assert_close(got, ref, atol=1e-2, rtol=1e-2)
        assert ((got - ref).norm() / ref.norm()).item() < 3e-3


def test_fullsize_rows_sum_to_one(setup):
    """v = 1 -> every output element is 1 (softmax rows sum to one), for sparse and dense masks."""
    nat, q, k, v, prm = setup
    ones = torch.ones_like(v)
    for m in (prm, O.dense_band_params(S, V + L)):
        o = nat.band_attention(q, k, ones, nat.BandMask(**m))
        assert (o.float() - 1).abs().max().item() <= 2 ** -7


def test_fullsize_linearity_in_v(setup):
    """attention is linear in V: o(v1 + v2) = o(v1) + o(v2) up to bf16 rounding."""
    nat, q, k, v, prm = setup
    mask = nat.BandMask(**prm)
    g = torch.Generator(device="cuda").manual_seed(9)
    v2 = torch.randn(v.shape, device="cuda", dtype=torch.bfloat16, generator=g)
    o1 = nat.band_attention(q, k, v, mask).float()
    o2 = nat.band_attention(q, k, v2, mask).float()
    o12 = nat.band_attention(q, k, (v.float() + v2.float()).to(torch.bfloat16), mask).float()
    err = ((o12 - (o1 + o2)).norm() / (o1 + o2).norm()).item()
    assert err < 6e-3, err


# ---------------------------------------------------------------------------------------------------------
# the other two SVG1 models at their production geometry (the reference's install hooks: svg/models/wan/inference.py:41-44,
# svg/models/cog/inference.py:31-34 with the v1.5 numbers F = 11, P = 4080)
# ---------------------------------------------------------------------------------------------------------
FULLSIZE = {
    # name: (D, F, P, ctx, text_first, sparsity)
    "wan720p": (128, 21, 3600, 0, False, 0.30),
    "cog15_768p": (64, 11, 4080, 226, True, 0.25),
}


@pytest.mark.parametrize("model", sorted(FULLSIZE))
def test_fullsize_spot_rows_wan_cog(model):
    """Wan 2.1 720p (S = 75600: `<=` band of ceil'ed width + first-frame sink columns, no text) and CogVideoX-v1.5 (S = 45106,
    D = 64, text FIRST) at full size: (1) the fused layout transformation equals placement -> attention -> inverse placement bit
    for bit, (2) spot rows of a spatial and a temporal head — band edges, sink / text columns, sequence ends — against the oracle
    under the REFERENCE's mask predicate (svg/models/wan/utils.py:25-41, svg/models/cog/utils.py:30-46) evaluated on those rows."""
    from svg import _native as nat

    nat.load()
    D_, F2, P2, ctx, text_first, sparsity = FULLSIZE[model]
    V2 = F2 * P2
    S2 = V2 + ctx
    vid0 = ctx if text_first else 0
    mul = O.sparsity_to_width(sparsity, ctx, F2, P2)
    if model == "wan720p":
        prm = O.wan_band_params(S2, F2, P2, mul)
        assert prm["band"] == 12416 + 1
        ref_mask = lambda rows: O.wan_mask(S2, F2, P2, mul, rows=rows)        # noqa: E731
        edge = 12416
    else:
        prm = O.cog_band_params(S2, ctx, F2, P2, mul)
        ref_mask = lambda rows: O.cog_mask(S2, ctx, F2, P2, mul, rows=rows)   # noqa: E731
        edge = prm["band"]
    g = torch.Generator(device="cuda").manual_seed(17)
    q, k, v = (torch.randn(1, 2, S2, D_, device="cuda", dtype=torch.bfloat16, generator=g) for _ in range(3))
    best = torch.tensor([[0, 1]], device="cuda")
    mask = nat.BandMask(**prm)
    o = nat.band_attention(q, k, v, mask, head_perm_flag=best, vid0=vid0, num_frame=F2, frame_size=P2)
    qp, kp, vp = (torch.empty_like(x) for x in (q, k, v))
    nat.head_placement([q, k, v], [qp, kp, vp], best, ctx, F2, P2, text_first, False)
    om = nat.band_attention(qp, kp, vp, mask)
    oi = torch.empty_like(om)
    nat.head_placement([om], [oi], best, ctx, F2, P2, text_first, True)
    assert torch.equal(oi, o)
    rows = [0, 1, vid0, vid0 + 1, P2 - 1, P2, P2 + 1, vid0 + edge - 1, vid0 + edge, vid0 + edge + 1, S2 // 2, S2 // 2 + 1,
            S2 - edge - 2, S2 - edge - 1, S2 - edge, S2 - 2, S2 - 1]
    if ctx:
        rows += [ctx - 1, ctx, ctx + 1]
    rows += torch.randint(0, S2, (24,), generator=torch.Generator().manual_seed(3)).tolist()
    m = ref_mask(rows)

    def logical(x):   # token order the mask is applied in: temporal heads are token-major inside the video part
        x = x.float().cpu()
        vidp = x[vid0:vid0 + V2].reshape(F2, P2, D_).transpose(0, 1).reshape(V2, D_)
        return torch.cat([x[:vid0], vidp, x[vid0 + V2:]])

    for h, temporal in ((0, False), (1, True)):
        qh, kh, vh = ((logical(x[0, h]) if temporal else x[0, h].float().cpu()) for x in (q, k, v))
        ref = O.masked_attention(qh[rows], kh, vh, m)
        got = om[0, h].float().cpu()[rows]     # om is in logical order for both heads
        torch.testing.assert_close(got, ref, atol=1e-2, rtol=1e-2)
        assert ((got - ref).norm() / ref.norm()).item() < 3e-3
