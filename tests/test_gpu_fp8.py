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
    ref_dq = O.masked_attention(dequant(q), dequant(k), dequant(v), mask)
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
