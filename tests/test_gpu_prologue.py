"""GPU parity of the pre-attention prologue (svg_rms_norm_forward, svg_layer_norm_forward, svg_apply_qk_rope_inplace_*,
svg_qk_norm_rope) through the C ABI.  Parameter grids follow the reference's tests (svg/kernels/test/test_rms_norm.py:38,
test_layer_norm.py:32, test_apply_rope*.py:39); tolerances are the reference's (bf16 rtol 3e-2 / atol 2e-2, fp16 5e-3) —
and on top of that bit-exactness against the oracle wherever the arithmetic has no reduction."""
from itertools import product
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import svg_oracle as O

pytestmark = pytest.mark.gpu
GOLD = np.load(str(Path(__file__).parent / "golden" / "prologue_golden.npz"))
TOL = {torch.float16: (5e-3, 5e-3), torch.bfloat16: (3e-2, 2e-2)}


@pytest.fixture(scope="module")
def nat():
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "sparse-videogen_amd"))
    from svg import _native
    _native.load()
    return _native


def dev(t):
    return t.cuda()


def close(a, b):
    rtol, atol = TOL[a.dtype]
    torch.testing.assert_close(a.float().cpu(), b.float().cpu(), rtol=rtol, atol=atol)


def ulp_equal(a, b, max_frac, max_ulp=1, abs_ok=0.0):
    """a, b 16-bit float tensors: identical except for at most max_frac of the elements, which differ by <= max_ulp ulps
    (or by <= abs_ok in absolute terms: results of a cancellation, where an ulp is meaningless)"""
    a, b = a.cpu(), b.cpu()
    ne = a != b
    ia, ib = a.view(torch.int16).int(), b.view(torch.int16).int()
    bad = ne & ((ia - ib).abs() > max_ulp) & ((a.float() - b.float()).abs() > abs_ok)
    assert not bad.any(), f"{int(bad.sum())} elements more than {max_ulp} ulp apart, e.g. {a[bad][:4]} vs {b[bad][:4]}"
    assert ne.float().mean().item() <= max_frac, f"{ne.float().mean().item():.2e} of the elements differ"


@pytest.mark.parametrize("m,n", list(product([1, 7, 31, 55, 95, 128, 512], [32, 64, 128, 256])))
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_rms_norm(nat, m, n, dtype):
    torch.manual_seed(m * 1000 + n)
    x, w = torch.randn(m, n).to(dtype), torch.randn(n).to(dtype)
    got = dev(x)
    nat.rms_norm_forward(got, dev(w), 1e-5)
    # the reduction order (and rsqrt's last bit) differs: the normalised value can land on the other side of a rounding
    # boundary (1 ulp), which the weight multiply turns into at most 2 ulps of the result
    ulp_equal(got, O.rms_norm(x, w), 2e-3, 2)
    close(got, torch.nn.functional.rms_norm(x.float(), [n], w.float(), 1e-5).to(dtype))   # the reference's check


@pytest.mark.parametrize("m,n", list(product([1, 7, 31, 55, 95, 128, 512], [32, 64, 128, 256])))
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_layer_norm(nat, m, n, dtype):
    torch.manual_seed(m * 1000 + n + 1)
    x, w, b = torch.randn(m, n).to(dtype), torch.randn(n).to(dtype), torch.randn(n).to(dtype)
    got = dev(x)
    nat.layer_norm_forward(got, dev(w), dev(b))
    ulp_equal(got, O.layer_norm(x, w, b), 5e-3, 2, 1e-4)   # (x - mean) * inv * w + b can cancel
    close(got, torch.nn.functional.layer_norm(x.float(), [n], w.float(), b.float(), 1e-5).to(dtype))


ROPE_GRID = [p for p in product([1, 3], [16], [151, 1037], [64, 128, 256], [15, 77])] + [(1, 32, 6778, 128, 35), (5, 16, 151, 64, 35)]


@pytest.mark.parametrize("bsz,H,S,D,L", ROPE_GRID)
@pytest.mark.parametrize("kind", ["cossin", "txtlast", "complex"])
def test_apply_rope(nat, kind, bsz, H, S, D, L):
    torch.manual_seed(S + D + L)
    dtype = torch.float16 if kind == "complex" else torch.bfloat16      # as in the reference's tests
    q, k = torch.randn(bsz, H, S, D).to(dtype), torch.randn(bsz, H, S, D).to(dtype)
    cols = D // 2 if kind == "complex" else D
    a, b = torch.randn(S - L, cols), torch.randn(S - L, cols)
    gq, gk = dev(q), dev(k)
    fn = {"cossin": nat.apply_qk_rope_inplace_cossin, "txtlast": nat.apply_qk_rope_inplace_cossin_txtlast,
          "complex": nat.apply_qk_rope_inplace_cossin_complex}[kind]
    fn(gq, gk, dev(a), dev(b), L)
    rq, rk = O.apply_qk_rope(q, k, a, b, L, kind)
    assert torch.equal(gq.cpu(), rq) and torch.equal(gk.cpu(), rk), "bit-exact (no reduction in this arithmetic)"


def test_rope_gqa_and_bf16_complex(nat):
    torch.manual_seed(5)
    bsz, Hq, Hkv, S, D, L = 2, 8, 2, 300, 128, 20
    q, k = torch.randn(bsz, Hq, S, D).to(torch.bfloat16), torch.randn(bsz, Hkv, S, D).to(torch.bfloat16)
    fr, fi = torch.randn(S - L, D // 2), torch.randn(S - L, D // 2)
    gq, gk = dev(q), dev(k)
    nat.apply_qk_rope_inplace_cossin_complex(gq, gk, dev(fr), dev(fi), L)
    assert torch.equal(gq.cpu(), O.apply_qk_rope(q, q, fr, fi, L, "complex")[0])
    assert torch.equal(gk.cpu(), O.apply_qk_rope(k, k, fr, fi, L, "complex")[0])


def test_goldens_from_the_reference(nat):
    """Vectors produced by the reference's own torch reference functions (tests/golden/make_golden_prologue.py)."""
    def t16(name, dtype=torch.bfloat16):
        return torch.from_numpy(GOLD[name].copy()).view(dtype)
    for tag in ["7x32", "31x64", "95x128", "128x256"]:
        x, w, b = t16(f"norm_x_{tag}"), t16(f"norm_w_{tag}"), t16(f"norm_b_{tag}")
        g = dev(x); nat.rms_norm_forward(g, dev(w), 1e-5)
        ulp_equal(g, t16(f"rms_replica_{tag}"), 2e-3, 2)
        close(g, t16(f"rms_ref_{tag}"))
        g = dev(x); nat.layer_norm_forward(g, dev(w), dev(b))
        ulp_equal(g, t16(f"ln_ref_{tag}"), 5e-3, 2, 1e-4)
    for tag in ["1_2_151_64_15", "2_1_151_128_35", "1_1_151_256_77"]:
        L = int(tag.split("_")[-1])
        q, cos, sin = t16(f"rope_q_{tag}"), torch.from_numpy(GOLD[f"rope_cos_{tag}"]), torch.from_numpy(GOLD[f"rope_sin_{tag}"])
        a, b = dev(q), dev(q)
        nat.apply_qk_rope_inplace_cossin(a, b, dev(cos), dev(sin), L)
        assert torch.equal(a.cpu()[:, :, L:], t16(f"rope_first_{tag}")) and torch.equal(a.cpu()[:, :, :L], q[:, :, :L])
        a, b = dev(q), dev(q)
        nat.apply_qk_rope_inplace_cossin_txtlast(a, b, dev(cos), dev(sin), L)
        assert torch.equal(a.cpu()[:, :, :-L], t16(f"rope_last_{tag}")) and torch.equal(a.cpu()[:, :, -L:], q[:, :, -L:])
        qh = t16(f"cplx_q_{tag}", torch.float16)
        fr, fi = torch.from_numpy(GOLD[f"cplx_fr_{tag}"]), torch.from_numpy(GOLD[f"cplx_fi_{tag}"])
        a, b = dev(qh), dev(qh)
        nat.apply_qk_rope_inplace_cossin_complex(a, b, dev(fr), dev(fi), L)
        assert torch.equal(a.cpu()[:, :, L:], t16(f"cplx_out_{tag}", torch.float16))


@pytest.mark.parametrize("norm,rope", [(1, 1), (2, 1), (1, 2), (1, 0), (0, 1)])
def test_fused_equals_sequence(nat, norm, rope):
    """svg_qk_norm_rope == norm entry point followed by rope entry point, bit for bit (Hunyuan: rms + txtlast; Cog: layer +
    text-first; Wan: rms over the full hidden size is NOT this op, only its rope is)."""
    torch.manual_seed(11)
    bsz, Hq, Hkv, S, D, L = 2, 6, 3, 777, 128, 64
    dt = torch.bfloat16
    q, k = torch.randn(bsz, Hq, S, D).to(dt), torch.randn(bsz, Hkv, S, D).to(dt)
    qw, qb, kw, kb = (torch.randn(D).to(dt) for _ in range(4))
    cols = D // 2 if rope == 2 else D
    cs, sn = torch.randn(S - L, cols), torch.randn(S - L, cols)
    fq, fk = dev(q), dev(k)
    nat.qk_norm_rope(fq, fk, norm, dev(qw), dev(qb) if norm == 2 else None, dev(kw), dev(kb) if norm == 2 else None, 1e-6,
                     rope, dev(cs) if rope else None, dev(sn) if rope else None, 0, S - L)
    sq, sk = dev(q), dev(k)
    if norm == 1:
        nat.rms_norm_forward(sq.view(-1, D), dev(qw), 1e-6); nat.rms_norm_forward(sk.view(-1, D), dev(kw), 1e-6)
    elif norm == 2:
        # the stand-alone layer-norm entry point fixes eps = 1e-5 like the reference; run the fused op without rope instead
        nat.qk_norm_rope(sq, sk, 2, dev(qw), dev(qb), dev(kw), dev(kb), 1e-6)
    if rope == 1:
        nat.apply_qk_rope_inplace_cossin_txtlast(sq, sk, dev(cs), dev(sn), L)
    elif rope == 2:
        nat.qk_norm_rope(sq, sk, 0, rope_kind=2, cos=dev(cs), sin=dev(sn), rope_lo=0, rope_hi=S - L)
    assert torch.equal(fq, sq) and torch.equal(fk, sk)


@pytest.mark.parametrize("norm,rope,D", [(1, 1, 128), (2, 1, 64), (0, 2, 128), (0, 0, 256)])
def test_transposing_form_equals_transpose_then_in_place(nat, norm, rope, D):
    """svg_qk_norm_rope_transpose([bsz, S, H*D]) == transpose(1, 2).contiguous() followed by svg_qk_norm_rope, bit for bit."""
    torch.manual_seed(13)
    bsz, Hq, Hkv, S, L = 2, 6, 3, 533, 40
    dt = torch.bfloat16
    q_in, k_in = torch.randn(bsz, S, Hq * D).to(dt).cuda(), torch.randn(bsz, S, Hkv * D).to(dt).cuda()
    qw, qb, kw, kb = (torch.randn(D).to(dt).cuda() for _ in range(4))
    cols = D // 2 if rope == 2 else D
    cs, sn = torch.randn(S - L, cols).cuda(), torch.randn(S - L, cols).cuda()
    args = (norm, qw, qb if norm == 2 else None, kw, kb if norm == 2 else None, 1e-6, rope, cs if rope else None,
            sn if rope else None, L, S)
    q_out, k_out = nat.qk_norm_rope_transpose(q_in, k_in, Hq, Hkv, *args)
    q_ref = q_in.unflatten(2, (Hq, D)).transpose(1, 2).contiguous()
    k_ref = k_in.unflatten(2, (Hkv, D)).transpose(1, 2).contiguous()
    if norm or rope:
        nat.qk_norm_rope(q_ref, k_ref, *args)
    assert torch.equal(q_out, q_ref) and torch.equal(k_out, k_ref)
    v_out, none = nat.qk_norm_rope_transpose(k_in, None, Hkv, 0)       # plain transpose of one tensor (V)
    assert none is None and torch.equal(v_out, k_in.unflatten(2, (Hkv, D)).transpose(1, 2).contiguous())


def test_full_size_hunyuan_prologue(nat):
    """HunyuanVideo 720p shape (H = 24, S = 119056, D = 128): fused rms-norm + text-last rope; spot rows against the oracle,
    text rows un-rotated, and the fused pass equals the two separate passes."""
    torch.manual_seed(3)
    H, S, D, L = 24, 119056, 128, 256
    dt = torch.bfloat16
    q = torch.randn(1, H, S, D, device="cuda", dtype=dt)
    k = torch.randn(1, H, S, D, device="cuda", dtype=dt)
    qw, kw = torch.randn(D, device="cuda").to(dt), torch.randn(D, device="cuda").to(dt)
    cos, sin = torch.randn(S - L, D, device="cuda"), torch.randn(S - L, D, device="cuda")
    q0, k0 = q.clone(), k.clone()
    nat.qk_norm_rope(q, k, 1, qw, None, kw, None, 1e-6, 1, cos, sin, 0, S - L)
    a, b = q0.clone(), k0.clone()
    nat.rms_norm_forward(a.view(-1, D), qw, 1e-6); nat.rms_norm_forward(b.view(-1, D), kw, 1e-6)
    nat.apply_qk_rope_inplace_cossin_txtlast(a, b, cos, sin, L)
    assert torch.equal(q, a) and torch.equal(k, b)
    rows = torch.tensor([0, 1, 63, 64, 4097, 59999, S - L - 1, S - L, S - 1])
    for x0, x1, w in ((q0, q, qw), (k0, k, kw)):
        sub = x0[:, :, rows].cpu()
        ref = O.rms_norm(sub, w.cpu(), 1e-6)
        pos = rows < S - L
        ref[:, :, pos] = O.rope_cossin(ref[:, :, pos], cos[rows[pos]].cpu(), sin[rows[pos]].cpu())
        ulp_equal(x1[:, :, rows], ref, 2e-3, 2)


# ---------------------------------------------------------------------------------------------------------
# the Wan 2.1 prologue in one pass (svg_rmsnorm_rope_transpose, round 6): RMSNorm across ALL heads + RoPE + head-major transpose
# ref: svg/models/wan/attention.py:99-148 (get_qk_norm / get_transpose_qkv / get_rotary_emb)
# ---------------------------------------------------------------------------------------------------------
def _triton_rmsnorm(x, w, eps):
    """the reference's Triton RMSNorm (svg/kernels/triton/rmsnorm.py:8-48): fp32 x * rstd * w, ONE rounding"""
    xf = x.float()
    y = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return (y * w.float() if w is not None else y).to(x.dtype)


@pytest.mark.parametrize("H,D,rope,dtype,wdt", [
    (40, 128, 2, torch.bfloat16, torch.bfloat16),      # Wan 2.1 14B: 5120 = 640 chunks, ten per lane
    (12, 128, 2, torch.bfloat16, torch.float32),       # Wan 2.1 1.3B: 1536 = 192 chunks; fp32 weights
    (5, 64, 1, torch.float16, torch.float16),          # 40 chunks: lanes without a chunk; cos-sin table
    (7, 128, 0, torch.bfloat16, torch.bfloat16),       # no rotation
    (64, 128, 2, torch.float16, torch.float16),        # 8192: the largest row
])
def test_rmsnorm_rope_transpose_equals_the_three_passes(nat, H, D, rope, dtype, wdt):
    """one launch == svg_rmsnorm_forward(q), svg_rmsnorm_forward(k), svg_qk_norm_rope_transpose(norm 0, rope), transpose(v) bit for bit
    (partial rotation range, q_scale folded into q's last rounding), and == the torch statement of the reference's three steps"""
    torch.manual_seed(H * D + rope)
    bsz, S, lo, hi = 2, 333, 7, 301
    q_in, k_in, v_in = (torch.randn(bsz, S, H * D).to(dtype).cuda() * 1.7 for _ in range(3))
    qw, kw = (torch.randn(H * D).mul(0.3).add(1.0).to(wdt).cuda() for _ in range(2))
    cols = D // 2 if rope == 2 else D
    cs, sn = (torch.randn(hi - lo, cols).cuda(), torch.randn(hi - lo, cols).cuda()) if rope else (None, None)
    for q_scale in (1.0, 0.1275):
        q, k, v = nat.rmsnorm_rope_transpose(q_in, k_in, v_in, H, qw, kw, 1e-6, rope, cs, sn, lo, hi, q_scale=q_scale)
        qn, kn = nat.rmsnorm_forward(q_in, qw, 1e-6), nat.rmsnorm_forward(k_in, kw, 1e-6)
        q_ref, k_ref = nat.qk_norm_rope_transpose(qn, kn, H, H, 0, None, None, None, None, 1e-6, rope, cs, sn, lo, hi, q_scale=q_scale)
        assert torch.equal(q, q_ref) and torch.equal(k, k_ref)
        assert torch.equal(v, v_in.unflatten(2, (H, D)).transpose(1, 2).contiguous())
    # partial calls: k or v absent, no weights
    q1, k1, v1 = nat.rmsnorm_rope_transpose(q_in, None, None, H, qw, None, 1e-6, rope, cs, sn, lo, hi)
    assert k1 is None and v1 is None
    q2, _, _ = nat.rmsnorm_rope_transpose(q_in, k_in, v_in, H, qw, kw, 1e-6, rope, cs, sn, lo, hi)
    assert torch.equal(q1, q2)
    q3, _, _ = nat.rmsnorm_rope_transpose(q_in, None, None, H, None, None, 1e-6, 0)
    assert torch.equal(q3, nat.rmsnorm_forward(q_in, None, 1e-6).unflatten(2, (H, D)).transpose(1, 2).contiguous())
    # the reference's statement of the same three steps in torch (fp32 norm with one rounding; fp64 complex rotation)
    ref = _triton_rmsnorm(q_in.cpu(), qw.cpu(), 1e-6).unflatten(2, (H, D)).transpose(1, 2).contiguous()
    if rope == 2:
        ref[:, :, lo:hi] = O.rope_complex(ref[:, :, lo:hi], cs.cpu(), sn.cpu())
    elif rope == 1:
        ref[:, :, lo:hi] = O.rope_cossin(ref[:, :, lo:hi], cs.cpu(), sn.cpu())
    # (the torch norm sums in another order: rstd may differ in its last fp32 bit, the rounded norm then by one ulp of T in a few elements, and
    #  the rotation — table entries ~N(0, 1) here — multiplies that: the reference's own tolerance, and nearly all elements equal)
    close(q2, ref)
    ulp_equal(q2, ref, 5e-3, 4, abs_ok=6e-2)


def test_full_size_wan_prologue(nat):
    """Wan 2.1 720p (S = 75600, 40 x 128): the fused pass equals the unfused sequence at full size; reports both times"""
    torch.manual_seed(4)
    H, S, D = 40, 75600, 128
    dt = torch.bfloat16
    q_in, k_in, v_in = (torch.randn(1, S, H * D, device="cuda", dtype=dt) for _ in range(3))
    qw, kw = (torch.randn(H * D, device="cuda").mul(0.2).add(1.5).to(dt) for _ in range(2))
    fr, fi = torch.randn(S, D // 2, device="cuda"), torch.randn(S, D // 2, device="cuda")

    def fused():
        return nat.rmsnorm_rope_transpose(q_in, k_in, v_in, H, qw, kw, 1e-6, 2, fr, fi, 0, S)

    def unfused():
        qn, kn = nat.rmsnorm_forward(q_in, qw, 1e-6), nat.rmsnorm_forward(k_in, kw, 1e-6)
        q, k = nat.qk_norm_rope_transpose(qn, kn, H, H, 0, None, None, None, None, 1e-6, 2, fr, fi, 0, S)
        v, _ = nat.qk_norm_rope_transpose(v_in, None, H, 0)
        return q, k, v

    a, b = fused(), unfused()
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    del a, b
    times = {}
    for name, fn in (("fused", fused), ("unfused", unfused)):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        times[name] = e0.elapsed_time(e1) / 5
    alg = 6 * S * H * D * 2      # q, k, v read once and written once
    print(f"\n[wan 720p prologue] fused {times['fused']:.3f} ms = {alg / times['fused'] / 1e6:.0f} GB/s algorithmic; the three-pass sequence "
          f"{times['unfused']:.3f} ms")
    assert times["fused"] < times["unfused"]
