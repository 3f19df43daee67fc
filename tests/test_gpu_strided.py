"""Attention straight out of / into the projection layout (include/svg_attn.h, svg_attn_layout_t; svg_band_attention_strided,
svg_band_attention_switch_strided, svg_varblock_attention_strided, svg_sample_mse_strided).

ref: the reference's processors hand `proj(x).unflatten(2, (heads, -1)).transpose(1, 2)` views to flex_attention / flash-attn and turn the
result back with `.transpose(1, 2).flatten(2, 3)` (svg/models/wan/attention.py:123-125,168-170, hyvideo/attention.py:83-85,202).

The strided entry points run the same arithmetic on the same values as the contiguous ones: every comparison here is BIT-EXACT
(torch.equal) against the contiguous call, which test_gpu_m16.py / test_gpu_kernels.py hold to the oracle; one case per kernel is also
checked against the oracle directly."""
import pytest
import torch

from oracle import svg_oracle as O
from test_gpu_kernels import _band_case, check_attn, dev, random_partition_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    from svg import _native

    _native.load()
    return _native


def _proj_views(B, H, S, D, dtype, fused_qkv, seed):
    """q, k, v as [B, H, S, D] views of projection outputs on the GPU: three [B, S, H * D] tensors, or slices of one [B, S, 3 * H * D]"""
    g = torch.Generator().manual_seed(seed)
    if fused_qkv:
        buf = dev(torch.randn(B, S, 3 * H * D, generator=g).to(dtype))
        return tuple(buf[:, :, i * H * D:(i + 1) * H * D].unflatten(2, (H, D)).transpose(1, 2) for i in range(3))
    return tuple(dev(torch.randn(B, S, H * D, generator=g).to(dtype)).unflatten(2, (H, D)).transpose(1, 2) for _ in range(3))


def _is_token_major(o):
    return o.transpose(1, 2).is_contiguous()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("fused_qkv", [False, True])
@pytest.mark.parametrize("D", [128, 64])
@pytest.mark.parametrize("model", ["hy", "wan", "cog", "dense2"])
def test_band_attention_strided_equals_contiguous(nat, model, D, fused_qkv, dtype):
    F_, P_, ctx, L, mul = 5, 150, 40, 11, 2.3
    S, prm, mask, _ = _band_case(model, F_, P_, ctx, L, mul)
    B, H = 2, 3
    q, k, v = _proj_views(B, H, S, D, dtype, fused_qkv, seed=7)
    assert not v.is_contiguous()
    m = nat.BandMask(**prm)
    ref = nat.band_attention(q.contiguous(), k.contiguous(), v.contiguous(), m)
    o = nat.band_attention(q, k, v, m, token_major_out=True)          # everything strided, o token-major
    assert o.shape == ref.shape and _is_token_major(o) and torch.equal(o, ref)
    o2 = nat.band_attention(q.contiguous(), k.contiguous(), v, m)      # what the processors pass: q, k head-major, v in place
    assert o2.is_contiguous() and torch.equal(o2, ref)
    o3 = nat.band_attention(q.contiguous(), k.contiguous(), v.contiguous(), m, token_major_out=True)
    assert _is_token_major(o3) and torch.equal(o3, ref)
    flat = o.transpose(1, 2).flatten(2, 3)                             # the processors' next line: a view now
    assert flat.data_ptr() == o.data_ptr() and flat.shape == (B, S, H * D)
    if model == "hy" and not fused_qkv:
        check_attn(o[:1], O.masked_attention(q[:1].float().cpu().to(dtype), k[:1].float().cpu().to(dtype), v[:1].float().cpu().to(dtype), mask), dtype)


@pytest.mark.parametrize("D", [128, 64])
@pytest.mark.parametrize("model", ["hy", "cog"])
def test_band_attention_strided_fused_placement(nat, model, D):
    """token-major heads (head_perm_flag) gather K / V rows and scatter O rows through the placement index AND the row strides"""
    F_, P_, ctx, L, mul, H, B = 6, 130, 24, 7, 1.6, 4, 2
    S, prm, mask, vid0 = _band_case(model, F_, P_, ctx, L, mul)
    q, k, v = _proj_views(B, H, S, D, torch.bfloat16, True, seed=3)
    best = dev(torch.tensor([[0, 1, 1, 0], [1, 0, 1, 1]]))
    kw = dict(head_perm_flag=best, vid0=vid0, num_frame=F_, frame_size=P_)
    ref = nat.band_attention(q.contiguous(), k.contiguous(), v.contiguous(), nat.BandMask(**prm), **kw)
    o = nat.band_attention(q, k, v, nat.BandMask(**prm), token_major_out=True, **kw)
    assert _is_token_major(o) and torch.equal(o, ref)


@pytest.mark.parametrize("D", [128, 64])
def test_band_attention_switch_strided(nat, D):
    F_, P_, ctx, L, mul, H, B = 5, 150, 40, 11, 2.3, 3, 1
    S, prm, mask, vid0 = _band_case("hy", F_, P_, ctx, L, mul)
    q, k, v = _proj_views(B, H, S, D, torch.bfloat16, False, seed=11)
    m, dm = nat.BandMask(**prm), nat.BandMask(**O.dense_band_params(S, F_ * P_ + L))
    best = dev(torch.tensor([[1, 0, 1]]))
    kw = dict(head_perm_flag=best, vid0=vid0, num_frame=F_, frame_size=P_)
    for flag in (0, 1):
        f = dev(torch.tensor([flag], dtype=torch.int32))
        ref = nat.band_attention_switch(q.contiguous(), k.contiguous(), v.contiguous(), m, dm, f, **kw)
        o = nat.band_attention_switch(q.contiguous(), k.contiguous(), v, m, dm, f, token_major_out=True, **kw)
        assert _is_token_major(o) and torch.equal(o, ref), flag


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("D", [128, 64])
@pytest.mark.parametrize("hq,hkv", [(4, 4), (8, 2)])
def test_varblock_attention_strided_equals_contiguous(nat, hq, hkv, D, dtype):
    S, MB, NB = 4096, 12, 40
    gen = torch.Generator().manual_seed(77 + hq)
    rsz, csz = random_partition_batch(S, MB, hkv, gen), random_partition_batch(S, NB, hkv, gen)
    bmap = torch.rand(hkv, MB, NB, generator=gen) > 0.5
    q = dev(torch.randn(1, S, hq * D, generator=gen).to(dtype)).unflatten(2, (hq, D)).transpose(1, 2)
    kvbuf = dev(torch.randn(1, S, 2 * hkv * D, generator=gen).to(dtype))
    k, v = (kvbuf[:, :, i * hkv * D:(i + 1) * hkv * D].unflatten(2, (hkv, D)).transpose(1, 2) for i in range(2))
    qi = dev(torch.stack([torch.randperm(S, generator=gen) for _ in range(hq)]).to(torch.int32))
    ki = dev(torch.stack([torch.randperm(S, generator=gen) for _ in range(hkv)]).to(torch.int32))
    args = (dev(bmap), dev(rsz), dev(csz))
    for kw in (dict(), dict(q_row_idx=qi, kv_row_idx=ki)):
        ref = nat.varblock_attention(q[0].contiguous(), k[0].contiguous(), v[0].contiguous(), *args, **kw)     # [H, S, D], zero-filled
        o = nat.varblock_attention(q, k, v, *args, token_major_out=True, rows_covered=True, **kw)            # [1, H, S, D]
        assert o.shape == (1, hq, S, D) and _is_token_major(o) and torch.equal(o[0], ref)
        o2 = nat.varblock_attention(q.contiguous(), k.contiguous(), v, *args, rows_covered=True, **kw)         # v in place only
        assert o2.is_contiguous() and torch.equal(o2[0], ref)
        o3 = nat.varblock_attention(q[0].contiguous(), k[0].contiguous(), v[0].contiguous(), *args, rows_covered=True, **kw)
        assert torch.equal(o3, ref)   # the partitions cover every row: no zero fill needed


def test_varblock_rows_covered_with_keyless_block_rows(nat):
    """rows_covered=True hands the kernel an uninitialised output: a block-row whose map row is empty (no key at all) still gets its tiles and
    writes zeros, and so does a block-row whose only key blocks are empty clusters"""
    S, D, H, MB, NB = 4096, 128, 2, 10, 16
    gen = torch.Generator().manual_seed(9)
    rsz = random_partition_batch(S, MB, H, gen)
    csz = random_partition_batch(S, NB - 1, H, gen)
    csz = torch.cat([csz, torch.zeros(H, 1, dtype=torch.int32)], dim=1)      # the last key cluster is empty
    bmap = torch.rand(H, MB, NB, generator=gen) > 0.5
    bmap[:, 3, :] = False                                                     # no key block at all
    bmap[:, 6, :] = False
    bmap[:, 6, NB - 1] = True                                                 # only the empty cluster
    q, k, v = (dev(torch.randn(H, S, D, generator=gen).to(torch.bfloat16)) for _ in range(3))
    args = (dev(bmap), dev(rsz), dev(csz))
    ref = nat.varblock_attention(q, k, v, *args)                               # zero-filled output
    poison = [dev(torch.full((H, S, D), float("nan")).to(torch.bfloat16)) for _ in range(3)]   # whatever torch.empty hands out next: NaNs
    del poison
    o = nat.varblock_attention(q, k, v, *args, rows_covered=True)
    assert torch.equal(o, ref)
    for h in range(H):
        off = torch.cumsum(torch.cat([torch.zeros(1, dtype=torch.int64), rsz[h].to(torch.int64)]), 0)
        for i in (3, 6):
            assert float(o[h, off[i]:off[i + 1]].float().abs().max()) == 0.0 if off[i + 1] > off[i] else True


def test_varblock_uncovered_rows_stay_zero_when_strided(nat):
    """q_sizes that do not cover Sq: without rows_covered the wrapper zero-fills the token-major output like the contiguous one"""
    S, D, H = 2048, 128, 2
    gen = torch.Generator().manual_seed(5)
    rsz = torch.tensor([[700, 600], [300, 900]], dtype=torch.int32)          # 1300 / 1200 of 2048 rows
    csz = random_partition_batch(S, 6, H, gen)
    bmap = torch.ones(H, 2, 6, dtype=torch.bool)
    q, k, v = (dev(torch.randn(1, S, H * D, generator=gen).to(torch.bfloat16)).unflatten(2, (H, D)).transpose(1, 2) for _ in range(3))
    ref = nat.varblock_attention(q[0].contiguous(), k[0].contiguous(), v[0].contiguous(), dev(bmap), dev(rsz), dev(csz))
    o = nat.varblock_attention(q, k, v, dev(bmap), dev(rsz), dev(csz), token_major_out=True)
    assert torch.equal(o[0], ref) and float(o[0, 0, 1300:].abs().max()) == 0.0


@pytest.mark.parametrize("D", [128, 64])
def test_sample_mse_strided_equals_contiguous(nat, D):
    from svg.models.hyvideo.utils import profile_desc

    F_, P_, ctx, H, B = 6, 260, 24, 3, 2
    S = F_ * P_ + ctx
    q, k, v = _proj_views(B, H, S, D, torch.bfloat16, True, seed=21)
    rows = dev(torch.randint(0, F_ * P_, (64,), generator=torch.Generator().manual_seed(1)))
    prof = profile_desc(ctx, F_, P_)
    ref = nat.sample_mse(q.contiguous().view(B * H, S, D), k.contiguous().view(B * H, S, D), v.contiguous().view(B * H, S, D), rows, prof)
    got = nat.sample_mse(q, k, v, rows, prof)                      # all three read in place
    got2 = nat.sample_mse(q.contiguous(), k.contiguous(), v, rows, prof)
    assert torch.equal(got, ref) and torch.equal(got2, ref) and torch.isfinite(ref).all()
    f = dev(torch.zeros(1, dtype=torch.int32))
    assert torch.equal(nat.sample_mse(q, k, v, rows, prof, skip_flag=f), ref)
    h = lambda t: t.to(torch.float16)  # noqa: E731  fp16 runs the first form of the kernel: strided inputs are copied, same result
    assert torch.equal(nat.sample_mse(h(q.contiguous()).transpose(1, 2).contiguous().transpose(1, 2), h(k), h(v), rows, prof),
                       nat.sample_mse(h(q).contiguous(), h(k).contiguous(), h(v).contiguous(), rows, prof))


def test_strided_fallbacks_copy(nat):
    """what no strided entry point takes is copied, as the reference does: an explicit schedule, a pre-scaled q"""
    F_, P_, ctx, L, mul = 5, 150, 40, 11, 2.3
    S, prm, mask, _ = _band_case("hy", F_, P_, ctx, L, mul)
    m = nat.BandMask(**prm)
    for D, kw in ((64, dict(variant=3)), (128, dict(variant=2)), (128, dict(q_prescaled=True))):
        q, k, v = _proj_views(1, 3, S, D, torch.bfloat16, False, seed=D)
        ref = nat.band_attention(q.contiguous(), k.contiguous(), v.contiguous(), m, **kw)
        o = nat.band_attention(q, k, v, m, token_major_out=True, **kw)
        assert torch.equal(o, ref)


def test_strided_abi_argument_checks(nat):
    """layout_from_abi (csrc/svg_common.h): strides that are not multiples of 16 bytes, rows shorter than D, k / v whose byte offsets do
    not fit 32 bits -> error codes, no launch"""
    import ctypes as C

    lib = nat.load()
    B, H, S, D = 1, 2, 512, 128
    x = dev(torch.zeros(B, H, S, D, dtype=torch.bfloat16))
    m = nat.BandMask(**O.dense_band_params(S))
    st = torch.cuda.current_stream().cuda_stream

    def call(lay, D_=D):
        return lib.svg_band_attention_strided(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), B * H, S, D_, 0, 0.1, C.byref(m), None,
                                              C.byref(lay) if lay is not None else None, st)

    good = nat.attn_layout(x, x, x, x)
    assert call(good) == 0
    assert call(None) == -1
    bad = nat.attn_layout(x, x, x, x)
    bad.v.row = D + 4
    assert call(bad) == -2
    bad = nat.attn_layout(x, x, x, x)
    bad.q.row = D - 8
    assert call(bad) == -1
    bad = nat.attn_layout(x, x, x, x)
    bad.k.row = (1 << 32) // (2 * S)
    assert call(bad) == -2
    bad = nat.attn_layout(x, x, x, x)
    bad.heads_per_batch = 3
    assert call(bad) == -1
    torch.cuda.synchronize()
    # the variable-block entry point on block-rows too small for the two-phase body: unsupported, the binding copies
    H2, S2, QB, KB = 2, 512, 8, 8
    ws = dev(torch.zeros(int(lib.svg_varblock_workspace_bytes(H2, H2, QB, KB, S2)), dtype=torch.uint8))
    sz = dev(torch.full((H2, QB), S2 // QB, dtype=torch.int32))
    bm = dev(torch.ones(H2, QB, KB, dtype=torch.uint8))
    rc = lib.svg_varblock_attention_strided(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), H2, H2, S2, S2, D, 0, 0.1, bm.data_ptr(),
                                            sz.data_ptr(), sz.data_ptr(), QB, KB, None, None, ws.data_ptr(), ws.numel(), C.byref(good), st)
    assert rc == -2
    torch.cuda.synchronize()


def test_core_svg1_token_major_io(nat, monkeypatch):
    """svg.models._core: with v a view of the projection output the sparse branch returns o stored token-major — the processors'
    `.transpose(1, 2).flatten(2, 3)` is a view — and the values equal the all-contiguous path (TOKEN_MAJOR_IO = False)"""
    from svg.models import _core
    from svg.models.hyvideo.utils import profile_desc

    F_, P_, ctx, L, mul, D, H = 6, 260, 24, 7, 1.6, 128, 4
    S, prm, mask, vid0 = _band_case("hy", F_, P_, ctx, L, mul)
    geo = _core.Geometry(ctx, F_, P_)
    q, k, v = _proj_views(1, H, S, D, torch.bfloat16, False, seed=31)
    q, k = q.contiguous(), k.contiguous()
    prof = profile_desc(ctx, F_, P_)
    outs = {}
    for on in (True, False):
        monkeypatch.setattr(_core, "TOKEN_MAJOR_IO", on)
        torch.manual_seed(0)
        outs[on], best = _core.svg1_sparse_attention(q, k, v if on else v.contiguous(), geo, nat.BandMask(**prm), prof, 32, F_ * P_)
    assert _is_token_major(outs[True]) and outs[False].is_contiguous() and torch.equal(outs[True], outs[False])
    flat = outs[True].transpose(1, 2).flatten(2, 3)
    assert flat.data_ptr() == outs[True].data_ptr()
    monkeypatch.setattr(_core, "TOKEN_MAJOR_IO", True)
    d = _core.dense_attention(q, k, v, valid_len=F_ * P_ + L)
    assert _is_token_major(d) and torch.equal(d, nat.band_attention(q, k, v.contiguous(), nat.BandMask(**O.dense_band_params(S, F_ * P_ + L))))
    assert _core.value_in_place(dev(torch.zeros(1, S, H * D, dtype=torch.bfloat16)), H) is not None
    assert _core.value_in_place(dev(torch.zeros(1, S, H * 64, dtype=torch.bfloat16)), H) is not None
    assert _core.value_in_place(dev(torch.zeros(1, S, H * 32, dtype=torch.bfloat16)), H) is None
    assert _core.value_in_place(dev(torch.zeros(1, S, H * D, dtype=torch.float32)), H) is None


def test_core_svg2_token_major_io(nat, monkeypatch):
    """svg2_sparse_attention (k-means -> block map -> variable-block attention): v in place, o token-major, no zero fill == the
    all-contiguous path bit for bit"""
    from svg.models import _core

    F_, P_, D, H = 4, 1024, 128, 2
    S = F_ * P_
    geo = _core.Geometry(0, F_, P_)
    g = torch.Generator().manual_seed(41)
    cent = torch.randn(8, H * D, generator=g) * 1.5
    x = cent[torch.randint(0, 8, (S,), generator=g)] + 0.35 * torch.randn(S, H * D, generator=g)
    q, k, v = (dev((x @ torch.randn(H * D, H * D, generator=g) / (H * D) ** 0.5)[None].to(torch.bfloat16)).unflatten(2, (H, D)).transpose(1, 2)
               for _ in range(3))
    q, k = q.contiguous(), k.contiguous()
    outs = {}
    for on in (True, False):
        monkeypatch.setattr(_core, "TOKEN_MAJOR_IO", on)
        store = _core.CentroidStore()
        torch.manual_seed(0)       # (the first call of a layer draws its initial centroids from the global generator)
        outs[on] = _core.svg2_sparse_attention(q, k, v if on else v.contiguous(), geo, store, 0, 16, 32, 0.9, 0.1, 3, 2)
    assert _is_token_major(outs[True]) and torch.equal(outs[True], outs[False]) and torch.isfinite(outs[True].float()).all()


@pytest.mark.parametrize("D", [128, 64])
def test_kmeans_loop_batch_strided_equals_contiguous(nat, D):
    """svg_kmeans_loop_strided: the video tokens `x[:, :V]` of a [H, S, D] tensor read in place (heads S * D apart) == the loop on the
    contiguous copy the reference makes (svg/models/hyvideo/attention.py:592-599), bit for bit; through batch_kmeans_Euclid as well"""
    from svg.kmeans_utils import batch_kmeans_Euclid

    H, S, V, K = 3, 2304, 2048, 24
    g = torch.Generator().manual_seed(3)
    cent = torch.randn(8, D, generator=g) * 1.5
    full = dev((cent[torch.randint(0, 8, (H, S), generator=g)] + 0.35 * torch.randn(H, S, D, generator=g)).to(torch.bfloat16))
    xv = full[:, :V]                       # [H, V, D], stride (S * D, D, 1)
    assert not xv.is_contiguous()
    c0 = xv[:, :K].contiguous()
    for iters in (1, 2, 5):
        ref = nat.kmeans_loop(xv.contiguous(), None, c0, iters, 1e-4)
        got = nat.kmeans_loop(xv, None, c0, iters, 1e-4)
        for a, b in zip(ref, got):
            assert torch.equal(a, b), iters
    a = batch_kmeans_Euclid(xv, K, max_iters=3, init_centroids=c0, return_sorted_indices=True, check_every=0)
    b = batch_kmeans_Euclid(xv.contiguous(), K, max_iters=3, init_centroids=c0, return_sorted_indices=True, check_every=0)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    # a batch stride below N * D or not a multiple of 8 elements is refused
    import ctypes as C

    lib = nat.load()
    z = dev(torch.zeros(64, dtype=torch.uint8))
    args = (c0.data_ptr(), c0.data_ptr(), c0.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), c0.data_ptr(), z.data_ptr(), H, V, K, D, 0, 1, 1e-4,
            z.data_ptr(), 64, torch.cuda.current_stream().cuda_stream)
    assert lib.svg_kmeans_loop_strided(xv.data_ptr(), V * D - 8, *args) == -1
    assert lib.svg_kmeans_loop_strided(xv.data_ptr(), S * D + 4, *args) == -2


def test_core_svg2_video_tokens_in_place(nat, monkeypatch):
    """svg2_sparse_attention on a text-last model (HunyuanVideo): the k-means reads the video tokens of q and k as views — same output as
    with the copies (the head-sharded path's form), bit for bit"""
    from svg.models import _core

    F_, P_, ctx, L, D, H = 4, 512, 64, 20, 128, 2
    V = F_ * P_
    S = V + ctx
    geo = _core.Geometry(ctx, F_, P_)
    g = torch.Generator().manual_seed(43)
    cent = torch.randn(8, H * D, generator=g) * 1.5
    x = cent[torch.randint(0, 8, (S,), generator=g)] + 0.35 * torch.randn(S, H * D, generator=g)
    q, k, v = (dev((x @ torch.randn(H * D, H * D, generator=g) / (H * D) ** 0.5)[None].to(torch.bfloat16)).unflatten(2, (H, D)).transpose(1, 2)
               for _ in range(3))
    q, k = q.contiguous(), k.contiguous()
    outs = []
    real_loop = nat.kmeans_loop
    for copies in (False, True):
        if copies:   # what the path did before: contiguous copies of the video tokens
            monkeypatch.setattr(nat, "kmeans_loop", lambda xx, *a, **kw: real_loop(xx.contiguous(), *a, **kw))
        store = _core.CentroidStore()
        torch.manual_seed(0)
        outs.append(_core.svg2_sparse_attention(q, k, v, geo, store, 0, 12, 24, 0.9, 0.1, 3, 2, prompt_length=L))
    assert torch.equal(outs[0], outs[1]) and torch.isfinite(outs[0].float()).all()
