"""Production-size parity of the SVG2 path (BASELINE.json configs[2] and, with fp8=True, configs[4]): Wan 2.1 720p (S = 75600,
QC = 300, KC = 1000, `scripts/wan/wan_t2v_720p_sap.sh:14-19`) and HunyuanVideo 720p (S = 119056, QC = 400, KC = 1000 plus the
prompt / unused-prompt pseudo clusters of `svg/models/hyvideo/attention.py:657-702`).

Everything upstream of the attention comes from the HIP path itself — labels and the stable token permutation from the flash-kmeans
kernels, the block map from `svg_identify_dynamic_map` — and `svg_varblock_attention` runs with the permutation fused in
(q_row_idx / kv_row_idx), i.e. exactly what the SAP processors call.  A dense CPU oracle of the whole problem would be tens of
TFLOP, so the output is checked on spot rows — >= 256 per head: first / last row of the largest and of the smallest q-cluster, the
neighbours of EMPTY clusters (forced: three initial centroids sit far outside the data, and an empty cluster
keeps its centroid, `svg/kmeans_utils.py:416-421`), rows of the block-rows with the longest and the shortest run lists, the text
rows (Hunyuan), random rows — against `O.masked_attention` under the element mask rebuilt from labels + map, the semantics of
`dynamic_block_sparse_fwd_flashinfer` (`svg/kmeans_utils.py:1319-1392`: q rows of block-row i attend the kv rows of the active
block-cols).  Plus: fused == permute -> attention -> inverse permute bit for bit at that size.
"""
import pytest
import torch

from oracle import svg_oracle as O

pytestmark = pytest.mark.gpu

CASES = {
    # name: (heads used, D, F, P, ctx, prompt, QC, KC)
    "wan720p": (4, 128, 21, 3600, 0, 0, 300, 1000),
    "hy720p": (3, 128, 33, 3600, 256, 64, 400, 1000),
}
N_ROWS = 256


def clustered(H, N, D, modes, gen, spread=0.35):
    """the bench_svg2.py data: a per-head mixture of `modes` Gaussians (iid data gives density ~1)"""
    centers = torch.randn(H, modes, D, device="cuda", generator=gen) * 1.5
    lab = torch.randint(0, modes, (H, N), device="cuda", generator=gen)
    x = torch.gather(centers, 1, lab[..., None].expand(-1, -1, D)) + spread * torch.randn(H, N, D, device="cuda", generator=gen)
    return x.to(torch.bfloat16)


def build_case(name):
    """-> dict with q, k, v [H, S, D] (original order), ext labels [H, S] (pseudo clusters for the text rows), map [H, QB, KB],
    sizes, sorted indices — all produced by the HIP path the processors use (svg.models._core)."""
    from svg import _native as nat
    from svg.kmeans_utils import identify_dynamic_map
    from svg.models import _core

    nat.load()
    H, D, F_, P_, ctx, L, QC, KC = CASES[name]
    V = F_ * P_
    S = V + ctx
    gen = torch.Generator(device="cuda").manual_seed(11)
    q = clustered(H, S, D, 64, gen)
    k = clustered(H, S, D, 64, gen)
    v = torch.randn(H, S, D, device="cuda", dtype=torch.bfloat16, generator=gen)
    qv, kv = q[None, :, :V].contiguous(), k[None, :, :V].contiguous()
    # warm start from the first QC / KC video rows (deterministic init) with three centroids pushed out to three
    # times their data point: every point of the mixture (|x| ~ 17, modes ~ 24 apart) is nearer to any other centroid than to a
    # point at |x| ~ 51, so these clusters are, and stay, EMPTY
    store = _core.CentroidStore()
    qi, ki = qv[0, :, :QC].clone(), kv[0, :, :KC].clone()
    for c_, n_ in ((qi, QC), (ki, KC)):
        c_[:, [5, n_ // 2, n_ - 1]] *= 3
    store.q[0], store.k[0] = qi, ki
    (ql, qc, qs, _, qidx), (kl, kc, ks, _, kidx) = _core.kmeans_clustering(store, 0, qv, kv, QC, KC, 50, 2)
    q_sizes, k_sizes = qs.view(1, H, QC), ks.view(1, H, KC)
    assert int((q_sizes == 0).sum()) >= 3 * H and int((k_sizes == 0).sum()) >= 3 * H, "the forced empty clusters are gone"
    dmap = identify_dynamic_map(qc.view(1, H, QC, D), kc.view(1, H, KC, D), q_sizes, k_sizes, 0.9, 0.1)
    ql, kl = ql.view(H, V), kl.view(H, V)
    if ctx:
        dmap, q_sizes, k_sizes, qidx, kidx = _core.dynamic_map_post_processing(dmap, q_sizes, k_sizes, qidx, kidx, V, ctx, L)

        def ext(lab, n):   # prompt rows -> pseudo cluster n, unused prompt rows -> n + 1 (hyvideo/attention.py:681-692)
            tail = torch.cat([torch.full((L,), n), torch.full((ctx - L,), n + 1)]).to(lab)
            return torch.cat([lab, tail.expand(H, -1)], dim=1)

        ql, kl = ext(ql, QC), ext(kl, KC)
    QB, KB = q_sizes.shape[-1], k_sizes.shape[-1]
    return dict(nat=nat, H=H, S=S, D=D, V=V, q=q, k=k, v=v, ql=ql, kl=kl, dmap=dmap.view(H, QB, KB).contiguous(),
                q_sizes=q_sizes.view(H, QB).contiguous(), k_sizes=k_sizes.view(H, KB).contiguous(),
                qidx=qidx.contiguous(), kidx=kidx.contiguous(), QB=QB, KB=KB)


@pytest.fixture(scope="module", params=sorted(CASES))
def case(request):
    c = build_case(request.param)
    c["name"] = request.param
    yield c
    torch.cuda.empty_cache()


def spot_rows(c, h):
    """>= N_ROWS original row indices of head h that cover the structural corner cases of the variable-block walk"""
    qs = c["q_sizes"][h].cpu()
    off = torch.cat([torch.zeros(1, dtype=torch.int64), qs.cumsum(0)])
    qidx = c["qidx"][h].cpu().long()
    keys_per_row = (c["dmap"][h].float() @ c["k_sizes"][h].float()[:, None])[:, 0].cpu()       # active keys of each block-row
    runs_per_row = (c["dmap"][h] & (c["k_sizes"][h] > 0)[None]).sum(1).cpu()                  # entries of its run list
    nonempty = (qs > 0).nonzero()[:, 0]
    pick = [nonempty[qs[nonempty].argmax()], nonempty[qs[nonempty].argmin()],
            nonempty[runs_per_row[nonempty].argmax()], nonempty[runs_per_row[nonempty].argmin()],
            nonempty[keys_per_row[nonempty].argmax()], nonempty[keys_per_row[nonempty].argmin()]]
    for e in (qs == 0).nonzero()[:, 0].tolist():          # block-rows on either side of an empty cluster
        pick += [x for x in (nonempty[nonempty < e][-1:], nonempty[nonempty > e][:1]) if len(x)]
    if c["QB"] > CASES[c["name"]][6]:                      # Hunyuan: the prompt and the unused-prompt pseudo clusters
        pick += [torch.tensor(c["QB"] - 2), torch.tensor(c["QB"] - 1)]
    rows = []
    for i in {int(x) for x in pick}:
        a, b = int(off[i]), int(off[i + 1])
        pos = {a, b - 1, (a + b) // 2, min(a + 255, b - 1), min(a + 256, b - 1), max(b - 33, a)}   # tile / wave boundaries
        rows += qidx[sorted(pos)].tolist()
    rows = list(dict.fromkeys(rows))
    g = torch.Generator().manual_seed(100 + h)
    extra = torch.randperm(c["S"], generator=g)[: max(0, N_ROWS - len(rows)) + 8].tolist()
    rows += [r for r in extra if r not in set(rows)]
    assert len(rows) >= N_ROWS
    return rows, int(runs_per_row.max())


def oracle_rows(c, h, rows):
    qh, kh, vh = (c[n][h].float().cpu() for n in ("q", "k", "v"))
    rl = c["ql"][h].cpu()[rows]
    em = c["dmap"][h].cpu()[rl][:, c["kl"][h].cpu()]     # [rows, S]: key j allowed iff map[label(row), label(j)]
    return O.masked_attention(qh[rows], kh, vh, em)


@pytest.mark.parametrize("fp8", [False, True], ids=["bf16", "fp8"])
def test_svg2_production_spot_rows(case, fp8):
    c, nat = case, case["nat"]
    o = nat.varblock_attention(c["q"], c["k"], c["v"], c["dmap"], c["q_sizes"], c["k_sizes"], q_row_idx=c["qidx"],
                               kv_row_idx=c["kidx"], fp8=fp8)
    torch.cuda.synchronize()
    assert torch.isfinite(o.float()).all()
    worst, longest = 0.0, 0
    for h in range(c["H"]):
        rows, nruns = spot_rows(c, h)
        longest = max(longest, nruns)
        ref = oracle_rows(c, h, rows)
        got = o[h].float().cpu()[rows]
        e = ((got - ref).norm() / ref.norm()).item()
        worst = max(worst, e)
        if fp8:
            # e4m3 QK^T / PV on the clustered bench distribution: 8.8 % against the 16-bit kernel measured on the whole output
            # (BENCH_r02 svg2_wan720p_fp8); there is no reference fp8 implementation (/root/reference/README.md:117)
            assert e < 0.12, (c["name"], h, e)
        else:
            torch.testing.assert_close(got, ref, atol=1e-2, rtol=1e-2)   # the reference's tolerance (test_sparse_attn_dyn_blk_wan.py:133)
            assert e < 3e-3, (c["name"], h, e)
    assert longest > 100, "the production regime has run lists of hundreds of entries"
    print(f"[svg2 {c['name']} {'fp8' if fp8 else 'bf16'}] worst rel L2 over heads {worst:.2e}; longest run list {longest} entries; "
          f"density {float(O.density_calculation(c['dmap'][None].cpu(), c['q_sizes'][None].cpu(), c['k_sizes'][None].cpu()).mean()):.3f}")


@pytest.mark.parametrize("fp8", [False, True], ids=["bf16", "fp8"])
def test_svg2_production_fused_equals_materialised(case, fp8):
    """the fused row gather / scatter == permute_tensor_by_labels -> attention -> apply_inverse_permutation through the same kernel,
    bit for bit (ref pipeline: hyvideo/attention.py:651-653,778-783)"""
    c, nat = case, case["nat"]
    o = nat.varblock_attention(c["q"], c["k"], c["v"], c["dmap"], c["q_sizes"], c["k_sizes"], q_row_idx=c["qidx"],
                               kv_row_idx=c["kidx"], fp8=fp8)
    qp, kp, vp = nat.permute_rows(c["q"], c["qidx"]), nat.permute_rows(c["k"], c["kidx"]), nat.permute_rows(c["v"], c["kidx"])
    op = nat.varblock_attention(qp, kp, vp, c["dmap"], c["q_sizes"], c["k_sizes"], fp8=fp8)
    assert torch.equal(nat.permute_rows(op, c["qidx"], inverse=True), o)
    # every row of a q-cluster without rows does not exist; every existing row was written (o starts as zeros: count exact zeros rows)
    zero_rows = (o.float().abs().sum(-1) == 0).sum().item()
    assert zero_rows == 0, zero_rows


def test_svg2_production_labels_partition(case):
    """index work is bit-exact: the sorted indices are the stable argsort of the labels and the sizes their histogram"""
    c = case
    V = c["V"]
    for h in range(c["H"]):
        lab = c["ql"][h, :V].cpu()
        assert torch.equal(c["qidx"][h, :V].cpu().long(), O.stable_argsort(lab))
        n = CASES[c["name"]][6]
        assert torch.equal(c["q_sizes"][h, :n].cpu().long(), torch.bincount(lab, minlength=n))


def _bf16_ulp(x):
    return torch.exp2(torch.floor(torch.log2(x.abs().clamp(min=2.0 ** -126))) - 7)


def test_kmeans_production_size_vs_oracle():
    """Two Lloyd iterations of the Wan 2.1 720p k-side k-means (N = 75600, K = 1000, D = 128; two heads) from the deterministic init
    (first K rows) against O.kmeans_iter, step by step on the HIP path's own previous state (run(m - 1)'s centroids
    are exactly what iteration m assigns with): labels equal to the oracle's argmin or a rounding-level near-tie (the two distances
    within 1e-2 relative: the bf16 norm rounding of SURVEY hazard 4), counts == bincount, sorted indices == stable argsort, centroids
    within ONE bf16 ulp of the fp32-sum update of those labels (the order of the fp32 sums is the only freedom), empty clusters keep
    their centroid.  The walking update kernel (round 5) had no oracle at this size.
    ref: svg/kmeans_utils.py:629-643 (_euclid_iter), :375-421 (sorted centroid update), :684-733 (loop)."""
    from svg import _native as nat
    from svg.kmeans_utils import batch_kmeans_Euclid

    nat.load()
    B, N, K, D = 2, 75600, 1000, 128
    gen = torch.Generator(device="cuda").manual_seed(31)
    xd = clustered(B, N, D, 64, gen)
    c0d = xd[:, :K].clone()
    c0d[:, [7, K - 1]] *= 3          # two centroids outside the data: empty clusters
    x, prev = xd.cpu(), c0d.cpu()
    xsq = O.kmeans_xsq(x)
    total_mism = 0
    for m in (1, 2):
        lab, cent, cnt, nit, sidx = batch_kmeans_Euclid(xd, K, max_iters=m, init_centroids=c0d, return_sorted_indices=True, check_every=0)
        assert int(nit) == m
        lab, cent, cnt, sidx = lab.cpu(), cent.cpu(), cnt.cpu(), sidx.cpu()
        for b in range(B):      # one head at a time: [N, K] fp32 distances are 302 MB
            dist = O.kmeans_distances(x[b:b + 1], xsq[b:b + 1], prev[b:b + 1])[0]
            ref_lab = dist.argmin(-1)
            mism = lab[b] != ref_lab
            d_got = dist.gather(1, lab[b][:, None])[:, 0]
            d_ref = dist.min(-1).values
            assert mism.float().mean() < 5e-3, mism.float().mean()
            assert torch.all((d_got - d_ref)[mism] <= 1e-2 * d_ref[mism].clamp(min=1.0))
            total_mism += int(mism.sum())
        c_ref, cnt_ref = O.kmeans_update(x, lab, prev)
        assert torch.equal(cnt, cnt_ref)
        assert torch.equal(sidx.long(), O.stable_argsort(lab))
        empty = cnt_ref == 0
        assert int(empty.sum()) >= 2 * B and torch.equal(cent[empty], prev[empty])
        # the oracle's update rounds the fp32 mean once; a different fp32 summation order may cross one rounding boundary
        sums = torch.zeros(B, K, D).scatter_add_(1, lab[..., None].expand(-1, -1, D), x.float())
        mean = torch.where(empty[..., None], prev.float(), sums / cnt_ref.float().clamp(min=1)[..., None])
        err = (cent.float() - mean).abs()
        assert torch.all(err <= _bf16_ulp(mean) * 0.5 * 1.02 + 1e-6), (err / _bf16_ulp(mean)).max()   # correctly rounded up to the sums' order
        assert torch.all((cent.float() - c_ref.float()).abs() <= _bf16_ulp(c_ref.float())), "more than one bf16 ulp from the oracle's centroids"
        assert (cent != c_ref).float().mean() < 2e-2
        prev = cent
    print(f"[kmeans Wan 720p k-side, 2 heads x 2 iterations] near-tie labels that differ from the oracle's argmin: {total_mism} of {2 * B * N}")
