#!/usr/bin/env python3
"""Randomised cross-check of oracle/svg_oracle.py against the reference's OWN torch functions, imported from /root/reference (build
container only: the reference does not travel to the GPU box).  The committed fixtures (tests/golden/make_golden.py) pin each function
on one or two geometries; this draws many random ones — frame counts, ragged frame sizes, text lengths, band multipliers, cluster
counts with empty clusters, top-p / min_kc_ratio — and demands equality:

  mask_mod predicates (hy / wan / cog) on the full index grid      bit-exact
  get_attention_mask profiling masks (hy / wan / cog)              bit-exact
  sparsity_to_width (hy / wan / cog)                               exact (same float expression)
  ref_*_sparse_head_placement / ref_*_hidden_states_placement      bit-exact
  weighted_softmax                                                 1e-6
  identify_dynamic_map (fp32 and bf16 centroids)                   bit-exact off exact ties of two probabilities (torch.sort is not stable: the
                                                                   reference's own result is undefined there; such rows are counted and reported)
  density_calculation                                              exact
  dynamic_block_sparse_fwd_torch (ragged + empty clusters)         1e-5
  sample_mse of the Hunyuan / Wan / Cog processors                 1e-6 (fp32 inputs; NaN positions equal for Cog)
  Wan BSR op: get_factor / ref_gen_temporal_mask                   bit-exact against the PRODUCT's host-side generator (svg.kernels.ops)
  Hunyuan BSR op: _gen_temporal_mask / _gen_spatial_mask           bit-exact against the PRODUCT's (row pointer, padded column indices, block size)
  the PRODUCT's torch-level placement helpers (ref_* functions, in-place token reorders) and sparsity_to_width             bit-exact
  the PRODUCT's mask descriptors (host code of svg.models.*.utils): profile_desc expanded as the device reads it == get_attention_mask;
      generate_temporal_head_mask_mod (svg_band_mask_t) expanded == the reference's mask_mod on the full grid             bit-exact

    python tools/fuzz_oracle_vs_reference.py [--trials 40] > profiles/<round>_fuzz_oracle_vs_reference.txt"""
import argparse
import sys
import types
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests" / "golden"))
import make_golden as MG  # noqa: E402

from oracle import svg_oracle as O  # noqa: E402


def grid_mask(mask_mod, S):
    q = torch.arange(S)[:, None].expand(S, S)
    k = torch.arange(S)[None, :].expand(S, S)
    return mask_mod(0, 0, q, k)


def expand_profile_variant(pv, vid0, F_, P_, S, rows=None):
    """svg_profile_variant_t as the device reads it (csrc/profiler.hip `ProfilePolicy::allowed`, documented in include/svg_attn.h) -> the
    float mask it stands for: rows `rows` (default all) x all S keys"""
    V = F_ * P_
    idx = torch.arange(S)

    def coord(t):
        if pv.coord != 1:
            return t
        v = t - vid0
        inside = (v >= 0) & (v < V)
        tm = vid0 + (v % P_) * F_ + v // P_
        return torch.where(inside, tm, t)

    q = idx if rows is None else torch.as_tensor(rows)
    x, y = (coord(q) - pv.origin)[:, None], (coord(idx) - pv.origin)[None, :]
    qtext = ((q >= pv.text_lo) & (q < pv.text_hi))[:, None]
    ktext = ((idx >= pv.text_lo) & (idx < pv.text_hi))[None, :]
    dom = (x >= 0) & (x < pv.span) & (y >= 0) & (y < pv.span)
    db = torch.div(x, 128, rounding_mode="floor") - torch.div(y, 128, rounding_mode="floor")
    band = (db < pv.band_blocks) & (-db < pv.band_blocks)
    sink = y < pv.sink_cols
    return (qtext | ktext | (dom & (band | sink))).float()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=40)
    ap.add_argument("--seed", type=int, default=2026)
    args = ap.parse_args()

    MG.install_stubs()
    MG._stub("diffusers.models.normalization", RMSNorm=type("RMSNorm", (), {}))
    if "matplotlib" not in sys.modules:
        try:
            import matplotlib  # noqa: F401
        except ImportError:
            MG._stub("matplotlib", pyplot=types.SimpleNamespace())
            MG._stub("matplotlib.pyplot")
    sys.path.insert(0, MG.REF)
    torch.Tensor.cuda = lambda self, *a, **k: self
    import svg.kernels.ops.attention_ops as ref_ops_hy
    import svg.kernels.ops.attention_ops_wan as ref_ops_wan
    import svg.kmeans_utils as KU
    import svg.models.cog.attention as cog_attn
    import svg.models.cog.placement as cog_pl
    import svg.models.cog.utils as cog_u
    import svg.models.hyvideo.attention as hy_attn
    import svg.models.hyvideo.placement as hy_pl
    import svg.models.hyvideo.utils as hy_u
    import svg.models.wan.attention as wan_attn
    import svg.models.wan.utils as wan_u

    # The product's package is called `svg` like the reference's, which is imported above: load it under another top-level name (its
    # modules import each other relatively).  Host code only — mask generators and descriptors; nothing here touches the GPU.
    import importlib
    import importlib.util

    pkg_dir = ROOT / "sparse-videogen_amd" / "svg"
    spec = importlib.util.spec_from_file_location("svg_amd", pkg_dir / "__init__.py", submodule_search_locations=[str(pkg_dir)])
    svg_amd = importlib.util.module_from_spec(spec)
    sys.modules["svg_amd"] = svg_amd
    spec.loader.exec_module(svg_amd)
    own_ops_wan = importlib.import_module("svg_amd.kernels.ops.attention_ops_wan")
    own_ops_hy = importlib.import_module("svg_amd.kernels.ops.attention_ops")
    own_hy_pl = importlib.import_module("svg_amd.models.hyvideo.placement")
    own_cog_pl = importlib.import_module("svg_amd.models.cog.placement")
    own_cog_u = importlib.import_module("svg_amd.models.cog.utils")
    own_hy_u = importlib.import_module("svg_amd.models.hyvideo.utils")
    own_wan_u = importlib.import_module("svg_amd.models.wan.utils")
    for m in (own_ops_wan, own_ops_hy, own_cog_u, own_hy_u, own_wan_u):
        assert str(pkg_dir) in m.__file__ and "/root/reference" not in m.__file__, m.__file__

    gen = torch.Generator().manual_seed(args.seed)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=gen))

    def rf(lo, hi):
        return lo + (hi - lo) * float(torch.rand(1, generator=gen))

    counts = {}

    def ok(name, cond, detail=""):
        c = counts.setdefault(name, [0, 0])
        c[0] += 1
        if not cond:
            c[1] += 1
            print(f"MISMATCH {name}: {detail}")

    ties = 0
    for trial in range(args.trials):
        F_, P_, ctx = ri(2, 6), ri(20, 170), ri(1, 40)
        L, mul = ri(1, ctx), rf(0.3, 3.0)
        V = F_ * P_
        S = V + ctx
        # ---- mask_mod predicates ----
        ok("mask_mod hy", torch.equal(grid_mask(hy_u.generate_temporal_head_mask_mod(ctx, L, F_, P_, mul=mul), S), O.hy_mask(S, ctx, L, F_, P_, mul)), (F_, P_, ctx, L, mul))
        ok("mask_mod wan", torch.equal(grid_mask(wan_u.generate_temporal_head_mask_mod(0, 0, F_, P_, mul=mul), V), O.wan_mask(V, F_, P_, mul)), (F_, P_, mul))
        for sink in (False, True):
            ok("mask_mod cog", torch.equal(grid_mask(cog_u.generate_temporal_head_mask_mod(ctx, F_, P_, mul=mul, attn_sink=sink), S),
                                           O.cog_mask(S, ctx, F_, P_, mul, attn_sink=sink)), (F_, P_, ctx, mul, sink))
        # ---- the band descriptor of the C ABI states the same predicate ----
        ok("band desc hy", torch.equal(O.band_mask(S, **O.hy_band_params(S, ctx, L, F_, P_, mul)), O.hy_mask(S, ctx, L, F_, P_, mul)))
        ok("band desc wan", torch.equal(O.band_mask(V, **O.wan_band_params(V, F_, P_, mul)), O.wan_mask(V, F_, P_, mul)))
        ok("band desc cog", torch.equal(O.band_mask(S, **O.cog_band_params(S, ctx, F_, P_, mul)), O.cog_mask(S, ctx, F_, P_, mul)))
        # ---- profiling masks ----
        for which, idx in (("spatial", 0), ("temporal", 1)):
            ok("profile mask hy", torch.equal(hy_u.get_attention_mask(which, V, ctx, F_, P_, device="cpu").float(), O.profile_masks("hy", ctx, F_, P_)[idx][:V]), (which, F_, P_, ctx))      # (the first sample_mse_max_row rows)
            ok("profile mask wan", torch.equal(wan_u.get_attention_mask(which, V, 0, F_, P_).float(), O.profile_masks("wan", 0, F_, P_)[idx][:V]), (which, F_, P_))
            ok("profile mask cog", torch.equal(cog_u.get_attention_mask(which, ctx, F_, P_).float(), O.profile_masks("cog", ctx, F_, P_)[idx]), (which, F_, P_, ctx))
        # ---- the PRODUCT's analytic profile descriptors (host code: svg.models.*.utils.profile_desc), expanded as the device reads them ----
        for model, mod_own, ref_masks, c_len in (("hy", own_hy_u, [hy_u.get_attention_mask(w, V, ctx, F_, P_, device="cpu").float() for w in ("spatial", "temporal")], ctx),
                                                 ("wan", own_wan_u, [wan_u.get_attention_mask(w, V, 0, F_, P_).float() for w in ("spatial", "temporal")], 0),
                                                 ("cog", own_cog_u, [cog_u.get_attention_mask(w, ctx, F_, P_).float() for w in ("spatial", "temporal")], ctx)):
            d = mod_own.profile_desc(c_len, F_, P_)
            for i in range(2):
                mine = expand_profile_variant(d.variant[i], d.vid0, F_, P_, V + c_len)
                ok(f"product profile descriptor {model}", torch.equal(mine[:ref_masks[i].shape[0]], ref_masks[i]), (model, i, F_, P_, c_len))
        # ---- the PRODUCT's band-mask descriptors (generate_temporal_head_mask_mod -> svg_band_mask_t) against the reference's mask_mod ----
        for name, desc, ref_mm, Sx in (("hy", own_hy_u.generate_temporal_head_mask_mod(ctx, L, F_, P_, mul=mul), hy_u.generate_temporal_head_mask_mod(ctx, L, F_, P_, mul=mul), S),
                                       ("wan", own_wan_u.generate_temporal_head_mask_mod(0, 0, F_, P_, mul=mul), wan_u.generate_temporal_head_mask_mod(0, 0, F_, P_, mul=mul), V),
                                       ("cog", own_cog_u.generate_temporal_head_mask_mod(ctx, F_, P_, mul=mul), cog_u.generate_temporal_head_mask_mod(ctx, F_, P_, mul=mul), S)):
            fields = {n: int(getattr(desc, n)) for n in ("real_len", "band", "colfull_lo", "colfull_hi", "rowfull_lo", "rowfull_hi")}
            ok(f"product band descriptor {name}", torch.equal(O.band_mask(Sx, **fields), grid_mask(ref_mm, Sx)), (name, F_, P_, ctx, L, mul))
        # ---- sparsity -> width ----
        sp = rf(0.1, 0.6)
        for mod in (hy_u, wan_u, cog_u):
            try:
                want = mod.sparsity_to_width(sp, ctx, F_, P_)
            except ValueError:      # sqrt of a negative number: the sparsity does not leave room for the text rows / columns
                continue
            ok("sparsity_to_width", want == O.sparsity_to_width(sp, ctx, F_, P_), (sp, ctx, F_, P_))
        # ---- placement ----
        cfg, H, D = ri(1, 2), ri(1, 3), 8 * ri(1, 4)
        x = [torch.randn(cfg, H, S, D, generator=gen).to(torch.bfloat16) for _ in range(3)]
        best = torch.randint(0, 2, (cfg, H), generator=gen)
        for name, fwd, inv, tf in (("hy", hy_pl.ref_hunyuan_sparse_head_placement, hy_pl.ref_hunyuan_hidden_states_placement, False),
                                   ("cog", cog_pl.ref_sparse_head_placement, cog_pl.ref_hidden_states_placement, True)):
            got = fwd(x[0], x[1], x[2], best, ctx, F_, P_)
            ok(f"placement {name} fwd", all(torch.equal(a, O.head_placement(b, best, ctx, F_, P_, text_first=tf)) for a, b in zip(got, x)))
            out = torch.zeros_like(x[0])
            inv(x[0], out, best, ctx, F_, P_)
            ok(f"placement {name} inv", torch.equal(out, O.head_placement(x[0], best, ctx, F_, P_, text_first=tf, inverse=True)))
        # ---- the PRODUCT's torch-level placement helpers (CPU tensors: the `ref_*` functions and the in-place token reorders) ----
        for name, r_mod, o_mod, fwd, inv, tm, fm in (
                ("hy", hy_pl, own_hy_pl, "ref_hunyuan_sparse_head_placement", "ref_hunyuan_hidden_states_placement", "hunyuan_token_reorder_to_token_major", "hunyuan_token_reorder_to_frame_major"),
                ("cog", cog_pl, own_cog_pl, "ref_sparse_head_placement", "ref_hidden_states_placement", "token_reorder_to_token_major", "token_reorder_to_frame_major")):
            a, b = getattr(r_mod, fwd)(x[0], x[1], x[2], best, ctx, F_, P_), getattr(o_mod, fwd)(x[0], x[1], x[2], best, ctx, F_, P_)
            ok(f"product placement {name} fwd", all(torch.equal(u, w) for u, w in zip(a, b)))
            oa, ob = torch.zeros_like(x[0]), torch.zeros_like(x[0])
            getattr(r_mod, inv)(x[0], oa, best, ctx, F_, P_)
            getattr(o_mod, inv)(x[0], ob, best, ctx, F_, P_)
            ok(f"product placement {name} inv", torch.equal(oa, ob))
            for fn in (tm, fm):
                ta, tb = x[1].clone(), x[1].clone()
                ra, rb = getattr(r_mod, fn)(ta, ctx, V, F_, P_), getattr(o_mod, fn)(tb, ctx, V, F_, P_)
                ok(f"product token reorder {name}", torch.equal(ra, rb) and torch.equal(ta, tb))
        for mod_r, mod_o in ((hy_u, own_hy_u), (wan_u, own_wan_u), (cog_u, own_cog_u)):
            try:
                want = mod_r.sparsity_to_width(sp, ctx, F_, P_)
            except ValueError:
                continue
            ok("product sparsity_to_width", want == mod_o.sparsity_to_width(sp, ctx, F_, P_))
        # ---- weighted softmax, dynamic map, density, variable-block attention ----
        B, Hh, QC, KC, Dd = 1, ri(1, 3), ri(2, 12), ri(2, 16), 16 * ri(1, 4)
        qc, kc = torch.randn(B, Hh, QC, Dd, generator=gen) * 2, torch.randn(B, Hh, KC, Dd, generator=gen) * 2
        N = ri(40, 200)

        def sizes(n):
            cut = torch.sort(torch.randint(0, N + 1, (B, Hh, n - 1), generator=gen), dim=-1)[0]
            edges = torch.cat([torch.zeros(B, Hh, 1, dtype=torch.long), cut, torch.full((B, Hh, 1), N)], -1)
            return (edges[..., 1:] - edges[..., :-1]).to(torch.int32)       # sums to N, zeros allowed (empty clusters)

        qsz, ksz = sizes(QC), sizes(KC)
        scores = torch.randn(B, Hh, QC, KC, generator=gen) * 3
        ok("weighted_softmax", torch.allclose(KU.weighted_softmax(scores, ksz.unsqueeze(-2).float()), O.weighted_softmax(scores, ksz.unsqueeze(-2).float()), atol=1e-6, rtol=1e-6))
        p_, r_ = rf(0.3, 0.99), (0.0 if ri(0, 1) else rf(0.0, 0.5))
        for dt in (torch.float32, torch.bfloat16):
            want = KU.identify_dynamic_map(qc.to(dt), kc.to(dt), qsz, ksz, p_, r_)
            got = O.identify_dynamic_map(qc.to(dt), kc.to(dt), qsz, ksz, p_, r_)
            if not torch.equal(want, got):
                # rows with two exactly equal probabilities: torch.sort (unstable) may order them either way
                sc = torch.matmul(qc.to(dt), kc.to(dt).transpose(-2, -1)) / (Dd ** 0.5)
                pr = KU.weighted_softmax(sc, ksz.unsqueeze(-2).float())
                srt = torch.sort(pr.float(), dim=-1)[0]
                tie_rows = (srt[..., 1:] == srt[..., :-1]).any(-1)
                bad_rows = (want != got).any(-1)
                if bool((bad_rows & ~tie_rows).any()):
                    ok("identify_dynamic_map", False, (dt, QC, KC, p_, r_))
                else:
                    ties += int(bad_rows.sum())
                    ok("identify_dynamic_map", True)
            else:
                ok("identify_dynamic_map", True)
        dmap = KU.identify_dynamic_map(qc, kc, qsz, ksz, p_, r_)
        ok("density_calculation", torch.equal(KU.density_calculation(dmap, qsz, ksz), O.density_calculation(dmap, qsz, ksz)))
        q3, k3, v3 = (torch.randn(B, Hh, N, Dd, generator=gen) for _ in range(3))
        dm2 = dmap.clone()
        dm2[..., 0] |= ~dm2.any(-1)                       # (rows without any block: the reference divides 0 / 0 there)
        first_nonempty = (ksz > 0).float().argmax(-1)     # make sure every q block sees a key block WITH rows
        dm2.scatter_(-1, first_nonempty[:, :, None, None].expand(B, Hh, QC, 1), True)
        want = KU.dynamic_block_sparse_fwd_torch(q3, k3, v3, dm2, qsz, ksz)
        got = O.dynamic_block_sparse_fwd(q3, k3, v3, dm2, qsz, ksz)
        ok("dynamic_block_sparse_fwd_torch", torch.allclose(want.float(), got, atol=1e-5, rtol=1e-5), float((want.float() - got).abs().max()))
        # ---- online profiler of the three processors (fp32 inputs) ----
        Hs, Ds = ri(1, 3), 16 * ri(1, 4)
        n_rows = 8
        for name, cls, ctx_m, mod, model in (("hy", hy_attn.Hunyuan_SVGAttn_Processor2_0, ctx, hy_u, "hy"), ("wan", wan_attn.WanAttn_SVGAttn_Processor2_0, 0, wan_u, "wan"),
                                             ("cog", cog_attn.CogVideoX_SparseAttn_Processor2_0, ctx, cog_u, "cog")):
            Sm = V + ctx_m
            q, k, v = (torch.randn(1, Hs, Sm, Ds, generator=gen) for _ in range(3))
            if model == "cog":
                cls.attention_masks = [mod.get_attention_mask("spatial", ctx_m, F_, P_), mod.get_attention_mask("temporal", ctx_m, F_, P_)]
            elif model == "hy":
                cls.attention_masks = [mod.get_attention_mask("spatial", V, ctx_m, F_, P_, device="cpu"), mod.get_attention_mask("temporal", V, ctx_m, F_, P_, device="cpu")]
            else:
                cls.attention_masks = [mod.get_attention_mask("spatial", V, 0, F_, P_), mod.get_attention_mask("temporal", V, 0, F_, P_)]
            cls.num_sampled_rows = n_rows
            if model != "cog":
                cls.sample_mse_max_row = V
            seed = ri(0, 10 ** 6)
            torch.manual_seed(seed)
            want = cls(0).sample_mse(q, k, v)
            torch.manual_seed(seed)
            rows = torch.randint(low=0, high=Sm if model == "cog" else V, size=(n_rows,))
            got = O.sample_mse(q, k, v, rows, O.profile_masks(model, ctx_m, F_, P_))
            same_nan = torch.equal(torch.isnan(want), torch.isnan(got))
            ok(f"sample_mse {name}", same_nan and torch.allclose(torch.nan_to_num(want), torch.nan_to_num(got), atol=1e-6, rtol=1e-5),
               (seed, float((torch.nan_to_num(want) - torch.nan_to_num(got)).abs().max())))
        # ---- Wan uniform-block op: the product's host-side generator against the reference's ----
        Fw, Pw, mw = ri(2, 8), ri(30, 600), rf(0.2, 2.5)
        # ---- Hunyuan uniform-block op (svg/kernels/ops/attention_ops.py): BSR row pointer, column indices (with the reference's padding) and block size ----
        Fh, Ph, mh = ri(1, 9), 10 * ri(1, 40), rf(0.1, 3.0)
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):                       # (the reference prints the whole mask)
            ref_t, ref_s = ref_ops_hy._gen_temporal_mask(Fh, Ph, mh), ref_ops_hy._gen_spatial_mask(Fh, Ph, int(mh))
        own_t, own_s = own_ops_hy._gen_temporal_mask(Fh, Ph, mh), own_ops_hy._gen_spatial_mask(Fh, Ph, int(mh))
        for nm, a, b in (("temporal", ref_t, own_t), ("spatial", ref_s, own_s)):
            ok(f"hunyuan bsr _gen_{nm}_mask", torch.equal(a[0].cpu(), b[0].cpu()) and torch.equal(a[1].cpu(), b[1].cpu()) and tuple(a[2]) == tuple(b[2]) and a[0].dtype == b[0].dtype and a[1].dtype == b[1].dtype,
               (Fh, Ph, mh))
        ok("wan bsr get_factor", ref_ops_wan.get_factor(Fw, Pw) == own_ops_wan.get_factor(Fw, Pw), (Fw, Pw))
        ok("wan bsr ref_gen_temporal_mask", torch.equal(torch.as_tensor(ref_ops_wan.ref_gen_temporal_mask(Fw, Pw, mw)), torch.as_tensor(own_ops_wan.ref_gen_temporal_mask(Fw, Pw, mw))), (Fw, Pw, mw))

    print(f"# fuzz of oracle/svg_oracle.py (and of the product's host-side mask descriptors and BSR generator, loaded under the name svg_amd) against the reference's own torch functions: {args.trials} random geometries, seed {args.seed}")
    print("| function | comparisons | mismatches |\n|---|---|---|")
    total_bad = 0
    for name, (n, bad) in counts.items():
        print(f"| {name} | {n} | {bad} |")
        total_bad += bad
    print(f"\nidentify_dynamic_map: {ties} rows differed only where two probabilities are exactly equal (torch.sort is unstable: undefined in the reference too)")
    print("RESULT:", "all equal" if total_bad == 0 else f"{total_bad} MISMATCHES")
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
