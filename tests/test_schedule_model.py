"""CPU checks of the *scheduling logic* the HIP kernels use (which KV tiles a workgroup visits, how a tile is
classified, the analytic profiler predicates).  The device code is mirrored line by line in Python here and checked
exhaustively against the oracle's dense masks on small geometries — this catches logic errors without a GPU.
Mirrors: BandPolicy::init / classify / allowed (csrc/attention.hip), ProfilePolicy::allowed (csrc/profiler.hip)."""
import math

import pytest
import torch

from oracle import svg_oracle as O

BN = 64
BIG = 1 << 28


def band_schedule(q0, q_end, S, real, band, cf_lo, cf_hi, rf_lo, rf_hi):
    alo = ahi = blo = bhi = clo = chi = BIG
    if q0 < real:
        qr1 = min(q_end, real)
        if q0 < rf_hi and qr1 > rf_lo:
            alo, ahi = 0, (real + BN - 1) // BN
        else:
            alo = max(0, q0 - band + 1) // BN
            ahi = (min(real, qr1 - 1 + band) + BN - 1) // BN
            ch = min(cf_hi, real)
            if ch > cf_lo:
                blo, bhi = cf_lo // BN, (ch + BN - 1) // BN
    if q_end > real:
        clo, chi = real // BN, (S + BN - 1) // BN
    if blo < alo:
        alo, ahi, blo, bhi = blo, bhi, alo, ahi
    if clo < blo:
        blo, bhi, clo, chi = clo, chi, blo, bhi
    if blo < alo:
        alo, ahi, blo, bhi = blo, bhi, alo, ahi
    if blo < BIG and blo <= ahi:
        ahi = max(ahi, bhi)
        blo, bhi, clo, chi = clo, chi, BIG, BIG
        if blo < BIG and blo <= ahi:
            ahi = max(ahi, bhi)
            blo, bhi = BIG, BIG
    elif clo < BIG and clo <= bhi:
        bhi = max(bhi, chi)
        clo, chi = BIG, BIG
    tiles = []
    for lo, hi in ((alo, ahi), (blo, bhi), (clo, chi)):
        tiles += list(range(lo, hi)) if lo < BIG else []
    return tiles


def classify(w0, q_end, k0, S, real, band, cf_lo, cf_hi, rf_lo, rf_hi, WR=32):
    # fast path (BandPolicy::init / classify): per-wave FULL range of first keys; WR = rows per wave (32, or 64 in attn_body_w4)
    w1f = min(w0 + WR, q_end)
    if w0 < q_end and w1f <= real:
        if max(w1f - band, 0) <= k0 <= min(w0 + band - BN, min(real, S) - BN):
            return 1
    if w0 >= q_end:
        return 0
    w1 = min(w0 + WR, q_end)
    k1 = min(k0 + BN, S)
    all_ = False
    if k0 + BN <= S:
        if w1 <= real and k1 <= real:
            all_ = ((k1 - 1 - w0 < band) and (w1 - 1 - k0 < band)) or (k0 >= cf_lo and k1 <= cf_hi) or \
                   (w0 >= rf_lo and w1 <= rf_hi)
        elif w0 >= real and k0 >= real:
            all_ = True
    if all_:
        return 1
    any_ = False
    if w0 < real and k0 < real:
        w1r, k1r = min(w1, real), min(k1, real)
        any_ = ((k0 - (w1r - 1) < band) and (w0 - (k1r - 1) < band)) or (k0 < cf_hi and k1r > cf_lo) or \
               (w0 < rf_hi and w1r > rf_lo)
    if w1 > real and k1 > real:
        any_ = True
    return 2 if any_ else 0


CASES = []
for (F_, P_, ctx, L, mul) in [(4, 140, 16, 9, 1.9), (5, 150, 40, 11, 2.3), (3, 100, 300, 120, 1.3), (2, 70, 10, 10, 4.0)]:
    V = F_ * P_
    CASES += [
        ("hy", V + ctx, O.hy_band_params(V + ctx, ctx, L, F_, P_, mul)),
        ("wan", V, O.wan_band_params(V, F_, P_, mul)),
        ("cog", V + ctx, O.cog_band_params(V + ctx, ctx, F_, P_, mul)),
        ("cog_sink", V + ctx, O.cog_band_params(V + ctx, ctx, F_, P_, mul, True)),
        ("dense", V + ctx, O.dense_band_params(V + ctx)),
        ("dense2", V + ctx, O.dense_band_params(V + ctx, V + L)),
    ]


@pytest.mark.parametrize("BM,WR", [(128, 32), (256, 32), (256, 64)])
@pytest.mark.parametrize("name,S,prm", CASES)
def test_band_schedule_and_classify(name, S, prm, BM, WR):
    mask = O.band_mask(S, **prm)
    p = (prm["real_len"], prm["band"], prm["colfull_lo"], prm["colfull_hi"], prm["rowfull_lo"], prm["rowfull_hi"])
    for q0 in range(0, S, BM):
        q_end = min(S, q0 + BM)
        tiles = band_schedule(q0, q_end, S, *p)
        assert tiles == sorted(set(tiles)), "tile ranges must be disjoint and ascending"
        visited = torch.zeros(S, dtype=torch.bool)
        for t in tiles:
            visited[t * BN: (t + 1) * BN] = True
        needed = mask[q0:q_end].any(dim=0)
        assert not (needed & ~visited).any(), f"{name}: q-tile {q0} misses allowed keys"
        for w0 in range(q0, q0 + BM, WR):
            for t in tiles:
                k0 = t * BN
                cls = classify(w0, q_end, k0, S, *p, WR=WR)
                if w0 >= q_end:
                    assert cls == 0
                    continue
                sub = mask[w0:min(w0 + WR, q_end), k0:min(k0 + BN, S)]
                if cls == 1:
                    assert sub.all() and sub.shape[1] == BN, f"{name}: FULL tile has a masked element"
                elif cls == 0:
                    assert not sub.any(), f"{name}: SKIP tile has an allowed element"


# ---- profiler predicate (ProfilePolicy::allowed) ----
def prof_allowed(q, k, S, vid0, F_, P_, var):
    coord, origin, span, bb, sink, tlo, thi = var
    V = F_ * P_

    def co(i):
        if coord == 1 and 0 <= i - vid0 < V:
            r = i - vid0
            f, pp = divmod(r, P_)
            return vid0 + pp * F_ + f
        return i

    tq = tlo <= q < thi
    tk = tlo <= k < thi
    x, y = co(q) - origin, co(k) - origin
    dom = 0 <= x < span and 0 <= y < span
    db = (x >> 7) - (y >> 7)
    band = db < bb and -db < bb
    return tq or tk or (dom and (band or y < sink))


def prof_variants(model, ctx, F_, P_):
    V = F_ * P_
    if model == "hy":
        bb = int((P_ * 1.5) // 128)
        return 0, ((0, 0, V, bb, 0, V, V + ctx), (1, 0, V, bb, 0, V, V + ctx))
    if model == "wan":
        bb = int((P_ * 2) // 128)
        return 0, ((0, 0, V, bb, P_, 0, 0), (1, 0, V, bb, P_, 0, 0))
    bb = int((P_ * 1.5) // 128)
    span = min(V + ctx, math.ceil(V / 128) * 128)
    return ctx, ((0, 0, span, bb, 0, 0, ctx), (1, ctx, V, bb, 0, 0, 0))


@pytest.mark.parametrize("model,ctx,F_,P_", [("hy", 16, 4, 140), ("wan", 0, 4, 140), ("cog", 16, 4, 140), ("cog", 26, 3, 200),
                                            ("hy", 40, 5, 100), ("wan", 0, 3, 260)])
def test_profile_predicates_equal_reference_masks(model, ctx, F_, P_):
    S = F_ * P_ + ctx
    vid0, variants = prof_variants(model, ctx, F_, P_)
    masks = O.profile_masks(model, ctx, F_, P_)
    gen = torch.Generator().manual_seed(0)
    rows = torch.randint(0, S, (24,), generator=gen).tolist()
    for var, m in zip(variants, masks):
        for q in rows:
            got = torch.tensor([prof_allowed(q, k, S, vid0, F_, P_, var) for k in range(S)])
            assert torch.equal(got, m[q] != 0), f"{model} variant coord={var[0]} row {q}"


# ---- profiler tile classes of the second form (csrc/profiler.hip: ProfilePolicy::classify, allowed_fast, profile16_kernel's row ranking and
#      its skip of fast tiles outside the band of a wave's rows) ----
def _asr7(v):
    return v >> 7            # Python's >> on a negative int is the arithmetic shift the kernel's `>> 7` on int is


def prof_coord(i, vid0, F_, P_, var):
    coord = var[0]
    if coord == 1 and 0 <= i - vid0 < F_ * P_:
        f, pp = divmod(i - vid0, P_)
        return vid0 + pp * F_ + f
    return i


def prof_wave_state(rows, vid0, F_, P_, var):
    """per-wave / per-row state of profile16_kernel for one mask: (xlo_blk, xhi_blk, any_text, [(fa0, falen, fblen)] per row)"""
    coord, origin, span, bb, sink, tlo, thi = var
    blks, lanes, any_text = [], [], False
    for q in rows:
        x = prof_coord(q, vid0, F_, P_, var) - origin
        qtext = tlo <= q < thi
        any_text |= qtext
        blks.append(_asr7(x))
        xdom = 0 <= x < span
        a0, a1 = max((_asr7(x) - bb + 1) * 128, 0), min((_asr7(x) + bb) * 128, span)
        falen = max(a1 - a0, 0) if xdom else 0
        fblen = min(sink, span) if (xdom and sink > 0) else 0
        lanes.append((-(1 << 30), 1 << 32, fblen) if qtext else (a0, falen, fblen))
    return min(blks), max(blks), any_text, lanes


def prof_classify(k0, S, vid0, F_, P_, var, xlo_blk, xhi_blk, any_text):
    """-> (class, ybase, ystride): 0 SKIP, 2 PARTIAL (general predicate), 3 PARTIAL_FAST — ProfilePolicy::classify plus the second form's skip"""
    coord, origin, span, bb, sink, tlo, thi = var
    V, BN = F_ * P_, 64
    k1 = min(k0 + BN, S)
    cls, ybase, ystride = 2, 0, (F_ if coord == 1 else 1)
    if coord == 0:
        text_keys = k0 < thi and k1 > tlo
        if not any_text and not text_keys:
            y0, y1 = k0 - origin, k1 - 1 - origin
            outside = y1 < 0 or y0 >= span
            far = (_asr7(y0) - xhi_blk >= bb) or (xlo_blk - _asr7(y1) >= bb)
            if outside or (far and y0 >= sink and y0 >= 0):
                return 0, 0, ystride
        if not text_keys and k0 + BN <= S and k0 - origin >= 0 and k0 + BN - origin <= span:
            cls, ybase = 3, k0 - origin
    else:
        i0 = max(k0 - vid0, 0)
        f0 = i0 // P_
        p0 = i0 - f0 * P_
        text_keys = k0 < thi and k0 + BN > tlo
        if k0 >= vid0 and k0 + BN <= vid0 + V and k0 + BN <= S and p0 + BN <= P_ and not text_keys:
            cls, ybase = 3, vid0 + p0 * F_ + f0 - origin
    if cls == 3 and not any_text and ybase >= sink:      # profile16_kernel: a fast tile farther than the band from every row of the wave
        b0, b1 = _asr7(ybase), _asr7(ybase + 63 * ystride)
        if b0 - xhi_blk >= bb or xlo_blk - b1 >= bb:
            cls = 0
    return cls, ybase, ystride


@pytest.mark.parametrize("model,ctx,F_,P_", [("hy", 16, 4, 140), ("wan", 0, 4, 140), ("cog", 16, 4, 140), ("cog", 26, 3, 200),
                                            ("hy", 40, 5, 100), ("wan", 0, 3, 260), ("hy", 64, 6, 330), ("wan", 0, 7, 192)])
@pytest.mark.parametrize("R,seed", [(64, 0), (37, 1), (5, 2)])
def test_profile_tile_classes_against_reference_masks(model, ctx, F_, P_, R, seed):
    """The index logic of the online profiler's second form against the reference's materialised masks: rows ranked by their coordinate
    under the second mask and cut into waves of 16; for every wave, mask and 64-key tile a SKIP class means no row of the wave sees a
    key of the tile, a FAST class means the two interval tests give exactly the mask's elements, anything else falls to the general
    predicate (test above)."""
    S = F_ * P_ + ctx
    vid0, variants = prof_variants(model, ctx, F_, P_)
    masks = [m != 0 for m in O.profile_masks(model, ctx, F_, P_)]
    gen = torch.Generator().manual_seed(seed)
    lo, hi = (vid0, S) if model == "cog" else (0, S)
    rows = torch.randint(lo, hi, (R,), generator=gen).tolist()
    order = sorted(range(R), key=lambda i: (prof_coord(rows[i], vid0, F_, P_, variants[1]), i))     # rank by counting, ties by lane
    ranked = [rows[i] for i in order]
    n_skip = n_fast = 0
    for w0 in range(0, R, 16):
        wave_rows = ranked[w0:w0 + 16]
        for var, m in zip(variants, masks):
            xlo, xhi, any_text, lanes = prof_wave_state(wave_rows, vid0, F_, P_, var)
            for k0 in range(0, S, 64):
                cls, ybase, ys = prof_classify(k0, S, vid0, F_, P_, var, xlo, xhi, any_text)
                sub = torch.stack([m[q, k0:min(k0 + 64, S)] for q in wave_rows])
                if cls == 0:
                    n_skip += 1
                    assert not sub.any(), f"{model} coord={var[0]} wave {w0} tile {k0}: SKIP tile has an allowed key"
                elif cls == 3:
                    n_fast += 1
                    assert sub.shape[1] == 64
                    for (fa0, falen, fblen), want in zip(lanes, sub):
                        got = torch.tensor([(0 <= ybase + off * ys - fa0 < falen) or (0 <= ybase + off * ys < fblen) for off in range(64)])
                        assert torch.equal(got, want), f"{model} coord={var[0]} wave {w0} tile {k0}: fast predicate differs from the mask"
                else:
                    for q, want in zip(wave_rows, sub):
                        got = torch.tensor([prof_allowed(q, k, S, vid0, F_, P_, var) for k in range(k0, min(k0 + 64, S))])
                        assert torch.equal(got, want)
    assert n_fast > 0


# ---- variable-block policy: run list and row cursor (csrc/attention.hip VarblockPolicy::init / kv_phys_at) ----
def vb_run_list(map_row, k_off):
    """host model of the LDS run list of one block-row: the active non-empty key blocks in ascending order as
    (end in compact coordinates, permuted start - compact start) pairs, two sentinels behind the last run"""
    runs, total = [], 0
    for j, on in enumerate(map_row):
        ln = k_off[j + 1] - k_off[j]
        if on and ln > 0:
            runs.append((total + ln, k_off[j] - total))
            total += ln
    return runs + [(0x7FFFFFFF, 0)] * 2, total


def vb_walk(runs, total, row, n_tiles, BN=64):
    """the cursor of one staging lane (key row `row` of every tile): current and next run in registers, the list is read only
    when the lane crosses into the next run; keys behind the last one are clamped to it (they are masked in the softmax)"""
    j, r, rn, reads, out = 0, runs[0], runs[1], 2, []
    for t in range(n_tiles):
        pos = min(t * BN + row, total - 1)
        while r[0] <= pos:
            r = rn
            j += 1
            rn = runs[j + 1]
            reads += 1
        out.append(pos + r[1])
    return out, reads


@pytest.mark.parametrize("seed", range(8))
def test_varblock_run_cursor_model(seed):
    """the cursor yields, for every tile and key row, the permuted position a brute-force expansion of the active blocks gives
    (ragged clusters, empty clusters, single-key clusters, the ragged last tile), and never reads behind the sentinels"""
    g = torch.Generator().manual_seed(seed)
    KB = int(torch.randint(1, 40, (1,), generator=g))
    sizes = torch.randint(0, 200, (KB,), generator=g)
    sizes[torch.rand(KB, generator=g) < 0.2] = 0          # empty clusters
    sizes[torch.rand(KB, generator=g) < 0.2] = 1          # single keys: several run crossings inside one tile
    k_off = [0] + torch.cumsum(sizes, 0).tolist()
    map_row = (torch.rand(KB, generator=g) < 0.5).tolist()
    runs, total = vb_run_list(map_row, k_off)
    flat = [p for j in range(KB) if map_row[j] for p in range(k_off[j], k_off[j + 1])]   # brute force: permuted position of every active key
    assert total == len(flat) and len(runs) <= KB + 2
    if total == 0:
        return   # no tiles: the kernel never calls the cursor (nT == 0)
    n_tiles = (total + 63) // 64
    for row in (0, 1, 17, 63):
        got, reads = vb_walk(runs, total, row, n_tiles)
        want = [flat[min(t * 64 + row, total - 1)] for t in range(n_tiles)]
        assert got == want
        assert reads <= len(runs)   # every list entry is read at most once per lane


# ---- remainder packing (round 3): a tile shared by the ragged last tiles of two block-rows (VarblockPolicy::init / classify /
#      allowed / row_intervals, csrc/attention.hip) ----
def vb_packed_run_list(row_a, row_b, k_off):
    """host model of the three-class run list: key blocks both members attend, then only A's, then only B's (each class in
    ascending block order); returns (runs, kC, kCA, total)"""
    runs, total, marks = [], 0, []
    for want in (3, 1, 2):
        for j in range(len(row_a)):
            cls = (1 if row_a[j] else 0) | (2 if row_b[j] else 0)
            ln = k_off[j + 1] - k_off[j]
            if cls == want and ln > 0:
                runs.append((total + ln, k_off[j] - total))
                total += ln
        marks.append(total)
    return runs + [(0x7FFFFFFF, 0)] * 2, marks[0], marks[1], total


def vb_packed_allowed(row, k, ra, kC, kCA, total):
    return k < kCA if row < ra else (k < kC or (kCA <= k < total))


def vb_packed_classify(w0, k0, ra, rb, kC, kCA, total, BN=64):
    """the per-wave FULL / PARTIAL / SKIP classification (0 skip, 1 full, 2 partial)"""
    if w0 >= ra + rb:
        return 0
    w1 = min(w0 + 32, ra + rb)
    only_a, only_b = w1 <= ra, w0 >= ra
    f1_hi = kCA if only_a else kC
    f2_lo, f2_hi = kCA, (total if only_b else kCA)
    if (k0 >= 0 and k0 + BN <= f1_hi) or (k0 >= f2_lo and k0 + BN <= f2_hi):
        return 1
    any1_hi = kCA if only_a else (kC if only_b else total)
    any2_lo = kCA if only_b else total
    return 2 if (k0 < any1_hi or (k0 + BN > any2_lo and k0 < total)) else 0


@pytest.mark.parametrize("seed", range(10))
def test_varblock_packed_tile_model(seed):
    """For random pairs of map rows: every key of the union appears exactly once in the run list; a row of member A may see exactly
    the keys of A's blocks, a row of member B exactly B's (brute force over all rows x keys, through the same cursor walk the
    kernel uses); FULL tiles are fully allowed and SKIP tiles fully masked for every row of the wave, whatever the member mix."""
    g = torch.Generator().manual_seed(100 + seed)
    KB = int(torch.randint(2, 30, (1,), generator=g))
    sizes = torch.randint(0, 150, (KB,), generator=g)
    sizes[torch.rand(KB, generator=g) < 0.15] = 0
    k_off = [0] + torch.cumsum(sizes, 0).tolist()
    row_a = (torch.rand(KB, generator=g) < 0.6).tolist()
    row_b = [(a if torch.rand(1, generator=g).item() < 0.8 else not a) for a in row_a]   # similar key lists, like real partners
    ra, rb = int(torch.randint(1, 200, (1,), generator=g)), int(torch.randint(1, 57, (1,), generator=g))
    runs, kC, kCA, total = vb_packed_run_list(row_a, row_b, k_off)
    owner = {}                                   # permuted key position -> which members attend it
    for j in range(KB):
        for p in range(k_off[j], k_off[j + 1]):
            if row_a[j] or row_b[j]:
                owner[p] = (row_a[j], row_b[j])
    assert total == len(owner) and 0 <= kC <= kCA <= total
    if total == 0:
        return
    n_tiles = (total + 63) // 64
    perm_of = {}                                 # compact position -> permuted position, through the kernel's cursor
    for row in range(64):
        got, _ = vb_walk(runs, total, row, n_tiles)
        for t in range(n_tiles):
            if t * 64 + row < total:
                perm_of[t * 64 + row] = got[t]
    assert sorted(perm_of.values()) == sorted(owner)          # a permutation of the union: nothing twice, nothing missing
    for row in (0, ra - 1, ra, ra + rb - 1):
        if not 0 <= row < ra + rb:
            continue
        for kpos in range(total):
            a_on, b_on = owner[perm_of[kpos]]
            assert vb_packed_allowed(row, kpos, ra, kC, kCA, total) == (a_on if row < ra else b_on), (row, kpos)
    for w0 in range(0, 256, 32):
        rows = [r for r in range(w0, w0 + 32) if r < ra + rb]
        for t in range(n_tiles):
            cls = vb_packed_classify(w0, t * 64, ra, rb, kC, kCA, total)
            cells = [vb_packed_allowed(r, k, ra, kC, kCA, total) and k < total for r in rows for k in range(t * 64, t * 64 + 64)]
            if not rows:
                assert cls == 0
            elif cls == 1:
                assert all(cells) and t * 64 + 64 <= total
            elif cls == 0:
                assert not any(cells)
