"""tools/native_harness on the CPU box: `--dry` prints the geometry and the svg_band_mask_t a run would use without touching the GPU.  The
harness is the suite's second, torch-free checker (tests/test_gpu_native_harness.py); these tests pin ITS mask constants and its pair count
to the product's mask builders (which the golden tests pin to the reference) and to the oracle's dense mask."""
import json
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import svg_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def harness():
    import __graft_entry__ as entry

    entry._load_build_module().build(force=False, asm=False, verbose=False)   # --dry still dlopens the library (ABI check)
    return entry.build_native_harness()


def dry(harness, geom):
    r = subprocess.run([str(harness), "--geom", geom, "--dry"], capture_output=True, text=True, timeout=60, cwd=str(ROOT))
    assert r.returncode == 0, (r.stdout, r.stderr)
    return json.loads(r.stdout.strip().splitlines()[-1])


def product_mask(kind, ctx, L, F, P, sparsity):
    from svg.models.cog import utils as cog
    from svg.models.hyvideo import utils as hy
    from svg.models.hyvideo.utils import sparsity_to_width
    from svg.models.wan import utils as wan

    w = sparsity_to_width(sparsity, ctx, F, P)
    if kind == "hy":
        return hy.generate_temporal_head_mask_mod(ctx, L, F, P, mul=w)
    if kind == "wan":
        return wan.generate_temporal_head_mask_mod(ctx, ctx, F, P, mul=w)
    return cog.generate_temporal_head_mask_mod(ctx, F, P, mul=w)


def interval_pairs(m, S):
    """#allowed pairs of a BandMask in its interval form (tools/svg1_models.py)"""
    q = np.arange(S, dtype=np.int64)
    real = m.real_len
    rq = q < real
    rowf = (q >= m.rowfull_lo) & (q < m.rowfull_hi)
    lo = np.where(rq, np.where(rowf, 0, np.maximum(q - m.band + 1, 0)), real)
    hi = np.where(rq, np.where(rowf, real, np.minimum(q + m.band, real)), S)
    alen = np.maximum(hi - lo, 0)
    ch = min(m.colfull_hi, real)
    b0, b1 = m.colfull_lo, max(ch, m.colfull_lo)
    inter = np.maximum(np.minimum(hi, b1) - np.maximum(lo, b0), 0)
    blen = np.where(rq & ~rowf, (b1 - b0) - inter, 0)
    return int((alen + blen).sum())


# the production geometries of the harness: (harness name, model kind, context length, prompt length, F, P, sparsity of the model's script)
@pytest.mark.parametrize("geom,kind,ctx,L,F,P,sparsity", [
    ("hy720p", "hy", 256, 64, 33, 3600, 0.25), ("hy480p", "hy", 256, 64, 33, 1350, 0.25), ("wan720p", "wan", 0, 0, 21, 3600, 0.30),
    ("cog15", "cog", 226, 226, 11, 4080, 0.25), ("cog480p", "cog", 226, 226, 13, 1350, 0.25)])
def test_harness_mask_is_the_products(harness, geom, kind, ctx, L, F, P, sparsity):
    d = dry(harness, geom)
    m = product_mask(kind, ctx, L, F, P, sparsity)
    assert d["S"] == F * P + ctx and d["vid0"] == (ctx if kind == "cog" else 0) and (d["F"], d["P"]) == (F, P)
    assert d["mask"] == [m.real_len, m.band, m.colfull_lo, m.colfull_hi, m.rowfull_lo, m.rowfull_hi]
    assert d["pairs"] == interval_pairs(m, d["S"])


@pytest.mark.parametrize("geom", ["small", "small64"])
def test_harness_pair_count_is_the_dense_masks(harness, geom):
    """the toy geometries against the oracle's dense [S, S] statement of the predicate (oracle.band_mask, pinned to the reference's
    mask_mods by tests/test_oracle_golden.py)"""
    d = dry(harness, geom)
    real, band, cl, ch, rl, rh = d["mask"]
    dense = O.band_mask(d["S"], real, band, cl, ch, rl, rh)
    assert d["pairs"] == int(dense.sum())
    assert 0 < d["pairs"] < d["S"] ** 2 and torch.is_tensor(dense)
