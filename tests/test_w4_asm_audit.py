"""Audit of the gfx950 assembly of the one-wave-per-SIMD attention kernels (csrc/attn_w4.h).  Their MFMAs are inline asm and their
accumulators live in hand-assigned AGPRs, so three things hipcc normally guarantees have to be checked on the listing instead
(tools/asm_hazards.py): no VALU write of an MFMA operand inside the MFMA's hazard window (a compiler spill reload in front of an
asm MFMA), no early read of an MFMA result, no transcendental result consumed by an asm instruction without its wait state,
and no compiler-generated access to the owned accumulator registers or to scratch (the kernels must compile without spills).
Runs on the listing build.py keeps for attention_w4.hip; builds it when it is missing or stale (hipcc cross-compiles: no GPU)."""
import importlib.util
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "sparse-videogen_amd"
LISTING = PKG / "build" / "attention_w4-hip-amdgcn-amd-amdhsa-gfx950.s"


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def listing():
    srcs = [PKG / "csrc" / n for n in ("attention_w4.hip", "attn_w4.h", "attn_w4_agpr.inc", "attn_core.h", "band_policy.h")]
    if not LISTING.exists() or LISTING.stat().st_mtime < max(s.stat().st_mtime for s in srcs):
        _load(PKG / "build.py", "svg_build").build(force=False, asm=False, verbose=False)
    assert LISTING.exists(), "build.py keeps the assembly of attention_w4.hip"
    return LISTING


def test_generated_register_helpers_are_current():
    """csrc/attn_w4_agpr.inc is generated (tools/gen_w4_agpr.py) and committed: the committed file must be what the generator writes."""
    inc = PKG / "csrc" / "attn_w4_agpr.inc"
    before = inc.read_text()
    _load(ROOT / "tools" / "gen_w4_agpr.py", "gen_w4_agpr")
    after = inc.read_text()
    inc.write_text(before)
    assert before == after, "re-run tools/gen_w4_agpr.py and commit csrc/attn_w4_agpr.inc"


def test_no_mfma_hazards_no_spills(listing, capsys):
    audit = _load(ROOT / "tools" / "asm_hazards.py", "asm_hazards")
    argv = sys.argv
    sys.argv = ["asm_hazards.py", "w4", str(listing)]
    try:
        rc = audit.main()
    finally:
        sys.argv = argv
    out = capsys.readouterr().out
    kernels = [l for l in out.splitlines() if l.startswith("_Z")]
    assert len(kernels) >= 4, out            # {bf16, f16} x {D 128, 64}  (the switch kernels of this body were retired in round 4)
    assert rc == 0, out
    for l in kernels:
        assert " 0 hazards, 0 spill moves" in l, l


F8_LISTINGS = [PKG / "build" / f"{stem}-hip-amdgcn-amd-amdhsa-gfx950.s" for stem in ("attention", "attention_f8")]


@pytest.fixture(scope="module")
def f8_listings():
    srcs = [PKG / "csrc" / n for n in ("attention.hip", "attention_f8.hip", "attn_f8.h", "attn_core.h", "band_policy.h")]
    newest = max(s.stat().st_mtime for s in srcs)
    if any(not f.exists() or f.stat().st_mtime < newest for f in F8_LISTINGS):
        _load(PKG / "build.py", "svg_build").build(force=False, asm=False, verbose=False)
    return F8_LISTINGS


def _audit(audit, capsys, pattern, listing):
    argv = sys.argv
    sys.argv = ["asm_hazards.py", pattern, str(listing), "--asm-mfma"]
    try:
        rc = audit.main()
    finally:
        sys.argv = argv
    out = capsys.readouterr().out
    return rc, out, [l for l in out.splitlines() if l.startswith("_Z")]


def test_first_step_mfmas_have_no_hazards(f8_listings, capsys):
    """The first contraction step of a tile of the pre-scaled-q two-phase body (attn_core.h, E::mfma_keep_c) and of the fp8 bodies
    (attn_f8.h, mfma_qk_first) is an inline-asm MFMA (D != C: saves a 16-register copy of the loop-invariant C per tile).  Nothing
    may write one of its operands — A, B, C or the block-scale words — within two wait states in front of it, and nothing but the
    next step's MFMA (as C, same tuple) may touch its destination inside the MFMA's latency."""
    audit = _load(ROOT / "tools" / "asm_hazards.py", "asm_hazards")
    for f in f8_listings:
        assert f.exists(), f"build.py keeps the assembly of {f.name}"
    seen = 0
    for f in f8_listings:                                    # {band, varblock} x {bf16, f16}: 2 asm MFMAs each
        rc, out, kernels = _audit(audit, capsys, "attn_f8", f)
        seen += len(kernels)
        assert rc == 0, out
        for l in kernels:
            assert "2 asm MFMAs, 0 hazards" in l, l
    assert seen == 4, seen
    # pre-scaled kernels: band {plain, switch} x dtype x head_dim
    rc, out, kernels = _audit(audit, capsys, "pp2q", f8_listings[0])
    assert rc == 0, out
    assert not any("trace" in l for l in kernels), "the kept listing is the PRODUCT build's (an -DSVG_ABLATIONS build has the trace kernels)"
    assert len(kernels) == 8 and not any("varblock" in l for l in kernels), out
    for l in kernels:
        assert (" 8 asm MFMAs, 0 hazards" if "switch" in l else " 4 asm MFMAs, 0 hazards") in l, l
    # the PRE form of the 16x16x32 body (attn_m16.h, Mfma16::mfma_keep_c): 8 first contraction steps per tile x two copies of the loop;
    # the switch kernels hold the body twice
    rc, out, kernels = _audit(audit, capsys, "band_attn_m16", f8_listings[0])
    assert rc == 0, out
    assert kernels and all("0 hazards" in l for l in kernels), out
    assert any(" 16 asm MFMAs" in l for l in kernels) and any(" 32 asm MFMAs" in l for l in kernels), out
