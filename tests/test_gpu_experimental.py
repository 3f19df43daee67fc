"""GPU tests of EXPERIMENTAL entry points — code that compiles and has not been validated on hardware yet.  They run only with
SVG_EXPERIMENTAL=1 (the first thing round 4 does); the default `-m gpu` run skips them, so an unvalidated kernel cannot turn the
parity suite red."""
import os
from pathlib import Path

import pytest
import torch

from oracle import svg_oracle as O

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("SVG_EXPERIMENTAL") != "1", reason="set SVG_EXPERIMENTAL=1")]


@pytest.fixture(scope="module")
def nat():
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "sparse-videogen_amd"))
    from svg import _native
    _native.load()
    return _native


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Hq,Hkv", [(2, 2), (4, 2)])
def test_varblock_mixed_precision_body(nat, dt, Hq, Hkv):
    """svg_varblock_attention_fp8pv (16-bit QK^T, e4m3 PV) against the fp32 oracle under the element mask: ragged and EMPTY clusters,
    GQA, partial tiles.  Expected error: the e4m3 rounding of P and V only (~2.5 - 3.7 % rel. L2, tools/fp8_precision_study.py) —
    well below the all-e4m3 kernel's on the same inputs."""
    gen = torch.Generator().manual_seed(7)
    D = 128
    q_sizes = torch.tensor([[300, 1, 0, 129, 70, 524]] * Hkv, dtype=torch.int32)
    k_sizes = torch.tensor([[64, 0, 200, 333, 1, 426]] * Hkv, dtype=torch.int32)
    S = int(q_sizes[0].sum())
    assert S == int(k_sizes[0].sum())
    q = torch.randn(Hq, S, D, generator=gen).to(dt)
    k, v = torch.randn(Hkv, S, D, generator=gen).to(dt), torch.randn(Hkv, S, D, generator=gen).to(dt)
    bmap = torch.rand(Hkv, 6, 6, generator=gen) < 0.6
    bmap[:, torch.arange(6), torch.tensor([5, 3, 2, 0, 3, 5])] = True       # every q block sees a key block with rows
    args = (q.cuda(), k.cuda(), v.cuda(), bmap.cuda(), q_sizes.cuda(), k_sizes.cuda())
    o_pv = nat.varblock_attention(*args, fp8="pv").float().cpu()
    o_f8 = nat.varblock_attention(*args, fp8=True).float().cpu()
    o_16 = nat.varblock_attention(*args).float().cpu()
    g = Hq // Hkv
    ref = torch.stack([O.masked_attention(q[h].float(), k[h // g].float(), v[h // g].float(),
                                          O.block_mask_to_element_mask(bmap[h // g], q_sizes[h // g], k_sizes[h // g])) for h in range(Hq)])
    rows = torch.repeat_interleave(torch.arange(6), q_sizes[0].long())       # rows of empty q clusters do not exist
    err = lambda o: ((o - ref).norm() / ref.norm()).item()      # noqa: E731
    e_pv, e_f8, e_16 = err(o_pv), err(o_f8), err(o_16)
    print(f"[mixed body {dt} Hq={Hq} Hkv={Hkv}] rel L2 vs fp32 oracle: 16-bit {e_16:.2e}, mixed {e_pv:.2e}, all-e4m3 {e_f8:.2e}")
    assert torch.isfinite(o_pv).all() and rows.numel() == S
    assert e_16 < 5e-3 and e_pv < 4.5e-2 and e_pv < e_f8
