"""GPU parity of the transformer-block glue kernels (svg_layernorm_forward, svg_modulate_shift_forward,
svg_modulate_gate_residual_forward, svg_layernorm_modulate_forward) through the C ABI, against the torch fall-back expressions
of the reference's Wan block (svg/models/wan/custom_models.py:44-108) restated in oracle/svg_oracle.py."""
from pathlib import Path

import pytest
import torch

from oracle import svg_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "sparse-videogen_amd"))
    from svg import _native
    _native.load()
    return _native


SHAPES = [(1, 37, 512), (2, 129, 1536), (1, 333, 3072), (2, 77, 5120), (1, 50, 8192), (1, 19, 40)]


@pytest.mark.parametrize("B,S,N", SHAPES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("affine", [True, False])
def test_layernorm_fp32(nat, B, S, N, dtype, affine):
    torch.manual_seed(N + S)
    x = (torch.randn(B, S, N) * 2 + 0.5).to(dtype)
    w, b = (torch.randn(N), torch.randn(N)) if affine else (None, None)
    y = nat.layernorm_forward(x.cuda(), w.cuda() if affine else None, b.cuda() if affine else None, 1e-6, torch.float32)
    ref = O.fp32_layernorm(x, w, b, 1e-6)
    assert y.dtype == torch.float32                     # the reference's kernel writes fp32 (layernorm.py:70)
    torch.testing.assert_close(y.cpu(), ref, atol=2e-5, rtol=2e-5)   # reduction order differs, nothing else


@pytest.mark.parametrize("B,S,N", SHAPES)
def test_modulate_and_gate_residual_bit_exact(nat, B, S, N):
    torch.manual_seed(N)
    xn = torch.randn(B, S, N)                                  # fp32 LayerNorm output
    scale, shift, gate = torch.randn(B, 1, N) * 0.3, torch.randn(B, 1, N) * 0.3, torch.randn(B, 1, N) * 0.3
    y = nat.modulate_shift_forward(xn.cuda(), scale.cuda(), shift.cuda(), torch.bfloat16)
    assert torch.equal(y.cpu(), O.modulate_shift(xn, scale, shift, torch.bfloat16))
    hidden, attn = torch.randn(B, S, N).to(torch.bfloat16), torch.randn(B, S, N).to(torch.bfloat16)
    z = nat.modulate_gate_residual_forward(hidden.cuda(), attn.cuda(), gate.cuda(), torch.bfloat16)
    assert torch.equal(z.cpu(), O.modulate_gate_residual(hidden, attn, gate, torch.bfloat16))
    z32 = nat.modulate_gate_residual_forward(hidden.cuda(), attn.float().cuda(), gate.cuda(), torch.float32)
    assert torch.equal(z32.cpu(), O.modulate_gate_residual(hidden, attn.float(), gate, torch.float32))


@pytest.mark.parametrize("B,S,N", SHAPES)
@pytest.mark.parametrize("affine,mod", [(False, True), (True, False), (True, True)])
def test_fused_layernorm_modulate(nat, B, S, N, affine, mod):
    """one pass == layernorm (fp32) followed by modulate, up to the reduction order of the statistics"""
    torch.manual_seed(S)
    x = torch.randn(B, S, N).to(torch.bfloat16)
    w, b = (torch.randn(N), torch.randn(N)) if affine else (None, None)
    scale, shift = (torch.randn(B, 1, N) * 0.3, torch.randn(B, 1, N) * 0.3) if mod else (None, None)
    c = lambda t: None if t is None else t.cuda()  # noqa: E731
    y = nat.layernorm_modulate_forward(c(x), c(w), c(b), c(scale), c(shift), 1e-6)
    ln = O.fp32_layernorm(x, w, b, 1e-6)
    ref = O.modulate_shift(ln, scale, shift, torch.bfloat16) if mod else ln.to(torch.bfloat16)
    assert y.dtype == torch.bfloat16
    ne = y.cpu() != ref
    assert ne.float().mean().item() < 2e-3
    torch.testing.assert_close(y.cpu().float(), ref.float(), atol=4e-2, rtol=1.6e-2)
    # and equal to the two separate kernels bit for bit
    two = nat.layernorm_forward(c(x), c(w), c(b), 1e-6, torch.float32)
    two = nat.modulate_shift_forward(two, c(scale), c(shift), torch.bfloat16) if mod else two.to(torch.bfloat16)
    assert torch.equal(y, two)


def test_reference_module_names(nat):
    from svg.kernels.triton.layernorm import triton_layernorm_forward
    from svg.kernels.triton.modulate import triton_modulate_gate_residual_forward, triton_modulate_shift_forward
    torch.manual_seed(0)
    x = torch.randn(1, 100, 5120, dtype=torch.bfloat16, device="cuda")
    n = triton_layernorm_forward(x, None, None, 1e-6, elementwise_affine=False)
    sc, sh, g = (torch.randn(1, 1, 5120, device="cuda") * 0.2 for _ in range(3))
    m = triton_modulate_shift_forward(n, sc, sh, output_dtype=torch.bfloat16)
    r = triton_modulate_gate_residual_forward(x, m, g, output_dtype=torch.bfloat16)
    ref_n = O.fp32_layernorm(x.cpu(), None, None, 1e-6)
    ref_m = O.modulate_shift(ref_n, sc.cpu(), sh.cpu(), torch.bfloat16)
    torch.testing.assert_close(m.cpu().float(), ref_m.float(), atol=4e-2, rtol=1.6e-2)
    assert torch.equal(r.cpu(), O.modulate_gate_residual(x.cpu(), m.cpu(), g.cpu(), torch.bfloat16))


def test_wan_block_forward_matches_reference_fallback(nat):
    """wan_block_forward (HIP glue) == the torch fall-back branch of the reference's block forward (custom_models.py:44-108)"""
    import torch.nn as nn
    from svg.models.wan.custom_models import install_block_forward

    torch.manual_seed(2)
    C, S, B = 512, 300, 2

    class Attn(nn.Module):
        def __init__(self):
            super().__init__()
            self.p = nn.Linear(C, C)

        def forward(self, hidden_states, **kw):
            return self.p(hidden_states)

    class FP32LN(nn.LayerNorm):
        def forward(self, x):
            return nn.functional.layer_norm(x.float(), self.normalized_shape, None if self.weight is None else self.weight.float(),
                                            None if self.bias is None else self.bias.float(), self.eps).to(x.dtype)

    class Blk(nn.Module):
        def __init__(self):
            super().__init__()
            self.scale_shift_table = nn.Parameter(torch.randn(1, 6, C) / C ** 0.5)
            self.norm1, self.norm3 = FP32LN(C, eps=1e-6, elementwise_affine=False), FP32LN(C, eps=1e-6, elementwise_affine=False)
            self.norm2 = FP32LN(C, eps=1e-6, elementwise_affine=True)
            self.attn1, self.attn2, self.ffn = Attn(), Attn(), Attn()

    blk = Blk()
    with torch.no_grad():
        blk.norm2.weight.normal_(); blk.norm2.bias.normal_()
    tr = nn.Module(); tr.blocks = nn.ModuleList([blk])
    assert install_block_forward(tr) == 1
    blk = blk.cuda()
    for m in (blk.attn1, blk.attn2, blk.ffn):
        m.to(torch.bfloat16)
    h = torch.randn(B, S, C).to(torch.bfloat16).cuda()
    enc = torch.randn(B, 7, C).to(torch.bfloat16).cuda()
    temb = torch.randn(B, 6, C).cuda() * 0.3
    with torch.no_grad():
        got = blk(h, enc, temb, None)
        # the reference's torch branch
        sh, sc, g, csh, csc, cg = (blk.scale_shift_table + temb.float()).chunk(6, dim=1)
        x = h
        n = (blk.norm1(x.float()) * (1 + sc) + sh).type_as(x)
        x = (x.float() + blk.attn1(hidden_states=n) * g).type_as(x)
        n = blk.norm2(x.float()).type_as(x)
        x = x + blk.attn2(hidden_states=n)
        n = (blk.norm3(x.float()) * (1 + csc) + csh).type_as(x)
        ref = (x.float() + blk.ffn(n).float() * cg).type_as(x)
    torch.testing.assert_close(got.float(), ref.float(), atol=6e-2, rtol=3e-2)
    assert (got != ref).float().mean().item() < 0.05


def test_full_size_wan_block_glue(nat):
    """Wan 2.1 720p hidden states [1, 75600, 5120]: fused == separate, spot rows against the oracle."""
    torch.manual_seed(1)
    x = torch.randn(1, 75600, 5120, device="cuda", dtype=torch.bfloat16)
    sc, sh = torch.randn(1, 1, 5120, device="cuda") * 0.2, torch.randn(1, 1, 5120, device="cuda") * 0.2
    y = nat.layernorm_modulate_forward(x, None, None, sc, sh, 1e-6)
    two = nat.modulate_shift_forward(nat.layernorm_forward(x, None, None, 1e-6), sc, sh, torch.bfloat16)
    assert torch.equal(y, two)
    rows = torch.tensor([0, 1, 4097, 75599])
    ref = O.modulate_shift(O.fp32_layernorm(x[:, rows].cpu(), None, None, 1e-6), sc.cpu(), sh.cpu(), torch.bfloat16)
    assert (y[:, rows].cpu() != ref).float().mean().item() < 2e-3


@pytest.mark.parametrize("sfx", ["h", "f"])
def test_layernorm_reference_padding_equals_the_references_triton_kernels(nat, sfx):
    """reference_padding=True against fixtures PRODUCED BY the reference's own Triton kernels (executed with Triton's interpreter,
    tests/golden/make_golden_triton.py; N = 1536 -> N2 = 2048): the HIP kernel reproduces layernorm.py:13-62 / :113-156 as they are,
    padding quirk included; reference_padding=False stays FP32LayerNorm — the two differ by what tests/test_triton_golden.py shows."""
    import numpy as np

    g = np.load(Path(__file__).resolve().parent / "golden" / "triton_golden.npz")
    x = torch.from_numpy(g[f"gl_{sfx}_x"]).cuda()
    w, b = torch.from_numpy(g[f"gl_{sfx}_w"]).cuda(), torch.from_numpy(g[f"gl_{sfx}_b"]).cuda()
    ln_p, ln_n = torch.from_numpy(g[f"gl_{sfx}_ln_p"]), torch.from_numpy(g[f"gl_{sfx}_ln_n"])
    for (ww, bb), ref in (((w, b), ln_p), ((None, None), ln_n)):
        quirk = nat.layernorm_forward(x, ww, bb, 1e-6, torch.float32, reference_padding=True).cpu()
        spec = nat.layernorm_forward(x, ww, bb, 1e-6, torch.float32).cpu()
        torch.testing.assert_close(quirk, ref, atol=2e-5, rtol=2e-5)
        torch.testing.assert_close(spec, O.fp32_layernorm(x.cpu(), None if ww is None else ww.cpu(), None if bb is None else bb.cpu(), 1e-6),
                                   atol=2e-5, rtol=2e-5)
        assert (spec - ref).abs().max().item() > 1e-3
    # the fused form honours the switch too, and the module-level switch of the reference-named shim reaches it
    sc, sh = torch.from_numpy(g[f"gl_{sfx}_scale"]).cuda(), torch.from_numpy(g[f"gl_{sfx}_shift"]).cuda()
    ms = torch.from_numpy(g[f"gl_{sfx}_ms"])
    fused = nat.layernorm_modulate_forward(x, None, None, sc, sh, 1e-6, x.dtype, reference_padding=True).cpu()
    if x.dtype == torch.float32:
        torch.testing.assert_close(fused, ms, atol=5e-5, rtol=5e-5)          # reduction order of the statistics
    else:
        ulp = torch.finfo(x.dtype).eps * ms.float().abs().clamp_min(1e-2)
        assert ((fused.float() - ms.float()).abs() <= 2 * ulp).all()
    import svg.kernels.triton.layernorm as LN

    try:
        LN.REFERENCE_PADDING = True
        torch.testing.assert_close(LN.triton_layernorm_forward(x, w, b, 1e-6, True).cpu(), ln_p, atol=2e-5, rtol=2e-5)
    finally:
        LN.REFERENCE_PADDING = False
    # a power-of-two row: the switch changes nothing, bit for bit
    x2 = torch.randn(3, 7, 2048, device="cuda").to(x.dtype) + 0.4
    assert torch.equal(nat.layernorm_forward(x2, None, None, 1e-6, torch.float32, reference_padding=True),
                       nat.layernorm_forward(x2, None, None, 1e-6, torch.float32))
