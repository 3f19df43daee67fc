"""Pre-attention prologue (QK norm + RoPE), CPU side: the oracle restatement against golden vectors produced by the
reference's own torch reference functions (tests/golden/make_golden_prologue.py), and the C ABI exports."""
import numpy as np
import pytest
import torch

from oracle import svg_oracle as O

GOLD = np.load(str(__import__("pathlib").Path(__file__).parent / "golden" / "prologue_golden.npz"))


def t16(name, dtype=torch.bfloat16):
    return torch.from_numpy(GOLD[name].copy()).view(dtype)


NORM_TAGS = ["7x32", "31x64", "95x128", "128x256"]
ROPE_TAGS = ["1_2_151_64_15", "2_1_151_128_35", "1_1_151_256_77"]


@pytest.mark.parametrize("tag", NORM_TAGS)
def test_oracle_rms_norm_equals_reference_replica(tag):
    x, w = t16(f"norm_x_{tag}"), t16(f"norm_w_{tag}")
    got = O.rms_norm(x, w)
    assert torch.equal(got, t16(f"rms_replica_{tag}")), "bit-exact vs replica_host_rms_norm (test_rms_norm.py:31-36)"
    # and within the reference's own tolerance of torch.nn.functional.rms_norm (test_rms_norm.py:13-18)
    torch.testing.assert_close(got.float(), t16(f"rms_ref_{tag}").float(), rtol=3e-2, atol=2e-2)


@pytest.mark.parametrize("tag", NORM_TAGS)
def test_oracle_layer_norm_equals_reference(tag):
    x, w, b = t16(f"norm_x_{tag}"), t16(f"norm_w_{tag}"), t16(f"norm_b_{tag}")
    got, ref = O.layer_norm(x, w, b), t16(f"ln_ref_{tag}")
    # F.layer_norm accumulates its statistics in a different order: equal up to one bf16 ulp on a few elements
    assert (got != ref).float().mean().item() < 5e-3
    torch.testing.assert_close(got.float(), ref.float(), rtol=8e-3, atol=8e-3)


@pytest.mark.parametrize("tag", ROPE_TAGS)
def test_oracle_rope_equals_reference(tag):
    L = int(tag.split("_")[-1])
    q, cos, sin = t16(f"rope_q_{tag}"), torch.from_numpy(GOLD[f"rope_cos_{tag}"]), torch.from_numpy(GOLD[f"rope_sin_{tag}"])
    first, _ = O.apply_qk_rope(q, q, cos, sin, L, "cossin")
    assert torch.equal(first[:, :, L:], t16(f"rope_first_{tag}")) and torch.equal(first[:, :, :L], q[:, :, :L])
    last, _ = O.apply_qk_rope(q, q, cos, sin, L, "txtlast")
    assert torch.equal(last[:, :, :-L], t16(f"rope_last_{tag}")) and torch.equal(last[:, :, -L:], q[:, :, -L:])
    for dt, key in ((torch.float16, "cplx_out"), (torch.bfloat16, "cplx_out_bf16")):
        qh = t16(f"cplx_q_{tag}", torch.float16).to(dt)
        fr, fi = torch.from_numpy(GOLD[f"cplx_fr_{tag}"]), torch.from_numpy(GOLD[f"cplx_fi_{tag}"])
        got, _ = O.apply_qk_rope(qh, qh, fr, fi, L, "complex")
        assert torch.equal(got[:, :, L:], t16(f"{key}_{tag}", dt)) and torch.equal(got[:, :, :L], qh[:, :, :L])


def test_kernels_module_mirrors_reference_names():
    import importlib
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "sparse-videogen_amd"))
    mod = importlib.import_module("svg.kernels.build._kernels")
    for name in ("rms_norm_forward", "layer_norm_forward", "apply_qk_rope_inplace_cossin", "apply_qk_rope_inplace_cossin_txtlast",
                 "apply_qk_rope_inplace_cossin_complex"):   # PYBIND11 module of the reference: svg/kernels/csrc/ops.cu
        assert callable(getattr(mod, name))
    with pytest.raises(RuntimeError, match="GPU"):
        mod.rms_norm_forward(torch.zeros(4, 64, dtype=torch.bfloat16), torch.ones(64, dtype=torch.bfloat16), 1e-5)
