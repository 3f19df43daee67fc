"""Host-side helpers added in round 4 (no GPU): the bounded workspace cache of svg._native, the shared registry behind the four
`custom_models.py` shims, the token shards of the N-rank step, and the Wan QK-norm type gate."""
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))


def test_workspace_cache_is_bounded_and_lru():
    from svg._native import WorkspaceCache

    c = WorkspaceCache(capacity=3)
    for i in range(3):
        c[("k", i)] = i
    assert c.get(("k", 0)) == 0          # touch 0: 1 is now the least recently used
    c[("k", 3)] = 3
    assert len(c) == 3 and c.get(("k", 1)) is None and c.get(("k", 0)) == 0 and c.get(("k", 3)) == 3
    c.clear()
    assert len(c) == 0


def test_transformer_registry_per_model_and_idempotent():
    from svg.models.cog import custom_models as cog
    from svg.models.context import TransformerRegistry
    from svg.models.hyvideo import custom_models as hy
    from svg.models.wan import custom_models as wan

    assert hy._REGISTRY is not cog._REGISTRY is not wan._REGISTRY        # one registry per model module

    class T(torch.nn.Module):
        def forward(self, x, timestep=None):
            from svg.models.context import current_timestep

            return current_timestep()

    calls = []
    reg = TransformerRegistry(also=lambda t: calls.append(t))
    t = T()
    reg.register_transformer(t)
    reg.register_transformer(t)                                          # registering twice keeps one entry
    reg.replace_sparse_forward()
    reg.replace_sparse_forward()                                         # the hook is installed once, `also` runs per call
    assert calls == [t, t]
    ts = torch.tensor([7.0])
    assert t.forward(torch.zeros(1), timestep=ts) is ts                  # the wrapper publishes the timestep during the call
    from svg.models.context import current_timestep

    assert current_timestep() is None


def test_wan_qk_norm_accepts_rms_norm_only():
    """ref: svg/models/wan/attention.py:105-120 raises ValueError for anything but torch.nn.RMSNorm / diffusers' RMSNorm"""
    from svg.models.wan.attention import WanAttn_SVGAttn_Processor2_0 as P

    class Attn:
        pass

    a = Attn()
    a.norm_q = torch.nn.RMSNorm(16, eps=1e-6)
    a.norm_k = torch.nn.RMSNorm(16, eps=1e-6)
    x = torch.randn(1, 5, 16)
    q, k = P(0).get_qk_norm(a, x, x)
    torch.testing.assert_close(q, a.norm_q(x))
    a.norm_q = torch.nn.LayerNorm(16)                                    # has a bias: never an RMSNorm
    with pytest.raises(ValueError):
        P(0).get_qk_norm(a, x, x)

    # a class that is merely CALLED RMSNorm (a third-party module with other semantics) is not diffusers' RMSNorm: rejected too
    class RMSNorm(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.weight, self.eps = torch.nn.Parameter(torch.ones(16)), 1e-6

        def forward(self, t):
            return t * (1 + self.weight)

    a.norm_q = RMSNorm()
    with pytest.raises(ValueError):
        P(0).get_qk_norm(a, x, x)
    # ... unless it lives in diffusers' modules (recognised by name: diffusers is not a dependency here) or declares the semantics
    RMSNorm.svg_rmsnorm_compatible = True
    q, _ = P(0).get_qk_norm(a, x, x)
    torch.testing.assert_close(q, a.norm_q(x))                           # (CPU tensors: the module's own forward)
    Fake = type("RMSNorm", (torch.nn.Module,), {"__module__": "diffusers.models.normalization", "forward": lambda self, t: t * 2.0})
    a.norm_q = Fake()
    a.norm_q.weight, a.norm_q.eps = torch.nn.Parameter(torch.ones(16)), 1e-6
    q, _ = P(0).get_qk_norm(a, x, x)
    torch.testing.assert_close(q, x * 2.0)


def test_token_range_units():
    from svg.distributed import token_range

    S = 119056
    tr = [token_range(S, r, 8, unit=128) for r in range(8)]
    assert tr[0] == (0, 14976) and tr[-1][1] == S and all(tr[i][1] == tr[i + 1][0] for i in range(7))
    assert max(b - a for a, b in tr) * 8 / S < 1.01
    assert token_range(10, 0, 1) == (0, 10) and token_range(10, 2, 3, unit=4) == (8, 10)   # a trailing partial unit goes to the last rank


def test_tuned_gemm_file_is_a_tunableop_results_file():
    """sparse-videogen_amd/tuning/tunableop_mi355x.csv (bench_step.enable_tuned_gemms): TunableOp's validators first — it rejects the file on any
    other library stack —, then one GemmTunableOp line per listed shape naming a library solution (no kernel of this repo)."""
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    lines = (root / "sparse-videogen_amd" / "tuning" / "tunableop_mi355x.csv").read_text().strip().splitlines()
    val = {l.split(",")[1] for l in lines if l.startswith("Validator,")}
    assert {"PT_VERSION", "HIP_VERSION", "HIPBLASLT_VERSION", "ROCBLAS_VERSION", "GCN_ARCH_NAME"} <= val
    ops = [l.split(",") for l in lines if not l.startswith("Validator,")]
    assert len(ops) == 4 and all(o[0] == "GemmTunableOp_BFloat16_TN" and o[2].startswith(("Gemm_Hipblaslt_", "Gemm_Rocblas_")) for o in ops)
    assert {o[1].split("_")[1] for o in ops} == {"3072"}          # the four 3072-wide shapes of the HunyuanVideo stack
    sys_path_added = str(root) not in __import__("sys").path
    if sys_path_added:
        __import__("sys").path.insert(0, str(root))
    import bench_step

    assert bench_step.TUNED_GEMMS.exists() and callable(bench_step.enable_tuned_gemms)
