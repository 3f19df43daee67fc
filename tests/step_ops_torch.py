"""A plain-torch statement of the op table `bench_step.run_step` runs on (bench_step.HipOps is the product: libsvgattn kernels) —
TEST INFRASTRUCTURE: lets the token / head sharding logic of the N-rank denoise step run on CPU tensors under gloo and be compared
with the single-process step.  Built on the oracle (norms, RoPE, masked attention)."""
import torch

from oracle import svg_oracle as O


class TorchOps:
    def __init__(self, geo, mul=1.6):
        self.geo = geo
        S = geo.S
        self.sparse_mask = O.hy_mask(S, geo.ctx, geo.L, geo.F, geo.P, mul)
        real = geo.V + geo.L
        idx = torch.arange(S)
        self.dense_mask = (idx[:, None] < real) == (idx[None, :] < real)     # two segments [0, real) and [real, S)

    def ln_mod(self, x, scale, shift):
        return O.modulate_shift(O.fp32_layernorm(x, eps=1e-6), scale[None], shift[None], x.dtype)

    def gate_res(self, res, x, gate):
        return O.modulate_gate_residual(res, x, gate[None], res.dtype)

    def prologue(self, st, q_buf, k_buf, v_buf, pos0, n_rot):
        g = self.geo
        q, k, v = (b.unflatten(2, (g.heads, -1)).transpose(1, 2).contiguous() for b in (q_buf, k_buf, v_buf))
        q, k = O.rms_norm(q, st.qn, 1e-6), O.rms_norm(k, st.kn, 1e-6)
        if n_rot == 0:      # a rank whose token range holds text tokens only (the 8-rank test geometry; never at production sizes)
            return q, k, v
        cos, sin = st.cos[pos0:pos0 + n_rot], st.sin[pos0:pos0 + n_rot]
        q = torch.cat([O.rope_cossin(q[:, :, :n_rot], cos, sin), q[:, :, n_rot:]], dim=2)
        k = torch.cat([O.rope_cossin(k[:, :, :n_rot], cos, sin), k[:, :, n_rot:]], dim=2)
        return q, k, v

    def attention(self, q, k, v, sparse):
        return O.masked_attention(q, k, v, self.sparse_mask if sparse else self.dense_mask).to(q.dtype)

    def gelu(self, x):
        return torch.nn.functional.gelu(x, approximate="tanh")
