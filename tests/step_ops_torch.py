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


class WanTorchOps:
    """torch statement of bench_step.WanHipOps (the Wan 2.1 block: RMSNorm across all heads, complex RoPE, SVG2 self attention, cross
    attention over the text tokens).  The sparse self attention is the oracle's SVG2 composition per head — k-means from the head's first
    rows (deterministic: a head's result does not depend on which rank computes it), top-p block map, variable-block attention."""

    def __init__(self, geo):
        self.geo = geo
        self.cent = {}     # (layer, global head) -> (q centroids, k centroids): the warm start of the next step

    def ln_mod(self, x, scale, shift):
        return O.modulate_shift(O.fp32_layernorm(x, eps=1e-6), scale[None], shift[None], x.dtype)

    def ln_affine(self, x, w, b):
        return O.fp32_layernorm(x, w, b, eps=1e-6).to(x.dtype)

    def gate_res(self, res, x, gate):
        return O.modulate_gate_residual(res, x, gate[None], res.dtype)

    def rms(self, x, w):
        xf = x.float()
        return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6) * w.float()).to(x.dtype)

    def prologue(self, blk, st, q_buf, k_buf, v_buf, pos0, n):
        g = self.geo
        q_buf, k_buf = self.rms(q_buf, blk["nq"]), self.rms(k_buf, blk["nk"])
        q, k, v = (b.unflatten(2, (g.heads, -1)).transpose(1, 2).contiguous() for b in (q_buf, k_buf, v_buf))
        re, im = st.rot_real[pos0:pos0 + n], st.rot_imag[pos0:pos0 + n]
        return O.rope_complex(q, re, im), O.rope_complex(k, re, im), v

    def self_attention(self, q, k, v, layer, sparse, head_shard=None):
        if not sparse:
            return O.masked_attention(q, k, v, None).to(q.dtype)
        g = self.geo
        h0 = head_shard[0] if head_shard is not None else 0
        out = torch.empty_like(q)
        for h in range(q.shape[1]):
            qh, kh = q[0, h][None], k[0, h][None]
            cq, ck = self.cent.get((layer, h0 + h), (qh[:, :g.qc].clone(), kh[:, :g.kc].clone()))
            ql, cq, qs, _ = O.batch_kmeans_euclid(qh, g.qc, max_iters=g.iter_step, init_centroids=cq)
            kl, ck, ks, _ = O.batch_kmeans_euclid(kh, g.kc, max_iters=g.iter_step, init_centroids=ck)
            self.cent[(layer, h0 + h)] = (cq, ck)
            qs, ks = torch.bincount(ql[0], minlength=g.qc), torch.bincount(kl[0], minlength=g.kc)
            dmap = O.identify_dynamic_map(cq[None], ck[None], qs[None, None], ks[None, None], g.top_p, g.min_kc_ratio)[0, 0]
            em = dmap[ql[0]][:, kl[0]]
            out[0, h] = O.masked_attention(q[0, h], k[0, h], v[0, h], em).to(q.dtype)
        return out

    def cross_attention(self, q, k, v):
        H = self.geo.heads
        qh, kh, vh = (x.unflatten(2, (H, -1)).transpose(1, 2) for x in (q, k, v))
        return O.masked_attention(qh, kh, vh, None).to(q.dtype).transpose(1, 2).flatten(2, 3)

    def linear_gelu(self, x, w, b):
        return torch.nn.functional.gelu(torch.addmm(b, x, w.t()), approximate="tanh")
