"""Duck-typed stand-ins for the diffusers modules the processors touch (diffusers is not installed here; SURVEY.md
Appendix B lists the attributes each processor reads)."""
import torch
import torch.nn as nn


class RMSNorm(nn.Module):
    svg_rmsnorm_compatible = True     # x * rsqrt(mean(x^2) + eps) * weight, like diffusers' RMSNorm (see WanAttn processors' get_qk_norm)

    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.eps = eps

    def forward(self, x):
        dt = x.dtype
        x = x.float()
        return (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + self.eps)).to(dt) * self.weight.to(dt)


class Attention(nn.Module):
    """`attn.to_q/.to_k/.to_v/.to_out/.heads/.norm_q/.norm_k/.add_*`, `.processor`, `.set_processor`, and a forward
    that calls `self.processor(self, hidden_states, **kwargs)` like diffusers' Attention.forward."""

    def __init__(self, dim, heads, qk_norm="rms", across_heads=False, added_kv=False, bias=True, dtype=torch.float32):
        super().__init__()
        self.heads = heads
        hd = dim // heads
        self.to_q, self.to_k, self.to_v = (nn.Linear(dim, dim, bias=bias) for _ in range(3))
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Dropout(0.0)])
        nd = dim if across_heads else hd
        if qk_norm == "rms":
            self.norm_q, self.norm_k = RMSNorm(nd), RMSNorm(nd)
        elif qk_norm == "layer":
            self.norm_q, self.norm_k = nn.LayerNorm(nd), nn.LayerNorm(nd)
        else:
            self.norm_q = self.norm_k = None
        self.add_q_proj = self.add_k_proj = self.add_v_proj = None
        self.norm_added_q = self.norm_added_k = None
        self.to_add_out = None
        if added_kv:
            self.add_q_proj, self.add_k_proj, self.add_v_proj = (nn.Linear(dim, dim) for _ in range(3))
            self.norm_added_q, self.norm_added_k = RMSNorm(hd), RMSNorm(hd)
            self.to_add_out = nn.Linear(dim, dim)
        self.processor = None
        self.to(dtype)

    def set_processor(self, p):
        self.processor = p

    def forward(self, hidden_states, **kw):
        return self.processor(self, hidden_states, **kw)


class Block(nn.Module):
    def __init__(self, attn, name="attn1"):
        super().__init__()
        setattr(self, name, attn)
        self._name = name

    @property
    def the_attn(self):
        return getattr(self, self._name)


class Transformer(nn.Module):
    """forward(hidden_states, encoder_hidden_states, timestep, ...) looping over blocks — enough for the timestep hook."""

    def __init__(self, blocks, blocks_attr="transformer_blocks", residual=True, returns_tuple=True):
        super().__init__()
        setattr(self, blocks_attr, nn.ModuleList(blocks))
        self._blocks_attr = blocks_attr
        self.returns_tuple = returns_tuple

    def forward(self, hidden_states, encoder_hidden_states=None, timestep=None, **kw):
        for b in getattr(self, self._blocks_attr):
            out = b.the_attn(hidden_states, encoder_hidden_states=encoder_hidden_states, **kw)
            if isinstance(out, tuple):
                h, e = out
                hidden_states = hidden_states + h
                if e is not None and encoder_hidden_states is not None:
                    encoder_hidden_states = encoder_hidden_states + e
            else:
                hidden_states = hidden_states + out
        return hidden_states, encoder_hidden_states


class Pipe:
    def __init__(self, transformer):
        self.transformer = transformer
        self.device = "cpu"
