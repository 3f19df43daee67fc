"""Cosmos attention processors — same class names / class-level configuration / call protocol as the reference module
svg/models/cosmos/attention.py (`proc(attn, hidden_states, encoder_hidden_states, attention_mask, image_rotary_emb, timestep)`
-> hidden_states).  The sparse core (online profiler, band attention with fused layout transformation, SVG2) is the Wan one:
the reference's cosmos/utils.py equals wan/utils.py and both `attention_core_logic`s are the same code."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from ...timer import time_logging_decorator
from .. import _core
from ..wan.attention import (  # noqa: F401
    WanAttn_SAPAttn_Processor,
    WanAttn_SVGAttn_Processor2_0,
    prepare_flashinfer_attention,
    prepare_flexattention,
)


def apply_rotary_emb_half(x: torch.Tensor, freqs_cis) -> torch.Tensor:
    """diffusers `apply_rotary_emb(use_real=True, use_real_unbind_dim=-2)` (ref: cosmos/attention.py:61-66): the channel
    halves are the real / imaginary parts, cos / sin: [S, D]."""
    cos, sin = freqs_cis
    cos, sin = cos[None, None].to(x.device), sin[None, None].to(x.device)
    x_real, x_imag = x.reshape(*x.shape[:-1], 2, -1).unbind(-2)
    x_rot = torch.cat([-x_imag, x_real], dim=-1)
    return (x.float() * cos + x_rot.float() * sin).to(x.dtype)


class _CosmosPlumbing:
    """QKV / per-head RMSNorm / half-split RoPE / output projection of the Cosmos processors (ref :40-124)."""

    @time_logging_decorator("Level 2 - qkv")
    def get_qkv(self, attn, hidden_states, encoder_hidden_states):
        return attn.to_q(hidden_states), attn.to_k(encoder_hidden_states), attn.to_v(encoder_hidden_states)

    @time_logging_decorator("Level 2 - transpose")
    def get_transpose_qkv(self, attn, query, key, value):
        return tuple(x.unflatten(2, (attn.heads, -1)).transpose(1, 2).contiguous() for x in (query, key, value))

    @time_logging_decorator("Level 2 - qk_norm")
    def get_qk_norm(self, attn, query, key):
        nq, nk = getattr(attn, "norm_q", None), getattr(attn, "norm_k", None)
        if _core.qk_norm_inplace(nq, nk, query, key):      # per-head RMSNorm on libsvgattn (GPU tensors)
            return query, key
        return (nq(query) if nq is not None else query), (nk(key) if nk is not None else key)

    @time_logging_decorator("Level 2 - rotary_emb")
    def get_rotary_emb(self, query, key, image_rotary_emb):
        if image_rotary_emb is not None:
            query, key = apply_rotary_emb_half(query, image_rotary_emb), apply_rotary_emb_half(key, image_rotary_emb)
        return query, key

    @time_logging_decorator("Level 2 - output")
    def get_o(self, attn, query, hidden_states):
        return attn.to_out[1](attn.to_out[0](hidden_states))

    def __call__(self, attn, hidden_states: torch.Tensor, encoder_hidden_states: Optional[torch.Tensor] = None,
                 attention_mask: Optional[torch.Tensor] = None, image_rotary_emb=None, timestep=None):
        cross = encoder_hidden_states is not None
        if timestep is None and not cross:
            from ..context import current_timestep

            timestep = current_timestep()
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        query, key, value = self.get_qkv(attn, hidden_states, encoder_hidden_states)
        query, key, value = self.get_transpose_qkv(attn, query, key, value)
        query, key = self.get_qk_norm(attn, query, key)
        query, key = self.get_rotary_emb(query, key, image_rotary_emb)
        assert query.shape[3] == key.shape[3] == value.shape[3], "Does not support GQA"
        if timestep is None or cross:   # cross attention in Cosmos (ref :104-107)
            hidden_states = F.scaled_dot_product_attention(query, key, value, attn_mask=attention_mask, dropout_p=0.0,
                                                           is_causal=False)
        else:
            assert query.is_contiguous() and key.is_contiguous() and value.is_contiguous(), "Query, key, value must be contiguous"
            hidden_states = self.attention_core_logic(query, key, value, timestep)
        hidden_states = hidden_states.transpose(1, 2).flatten(2, 3).type_as(query)
        return self.get_o(attn, query, hidden_states)


class Cosmos_SVG_AttnProcessor2_0(_CosmosPlumbing, WanAttn_SVGAttn_Processor2_0):
    """Sparse VideoGen 1 for Cosmos (ref: cosmos/attention.py:30-238)."""


class Cosmos_SAPAttn_Processor(_CosmosPlumbing, WanAttn_SAPAttn_Processor):
    """Sparse VideoGen 2 for Cosmos (ref: cosmos/attention.py:289-469)."""
