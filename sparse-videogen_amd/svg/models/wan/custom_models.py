"""ref: svg/models/wan/custom_models.py — make `timestep` reach every self-attention processor.

`replace_sparse_forward()` keeps the reference's name and zero-argument call; it needs the pipeline's transformer, which
the install hook (`replace_wan_attention`) registers through `register_transformer`."""
from __future__ import annotations

import types

import torch

from ... import _native
from ..context import TransformerRegistry



def wan_block_forward(self, hidden_states, encoder_hidden_states, temb, rotary_emb, timestep=None, **kwargs):
    """Forward of a Wan transformer block (ref: WanTransformerBlock_Sparse.forward, svg/models/wan/custom_models.py:23-111) with
    the block glue on libsvgattn: fused fp32 LayerNorm + modulate (one pass instead of the reference's two Triton kernels) and
    gate-residual.  Duck-typed on diffusers' WanTransformerBlock attributes (scale_shift_table, norm1/2/3, attn1/2, ffn);
    CPU tensors take the torch expressions of the reference's fall-back branch."""
    shift_msa, scale_msa, gate_msa, c_shift_msa, c_scale_msa, c_gate_msa = (self.scale_shift_table + temb.float()).chunk(6, dim=1)
    fast = hidden_states.is_cuda and hidden_states.dtype in (torch.bfloat16, torch.float16) and hidden_states.shape[-1] % 8 == 0 \
        and hidden_states.shape[-1] <= 8192

    def norm_mod(norm, scale, shift):
        x = hidden_states.contiguous()
        affine = getattr(norm, "elementwise_affine", getattr(norm, "weight", None) is not None)
        w, b = (norm.weight, norm.bias) if affine else (None, None)
        if fast:
            from ...kernels.triton import layernorm as _ln   # (REFERENCE_PADDING: the reference's fast-path variance, opt-in)

            return _native.layernorm_modulate_forward(x, w, b, scale, shift, norm.eps, hidden_states.dtype,
                                                      reference_padding=_ln.REFERENCE_PADDING)
        y = norm(x.float())
        return (y * (1 + scale) + shift).type_as(hidden_states) if scale is not None else y.type_as(hidden_states)

    def gate_res(x, gate):
        if fast:
            return _native.modulate_gate_residual_forward(hidden_states.contiguous(), x.contiguous(), gate, hidden_states.dtype)
        return (hidden_states.float() + x.float() * gate).type_as(hidden_states)

    attn_kw = {"timestep": timestep} if timestep is not None else {}
    attn_output = self.attn1(hidden_states=norm_mod(self.norm1, scale_msa, shift_msa), rotary_emb=rotary_emb, **attn_kw)
    hidden_states = gate_res(attn_output, gate_msa)
    attn_output = self.attn2(hidden_states=norm_mod(self.norm2, None, None), encoder_hidden_states=encoder_hidden_states)
    hidden_states = hidden_states + attn_output
    ff_output = self.ffn(norm_mod(self.norm3, c_scale_msa, c_shift_msa))
    return gate_res(ff_output, c_gate_msa)


def _is_wan_block(b) -> bool:
    return all(hasattr(b, a) for a in ("scale_shift_table", "norm1", "attn1", "norm2", "attn2", "norm3", "ffn"))


def install_block_forward(transformer) -> int:
    """Bind wan_block_forward to every Wan-style block of `transformer.blocks`; returns the number of blocks patched."""
    n = 0
    for b in getattr(transformer, "blocks", []):
        if _is_wan_block(b):
            b.forward = types.MethodType(wan_block_forward, b)
            n += 1
    return n


_REGISTRY = TransformerRegistry(also=install_block_forward)
register_transformer = _REGISTRY.register_transformer
replace_sparse_forward = _REGISTRY.replace_sparse_forward
