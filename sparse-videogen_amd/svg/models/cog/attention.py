"""CogVideoX attention processor — same class name / configuration / call protocol as the reference module
svg/models/cog/attention.py (text tokens first; returns (hidden_states, encoder_hidden_states))."""
from __future__ import annotations

from typing import Optional

import torch

from .. import _core
from .._core import Geometry
from ..hyvideo.attention import apply_rotary_emb
from .utils import dense_mask, generate_temporal_head_mask_mod, profile_desc


def qk_norm(attn, query, key):
    """ref: cog/attention.py:40-45 (LayerNorm over head_dim); HIP fast path = `_kernels.layer_norm_forward` (:23-29)"""
    if _core.qk_norm_inplace(getattr(attn, "norm_q", None), getattr(attn, "norm_k", None), query, key):
        return query, key
    if getattr(attn, "norm_q", None) is not None:
        query = attn.norm_q(query)
    if getattr(attn, "norm_k", None) is not None:
        key = attn.norm_k(key)
    return query, key


def rotary_emb(image_rotary_emb, query, key, text_seq_length, q_scale: float = 1.0, scaled: Optional[list] = None):
    """ref: cog/attention.py:47-50 — RoPE on the video tokens only (text first).  q_scale != 1: folded into the HIP pass's last rounding
    of q (the text tokens, which the pass does not rotate, are multiplied and rounded once more) IF that pass takes the call — `scaled`
    (a list) receives True / False; when it declines, the torch RoPE below runs on the plain q."""
    if image_rotary_emb is not None:
        cos, sin = image_rotary_emb   # HIP fast path = `_kernels.apply_qk_rope_inplace_cossin` (text first), :31-34
        took = bool(_core.qk_rope_inplace(query, key, cos, sin, text_seq_length, query.shape[2], q_scale=q_scale))
        if scaled is not None:
            scaled.append(took and q_scale != 1.0)
        if took:
            return query, key
        query[:, :, text_seq_length:] = apply_rotary_emb(query[:, :, text_seq_length:], image_rotary_emb)
        key[:, :, text_seq_length:] = apply_rotary_emb(key[:, :, text_seq_length:], image_rotary_emb)
    return query, key


class CogVideoX_SparseAttn_Processor2_0:
    """Sparse VideoGen 1 for CogVideoX (ref: cog/attention.py:59-224)."""

    version = None
    context_length = 0
    num_frame = 0
    frame_size = 0

    first_layers_fp = 0
    first_times_fp = 0

    num_sampled_rows = 32
    attention_masks = None
    block_mask = None
    fused_placement = True
    device_switch = True    # dense / sparse decision on the device when the timestep is a GPU tensor
    prescale_q = False      # opt-in (flex_attention's PRESCALE_QK trade-off, see Hunyuan's processor): the HIP RoPE pass folds sm_scale * log2(e) into its rounding of q

    def __init__(self, layer_idx):
        self.layer_idx = layer_idx
        self.last_best_mask_idx = None
        self._q_prescaled = False

    @classmethod
    def geometry(cls) -> Geometry:
        return Geometry(cls.context_length, cls.num_frame, cls.frame_size, text_first=True)

    def get_qkv(self, attn, hidden_states):
        return attn.to_q(hidden_states), attn.to_k(hidden_states), attn.to_v(hidden_states)

    def process_before_linear(self, attn, hidden_states, encoder_hidden_states):
        hidden_states = torch.cat([encoder_hidden_states, hidden_states], dim=1)
        batch_size, sequence_length, _ = hidden_states.shape
        return hidden_states, batch_size, sequence_length

    def transpose_qkv(self, attn, query, key, value, batch_size):
        head_dim = key.shape[-1] // attn.heads
        # v: a view of the projection's output where the attention kernels read it in place (svg_attn_layout_t), a head-major copy otherwise
        vv = _core.value_in_place(value, attn.heads) if not self.prescale_q else None
        query, key = (x.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2).contiguous() for x in (query, key))
        value = vv if vv is not None else value.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2).contiguous()
        return query, key, value, head_dim

    def get_o(self, attn, hidden_states, batch_size, head_dim):
        hidden_states = hidden_states.transpose(1, 2).reshape(batch_size, -1, attn.heads * head_dim)
        return attn.to_out[1](attn.to_out[0](hidden_states))

    def split_hidden_states(self, hidden_states, text_seq_length):
        return hidden_states.split([text_seq_length, hidden_states.size(1) - text_seq_length], dim=1)

    def flash_attention(self, query, key, value):
        return _core.dense_attention(query, key, value, q_prescaled=self._q_prescaled)

    def sample_mse(self, query, key, value):
        """Cog samples rows from the WHOLE sequence (ref :126)"""
        geo = self.geometry()
        return _core.sample_mse(query, key, value, geo, profile_desc(geo.context_length, geo.num_frame, geo.frame_size),
                                self.num_sampled_rows, query.shape[2], q_prescaled=self._q_prescaled)

    def attention_core_logic(self, query, key, value, timestep):
        cfg, num_heads, seq_len, dim = query.size()
        geo = self.geometry()
        assert seq_len == geo.seq_len, (
            f"Query Shape: {seq_len} is not equivalent to {geo.context_length} + {geo.num_frame} * {geo.frame_size}")
        # ref :173-176 — Cog's warm-up thresholds are fractions of 42 layers / 1000 timesteps
        first_layers, first_times = 42 * self.first_layers_fp, 1000 * (1 - self.first_times_fp)
        dense_flag = None   # (device-side dense / sparse switch, see the Hunyuan processor)
        if self.device_switch and _core.attention_dtype() == "bf16" and self.fused_placement and self.layer_idx >= first_layers and self.block_mask is not None \
                and query.is_cuda:
            dense_flag = _core.dense_flag_on_device(timestep, first_times)
        if dense_flag is None and _core.is_full_attention(self.layer_idx, timestep, first_layers, first_times):
            return self.flash_attention(query, key, value).reshape(cfg, num_heads, seq_len, dim)
        if self.block_mask is None:
            raise RuntimeError("CogVideoX_SparseAttn_Processor2_0.block_mask is not set: call replace_cog_attention first")
        prof = profile_desc(geo.context_length, geo.num_frame, geo.frame_size)
        if dense_flag is not None:
            out, best = _core.svg1_attention_device_switch(query, key, value, geo, self.block_mask, dense_mask(seq_len), prof,
                                                           self.num_sampled_rows, seq_len, dense_flag, q_prescaled=self._q_prescaled)
            self.last_best_mask_idx = best
            return out.reshape(cfg, num_heads, seq_len, dim)
        out, best = _core.svg1_sparse_attention(query, key, value, geo, self.block_mask, prof, self.num_sampled_rows, seq_len,
                                                fused=self.fused_placement, q_prescaled=self._q_prescaled)
        self.last_best_mask_idx = best
        return out.reshape(cfg, num_heads, seq_len, dim)

    def __call__(self, attn, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor,
                 attention_mask: Optional[torch.Tensor] = None, image_rotary_emb=None, timestep=None):
        if timestep is None:
            from ..context import current_timestep

            timestep = current_timestep()
        text_seq_length = encoder_hidden_states.size(1)
        hidden_states, batch_size, _ = self.process_before_linear(attn, hidden_states, encoder_hidden_states)
        query, key, value = self.get_qkv(attn, hidden_states)
        query, key, value, head_dim = self.transpose_qkv(attn, query, key, value, batch_size)
        query, key = qk_norm(attn, query, key)
        q_scale, flag = 1.0, []
        if self.prescale_q and image_rotary_emb is not None and _core.prescale_supported(query):
            q_scale = _core._native.softmax_q_scale(query.shape[-1])     # a request: the RoPE pass reports whether it folded it in
        query, key = rotary_emb(image_rotary_emb, query, key, text_seq_length, q_scale=q_scale, scaled=flag)
        self._q_prescaled = bool(flag and flag[0])
        try:
            hidden_states = self.attention_core_logic(query, key, value, timestep)
        finally:
            self._q_prescaled = False
        hidden_states = self.get_o(attn, hidden_states, batch_size, head_dim)
        encoder_hidden_states, hidden_states = self.split_hidden_states(hidden_states, text_seq_length)
        return hidden_states, encoder_hidden_states


def prepare_flexattention(cfg_size, num_head, head_dim, dtype, device, context_length, num_frame, frame_size, diag_width=1,
                          multiplier=2):
    """ref: cog/attention.py:227-239 — the band-mask descriptor (attn_sink=False like the reference)."""
    assert diag_width == multiplier
    return generate_temporal_head_mask_mod(context_length, num_frame, frame_size, mul=multiplier, attn_sink=False)
