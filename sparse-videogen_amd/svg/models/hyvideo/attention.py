"""HunyuanVideo attention processors — same class names, class-level configuration attributes and call protocol as the
reference module svg/models/hyvideo/attention.py (diffusers attention-processor protocol: `proc(attn, hidden_states,
encoder_hidden_states, attention_mask, image_rotary_emb, timestep)` -> (hidden_states, encoder_hidden_states)).

Everything inside `attention_core_logic` runs on libsvgattn (HIP, gfx950); the projections / QK-norm / RoPE around it
are the model's own torch modules, as in the reference's torch fall-back path (hyvideo/attention.py:195-225).
"""
from __future__ import annotations

import weakref
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from ...timer import time_logging_decorator
from .. import _core
from .._core import CentroidStore, Geometry
from .utils import dense_mask, generate_temporal_head_mask_mod, profile_desc


# ---- pre-attention helpers (torch; ref: hyvideo/attention.py:195-225) -------------------------------------------
def apply_rotary_emb(x: torch.Tensor, freqs_cis) -> torch.Tensor:
    """diffusers `apply_rotary_emb(use_real=True, use_real_unbind_dim=-1)`: interleaved pairs, cos/sin [S, D]."""
    cos, sin = freqs_cis
    cos, sin = cos[None, None].to(x.device), sin[None, None].to(x.device)
    x_real, x_imag = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    x_rot = torch.stack([-x_imag, x_real], dim=-1).flatten(3)
    return (x.float() * cos + x_rot.float() * sin).to(x.dtype)


def apply_qk_norm(norm_q, norm_k, query, key):
    if _core.qk_norm_inplace(norm_q, norm_k, query, key):   # HIP, in place (ref: the `_kernels.rms_norm_forward` path :162-170)
        return query, key
    if norm_q is not None:
        query = norm_q(query)
    if norm_k is not None:
        key = norm_k(key)
    return query, key


def apply_qk_rope_single(query, key, image_rotary_emb, encoder_hidden_states):
    n_txt = encoder_hidden_states.shape[1]
    cos, sin = image_rotary_emb
    if _core.qk_rope_inplace(query, key, cos, sin, 0, query.shape[2] - n_txt):   # `apply_qk_rope_inplace_cossin_txtlast` :172-178
        return query, key
    query = torch.cat([apply_rotary_emb(query[:, :, :-n_txt], image_rotary_emb), query[:, :, -n_txt:]], dim=2)
    key = torch.cat([apply_rotary_emb(key[:, :, :-n_txt], image_rotary_emb), key[:, :, -n_txt:]], dim=2)
    return query, key


def apply_qk_rope_double(query, key, image_rotary_emb):
    cos, sin = image_rotary_emb
    if _core.qk_rope_inplace(query, key, cos, sin, 0, query.shape[2]):           # txtlast with len_text_prompt = 0, :180-188
        return query, key
    return apply_rotary_emb(query, image_rotary_emb), apply_rotary_emb(key, image_rotary_emb)


class _HunyuanProcessorBase:
    """QKV projection / norm / RoPE / text concat / output projection shared by the three Hunyuan processors
    (ref: hyvideo/attention.py:252-373)."""

    fused_prologue = True   # QK-norm + RoPE in one HIP pass (False: two stages, as in the reference)
    # prescale_q = what flex_attention calls its PRESCALE_QK kernel option ("pre-scale QK by 1/sqrt(d) and change of base. Has about
    # 20% more numerical error, but slightly faster", torch/_inductor/kernel/flex/templates/flex_attention.py.jinja; default False,
    # which is what the reference runs: it passes no kernel_options, svg/models/hyvideo/attention.py:401-403).  True: the fused
    # prologue folds sm_scale * log2(e) into its last rounding of q and the attention / profiler kernels run their pre-scaled forms
    # (svg_band_attention_prescaled: one FMA per score less, -4 % kernel time) — the scores then come from round(c * q) instead of
    # c * round(q): against the reference's formulation the bf16 output moves by 3e-3 (scores of O(1)) to 6e-3 rel. L2 (|score| 50 - 80),
    # tests/test_gpu_prescaled.py::test_prescaled_vs_reference_formulation_on_large_logits.  OFF by default since round 4 (parity first);
    # only when the fused HIP prologue applies (GPU tensors, head_dim 128); q never leaves the processor, so nothing outside sees it.
    prescale_q = False
    _valid_len_cache: dict = {}

    def __init__(self, layer_idx: int = 0):
        self.layer_idx = layer_idx
        self._q_prescaled = False   # set by __call__ around attention_core_logic (whose signature is the reference's)

    @time_logging_decorator("Level 2 - get_qkv")
    def get_qkv(self, attn, hidden_states):
        return attn.to_q(hidden_states), attn.to_k(hidden_states), attn.to_v(hidden_states)

    @time_logging_decorator("Level 2 - get_transpose_qkv")
    def get_transpose_qkv(self, attn, query, key, value):
        return tuple(x.unflatten(2, (attn.heads, -1)).transpose(1, 2).contiguous() for x in (query, key, value))

    @time_logging_decorator("Level 2 - get_qk_norm")
    def get_qk_norm(self, attn, query, key):
        return apply_qk_norm(getattr(attn, "norm_q", None), getattr(attn, "norm_k", None), query, key)

    @time_logging_decorator("Level 2 - get_rotary_emb")
    def get_rotary_emb(self, attn, query, key, image_rotary_emb, encoder_hidden_states):
        if image_rotary_emb is not None:
            if getattr(attn, "add_q_proj", None) is None and encoder_hidden_states is not None:
                query, key = apply_qk_rope_single(query, key, image_rotary_emb, encoder_hidden_states)
            else:
                query, key = apply_qk_rope_double(query, key, image_rotary_emb)
        return query, key

    @time_logging_decorator("Level 2 - get_transpose_norm_rope")
    def get_transpose_norm_rope(self, attn, query, key, value, image_rotary_emb, encoder_hidden_states, q_scale: float = 1.0):
        """get_transpose_qkv + get_qk_norm + get_rotary_emb in one read and one write per element
        (svg_qk_norm_rope_transpose); None when the HIP path does not apply."""
        if not self.fused_prologue:
            return None
        single = getattr(attn, "add_q_proj", None) is None and encoder_hidden_states is not None
        S = query.shape[1]
        hi = S - (encoder_hidden_states.shape[1] if single else 0)
        cos, sin = image_rotary_emb if image_rotary_emb is not None else (None, None)
        return _core.qkv_from_projections(query, key, value, attn.heads, getattr(attn, "norm_q", None),
                                          getattr(attn, "norm_k", None), cos, sin, 0, hi, q_scale=q_scale)

    @time_logging_decorator("Level 2 - get_fused_prologue")
    def get_fused_prologue(self, attn, query, key, image_rotary_emb, encoder_hidden_states) -> bool:
        """QK-norm + RoPE in ONE pass over q and k (svg_qk_norm_rope) when both apply and the tensors are on the GPU;
        bit-identical to get_qk_norm followed by get_rotary_emb on the HIP path.  False: run the two stages."""
        if not self.fused_prologue or image_rotary_emb is None:
            return False
        nq, nk = getattr(attn, "norm_q", None), getattr(attn, "norm_k", None)
        if nq is None or nk is None:
            return False
        single = getattr(attn, "add_q_proj", None) is None and encoder_hidden_states is not None
        hi = query.shape[2] - (encoder_hidden_states.shape[1] if single else 0)
        cos, sin = image_rotary_emb
        return _core.qk_rope_inplace(query, key, cos, sin, 0, hi, norm_q=nq, norm_k=nk)

    @time_logging_decorator("Level 2 - get_encoder_condition_and_concat")
    def get_encoder_condition_and_concat(self, attn, query, key, value, encoder_hidden_states, q_scale: float = 1.0):
        if getattr(attn, "add_q_proj", None) is not None and encoder_hidden_states is not None:
            eq = attn.add_q_proj(encoder_hidden_states).unflatten(2, (attn.heads, -1)).transpose(1, 2)
            ek = attn.add_k_proj(encoder_hidden_states).unflatten(2, (attn.heads, -1)).transpose(1, 2)
            ev = attn.add_v_proj(encoder_hidden_states).unflatten(2, (attn.heads, -1)).transpose(1, 2)
            if getattr(attn, "norm_added_q", None) is not None:
                eq = attn.norm_added_q(eq)
            if getattr(attn, "norm_added_k", None) is not None:
                ek = attn.norm_added_k(ek)
            if q_scale != 1.0:   # the text stream's q joins a pre-scaled q (256 tokens; the torch modules rounded it already)
                eq = (eq.float() * q_scale).to(eq.dtype)
            query = torch.cat([query, eq], dim=2)
            key = torch.cat([key, ek], dim=2)
            value = torch.cat([value, ev], dim=2)
        return query, key, value

    @time_logging_decorator("Level 2 - get_cu_max_seqlen")
    def get_cu_max_seqlen(self, attention_mask, device):
        """ref :308-316.  Returns (valid_len, seq_len): the two dense segments are [0, valid) and [valid, S)."""
        if attention_mask is None:
            return None, None
        # The mask is the same tensor object for every layer of a forward pass: its popcount is read back to the host once per
        # tensor object (the reference synchronises on it in every layer-call).  Keyed by object identity, guarded by a weak
        # reference and the in-place version counter, so a recycled address or an edited mask never hits.
        cache = _HunyuanProcessorBase._valid_len_cache
        ent = cache.get(id(attention_mask))
        if ent is not None and ent[0]() is attention_mask and ent[1] == attention_mask._version:
            return ent[2], attention_mask.numel()
        if len(cache) > 8:
            cache.clear()
        n = int(attention_mask.sum())
        cache[id(attention_mask)] = (weakref.ref(attention_mask), attention_mask._version, n)
        return n, attention_mask.numel()

    @time_logging_decorator("Level 2 - get_o")
    def get_o(self, attn, hidden_states, encoder_hidden_states):
        if encoder_hidden_states is not None:
            n_txt = encoder_hidden_states.shape[1]
            hidden_states, encoder_hidden_states = hidden_states[:, :-n_txt], hidden_states[:, -n_txt:]
            if getattr(attn, "to_out", None) is not None:
                hidden_states = attn.to_out[1](attn.to_out[0](hidden_states))
            if getattr(attn, "to_add_out", None) is not None:
                encoder_hidden_states = attn.to_add_out(encoder_hidden_states)
        return hidden_states, encoder_hidden_states

    def __call__(self, attn, hidden_states: torch.Tensor, encoder_hidden_states: Optional[torch.Tensor] = None,
                 attention_mask: Optional[torch.Tensor] = None, image_rotary_emb=None, timestep=None):
        if timestep is None:
            from ..context import current_timestep

            timestep = current_timestep()
        if getattr(attn, "add_q_proj", None) is None and encoder_hidden_states is not None:
            hidden_states = torch.cat([hidden_states, encoder_hidden_states], dim=1)
        query, key, value = self.get_qkv(attn, hidden_states)
        q_scale = 1.0
        if self.prescale_q and self.fused_prologue and query.is_cuda and query.shape[-1] == attn.heads * 128 \
                and query.dtype in (torch.bfloat16, torch.float16):
            q_scale = _core._native.softmax_q_scale(128)
        fused = self.get_transpose_norm_rope(attn, query, key, value, image_rotary_emb, encoder_hidden_states, q_scale=q_scale)
        if fused is not None:
            query, key, value = fused
        else:
            q_scale = 1.0   # the staged prologue (torch modules or the in-place HIP ops) delivers a plain q
            query, key, value = self.get_transpose_qkv(attn, query, key, value)
            if not self.get_fused_prologue(attn, query, key, image_rotary_emb, encoder_hidden_states):
                query, key = self.get_qk_norm(attn, query, key)
                query, key = self.get_rotary_emb(attn, query, key, image_rotary_emb, encoder_hidden_states)
        query, key, value = self.get_encoder_condition_and_concat(attn, query, key, value, encoder_hidden_states, q_scale=q_scale)
        cu_max_seqlens = self.get_cu_max_seqlen(attention_mask, query.device)
        self._q_prescaled = q_scale != 1.0
        try:
            hidden_states = self.attention_core_logic(query, key, value, timestep, self.layer_idx, cu_max_seqlens)
        finally:
            self._q_prescaled = False
        hidden_states = hidden_states.transpose(1, 2).flatten(2, 3).to(query.dtype)
        return self.get_o(attn, hidden_states, encoder_hidden_states)


class HunyuanVideoAttnProcessor2_0_FlashAttention(_HunyuanProcessorBase):
    """Dense baseline (ref: hyvideo/attention.py:35-152): two segments, pad tokens attend only among themselves."""

    def __init__(self, layer_idx: int = 0):
        super().__init__(layer_idx)

    @time_logging_decorator("Level 2 - attention core logic")
    def attention_core_logic(self, query, key, value, timestep, layer_idx, cu_max_seqlens):
        valid = cu_max_seqlens[0] if cu_max_seqlens is not None else None
        return _core.dense_attention(query, key, value, valid)


class Hunyuan_SVGAttn_Processor2_0(_HunyuanProcessorBase):
    """Sparse VideoGen 1 (ref: hyvideo/attention.py:228-524).  Configuration lives in CLASS attributes set by
    replace_hyvideo_attention, exactly like the reference."""

    num_sampled_rows = 32
    attention_masks = None  # [spatial, temporal] analytic profiling-mask variants (see utils.get_attention_mask)

    prompt_length = 0
    context_length = 256
    num_frame = 33
    frame_size = 3600

    first_layers_fp = 0
    first_times_fp = 0

    sample_mse_max_row = 10000
    block_mask = None      # svg_band_mask_t descriptor (what the reference's flex BlockMask encodes)
    fused_placement = True  # fold both layout transformations into the attention kernel (bit-identical result)
    device_switch = True    # dense / sparse decision on the device when the timestep is a GPU tensor (no read-back per forward)
    prescale_q = False      # opt-in (flex_attention's PRESCALE_QK trade-off, see _HunyuanProcessorBase.prescale_q): q leaves the fused prologue carrying the softmax scale

    def __init__(self, layer_idx):
        super().__init__(layer_idx)
        self.last_best_mask_idx = None

    @classmethod
    def geometry(cls) -> Geometry:
        return Geometry(cls.context_length, cls.num_frame, cls.frame_size, text_first=False)

    @time_logging_decorator("Level 3 - sample_mse")
    def sample_mse(self, query, key, value):
        """-> [2, cfg, H] mean-squared errors of the spatial / temporal mask on sampled rows (ref :376-399)"""
        geo = self.geometry()
        return _core.sample_mse(query, key, value, geo, profile_desc(geo.context_length, geo.num_frame, geo.frame_size),
                                self.num_sampled_rows, self.sample_mse_max_row, q_prescaled=self._q_prescaled)

    @time_logging_decorator("Level 2 - attention core logic")
    def attention_core_logic(self, query, key, value, timestep, layer_idx, cu_max_seqlens):
        cfg, num_heads, seq_len, dim = query.size()
        geo = self.geometry()
        assert seq_len == geo.seq_len, (
            f"Query Shape: {seq_len} is not equivalent to {geo.context_length} + {geo.num_frame} * {geo.frame_size}")
        valid = cu_max_seqlens[0] if cu_max_seqlens is not None and cu_max_seqlens[0] is not None else (
            geo.video_length + self.prompt_length)
        # the layer test is host data; the timestep test stays on the device when the timestep is a GPU tensor (device_switch)
        dense_flag = None
        if self.device_switch and _core.attention_dtype() == "bf16" and self.fused_placement and self.layer_idx >= self.first_layers_fp and self.block_mask is not None \
                and query.is_cuda:
            dense_flag = _core.dense_flag_on_device(timestep, self.first_times_fp)
        pre = self._q_prescaled
        if dense_flag is None and _core.is_full_attention(self.layer_idx, timestep, self.first_layers_fp, self.first_times_fp):
            return _core.dense_attention(query, key, value, valid, q_prescaled=pre).reshape(cfg, num_heads, seq_len, dim)
        mask = self.block_mask
        if mask is None:
            raise RuntimeError("Hunyuan_SVGAttn_Processor2_0.block_mask is not set: call replace_hyvideo_attention first")
        prof = profile_desc(geo.context_length, geo.num_frame, geo.frame_size)
        if dense_flag is not None:
            out, best = _core.svg1_attention_device_switch(query, key, value, geo, mask, dense_mask(seq_len, int(valid)), prof,
                                                           self.num_sampled_rows, min(self.sample_mse_max_row, seq_len), dense_flag,
                                                           q_prescaled=pre)
            self.last_best_mask_idx = best
            return out.reshape(cfg, num_heads, seq_len, dim)
        out, best = _core.svg1_sparse_attention(query, key, value, geo, mask, prof, self.num_sampled_rows,
                                                min(self.sample_mse_max_row, seq_len), fused=self.fused_placement, q_prescaled=pre)
        self.last_best_mask_idx = best
        return out.reshape(cfg, num_heads, seq_len, dim)


def prepare_flexattention(cfg_size, num_head, head_dim, dtype, device, context_length, prompt_length, num_frame, frame_size,
                          diag_width=1, multiplier=2):
    """ref: hyvideo/attention.py:527-551.  The reference builds and warm-compiles a flex_attention BlockMask (minutes of
    Inductor time); here the mask is six integers and nothing needs compiling."""
    assert diag_width == multiplier
    return generate_temporal_head_mask_mod(context_length, prompt_length, num_frame, frame_size, mul=multiplier)


class Hunyuan_SAPAttn_Processor2_0(Hunyuan_SVGAttn_Processor2_0):
    """Sparse VideoGen 2 — semantic-aware permutation (ref: hyvideo/attention.py:555-804)."""

    num_q_centroids = 0
    num_k_centroids = 0
    top_p_kmeans = 0
    min_kc_ratio = 0

    kmeans_iter_init = 0
    kmeans_iter_step = 0
    zero_step_kmeans_init = False

    logging_file = None
    prescale_q = False   # k-means and the block map work on the plain q
    centroid_store = CentroidStore()  # class-level like the reference's dicts; `reset_state()` clears it

    @classmethod
    def reset_state(cls):
        """Forget the per-layer centroids (call between videos; the reference never resets them)."""
        cls.centroid_store.clear()

    @time_logging_decorator("Level 2 - attention core logic")
    def attention_core_logic(self, query, key, value, timestep, layer_idx, cu_max_seqlens):
        cfg, num_heads, seq_len, dim = query.size()
        assert cfg == 1, "Batch size must be 1 for kmeans block sparse attention"
        geo = self.geometry()
        assert seq_len == geo.seq_len, (
            f"Query Shape: {seq_len} is not equivalent to {geo.context_length} + {geo.num_frame} * {geo.frame_size}")
        if _core.is_full_attention(self.layer_idx, timestep, self.first_layers_fp, self.first_times_fp):
            if self.zero_step_kmeans_init and query.is_cuda:
                V = geo.video_length
                _core.kmeans_clustering(self.centroid_store, layer_idx, query[:, :, :V], key[:, :, :V],   # (views: read in place)
                                        self.num_q_centroids, self.num_k_centroids,
                                        self.kmeans_iter_init, self.kmeans_iter_step)
            valid = cu_max_seqlens[0] if cu_max_seqlens is not None and cu_max_seqlens[0] is not None else (
                geo.video_length + self.prompt_length)
            return _core.dense_attention(query, key, value, valid).reshape(cfg, num_heads, seq_len, dim)
        out = _core.svg2_sparse_attention(query, key, value, geo, self.centroid_store, layer_idx, self.num_q_centroids,
                                          self.num_k_centroids, self.top_p_kmeans, self.min_kc_ratio, self.kmeans_iter_init,
                                          self.kmeans_iter_step, prompt_length=int(self.prompt_length),
                                          logging_file=self.logging_file, timestep=timestep)
        return out.reshape(cfg, num_heads, seq_len, dim)


def replace_hyvideo_flashattention(pipe):
    """hyvideo_i2v_inference.py:14 imports this name from `svg.models.hyvideo.attention` (the reference defines it in
    `.inference`, ref: svg/models/hyvideo/inference.py:16-30); both module paths work here."""
    from .inference import replace_hyvideo_flashattention as _impl

    return _impl(pipe)
