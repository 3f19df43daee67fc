"""Wan 2.1 attention processors — same class names / class-level configuration / call protocol as the reference module
svg/models/wan/attention.py (`proc(attn, hidden_states, encoder_hidden_states, attention_mask, rotary_emb, timestep)`
-> hidden_states).  `attention_core_logic` runs on libsvgattn (HIP)."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from ...kernels.triton.rmsnorm import triton_rmsnorm_forward
from ...timer import time_logging_decorator
from .. import _core
from .._core import CentroidStore, Geometry
from .utils import dense_mask, generate_temporal_head_mask_mod, profile_desc


def apply_rotary_emb(query: torch.Tensor, key: torch.Tensor, freqs, q_scale: float = 1.0, scaled: Optional[list] = None):
    """ref: wan/attention.py:40-66.  `freqs` is either the complex tensor [1, 1, S, D/2] of diffusers or the
    (real, imag) fp32 pair [S, D/2] the reference's patched model forward produces for its CUDA kernel.
    q_scale != 1: folded into the last rounding of q IF the HIP pass takes the call — `scaled` (a list) receives True / False, and when the
    pass declines (its own acceptance rules decide, nobody re-states them) the torch RoPE below runs on the plain q."""
    # HIP fast path (ref: `_kernels.apply_qk_rope_inplace_cossin_complex(query, key, freqs_real, freqs_imag, 0)`, :45-48)
    S = query.shape[2]
    if isinstance(freqs, (tuple, list)):
        fr, fi = freqs
    else:
        fr, fi = freqs.real, freqs.imag
    took = bool(_core.qk_rope_inplace(query, key, fr, fi, 0, S, complex_pairs=True, q_scale=q_scale))
    if scaled is not None:
        scaled.append(took and q_scale != 1.0)
    if took:
        return query, key
    if isinstance(freqs, (tuple, list)):
        freqs = torch.complex(fr.double(), fi.double())[None, None]

    def rot(x):
        xc = torch.view_as_complex(x.to(torch.float64).unflatten(3, (-1, 2)))
        return torch.view_as_real(xc * freqs).flatten(3, 4).type_as(x)

    return rot(query), rot(key)


def _is_rms(mod) -> bool:
    """the reference takes torch.nn.RMSNorm or diffusers' RMSNorm (and their subclasses) and raises on anything else (ref :107-119).
    diffusers is not a dependency here, so its class is recognised by NAME — of a class that lives in diffusers' own modules, or of one that
    says it computes x * rsqrt(mean(x^2) + eps) * weight (`svg_rmsnorm_compatible = True`: the duck-typed stand-ins of the tests).  An
    unrelated third-party class that merely happens to be called RMSNorm ((1 + weight) scaling, a bias, ...) is rejected like any other
    module instead of being sent through the kernel."""
    if isinstance(mod, torch.nn.RMSNorm):
        return True
    for c in type(mod).__mro__:
        if c.__name__ == "RMSNorm" and (c.__module__.split(".")[0] == "diffusers" or getattr(c, "svg_rmsnorm_compatible", False)):
            return getattr(mod, "weight", None) is not None
    return False


def _rms_kernel_ok(mod, x) -> bool:
    """the HIP RMSNorm (weight and eps only, like the reference's kernel call :105-120) applies to this module and tensor"""
    w = getattr(mod, "weight", None)
    return bool(w is not None and x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and x.shape[-1] % 8 == 0 and x.shape[-1] <= 8192
                and tuple(w.shape) == (x.shape[-1],) and getattr(mod, "eps", None) is not None and getattr(mod, "bias", None) is None)


class WanAttn_SVGAttn_Processor2_0:
    """Sparse VideoGen 1 for Wan 2.1 (ref: wan/attention.py:77-328)."""

    version = None
    context_length = 0
    num_frame = 0
    frame_size = 0

    first_layers_fp = 0
    first_times_fp = 0

    num_sampled_rows = 32
    sample_mse_max_row = 10000
    device_switch = True    # dense / sparse decision on the device when the timestep is a GPU tensor
    attention_masks = None
    sparsity = 0

    block_mask = None
    temporal_mask_metadata = None
    fused_placement = True
    # opt-in, the PRESCALE_QK trade-off of flex_attention (see Hunyuan's processor; off by default since round 4: parity with the
    # reference's formulation first).  True, self attention only: the HIP RoPE pass folds sm_scale * log2(e) into its (last) rounding
    # of q and the attention core runs its pre-scaled kernels; never for the cross attention and the I2V image branch (torch SDPA)
    prescale_q = False

    def __init__(self, layer_idx):
        self.layer_idx = layer_idx
        self.last_best_mask_idx = None
        self._q_prescaled = False
        self._rope_scaled = False

    @classmethod
    def geometry(cls) -> Geometry:
        return Geometry(cls.context_length, cls.num_frame, cls.frame_size, text_first=False)

    @time_logging_decorator("Level 2 - qkv")
    def get_qkv(self, attn, hidden_states, encoder_hidden_states):
        return attn.to_q(hidden_states), attn.to_k(encoder_hidden_states), attn.to_v(encoder_hidden_states)

    @time_logging_decorator("Level 2 - qk_norm")
    def get_qk_norm(self, attn, query, key):
        # Wan normalises across all heads, before the head split, and the reference does it with its Triton RMSNorm kernel whatever the
        # module is (ref :105-120: `triton_rmsnorm_forward(query, attn.norm_q.weight, attn.norm_q.eps)` — fp32, one rounding), not with
        # the module's forward (diffusers rounds before the weight): same here on the GPU; other tensors take the module
        def norm(mod, x):
            if not _is_rms(mod):
                raise ValueError(f"Unsupported norm type: {type(mod)}")
            # the reference's kernel call takes weight and eps only (:105-120); a module that also carries a bias runs its own forward
            if _rms_kernel_ok(mod, x):
                return triton_rmsnorm_forward(x.contiguous(), mod.weight, mod.eps)
            return mod(x)

        if getattr(attn, "norm_q", None) is not None:
            query = norm(attn.norm_q, query)
        if getattr(attn, "norm_k", None) is not None:
            key = norm(attn.norm_k, key)
        return query, key

    @time_logging_decorator("Level 2 - transpose")
    def get_transpose_qkv(self, attn, query, key, value):
        return tuple(x.unflatten(2, (attn.heads, -1)).transpose(1, 2).contiguous() for x in (query, key, value))

    @time_logging_decorator("Level 2 - rotary_emb")
    def get_rotary_emb(self, query, key, rotary_emb, q_scale: float = 1.0):
        self._rope_scaled = False       # set by the RoPE pass itself: True only when the HIP pass folded q_scale into q
        if rotary_emb is not None:
            flag = []
            query, key = apply_rotary_emb(query, key, rotary_emb, q_scale=q_scale, scaled=flag)
            self._rope_scaled = bool(flag and flag[0])
        return query, key

    fused_prologue = True    # self attention on the GPU: qk_norm + transpose + rotary_emb as ONE pass over q, k, v (svg_rmsnorm_rope_transpose)

    @time_logging_decorator("Level 2 - qk_norm + transpose + rotary_emb (fused)")
    def get_fused_prologue(self, attn, query, key, value, rotary_emb, q_scale: float = 1.0):
        """get_qk_norm -> get_transpose_qkv -> get_rotary_emb in one kernel (bit-identical to the three steps on the HIP path; ref :99-148):
        returns (q, k, v) head-major, or None when the pass does not apply — then the caller runs the three steps."""
        if not self.fused_prologue or rotary_emb is None:
            return None
        nq, nk = getattr(attn, "norm_q", None), getattr(attn, "norm_k", None)
        if nq is None or nk is None or not (_is_rms(nq) and _is_rms(nk)):
            return None          # (an unsupported module raises in get_qk_norm, exactly as before)
        ts = (query, key, value)
        if not all(t.is_cuda and t.dim() == 3 and t.is_contiguous() and t.dtype == query.dtype and t.shape == query.shape for t in ts):
            return None
        if not (_rms_kernel_ok(nq, query) and _rms_kernel_ok(nk, key)) or float(nq.eps) != float(nk.eps):
            return None
        H = attn.heads
        D = query.shape[-1] // H
        if D not in (64, 128) or H * D != query.shape[-1]:
            return None
        fr, fi = rotary_emb if isinstance(rotary_emb, (tuple, list)) else (rotary_emb.real, rotary_emb.imag)
        S = query.shape[1]
        tb = _core._tables(fr, fi, S, D // 2, query.device)
        if tb is None:
            return None
        # weights in their own dtype, as get_qk_norm hands them to the RMSNorm kernel (an fp32 weight is used in fp32 there too)
        qw, kw = (m.weight.detach().to(device=query.device).contiguous() for m in (nq, nk))
        if qw.dtype != kw.dtype or qw.dtype not in (torch.bfloat16, torch.float16, torch.float32):
            return None
        # v: read in place by the attention kernels where they take strides (a view of the projection's output), else transposed in the same pass
        vv = _core.value_in_place(value, H) if q_scale == 1.0 else None
        q, k, v = _core._native.rmsnorm_rope_transpose(query, key, None if vv is not None else value, H, qw, kw, float(nq.eps), 2, tb[0], tb[1],
                                                       0, S, q_scale=q_scale)
        return q, k, (vv if vv is not None else v)

    @time_logging_decorator("Level 2 - output")
    def get_o(self, attn, query, hidden_states, hidden_states_img):
        hidden_states = hidden_states.transpose(1, 2).flatten(2, 3).type_as(query)
        if hidden_states_img is not None:
            hidden_states = hidden_states + hidden_states_img
        return attn.to_out[1](attn.to_out[0](hidden_states))

    def __call__(self, attn, hidden_states: torch.Tensor, encoder_hidden_states: Optional[torch.Tensor] = None,
                 attention_mask: Optional[torch.Tensor] = None, rotary_emb=None, timestep=None):
        cross = encoder_hidden_states is not None
        if timestep is None and not cross:
            from ..context import current_timestep

            timestep = current_timestep()
        encoder_hidden_states_img = None
        if getattr(attn, "add_k_proj", None) is not None and encoder_hidden_states is not None:
            encoder_hidden_states_img = encoder_hidden_states[:, :257]
            encoder_hidden_states = encoder_hidden_states[:, 257:]
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        query, key, value = self.get_qkv(attn, hidden_states, encoder_hidden_states)
        q_scale = 1.0
        D_head = query.shape[-1] // attn.heads
        if (self.prescale_q and not cross and timestep is not None and rotary_emb is not None and query.is_cuda
                and query.dtype in (torch.bfloat16, torch.float16) and D_head in (64, 128)):
            q_scale = _core._native.softmax_q_scale(D_head)     # a request: the RoPE pass reports whether it folded it in
        fused = None if cross else self.get_fused_prologue(attn, query, key, value, rotary_emb, q_scale)
        if fused is not None:
            query, key, value = fused
            self._rope_scaled = q_scale != 1.0
        else:
            query, key = self.get_qk_norm(attn, query, key)
            if cross and rotary_emb is None and query.is_cuda:
                # cross attention (512 text keys, torch SDPA below): head views instead of three contiguous copies — the q copy alone is
                # 2 x 774 MB of traffic per layer at Wan 2.1 720p; SDPA takes strided operands and returns q's layout
                query, key, value = (x.unflatten(2, (attn.heads, -1)).transpose(1, 2) for x in (query, key, value))
            else:
                query, key, value = self.get_transpose_qkv(attn, query, key, value)
            query, key = self.get_rotary_emb(query, key, rotary_emb, q_scale=q_scale)
        hidden_states_img = None
        if encoder_hidden_states_img is not None:  # I2V: CLIP image tokens, small dense cross attention (ref :174-188)
            key_img = attn.norm_added_k(attn.add_k_proj(encoder_hidden_states_img))
            value_img = attn.add_v_proj(encoder_hidden_states_img)
            key_img = key_img.unflatten(2, (attn.heads, -1)).transpose(1, 2)
            value_img = value_img.unflatten(2, (attn.heads, -1)).transpose(1, 2)
            hidden_states_img = F.scaled_dot_product_attention(query, key_img, value_img, attn_mask=None, dropout_p=0.0,
                                                               is_causal=False)
            hidden_states_img = hidden_states_img.transpose(1, 2).flatten(2, 3).type_as(query)
        if timestep is None or cross:  # cross attention in Wan (ref :198-201)
            hidden_states = F.scaled_dot_product_attention(query, key, value, attn_mask=attention_mask, dropout_p=0.0,
                                                           is_causal=False)
        else:
            self._q_prescaled = self._rope_scaled
            try:
                hidden_states = self.attention_core_logic(query, key, value, timestep)
            finally:
                self._q_prescaled = False
        return self.get_o(attn, query, hidden_states, hidden_states_img)

    @time_logging_decorator("Level 3 - sample mse")
    def sample_mse(self, query, key, value):
        geo = self.geometry()
        return _core.sample_mse(query, key, value, geo, profile_desc(geo.context_length, geo.num_frame, geo.frame_size),
                                self.num_sampled_rows, min(self.sample_mse_max_row, query.shape[2]), q_prescaled=self._q_prescaled)

    @time_logging_decorator("Level 3 - Dense Flash Attention")
    def flash_attention(self, query, key, value):
        return _core.dense_attention(query, key, value, q_prescaled=self._q_prescaled)

    @time_logging_decorator("Level 2 - attention core logic")
    def attention_core_logic(self, query, key, value, timestep):
        cfg, num_heads, seq_len, dim = query.size()
        geo = self.geometry()
        assert seq_len == geo.seq_len, (
            f"Query Shape: {seq_len} is not equivalent to {geo.context_length} + {geo.num_frame} * {geo.frame_size}")
        dense_flag = None   # (device-side dense / sparse switch, see the Hunyuan processor)
        if self.device_switch and _core.attention_dtype() == "bf16" and self.fused_placement and self.layer_idx >= self.first_layers_fp and self.block_mask is not None \
                and query.is_cuda:
            dense_flag = _core.dense_flag_on_device(timestep, self.first_times_fp)
        if dense_flag is None and _core.is_full_attention(self.layer_idx, timestep, self.first_layers_fp, self.first_times_fp):
            return self.flash_attention(query, key, value).reshape(cfg, num_heads, seq_len, dim)
        if self.block_mask is None:
            raise RuntimeError("WanAttn_SVGAttn_Processor2_0.block_mask is not set: call replace_wan_attention first")
        prof = profile_desc(geo.context_length, geo.num_frame, geo.frame_size)
        if dense_flag is not None:
            out, best = _core.svg1_attention_device_switch(query, key, value, geo, self.block_mask, dense_mask(seq_len), prof,
                                                           self.num_sampled_rows, min(self.sample_mse_max_row, seq_len), dense_flag,
                                                           q_prescaled=self._q_prescaled)
            self.last_best_mask_idx = best
            return out.reshape(cfg, num_heads, seq_len, dim)
        out, best = _core.svg1_sparse_attention(query, key, value, geo, self.block_mask, prof, self.num_sampled_rows,
                                                min(self.sample_mse_max_row, seq_len), fused=self.fused_placement,
                                                q_prescaled=self._q_prescaled)
        self.last_best_mask_idx = best
        return out.reshape(cfg, num_heads, seq_len, dim)


def prepare_flexattention(cfg_size, num_head, head_dim, dtype, device, context_length, prompt_length, num_frame, frame_size,
                          diag_width=1, multiplier=2):
    """ref: wan/attention.py:331-355 — returns the band-mask descriptor instead of a compiled flex BlockMask."""
    assert diag_width == multiplier, f"{diag_width} is not equivalent to {multiplier}"
    return generate_temporal_head_mask_mod(context_length, prompt_length, num_frame, frame_size, mul=multiplier)


def prepare_flashinfer_attention(cfg_size, num_head, head_dim, dtype, device, context_length, prompt_length, num_frame, frame_size,
                                 diag_width=1, multiplier=2):
    """ref: wan/attention.py:358-375 — the BSR metadata (row pointer, column indices, block size) of the temporal mask for the
    uniform-block alternative backend (`flashinfer_sparse_attn_forward`, svg/models/wan/utils.py)"""
    from .utils import gen_temporal_mask

    assert diag_width == multiplier, f"{diag_width} is not equivalent to {multiplier}"
    return gen_temporal_mask(num_frame, frame_size, multiplier)


class WanAttn_SAPAttn_Processor(WanAttn_SVGAttn_Processor2_0):
    """Sparse VideoGen 2 for Wan 2.1 (ref: wan/attention.py:379-559).  Centroids are kept per processor instance (one
    per layer), as in the reference."""

    num_layers = 0
    num_q_centroids = 0
    num_k_centroids = 0
    top_p_kmeans = 0
    min_kc_ratio = 0

    kmeans_iter_init = 0
    kmeans_iter_step = 0
    zero_step_kmeans_init = False

    logging_file = None
    prescale_q = False   # k-means and the block map work on the plain q

    def __init__(self, layer_idx):
        super().__init__(layer_idx)
        self.centroid_store = CentroidStore()

    def reset_state(self):
        self.centroid_store.clear()

    @time_logging_decorator("Level 2 - attention core logic")
    def attention_core_logic(self, query, key, value, timestep):
        cfg, num_heads, seq_len, dim = query.size()
        assert cfg == 1, "Batch size must be 1 for kmeans block sparse attention"
        geo = self.geometry()
        assert seq_len == geo.seq_len, (
            f"Query Shape: {seq_len} is not equivalent to {geo.context_length} + {geo.num_frame} * {geo.frame_size}")
        if _core.is_full_attention(self.layer_idx, timestep, self.first_layers_fp, self.first_times_fp):
            if self.zero_step_kmeans_init and query.is_cuda:
                V = geo.video_length
                _core.kmeans_clustering(self.centroid_store, self.layer_idx, query[:, :, :V].contiguous(),
                                        key[:, :, :V].contiguous(), self.num_q_centroids, self.num_k_centroids,
                                        self.kmeans_iter_init, self.kmeans_iter_step)
            return self.flash_attention(query, key, value).reshape(cfg, num_heads, seq_len, dim)
        out = _core.svg2_sparse_attention(query, key, value, geo, self.centroid_store, self.layer_idx, self.num_q_centroids,
                                          self.num_k_centroids, self.top_p_kmeans, self.min_kc_ratio, self.kmeans_iter_init,
                                          self.kmeans_iter_step, prompt_length=0, logging_file=self.logging_file,
                                          timestep=timestep)
        return out.reshape(cfg, num_heads, seq_len, dim)
