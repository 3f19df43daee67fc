"""Cosmos helpers of the SVG1 path — the reference module svg/models/cosmos/utils.py is identical to svg/models/wan/utils.py
(context_length = 0, 2-frame profiling band, first-frame sink), so the analytic descriptors are shared."""
from ..wan.utils import (  # noqa: F401
    flashinfer_sparse_attn_forward,
    gen_temporal_mask,
    generate_dense_mask_mod,
    generate_temporal_head_mask_mod,
    get_attention_mask,
    get_factor,
    profile_desc,
    sparsity_to_width,
)
