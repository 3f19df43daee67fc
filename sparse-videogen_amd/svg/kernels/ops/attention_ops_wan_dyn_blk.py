"""Variable-block sparse attention as an operator — same module path and function as the reference's
svg/kernels/ops/attention_ops_wan_dyn_blk.py (flashinfer.sparse.VariableBlockSparseAttentionWrapper plan + run), on
svg_varblock_attention.  The reference's own test (svg/kernels/test/test_sparse_attn_dyn_blk_wan.py) calls exactly this function."""
from __future__ import annotations

import torch

from ... import _native


def _test_variable_block_sparse_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, num_qo_heads: int, num_kv_heads: int,
                                          head_dim: int, block_mask_map: torch.Tensor, block_row_sz: torch.Tensor,
                                          block_col_sz: torch.Tensor) -> torch.Tensor:
    """q [num_qo_heads, S, D], k / v [num_kv_heads, S, D] (HND); block_mask_map bool [num_kv_heads, MB, NB]; block_row_sz [num_kv_heads, MB],
    block_col_sz [num_kv_heads, NB] -> o [num_kv_heads, group, S, D] (ref :8-43)"""
    assert torch.all(block_col_sz.sum(dim=1) == block_col_sz.sum(dim=1)[0])
    assert torch.all(block_row_sz.sum(dim=1) == block_row_sz.sum(dim=1)[0])
    assert q.shape == (num_qo_heads, q.shape[1], head_dim) and k.shape[0] == num_kv_heads
    dev = q.device
    o = _native.varblock_attention(q.contiguous(), k.contiguous(), v.contiguous(), block_mask_map.to(dev).contiguous(),
                                   block_row_sz.to(device=dev, dtype=torch.int32).contiguous(),
                                   block_col_sz.to(device=dev, dtype=torch.int32).contiguous())
    return o.reshape(num_kv_heads, -1, *o.shape[-2:])
