"""Token permutation operators — same module path and function names as the reference (svg/kernels/triton/permute.py),
implemented with the HIP kernels of csrc/permute.hip instead of Triton.

`sorted_indices` are produced by a stable counting sort on device (svg_argsort_labels), i.e. they equal
torch.argsort(labels, stable=True): the within-cluster order the reference leaves unspecified is fixed to 'by index'."""
from __future__ import annotations

from typing import Optional

import torch

from ... import _native
from ...timer import time_logging_decorator


@time_logging_decorator("Level 4 - permute tensor by labels triton")
def permute_tensor_by_labels_triton(tensor: torch.Tensor, labels: Optional[torch.Tensor], dim: int, *,
                                    sorted_indices: Optional[torch.Tensor] = None, num_clusters: Optional[int] = None):
    """ref: svg/kernels/triton/permute.py:82-128.  tensor [B,H,S,D]; labels [B*H,S] (or [B,H,S]) -> (permuted, idx int32)"""
    assert dim == 2, "permute_tensor_by_labels currently only supports dim==2 (sequence dimension)"
    assert tensor.dim() == 4, "Expected tensor shape [B,H,S,D]"
    assert tensor.is_cuda, "permute_tensor_by_labels requires GPU tensors"
    B, H, S, D = tensor.shape
    if sorted_indices is not None:
        sorted_indices = sorted_indices.to(torch.int32).reshape(B * H, S).contiguous()
    else:
        assert labels is not None, "Either `labels` or `sorted_indices` must be provided."
        lab = labels.to(tensor.device).reshape(B * H, S).to(torch.int32).contiguous()
        K = int(num_clusters) if num_clusters is not None else int(lab.max().item()) + 1
        sorted_indices, _ = _native.argsort_labels(lab, K)
    out = _native.permute_rows(tensor.reshape(B * H, S, D).contiguous(), sorted_indices)
    return out.reshape(B, H, S, D), sorted_indices


@time_logging_decorator("Level 4 - apply inverse permutation triton")
def apply_inverse_permutation_triton(permuted_tensor: torch.Tensor, sorted_indices: torch.Tensor, dim: int):
    """ref: svg/kernels/triton/permute.py:131-170: out[bh, idx[bh, s]] = in[bh, s]"""
    assert dim == 2, "apply_inverse_permutation currently only supports dim==2"
    assert permuted_tensor.dim() == 4, "Expected tensor shape [B,H,S,D]"
    assert permuted_tensor.is_cuda, "apply_inverse_permutation requires GPU tensors"
    B, H, S, D = permuted_tensor.shape
    idx = sorted_indices.to(torch.int32).reshape(B * H, S).contiguous()
    out = _native.permute_rows(permuted_tensor.reshape(B * H, S, D).contiguous(), idx, inverse=True)
    return out.reshape(B, H, S, D)
