"""Layout transformation for wan — same function names as the reference module svg/models/wan/placement.py
(ref: hyvideo/placement.py:124-153,360-387; cog/placement.py text-first variant).  The Triton kernels of the reference
are replaced by the HIP kernel `svg_head_placement` (csrc/placement.hip)."""
from .._placement_common import hidden_states_placement as _hsp
from .._placement_common import sparse_head_placement as _shp
from .._placement_common import torch_placement as _tp

TEXT_FIRST = False


def wan_sparse_head_placement(query, key, value, query_out, key_out, value_out, best_mask_idx, context_length, num_frame,
                              frame_size):
    _shp(query, key, value, query_out, key_out, value_out, best_mask_idx, context_length, num_frame, frame_size, TEXT_FIRST)


def wan_hidden_states_placement(hidden_states, hidden_states_out, best_mask_idx, context_length, num_frame, frame_size):
    return _hsp(hidden_states, hidden_states_out, best_mask_idx, context_length, num_frame, frame_size, TEXT_FIRST)


def ref_wan_sparse_head_placement(query, key, value, best_mask_idx, context_length, num_frame, frame_size):
    return tuple(_tp(x, best_mask_idx, context_length, num_frame, frame_size, TEXT_FIRST, False) for x in (query, key, value))


def ref_wan_hidden_states_placement(hidden_states, output_hidden_states, best_mask_idx, context_length, num_frame,
                                    frame_size):
    output_hidden_states.copy_(_tp(hidden_states, best_mask_idx, context_length, num_frame, frame_size, TEXT_FIRST, True))
    return output_hidden_states


def wan_token_reorder_to_token_major(tensor, fix_len, reorder_len, reorder_num_frame, frame_size):
    """ref: wan/placement.py:6-17 — frame major -> token major, in place"""
    from .._placement_common import token_reorder

    return token_reorder(tensor, fix_len, reorder_len, reorder_num_frame, frame_size, TEXT_FIRST, True)


def wan_token_reorder_to_frame_major(tensor, fix_len, reorder_len, reorder_num_frame, frame_size):
    """ref: wan/placement.py:20-31 — token major -> frame major, in place"""
    from .._placement_common import token_reorder

    return token_reorder(tensor, fix_len, reorder_len, reorder_num_frame, frame_size, TEXT_FIRST, False)
