"""Head-sharded execution of one attention layer-call over the GPUs of a node (one process per GPU, RCCL over xGMI).

The hot path has no cross-head dependency at any stage (profiler, layout transformation, k-means, block map, attention
are all per head — SURVEY.md §8e), so a layer-call splits by heads with NO collective inside the path.  The only
exchange is the one the next operator needs: `to_out` consumes all heads of a token, so the per-rank outputs
[1, H/N, S, D] are all-gathered once per layer-call (N-1 peers x H/N x S x D x 2 B each; at Hunyuan 720p and N = 8 that is
7 x 91 MB inbound per GPU, spread over the 7 xGMI links of the full mesh).  The reference has no multi-GPU version of
this path (its deprecated forks use xfuser ring/Ulysses with dense attention); parity is defined as equality with the
single-GPU result, which is bitwise because per-head arithmetic does not change.
"""
from __future__ import annotations

from typing import Callable, List, Sequence

import torch
import torch.distributed as dist


def shard_heads(num_heads: int, rank: int, world: int) -> List[int]:
    """Heads owned by `rank`: a contiguous block, sizes differing by at most one (24 heads / 8 GPUs -> 3 each)."""
    base, rem = divmod(num_heads, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def head_counts(num_heads: int, world: int) -> List[int]:
    return [len(shard_heads(num_heads, r, world)) for r in range(world)]


def all_gather_heads(o_local: torch.Tensor, num_heads: int, group=None) -> torch.Tensor:
    """[cfg, H_local, S, D] on every rank -> [cfg, H, S, D] on every rank (heads in global order)."""
    world = dist.get_world_size(group)
    counts = head_counts(num_heads, world)
    cfg, _, S, D = o_local.shape
    if len(set(counts)) == 1:
        out = torch.empty((world * cfg, counts[0], S, D), dtype=o_local.dtype, device=o_local.device)
        dist.all_gather_into_tensor(out, o_local.contiguous(), group=group)  # concatenated along dim 0
        return out.view(world, cfg, counts[0], S, D).permute(1, 0, 2, 3, 4).reshape(cfg, num_heads, S, D)
    # ragged split: pad every shard to the largest one
    mx = max(counts)
    pad = torch.zeros((cfg, mx, S, D), dtype=o_local.dtype, device=o_local.device)
    pad[:, : o_local.shape[1]] = o_local
    out = torch.empty((world * cfg, mx, S, D), dtype=o_local.dtype, device=o_local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    out = out.view(world, cfg, mx, S, D)
    return torch.cat([out[r, :, : counts[r]] for r in range(world)], dim=1)


def sharded_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, attn_fn: Callable[..., torch.Tensor],
                      per_head_args: Sequence[torch.Tensor] = (), group=None) -> torch.Tensor:
    """Run `attn_fn(q_h, k_h, v_h, *args_h)` on this rank's heads of full [cfg, H, S, D] tensors and all-gather.
    per_head_args are tensors with a head dimension at dim 1 (e.g. best_mask_idx [cfg, H]) sliced the same way."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    H = q.shape[1]
    mine = shard_heads(H, rank, world)
    sl = slice(mine[0], mine[-1] + 1) if mine else slice(0, 0)
    o_local = attn_fn(q[:, sl].contiguous(), k[:, sl].contiguous(), v[:, sl].contiguous(), *[a[:, sl] for a in per_head_args])
    return all_gather_heads(o_local, H, group)


def chunked_head_layout(num_heads: int, rank: int, world: int, max_chunks: int = 3):
    """Head ownership for overlapping the output all-gather with compute: heads are taken in super-groups of world * n
    consecutive heads, n per rank, so the all-gather of local chunk c (n heads per rank) is the contiguous, naturally ordered
    slice [c * world * n, (c + 1) * world * n) of the full [H, ...] output.  Returns (n_chunks, n_per_chunk, my_heads)."""
    assert num_heads % world == 0, f"{num_heads} heads do not split over {world} ranks"
    local = num_heads // world
    n_chunks = 1 if world == 1 else max(c for c in range(max_chunks, 0, -1) if local % c == 0)
    n_per = local // n_chunks
    mine = [c * world * n_per + rank * n_per + i for c in range(n_chunks) for i in range(n_per)]
    return n_chunks, n_per, mine


def gather_chunk(full: torch.Tensor, o_chunk: torch.Tensor, chunk: int, n_per: int, world: int, group=None, async_op: bool = True):
    """All-gather local chunk `chunk` ([n_per, S, D] on every rank) into its slice of `full` [H, S, D]; returns the work handle."""
    dst = full[chunk * world * n_per:(chunk + 1) * world * n_per]
    return dist.all_gather_into_tensor(dst, o_chunk.contiguous(), group=group, async_op=async_op)


def overlapped_sharded_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, attn_fn: Callable[..., torch.Tensor],
                                 per_head_args: Sequence[torch.Tensor] = (), group=None, max_chunks: int = 3) -> torch.Tensor:
    """`sharded_attention` with the exchange hidden behind compute (what `bench.py --gpus N` times): this rank's heads are
    processed in up to `max_chunks` chunks (`chunked_head_layout`); the all-gather of chunk c runs on the communicator's stream
    while chunk c + 1 computes, and on CUDA the chunk launches alternate between two streams so that the next chunk's workgroups
    fill the idle CUs of a launch's last round.  Needs num_heads % world == 0 and cfg == 1 (one contiguous [H, S, D] output).
    attn_fn(q_c, k_c, v_c, *args_c) -> [1, n, S, D] for tensors sliced to the chunk's n heads."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    cfg, H, S, D = q.shape
    assert cfg == 1, "overlapped_sharded_attention: cfg must be 1 (use sharded_attention otherwise)"
    n_chunks, n_per, _ = chunked_head_layout(H, rank, world, max_chunks)
    full = torch.empty((H, S, D), dtype=q.dtype, device=q.device)
    cuda = q.is_cuda
    main = torch.cuda.current_stream() if cuda else None
    side = [torch.cuda.Stream(device=q.device) for _ in range(2)] if cuda and n_chunks > 1 else None
    works = []
    for c in range(n_chunks):
        h0 = c * world * n_per + rank * n_per
        sl = slice(h0, h0 + n_per)
        st = side[c % 2] if side else main
        if side:
            st.wait_stream(main)
        ctx = torch.cuda.stream(st) if cuda else _NullCtx()
        with ctx:
            o_c = attn_fn(q[:, sl].contiguous(), k[:, sl].contiguous(), v[:, sl].contiguous(), *[a[:, sl] for a in per_head_args])
            works.append((gather_chunk(full, o_c[0], c, n_per, world, group), o_c))   # keep o_c alive until its gather is done
    if side:
        for st in side:
            main.wait_stream(st)
    for w, _ in works:
        w.wait()
    return full.unsqueeze(0)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False
