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

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

# ---------------------------------------------------------------------------------------------------------------
# opt-in switch for the processors (svg/models/_core.py): svg.distributed.enable(group) makes every SVG1 / SVG2 layer-call run
# this rank's heads only and all-gather the result; nothing changes for a process that never calls it.
# ---------------------------------------------------------------------------------------------------------------
_STATE: Dict[str, object] = {"enabled": False, "group": None}


def enable(group=None) -> None:
    """Shard every attention layer-call of the installed processors over the ranks of `group` (default: the world group).
    The processors keep receiving full [cfg, H, S, D] tensors (replicated activations); each rank computes
    `shard_heads(H, rank, world)` and the outputs are all-gathered, so the caller sees the same tensors as on one GPU —
    bitwise, because nothing in the path mixes heads (the k-means stopping rule, a maximum over all heads, is all-reduced)."""
    assert dist.is_available() and dist.is_initialized(), "svg.distributed.enable: call torch.distributed.init_process_group first"
    _STATE["enabled"], _STATE["group"] = True, group


def disable() -> None:
    _STATE["enabled"], _STATE["group"] = False, None


def active() -> bool:
    return bool(_STATE["enabled"]) and dist.is_initialized() and dist.get_world_size(_STATE["group"]) > 1


def current_group():
    return _STATE["group"]


def all_reduce_max_(t: torch.Tensor) -> torch.Tensor:
    """in-place MAX over the ranks of the enabled group (identity when sharding is off) — the k-means stopping rule's
    `center_shift` is a maximum over ALL heads (ref: svg/kmeans_utils.py:721-723)"""
    if active():
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_STATE["group"])
    return t


def run_sharded(fn: Callable[..., Tuple[torch.Tensor, ...]], head_tensors: Sequence[torch.Tensor], group=None):
    """fn(*[t[:, my heads] for t in head_tensors]) -> tensor or tuple of tensors with the head dimension at dim 1; every result is
    all-gathered back to all heads (global order).  Ragged splits (H % world != 0) are allowed."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    H = head_tensors[0].shape[1]
    mine = shard_heads(H, rank, world)
    if not mine:
        raise ValueError(f"svg.distributed: rank {rank} of {world} owns no head of {H} — head sharding needs num_heads >= world size "
                         f"(use a smaller process group for this model)")
    sl = slice(mine[0], mine[-1] + 1)
    res = fn(*[t[:, sl].contiguous() for t in head_tensors])
    single = not isinstance(res, (tuple, list))
    outs = []
    for r in ([res] if single else res):
        extra = r.shape[2:]
        r4 = r.reshape(r.shape[0], r.shape[1], -1, 1) if r.dim() != 4 else r
        g = all_gather_heads(r4, H, group)
        outs.append(g.reshape(g.shape[0], H, *extra))
    return outs[0] if single else tuple(outs)


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


# ---------------------------------------------------------------------------------------------------------------
# token shards <-> head shards (the two exchanges either side of the attention when activations live token-sharded)
# ---------------------------------------------------------------------------------------------------------------
def token_range(num_tokens: int, rank: int, world: int, unit: int = 1) -> Tuple[int, int]:
    """Tokens [a, b) of the sequence that live on `rank` between attention calls: contiguous, in whole `unit`s, a trailing partial
    unit goes to the last rank.  Nothing on the token-sharded side (norms, GEMMs, prologue with absolute RoPE positions, glue) needs
    frames kept together, so the unit is a tile of tokens, not a frame: HunyuanVideo 720p, S = 119056 over 8 ranks with unit = 128 gives
    14976 / 14848 (+16) tokens, largest / mean 1.006; unit = tokens per frame (33 frames -> 5, 4, 4, ...) gives 1.21 and costs the
    GEMM / glue side of an 8-GPU step about 8 %."""
    n_units = num_tokens // unit
    base, rem = divmod(n_units, world)
    a = (rank * base + min(rank, rem)) * unit
    b = a + (base + (1 if rank < rem else 0)) * unit
    if rank == world - 1:
        b = num_tokens
    return a, b


def _owner_lists(num_heads: int, world: int, head_lists) -> List[List[int]]:
    return [list(h) for h in head_lists] if head_lists is not None else [shard_heads(num_heads, r, world) for r in range(world)]


def tokens_to_heads(x_local: torch.Tensor, num_tokens: int, group=None, unit: int = 1, head_lists=None, presorted: bool = False,
                    out: Optional[torch.Tensor] = None, recv: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Inbound exchange of a layer-call.  x_local [H, S_r, ...]: ALL heads of this rank's tokens `token_range(num_tokens, rank,
    world, unit)` (head-major, as the fused prologue writes it) -> [H_local, num_tokens, ...]: this rank's heads of ALL tokens.
    One all_to_all_single (every peer sends to every peer directly: the 7 xGMI links of a rank run concurrently), then one
    strided copy per source rank into the sequence dimension.  Received per rank: (world - 1) / world of H_local * S * E bytes.
    head_lists: per-rank head ids when ownership is not `shard_heads` (bench.py: chunked_head_layout); presorted=True says
    x_local's heads are already ordered owner by owner."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    H = x_local.shape[0]
    rest = tuple(x_local.shape[2:])
    E = 1
    for r_ in rest:
        E *= r_
    owners = _owner_lists(H, world, head_lists)
    tr = [token_range(num_tokens, r, world, unit) for r in range(world)]
    S_me = tr[rank][1] - tr[rank][0]
    assert x_local.shape[1] == S_me, f"rank {rank} holds {x_local.shape[1]} tokens, token_range says {S_me}"
    Hl = len(owners[rank])
    order = [h for o in owners for h in o]
    if not presorted and order != list(range(H)):
        x_local = x_local.index_select(0, torch.tensor(order, device=x_local.device))
    send = x_local.contiguous().view(-1)   # (no copy for the head-major tensor the fused prologue writes)
    # recv: optional caller-owned staging buffer (a step that runs this exchange 180 times allocates nothing: ExchangeBuffers)
    if recv is None:
        recv = torch.empty(Hl * num_tokens * E, dtype=x_local.dtype, device=x_local.device)
    else:
        assert recv.dtype == x_local.dtype and recv.numel() >= Hl * num_tokens * E
        recv = recv.view(-1)[: Hl * num_tokens * E]
    dist.all_to_all_single(recv, send, output_split_sizes=[Hl * (b - a) * E for a, b in tr],
                           input_split_sizes=[len(o) * S_me * E for o in owners], group=group)
    if out is None:
        out = torch.empty((Hl, num_tokens) + rest, dtype=x_local.dtype, device=x_local.device)
    off = 0
    for a, b in tr:
        n = Hl * (b - a) * E
        out[:, a:b].copy_(recv[off:off + n].view((Hl, b - a) + rest))
        off += n
    return out


def heads_to_tokens(o_local: torch.Tensor, num_heads: int, group=None, unit: int = 1, head_lists=None,
                    send: Optional[torch.Tensor] = None, recv: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Outbound exchange, the inverse of tokens_to_heads: o_local [H_local, S, ...] (this rank's heads, all tokens) ->
    [H, S_r, ...] (all heads of this rank's tokens; heads in global order) — what a token-sharded `to_out` consumes, world
    times fewer received bytes than all-gathering the heads."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    S = o_local.shape[1]
    rest = tuple(o_local.shape[2:])
    E = 1
    for r_ in rest:
        E *= r_
    owners = _owner_lists(num_heads, world, head_lists)
    tr = [token_range(S, r, world, unit) for r in range(world)]
    Hl = len(owners[rank])
    assert o_local.shape[0] == Hl
    S_me = tr[rank][1] - tr[rank][0]
    if send is None:
        send = torch.cat([o_local[:, a:b].reshape(-1) for a, b in tr])
    else:   # caller-owned staging (see ExchangeBuffers): one strided copy per destination rank, no allocation
        assert send.dtype == o_local.dtype and send.numel() >= Hl * S * E
        send = send.view(-1)[: Hl * S * E]
        off = 0
        for a, b in tr:
            n = Hl * (b - a) * E
            send[off:off + n].view((Hl, b - a) + rest).copy_(o_local[:, a:b])
            off += n
    if recv is None:
        recv = torch.empty(num_heads * S_me * E, dtype=o_local.dtype, device=o_local.device)
    else:
        assert recv.dtype == o_local.dtype and recv.numel() >= num_heads * S_me * E
        recv = recv.view(-1)[: num_heads * S_me * E]
    dist.all_to_all_single(recv, send, output_split_sizes=[len(o) * S_me * E for o in owners],
                           input_split_sizes=[Hl * (b - a) * E for a, b in tr], group=group)
    out = recv.view((num_heads, S_me) + rest)
    order = [h for o in owners for h in o]
    if order != list(range(num_heads)):
        inv = torch.empty(num_heads, dtype=torch.long)
        inv[torch.tensor(order)] = torch.arange(num_heads)
        out = out.index_select(0, inv.to(out.device))
    return out


class ExchangeBuffers:
    """Staging of the two exchanges either side of a head-sharded attention inside a token-sharded stack, allocated ONCE for a
    (heads, tokens, head_dim, dtype, group) and reused by every layer of every step: tokens_to_heads receives into `recv_in`
    and scatters into one of three persistent [H_local, S, D] tensors (q, k, v); heads_to_tokens packs into `send_out` and
    receives into `recv_out` ([H, S_r, D], returned as a view — consume it before the next call)."""

    def __init__(self, num_heads: int, num_tokens: int, head_dim: int, dtype, device, group=None, unit: int = 1):
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        assert num_heads % world == 0, "ExchangeBuffers: equal head shards (use run_sharded for ragged splits)"
        self.group, self.unit, self.H, self.S, self.D = group, unit, num_heads, num_tokens, head_dim
        self.Hl = num_heads // world
        a, b = token_range(num_tokens, rank, world, unit)
        self.S_me = b - a
        mk = lambda *shape: torch.empty(shape, dtype=dtype, device=device)   # noqa: E731
        self.qkv = [mk(self.Hl, num_tokens, head_dim) for _ in range(3)]
        self.recv_in = mk(self.Hl * num_tokens * head_dim)
        self.send_out = mk(self.Hl * num_tokens * head_dim)
        self.recv_out = mk(num_heads * self.S_me * head_dim)
        self.bytes_in = 0     # received by this rank from OTHER ranks, accumulated (reset by the caller per step)
        self.bytes_out = 0
        self._esz = self.recv_in.element_size()

    def tokens_to_heads(self, x_local: torch.Tensor, which: int) -> torch.Tensor:
        """x_local [H, S_r, D] (all heads of my tokens) -> persistent [H_local, S, D] buffer number `which` (0 q, 1 k, 2 v)"""
        out = tokens_to_heads(x_local, self.S, self.group, self.unit, out=self.qkv[which], recv=self.recv_in)
        self.bytes_in += self.Hl * (self.S - self.S_me) * self.D * self._esz
        return out

    def heads_to_tokens(self, o_local: torch.Tensor) -> torch.Tensor:
        """o_local [H_local, S, D] -> [H, S_r, D] (a view of the receive buffer)"""
        out = heads_to_tokens(o_local, self.H, self.group, self.unit, send=self.send_out, recv=self.recv_out)
        self.bytes_out += (self.H - self.Hl) * self.S_me * self.D * self._esz
        return out


_SIDE_STREAMS: Dict[object, list] = {}


def side_streams(device) -> list:
    """two side streams per device, created once (a torch.cuda.Stream per call would leak HIP streams over a long run)"""
    key = (device.type, device.index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = [torch.cuda.Stream(device=device) for _ in range(2)]
    return _SIDE_STREAMS[key]


def chunked_head_layout(num_heads: int, rank: int, world: int, max_chunks: int = 3):
    """Head ownership for overlapping the output all-gather with compute: heads are taken in super-groups of world * n
    consecutive heads, n per rank, so the all-gather of local chunk c (n heads per rank) is the contiguous, naturally ordered
    slice [c * world * n, (c + 1) * world * n) of the full [H, ...] output.  Returns (n_chunks, n_per_chunk, my_heads)."""
    assert num_heads % world == 0, (f"chunked_head_layout: {num_heads} heads do not split evenly over {world} ranks — the "
                                    "overlapped scheme needs equal chunks; use sharded_attention / run_sharded (ragged splits) instead")
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
    side = side_streams(q.device) if cuda and n_chunks > 1 else None
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
