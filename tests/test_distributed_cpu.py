"""world_size-2 gloo test of the head-sharded layer-call (svg/distributed.py): every rank computes its heads, one
all-gather rebuilds [cfg, H, S, D]; the result must be bitwise equal to the single-process result.  The per-head
attention function here is torch SDPA (CPU) — what is under test is the sharding / exchange, which is device-agnostic."""
import os
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, H, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
    from svg.distributed import shard_heads, sharded_attention

    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    q, k, v = (torch.randn(2, H, 96, 32) for _ in range(3))
    flag = torch.arange(H)[None].expand(2, H) % 2

    def attn(qh, kh, vh, fl):
        o = torch.nn.functional.scaled_dot_product_attention(qh, kh, vh)
        return o * (1 + fl[:, :, None, None].float())  # a per-head argument must follow its head

    o = sharded_attention(q, k, v, attn, per_head_args=(flag,))
    ref = attn(q, k, v, flag)
    ok = torch.equal(o, ref) and sum(len(shard_heads(H, r, world)) for r in range(world)) == H
    ret[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("H", [4, 5])  # even and ragged head split
def test_head_sharded_layer_call_gloo(H):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000) + H
    mp.spawn(_worker, args=(world, port, H, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_shard_heads_partition():
    sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
    from svg.distributed import head_counts, shard_heads

    assert head_counts(24, 8) == [3] * 8 and head_counts(40, 8) == [5] * 8 and head_counts(24, 5) == [5, 5, 5, 5, 4]
    assert [h for r in range(5) for h in shard_heads(24, r, 5)] == list(range(24))


def _worker_chunked(rank, world, port, H, ret):
    """the bench's N > 1 scheme: per-chunk attention + asynchronous all-gather into contiguous slices of the full output"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
    from svg.distributed import chunked_head_layout, gather_chunk

    dist.init_process_group("gloo", rank=rank, world_size=world)
    S, D = 40, 8
    n_chunks, n_per, mine = chunked_head_layout(H, rank, world)
    torch.manual_seed(0)
    q_all = torch.randn(H, S, D)                      # every rank derives the same global tensor, then keeps its heads
    fn = lambda x: torch.tanh(x) * 2.0                # noqa: E731  stand-in for the per-head attention
    o = fn(q_all[mine])
    full = torch.empty(H, S, D)
    works = [gather_chunk(full, o[c * n_per:(c + 1) * n_per], c, n_per, world) for c in range(n_chunks)]
    for w in works:
        w.wait()
    owners = sorted(h for r in range(world) for h in chunked_head_layout(H, r, world)[2])
    ret[rank] = bool(torch.equal(full, fn(q_all)) and owners == list(range(H)) and len(mine) == H // world)
    dist.destroy_process_group()


@pytest.mark.parametrize("H", [24, 8, 6])   # 24 heads / 2 ranks -> 3 chunks of 4; 8 -> 2 chunks of 2; 6 -> 3 chunks of 1
def test_chunked_overlapped_gather_gloo(H):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + (os.getpid() % 2000) + H
    mp.spawn(_worker_chunked, args=(world, port, H, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_chunked_head_layout_partitions():
    sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
    from svg.distributed import chunked_head_layout

    for H, world in ((24, 8), (24, 4), (24, 2), (24, 1), (40, 8), (48, 8)):
        seen = []
        for r in range(world):
            n_chunks, n_per, mine = chunked_head_layout(H, r, world)
            assert n_chunks * n_per == H // world
            seen += mine
        assert sorted(seen) == list(range(H))
    assert chunked_head_layout(24, 3, 8) == (3, 1, [3, 11, 19])
    assert chunked_head_layout(24, 1, 2) == (3, 4, [4, 5, 6, 7, 12, 13, 14, 15, 20, 21, 22, 23])


def _worker_overlapped(rank, world, port, H, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
    from svg.distributed import overlapped_sharded_attention

    dist.init_process_group("gloo", rank=rank, world_size=world)

    torch.manual_seed(0)
    q, k, v = (torch.randn(1, H, 48, 16) for _ in range(3))
    scale_h = torch.arange(H, dtype=torch.float32)[None]          # a per-head argument, sliced like the heads

    def attn(qh, kh, vh, sc):
        return torch.nn.functional.scaled_dot_product_attention(qh, kh, vh) * (1.0 + sc[:, :, None, None])

    out = overlapped_sharded_attention(q, k, v, attn, per_head_args=(scale_h,))
    ref = attn(q, k, v, scale_h)
    ret[rank] = bool(torch.equal(out, ref))
    dist.destroy_process_group()


@pytest.mark.parametrize("H", [24, 6])
def test_overlapped_sharded_attention_gloo(H):
    """library form of the bench's N > 1 scheme: chunked heads, per-chunk asynchronous all-gather, full output on every rank"""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 33500 + (os.getpid() % 2000) + H
    mp.spawn(_worker_overlapped, args=(world, port, H, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _worker_exchange(rank, world, port, H, S, unit, chunked, ret):
    """inbound (token shards -> head shards) and outbound (head shards -> token shards) all-to-all of a layer-call"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
    from svg.distributed import chunked_head_layout, heads_to_tokens, shard_heads, token_range, tokens_to_heads

    dist.init_process_group("gloo", rank=rank, world_size=world)
    D = 4
    torch.manual_seed(0)
    x_all = torch.randn(H, S, D)                                   # the same global tensor on every rank
    a, b = token_range(S, rank, world, unit)
    lists = [chunked_head_layout(H, r, world, max_chunks=24)[2] for r in range(world)] if chunked else None
    mine = lists[rank] if chunked else shard_heads(H, rank, world)
    ok = sorted(t for r in range(world) for t in range(*token_range(S, r, world, unit))) == list(range(S))
    ok &= (a % unit == 0) and (b % unit == 0 or rank == world - 1)
    got = tokens_to_heads(x_all[:, a:b].contiguous(), S, unit=unit, head_lists=lists)
    ok &= torch.equal(got, x_all[mine])
    if chunked:   # bench.py's form: heads already ordered owner by owner, result written into a preallocated buffer
        order = [h for o in lists for h in o]
        buf = torch.full((len(mine), S, D), float("nan"))
        tokens_to_heads(x_all[order][:, a:b].contiguous(), S, unit=unit, head_lists=lists, presorted=True, out=buf)
        ok &= torch.equal(buf, x_all[mine])
    fn = lambda t: torch.tanh(t) * 3.0                             # noqa: E731  stand-in for the per-head attention
    back = heads_to_tokens(fn(got), H, unit=unit, head_lists=lists)
    ok &= torch.equal(back, fn(x_all)[:, a:b])
    ret[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,H,S,unit,chunked", [
    (2, 6, 50, 1, False),        # even heads, even tokens
    (2, 5, 37, 1, False),        # ragged heads (3 + 2), ragged tokens
    (2, 24, 5 * 7 + 3, 7, True),  # frames kept whole (unit = tokens per frame), text tokens on the last rank; chunked ownership
    (3, 40, 4 * 9 + 2, 9, False),  # 40 heads over 3 ranks (14, 13, 13), 4 frames over 3 ranks
    (8, 40, 21 * 3, 3, True),     # Wan: 40 heads / 8 ranks (5 each), 21 frames / 8 ranks (3, 3, 3, 3, 3, 2, 2, 2)
])
def test_token_head_exchange_gloo(world, H, S, unit, chunked):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 35500 + (os.getpid() % 2000) + H + world
    mp.spawn(_worker_exchange, args=(world, port, H, S, unit, chunked, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def _worker_enable(rank, world, port, H, ret):
    """svg.distributed.enable(): run_sharded is what the processors' attention cores go through (svg/models/_core.py)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
    from svg import distributed as sd

    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = not sd.active()
    sd.enable()
    ok &= sd.active()
    torch.manual_seed(0)
    q, k, v = (torch.randn(2, H, 24, 8) for _ in range(3))

    def core(qh, kh, vh):   # (output [cfg, h, S, D], per-head index [cfg, h]) like svg1_sparse_attention
        o = torch.nn.functional.scaled_dot_product_attention(qh, kh, vh)
        return o, o.sum(dim=(2, 3)).argsort(dim=1)[:, : qh.shape[1]] * 0 + (qh[:, :, 0, 0] > 0).long()

    o, idx = sd.run_sharded(core, (q, k, v), sd.current_group())
    ro, ridx = core(q, k, v)
    ok &= torch.equal(o, ro) and torch.equal(idx, ridx) and idx.dtype == torch.int64
    t = torch.tensor(float(rank))
    ok &= float(sd.all_reduce_max_(t)) == world - 1
    sd.disable()
    ok &= (not sd.active()) and float(sd.all_reduce_max_(torch.tensor(-1.0))) == -1.0
    ret[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,H", [(2, 6), (3, 40)])
def test_enable_run_sharded_gloo(world, H):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 37500 + (os.getpid() % 2000) + H
    mp.spawn(_worker_enable, args=(world, port, H, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}
