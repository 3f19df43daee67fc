"""The N-rank denoise step of bench_step.py (BASELINE.json configs[3]: hidden states token-sharded in units of tokens, attention
head-sharded, tokens_to_heads / heads_to_tokens around every attention, one all-gather of the hidden states per step) against the
single-process step on the same stack: 2, 3 and 8 processes (24 heads) over gloo on CPU with a torch statement of the op table
(tests/step_ops_torch.py) — what is under test is the sharding: token ranges, RoPE table slices, the text stream living on the last
rank, head shards, exchange layouts, the final gather.  The HIP ops themselves are tested in tests/test_gpu_*.py."""
import os
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _setup_paths():
    for p in (str(ROOT), str(ROOT / "sparse-videogen_amd"), str(ROOT / "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _geo(heads):
    _setup_paths()
    import bench_step

    return bench_step.StepGeo(F=5, P=40, ctx=24, L=7, hid=heads * 32, heads=heads, hd=32, mlp=96, unit=8)


def _single(heads, sparse):
    _setup_paths()
    import bench_step
    from step_ops_torch import TorchOps

    geo = _geo(heads)
    st = bench_step.Stack(1, 1, torch.device("cpu"), geo, dtype=torch.float32)
    g = torch.Generator().manual_seed(1)
    img = torch.randn(geo.V, geo.hid, generator=g) * 0.5
    txt = torch.randn(geo.ctx, geo.hid, generator=g) * 0.5
    return bench_step.run_step(st, img, txt, sparse, 1, [], TorchOps(geo), None, events=False), img, txt


def _worker(rank, world, port, heads, sparse, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    _setup_paths()
    import bench_step
    from step_ops_torch import TorchOps

    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    geo = _geo(heads)
    ref, img, txt = _single(heads, sparse)
    st = bench_step.Stack(1, 1, torch.device("cpu"), geo, dtype=torch.float32)
    sh = bench_step.Sharding(geo, rank, world, torch.device("cpu"), torch.float32)
    x = bench_step.run_step(st, img[sh.a: sh.a + sh.nv].contiguous(), txt[sh.t0: sh.t0 + sh.nt].contiguous(), sparse, 1,
                            [], TorchOps(geo), sh, events=False)
    full, nbytes = sh.gather_tokens(x)
    ok = (tuple(x.shape) == (sh.b - sh.a, geo.hid) and torch.allclose(x, ref[sh.a:sh.b], atol=2e-5, rtol=2e-5)
          and torch.allclose(full, ref, atol=2e-5, rtol=2e-5))
    # exchange volume bookkeeping: 2 layers x (3 inbound + 1 outbound) all-to-alls
    Hl = geo.heads // world
    want_in = 2 * 3 * Hl * (geo.S - (sh.b - sh.a)) * geo.hd * 4
    want_out = 2 * (geo.heads - Hl) * (sh.b - sh.a) * geo.hd * 4
    ok = ok and sh.buf.bytes_in == want_in and sh.buf.bytes_out == want_out and nbytes > 0
    # the text tokens are the tail of the sequence: whichever ranks' ranges reach past V hold them, in order
    ok = ok and sh.nt == max(0, sh.b - max(sh.a, geo.V)) and sh.t0 == max(sh.a, geo.V) - geo.V
    ret[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,heads,sparse", [(2, 4, True), (2, 4, False), (3, 6, True), (8, 24, True)])
def test_token_sharded_step_equals_single_process(world, heads, sparse):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29650 + world * 7 + heads + int(sparse)
    mp.spawn(_worker, args=(world, port, heads, sparse, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_production_token_shards_are_balanced():
    """HunyuanVideo 720p over 8 ranks: the shards the step bench uses differ by at most 1 % from the mean
    (whole frames: 5 / 4 frames, 21 % over), cover the sequence exactly once, and every boundary but the last sits on a unit."""
    _setup_paths()
    import bench_step
    from svg import distributed as D

    geo = bench_step.HY720P
    for world in (2, 4, 8):
        tr = [D.token_range(geo.S, r, world, unit=geo.unit) for r in range(world)]
        assert tr[0][0] == 0 and tr[-1][1] == geo.S and all(tr[i][1] == tr[i + 1][0] for i in range(world - 1))
        assert all(a % geo.unit == 0 for a, _ in tr)
        sizes = [b - a for a, b in tr]
        assert max(sizes) * world / geo.S <= 1.01, (world, sizes)
    frames = [b - a for a, b in (D.token_range(geo.S, r, 8, unit=geo.P) for r in range(8))]
    assert max(frames) * 8 / geo.S > 1.2      # what the frame-granular split cost


# ---------------------------------------------------------------------------------------------------------
# the Wan 2.1 SVG2 step (bench_step.run_step_wan, BASELINE.json configs[2]) token-sharded over 2 ranks == one process
# ---------------------------------------------------------------------------------------------------------
def _wan_geo():
    _setup_paths()
    import bench_step

    return bench_step.WanGeo(F=4, P=40, hid=4 * 32, heads=4, hd=32, ffn=96, text=12, layers=2, qc=4, kc=6, unit=8, iter_step=2)


def _wan_single(sparse):
    _setup_paths()
    import bench_step
    from step_ops_torch import WanTorchOps

    geo = _wan_geo()
    st = bench_step.WanStack(torch.device("cpu"), geo, dtype=torch.float32)
    x = torch.randn(geo.V, geo.hid, generator=torch.Generator().manual_seed(1)) * 0.5
    ops = WanTorchOps(geo)
    outs = [bench_step.run_step_wan(st, x, sparse, 1, ops, None) for _ in range(2)]     # two steps: the second one warm-starts the k-means
    return outs, x


def _wan_worker(rank, world, port, sparse, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    _setup_paths()
    import bench_step
    from step_ops_torch import WanTorchOps

    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    geo = _wan_geo()
    refs, x = _wan_single(sparse)
    st = bench_step.WanStack(torch.device("cpu"), geo, dtype=torch.float32)
    sh = bench_step.Sharding(geo, rank, world, torch.device("cpu"), torch.float32)
    ops = WanTorchOps(geo)
    ok = True
    for ref in refs:
        y = bench_step.run_step_wan(st, x[sh.a:sh.b].contiguous(), sparse, 1, ops, sh)
        full, _ = sh.gather_tokens(y)
        ok = ok and tuple(y.shape) == (sh.b - sh.a, geo.hid) and torch.allclose(y, ref[sh.a:sh.b], atol=3e-5, rtol=3e-5)
        ok = ok and torch.allclose(full, ref, atol=3e-5, rtol=3e-5)
    Hl = geo.heads // world     # 2 layers x (3 inbound + 1 outbound) all-to-alls in the last step
    ok = ok and sh.nt == 0 and sh.nv == sh.b - sh.a
    ret[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,sparse", [(2, True), (2, False), (4, True)])
def test_wan_svg2_token_sharded_step_equals_single_process(world, sparse):
    """layer 0 dense, layer 1 SVG2 (k-means per head from the head's own first rows -> block map -> variable-block attention), cross attention over
    replicated text tokens, two consecutive steps (warm start): every rank's token shard and the gathered hidden states equal the one-process step"""
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29750 + world * 5 + int(sparse)
    mp.spawn(_wan_worker, args=(world, port, sparse, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)
