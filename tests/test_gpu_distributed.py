"""svg.distributed.enable() on a GPU box: the processors' attention cores (svg/models/_core.py) run this rank's heads only and
all-gather — the result must equal the unsharded call bit for bit (SVG1: profiler + fused placement + band attention; SVG2: k-means
with random initial points and warm start, block map, variable-block attention).  Two ranks share cuda:0 over gloo because a gpurun
box has one GPU; RCCL is what the same code uses on a node."""
import os
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
    from svg import _native as nat
    from svg import distributed as sd
    from svg.models import _core

    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    ok = True
    # ---- SVG1 (Hunyuan-like geometry, 5 heads: ragged 3 + 2) ----
    H, D, F_, P_, ctx, L = 5, 128, 4, 200, 40, 11
    V = F_ * P_
    S = V + ctx
    g = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(1, H, S, D, generator=g).to(torch.bfloat16).to(dev) for _ in range(3))
    geo = _core.Geometry(ctx, F_, P_)
    mask = nat.BandMask(real_len=V + L, band=256, colfull_lo=V, colfull_hi=V + L, rowfull_lo=V, rowfull_hi=V + L)
    prof = nat.ProfileDesc(0, F_, P_, 1)
    prof.variant[0] = nat.ProfileVariant(0, 0, V, 2, 0, V, S)
    prof.variant[1] = nat.ProfileVariant(1, 0, V, 2, 0, V, S)

    def svg1():
        torch.manual_seed(11)     # sample_mse draws its rows from the CPU generator: the same on every rank
        return _core.svg1_sparse_attention(q, k, v, geo, mask, prof, 32, V)

    o_ref, best_ref = svg1()
    sd.enable()
    o_sh, best_sh = svg1()
    sd.disable()
    ok &= torch.equal(o_ref, o_sh) and torch.equal(best_ref, best_sh) and bool(best_ref.float().std() >= 0)
    # ---- SVG2 (Wan-like: no text), first call (random initial points) and warm-started call ----
    H2, F2, P2 = 5, 5, 300
    S2 = F2 * P2
    q2, k2, v2 = (torch.randn(1, H2, S2, D, generator=g).to(torch.bfloat16).to(dev) for _ in range(3))
    geo2 = _core.Geometry(0, F2, P2)

    def svg2(store):
        outs = []
        for call in range(2):
            torch.manual_seed(5 + call)
            torch.cuda.manual_seed(5 + call)
            outs.append(_core.svg2_sparse_attention(q2, k2, v2, geo2, store, 0, 12, 30, 0.9, 0.1, 4, 2))
        return outs

    ref2 = svg2(_core.CentroidStore())
    sd.enable()
    sh2 = svg2(_core.CentroidStore())
    sd.disable()
    ok &= all(torch.equal(a, b) for a, b in zip(ref2, sh2)) and all(torch.isfinite(a.float()).all() for a in sh2)
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_processors_head_sharded_equals_single_gpu():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 39500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}
