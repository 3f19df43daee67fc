#!/usr/bin/env python3
"""What one rank of `bench.py --gpus N` computes, on one GPU: H / N heads of the HunyuanVideo 720p layer-call as chunk launches on
two alternating streams (no collective) next to the same heads as a single launch.  python tools/nrank_emulate.py"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
import torch  # noqa: E402
from svg import _native as nat  # noqa: E402
from svg.distributed import chunked_head_layout  # noqa: E402

H, D, F_, P_, ctx, L, band = 24, 128, 33, 3600, 256, 64, 15616
V = F_ * P_
S = V + ctx
dev = torch.device("cuda", 0)
mask = nat.BandMask(real_len=V + L, band=band, colfull_lo=V, colfull_hi=V + L, rowfull_lo=V, rowfull_hi=V + L)
full_ms = None
for N in (1, 2, 4, 8):
    n_chunks, n_per, mine = chunked_head_layout(H, 0, N)
    Hl = len(mine)
    q, k, v = (torch.randn(1, Hl, S, D, device=dev, dtype=torch.bfloat16) for _ in range(3))
    o = torch.empty_like(q)
    best = torch.tensor([[h % 2 for h in mine]], device=dev, dtype=torch.int64)
    side = [torch.cuda.Stream(device=dev) for _ in range(2)]

    def chunked():
        main = torch.cuda.current_stream()
        for c in range(n_chunks):
            sl = slice(c * n_per, (c + 1) * n_per)
            st = side[c % 2] if n_chunks > 1 else main
            st.wait_stream(main)
            with torch.cuda.stream(st):
                nat.band_attention(q[:, sl], k[:, sl], v[:, sl], mask, head_perm_flag=best[:, sl].contiguous(), vid0=0, num_frame=F_,
                                   frame_size=P_, out=o[:, sl])
        for st in side:
            main.wait_stream(st)

    def single():
        nat.band_attention(q, k, v, mask, head_perm_flag=best, vid0=0, num_frame=F_, frame_size=P_, out=o)

    done = nat.notify_counters(Hl, 1, dev)
    target = nat.band_notify_target(S, mask)

    def notify():   # what bench.py --gpus N does: one launch + per-chunk waiters on the side streams (the gathers would follow them)
        main = torch.cuda.current_stream()
        done.zero_()
        ev = torch.cuda.Event()
        ev.record()
        nat.band_attention(q, k, v, mask, head_perm_flag=best, vid0=0, num_frame=F_, frame_size=P_, out=o, done=done)
        for c in range(n_chunks):
            st = side[c % 2]
            st.wait_event(ev)
            with torch.cuda.stream(st):
                nat.wait_counters(done[c * n_per:(c + 1) * n_per], target)
        for st in side:
            main.wait_stream(st)

    res = []
    for fn in (chunked, single, notify):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res.append(min(ts))
    if N == 1:
        full_ms = res[1]
    print(f"N = {N}: {Hl:2d} heads per rank, {n_chunks} chunks of {n_per}: {res[0]:7.3f} ms as chunk launches on two streams, {res[1]:7.3f} ms as one "
          f"launch, {res[2]:7.3f} ms as one launch with completion counters + waiters (shipped); ideal {full_ms / N:7.3f} ms -> "
          f"compute-only efficiency {full_ms / N / res[2]:.3f}", flush=True)
    del q, k, v, o
