"""A/B on one box: the attention kernels on contiguous [H, S, D] tensors against the same call with v read in place from the
projection layout and o written token-major (svg_attn_layout_t), at the BASELINE geometries.

    python tools/ab_strided.py [hy|cog|wan|both] [reps]

Prints one JSON line per case: kernel ms (mean of `reps` launches after 2 warm-up launches, HIP events on the launch stream), the
output checksum equality, and the cost of the two copies the strided call makes unnecessary (V transpose in, O transpose out)."""
import json
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "sparse-videogen_amd"))
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def timed(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def hy(reps):
    from svg import _native as nat
    from svg.models.hyvideo.utils import sparsity_to_width

    dev = torch.device("cuda:0")
    F_, P_, ctx, L, H, D = 33, 3600, 256, 64, 24, 128
    V = F_ * P_
    S = V + ctx
    width = sparsity_to_width(0.25, ctx, F_, P_)
    band = math.floor(width * P_ / 128) * 128
    mask = nat.BandMask(real_len=V + L, band=band, colfull_lo=V, colfull_hi=V + L, rowfull_lo=V, rowfull_hi=V + L)
    g = torch.Generator(device=dev).manual_seed(0)
    q, k = (torch.randn(1, H, S, D, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(2))
    best = (torch.arange(H, device=dev) % 2).reshape(1, H)
    pk = dict(head_perm_flag=best, vid0=0, num_frame=F_, frame_size=P_)
    out = {}
    for name, width_mult in (("v_separate_projection", 1), ("v_slice_of_fused_qkv", 3)):
        buf = torch.randn(1, S, width_mult * H * D, device=dev, dtype=torch.bfloat16, generator=g)
        v_view = buf[:, :, (width_mult - 1) * H * D:].unflatten(2, (H, D)).transpose(1, 2)
        v_c = v_view.contiguous()
        o_c = torch.empty_like(q)
        o_t = nat.token_major_empty(q)
        t_c = timed(lambda: nat.band_attention(q, k, v_c, mask, out=o_c, **pk), reps)
        t_s = timed(lambda: nat.band_attention(q, k, v_view, mask, out=o_t, **pk), reps)
        t_vo = timed(lambda: nat.band_attention(q, k, v_view, mask, out=o_c, **pk), reps)
        t_oo = timed(lambda: nat.band_attention(q, k, v_c, mask, out=o_t, **pk), reps)
        t_vcopy = timed(lambda: v_view.contiguous(), reps)
        t_ocopy = timed(lambda: o_c.transpose(1, 2).contiguous(), reps)
        out[name] = dict(contiguous_ms=round(t_c, 3), strided_ms=round(t_s, 3), v_in_place_only_ms=round(t_vo, 3), o_token_major_only_ms=round(t_oo, 3),
                         v_transpose_copy_ms=round(t_vcopy, 3), o_transpose_copy_ms=round(t_ocopy, 3), equal=bool(torch.equal(o_c, o_t)),
                         v_row_stride_bytes=int(v_view.stride(2) * 2))
    print(json.dumps({"case": "hy720p_svg1_band", "S": S, "H": H, "band": band, "reps": reps, **out}), flush=True)


def hy_pmc(strided: bool, launches: int = 2):
    """the HunyuanVideo 720p launch alone, for counter passes (tools/gpu_pmc.sh with PMC_CMD="python tools/ab_strided.py pmc-strided"):
    q, k head-major; v in place behind a fused-QKV row stride and o token-major (strided) or both contiguous"""
    from svg import _native as nat
    from svg.models.hyvideo.utils import sparsity_to_width

    dev = torch.device("cuda:0")
    F_, P_, ctx, L, H, D = 33, 3600, 256, 64, 24, 128
    V = F_ * P_
    S = V + ctx
    band = math.floor(sparsity_to_width(0.25, ctx, F_, P_) * P_ / 128) * 128
    mask = nat.BandMask(real_len=V + L, band=band, colfull_lo=V, colfull_hi=V + L, rowfull_lo=V, rowfull_hi=V + L)
    g = torch.Generator(device=dev).manual_seed(0)
    q, k = (torch.randn(1, H, S, D, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(2))
    buf = torch.randn(1, S, 3 * H * D, device=dev, dtype=torch.bfloat16, generator=g)
    v = buf[:, :, 2 * H * D:].unflatten(2, (H, D)).transpose(1, 2)
    best = (torch.arange(H, device=dev) % 2).reshape(1, H)
    pk = dict(head_perm_flag=best, vid0=0, num_frame=F_, frame_size=P_)
    if not strided:
        v = v.contiguous()
    o = nat.token_major_empty(q) if strided else torch.empty_like(q)
    for _ in range(launches):
        nat.band_attention(q, k, v, mask, out=o, **pk)
    torch.cuda.synchronize()
    print(json.dumps({"case": "hy720p_pmc", "strided": strided, "launches": launches}), flush=True)


def cog(reps):
    """CogVideoX-v1.5 768p SVG1 (cfg 2 x 48 heads, head_dim 64: the 32x32x16 two-phase body), text first"""
    from svg import _native as nat
    from svg.models.cog import utils as cog_u
    from svg.models.hyvideo.utils import sparsity_to_width

    dev = torch.device("cuda:0")
    B, H, D, F_, P_, ctx = 2, 48, 64, 11, 4080, 226
    S = ctx + F_ * P_
    mask = cog_u.generate_temporal_head_mask_mod(ctx, F_, P_, mul=sparsity_to_width(0.25, ctx, F_, P_))
    g = torch.Generator(device=dev).manual_seed(0)
    q, k = (torch.randn(B, H, S, D, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(2))
    v_view = torch.randn(B, S, H * D, device=dev, dtype=torch.bfloat16, generator=g).unflatten(2, (H, D)).transpose(1, 2)
    v_c = v_view.contiguous()
    best = (torch.arange(B * H, device=dev) % 2).reshape(B, H)
    pk = dict(head_perm_flag=best, vid0=ctx, num_frame=F_, frame_size=P_)
    o_c, o_t = torch.empty_like(q), nat.token_major_empty(q)
    t_c = timed(lambda: nat.band_attention(q, k, v_c, mask, out=o_c, **pk), reps)
    t_s = timed(lambda: nat.band_attention(q, k, v_view, mask, out=o_t, **pk), reps)
    t_vcopy = timed(lambda: v_view.contiguous(), reps)
    t_ocopy = timed(lambda: o_c.transpose(1, 2).contiguous(), reps)
    print(json.dumps({"case": "cogvideox15_768p_svg1_band", "S": S, "BH": B * H, "D": D, "reps": reps, "contiguous_ms": round(t_c, 3),
                      "strided_ms": round(t_s, 3), "v_transpose_copy_ms": round(t_vcopy, 3), "o_transpose_copy_ms": round(t_ocopy, 3),
                      "equal": bool(torch.equal(o_c, o_t))}), flush=True)


def wan(reps):
    import bench_svg2 as B

    r = B.measure("wan720p", steps=max(1, reps // 2), warmup=1)
    print(json.dumps({"case": "wan720p_svg2_varblock", "ms": r["ms"], "io_layout_ab": r.get("io_layout_ab")}), flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    if which in ("pmc-strided", "pmc-contiguous"):
        hy_pmc(which == "pmc-strided")
        sys.exit(0)
    if which in ("hy", "both"):
        hy(reps)
    if which in ("cog", "both"):
        cog(reps)
    if which in ("wan", "both"):
        wan(reps)
