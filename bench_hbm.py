#!/usr/bin/env python3
"""The HBM-bound rows of SURVEY.md §8(d) on one MI355X, each against the HBM roofline: head placement of q, k, v and its inverse
(HunyuanVideo 720p), the SVG2 token gather / scatter and the stable label sort (Wan 2.1 720p), the pre-attention prologue
`svg_qk_norm_rope_transpose` (+ the in-place `svg_qk_norm_rope`), and the Wan block glue (`svg_layernorm_modulate_forward`,
`svg_modulate_gate_residual_forward`).  Algorithmic bytes = every tensor row read once + written once; peaks from
MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured float4 copy.  A torch device-to-device copy of one Hunyuan tensor is timed in
the same run as this box's own reference point.

    python bench_hbm.py [--size full|small] [--reps N]     -> one JSON line
`bench.py` carries the same block as `hbm_kernels` in its line (extras), so the driver's BENCH file has a roofline for every one of
these rows.  Reference micro-benchmarks this mirrors: svg/models/hyvideo/placement.py:224-282, svg/kernels/triton/permute.py:203-260,
svg/kernels/test/bench_rms_norm.py:12-68 (time of one call on production shapes)."""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "sparse-videogen_amd"))
import torch  # noqa: E402

PEAK_SPEC_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PEAK_COPY_GBS = 6290.0      # MI355X_MICROARCH.md: 6.29 TB/s measured (float4 copy)

GEOMS = {   # Hunyuan side: H, D, F, P, ctx; Wan side: H, S, D, K (k clusters), hidden
    "full": {"hy": (24, 128, 33, 3600, 256), "wan": (40, 75600, 128, 1000, 5120)},
    "small": {"hy": (4, 128, 5, 160, 256), "wan": (4, 2048, 128, 40, 512)},
}


def _time(fn, reps: int) -> float:
    """mean ms of `reps` back-to-back calls between two events, after one warm call"""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def measure(size: str = "full", reps: int = 10, device: int = 0) -> dict:
    from svg import _native as nat

    nat.load()
    dev = torch.device("cuda", device)
    torch.cuda.set_device(dev)
    H, D, F_, P_, ctx = GEOMS[size]["hy"]
    S = F_ * P_ + ctx
    Hw, Sw, Dw, K, hid = GEOMS[size]["wan"]
    dt = torch.bfloat16
    rows = {}

    def rec(name, kernel, workload, ms, nbytes, ref):
        gbs = nbytes / ms / 1e6
        rows[name] = {"kernel": kernel, "workload": workload, "ms": round(ms, 4), "algorithmic_bytes": int(nbytes), "GBs": round(gbs, 1),
                      "frac_of_8TBs": round(gbs / PEAK_SPEC_GBS, 4), "frac_of_6.29TBs_copy": round(gbs / PEAK_COPY_GBS, 4), "reference": ref}

    # ---- HunyuanVideo: placement (a6), inverse placement (a8), prologue (f1) ----
    hy = f"HunyuanVideo {'720p 129f' if size == 'full' else 'toy'} cfg=1 H={H} S={S} D={D} bf16"
    q, k, v = (torch.randn(1, H, S, D, device=dev, dtype=dt) for _ in range(3))
    qo, ko, vo = (torch.empty_like(q) for _ in range(3))
    tensor_bytes = H * S * D * 2
    ms_copy = _time(lambda: qo.copy_(q), reps)
    copy_gbs = 2 * tensor_bytes / ms_copy / 1e6
    best = torch.tensor([[h % 2 for h in range(H)]], device=dev)
    rec("placement_qkv", "placement_kernel (svg_head_placement)", hy,
        _time(lambda: nat.head_placement([q, k, v], [qo, ko, vo], best, ctx, F_, P_, False, inverse=False), reps), 6 * tensor_bytes,
        "svg/models/hyvideo/placement.py:35-153 (hunyuan_sparse_head_placement)")
    rec("inverse_placement_o", "placement_kernel (inverse)", hy,
        _time(lambda: nat.head_placement([q], [qo], best, ctx, F_, P_, False, inverse=True), reps), 2 * tensor_bytes,
        "svg/models/hyvideo/placement.py:286-387 (hunyuan_hidden_states_placement)")
    w = [torch.randn(D, device=dev).to(dt) for _ in range(2)]
    cs, sn = torch.randn(S - ctx, D, device=dev), torch.randn(S - ctx, D, device=dev)
    table_bytes = 2 * (S - ctx) * D * 4
    rec("qk_norm_rope_inplace", f"qk_prologue_kernel<bf16,{D}> (svg_qk_norm_rope)", hy + ", RMSNorm + text-last RoPE in place on q, k",
        _time(lambda: nat.qk_norm_rope(q, k, 1, w[0], None, w[1], None, 1e-6, 1, cs, sn, 0, S - ctx), reps), 4 * tensor_bytes + table_bytes,
        "svg/models/hyvideo/attention.py:166-176 (rms_norm_forward x2 + apply_qk_rope_inplace_cossin_txtlast)")
    del qo, ko, vo
    qin, kin, vin = (x.transpose(1, 2).reshape(1, S, H * D).contiguous() for x in (q, k, v))

    def transposing():
        nat.qk_norm_rope_transpose(qin, kin, H, H, 1, w[0], None, w[1], None, 1e-6, 1, cs, sn, 0, S - ctx)
        nat.qk_norm_rope_transpose(vin, None, H, 0)

    rec("qk_norm_rope_transpose", f"qk_prologue_transpose_kernel (svg_qk_norm_rope_transpose: q, k with norm + RoPE, then v)",
        hy + ", [1, S, H*D] -> [1, H, S, D]", _time(transposing, reps), 6 * tensor_bytes + table_bytes,
        "svg/models/hyvideo/attention.py:260-264, 166-176 (get_transpose_qkv + norm + rope)")
    del q, k, v, qin, kin, vin, cs, sn
    torch.cuda.empty_cache()

    # ---- Wan 2.1: SVG2 label sort (a13), gather (a13) / scatter (a17), block glue (f2) ----
    wan = f"Wan 2.1 {'720p' if size == 'full' else 'toy'} H={Hw} S={Sw} D={Dw} bf16"
    x = torch.randn(Hw, Sw, Dw, device=dev, dtype=dt)
    labels = torch.randint(0, K, (Hw, Sw), device=dev, dtype=torch.int32)
    sidx, _counts = nat.argsort_labels(labels, K)
    rec("label_sort", "counting sort (svg_argsort_labels)", wan + f", K={K} labels", _time(lambda: nat.argsort_labels(labels, K), reps),
        Hw * Sw * 4 * 2, "svg/kmeans_utils.py:831 (torch.argsort of the labels)")
    wbytes = Hw * Sw * Dw * 2
    rec("svg2_gather", "permute_rows_kernel (svg_permute_rows)", wan, _time(lambda: nat.permute_rows(x, sidx), reps),
        2 * wbytes + Hw * Sw * 4, "svg/kernels/triton/permute.py:13-129 (permute_tensor_by_labels_triton)")
    y = nat.permute_rows(x, sidx)
    rec("svg2_scatter", "permute_rows_kernel (inverse)", wan, _time(lambda: nat.permute_rows(y, sidx, inverse=True), reps),
        2 * wbytes + Hw * Sw * 4, "svg/kernels/triton/permute.py:47-80, 132-200 (apply_inverse_permutation_triton)")
    del x, y, labels, sidx
    torch.cuda.empty_cache()
    hs = torch.randn(1, Sw, hid, device=dev, dtype=dt)
    att = torch.randn(1, Sw, hid, device=dev, dtype=dt)
    sc, sh, g = (torch.randn(1, 1, hid, device=dev) * 0.2 for _ in range(3))
    el = Sw * hid
    glue = f"Wan 2.1 {'720p' if size == 'full' else 'toy'} hidden [1, {Sw}, {hid}] bf16"
    rec("layernorm_modulate", "row_glue_kernel (svg_layernorm_modulate_forward)", glue,
        _time(lambda: nat.layernorm_modulate_forward(hs, None, None, sc, sh, 1e-6), reps), 4 * el,
        "svg/models/wan/custom_models.py:44-57 (layernorm + triton_modulate_shift_forward: two kernels, 12 B per element)")
    rec("modulate_gate_residual", "gate_residual_kernel (svg_modulate_gate_residual_forward)", glue,
        _time(lambda: nat.modulate_gate_residual_forward(hs, att, g, dt), reps), 6 * el,
        "svg/kernels/triton/modulate.py:91-160 (triton_modulate_gate_residual_forward), svg/models/wan/custom_models.py:60")
    del hs, att
    torch.cuda.empty_cache()
    # ---- Wan 2.1 self-attention prologue (f1): RMSNorm across all heads + complex RoPE + head-major transpose in ONE pass; with v and o in the
    #      projection layout (svg_attn_layout_t) only q and k go through it ----
    qin, kin, vin = (torch.randn(1, Sw, hid, device=dev, dtype=dt) for _ in range(3))
    qw, kw = (torch.randn(hid, device=dev).mul(0.2).add(1.5).to(dt) for _ in range(2))
    fr, fi = torch.randn(Sw, Dw // 2, device=dev), torch.randn(Sw, Dw // 2, device=dev)
    rope_bytes = 2 * Sw * (Dw // 2) * 4

    def three_pass():
        qn, kn = nat.rmsnorm_forward(qin, qw, 1e-6), nat.rmsnorm_forward(kin, kw, 1e-6)
        nat.qk_norm_rope_transpose(qn, kn, Hw, Hw, 0, None, None, None, None, 1e-6, 2, fr, fi, 0, Sw)
        nat.qk_norm_rope_transpose(vin, None, Hw, 0)

    ms3 = _time(three_pass, reps)
    pro = f"Wan 2.1 {'720p' if size == 'full' else 'toy'} q / k / v projections [1, {Sw}, {hid}] bf16 -> [1, {Hw}, {Sw}, {Dw}]"
    rec("wan_rmsnorm_rope_transpose_qkv", "rmsall_rope_transpose_kernel (svg_rmsnorm_rope_transpose: q, k, v)", pro,
        _time(lambda: nat.rmsnorm_rope_transpose(qin, kin, vin, Hw, qw, kw, 1e-6, 2, fr, fi, 0, Sw), reps), 6 * el * 2 + rope_bytes,
        "svg/models/wan/attention.py:99-148 (get_qk_norm -> get_transpose_qkv -> get_rotary_emb: three passes)")
    rows["wan_rmsnorm_rope_transpose_qkv"]["three_pass_sequence_ms"] = round(ms3, 4)
    rec("wan_rmsnorm_rope_transpose_qk", "rmsall_rope_transpose_kernel (svg_rmsnorm_rope_transpose: q, k; v read in place by the attention kernels)", pro,
        _time(lambda: nat.rmsnorm_rope_transpose(qin, kin, None, Hw, qw, kw, 1e-6, 2, fr, fi, 0, Sw), reps), 4 * el * 2 + rope_bytes,
        "svg/models/wan/attention.py:123-125 with svg_attn_layout_t (one GPU): the value copy of get_transpose_qkv is not made")
    del qin, kin, vin
    torch.cuda.empty_cache()
    return {"what": "HBM-bound kernels of SURVEY §8(d): algorithmic bytes (rows read once + written once) / time", "size": size, "reps": reps,
            "peak_GBs": {"spec": PEAK_SPEC_GBS, "measured_copy_MI355X_MICROARCH": PEAK_COPY_GBS},
            "torch_copy_this_box": {"ms": round(ms_copy, 4), "GBs": round(copy_gbs, 1), "bytes": 2 * tensor_bytes},
            "kernels": rows}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="full", choices=sorted(GEOMS))
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    print(json.dumps({"metric": "hbm_bound_kernels", **measure(a.size, a.reps)}))


if __name__ == "__main__":
    main()
