#!/usr/bin/env python3
"""Turn the counter summary of tools/gpu_pmc.sh into profiles/<tag>_pmc_traffic.json (what bench.py reports as
roofline.traffic): HBM-side bytes per launch of the dominant attention kernel = 2 x FETCH_SIZE (gfx950 correction of
MI355X_MICROARCH.md: the counter counts 64-B units of 128-B requests) + WRITE_SIZE, both in KB.

    python tools/pmc_traffic.py gpurun_out/pmc_<tag>/summary.txt <tag> [kernel substring] [source text]  >  profiles/<tag>_pmc_traffic.json
"""
import json
import re
import sys

txt = open(sys.argv[1]).read()
tag = sys.argv[2] if len(sys.argv) > 2 else "?"
blocks = {}
cur = None
for line in txt.splitlines():
    if line and not line.startswith(" "):
        cur = blocks.setdefault(line.strip(), {})
    else:
        m = re.match(r"\s+(\S+)\s+n=(\d+)\s+mean=(\S+)", line)
        if m and cur is not None:
            cur[m.group(1)] = float(m.group(3))
# dominant kernel: the band attention kernel with the most wave cycles (or the only one)
# (the 16-bit headline kernel when it is there: the bench's fp8 extras block launches band_attn_f8_kernel in the same run)
want = (sys.argv[3] if len(sys.argv) > 3 else None) or None      # optional: substring of the kernel to report (e.g. band_attn_f8)
cands = [(k, v) for k, v in blocks.items() if want and want in k] or \
        [(k, v) for k, v in blocks.items() if "band_attn_m16" in k] or \
        [(k, v) for k, v in blocks.items() if "band_attn_pp2q" in k] or \
        [(k, v) for k, v in blocks.items() if "band_attn_pp2" in k or "band_attn_w4" in k] or [(k, v) for k, v in blocks.items() if "band_attn" in k]
name, c = max(cands, key=lambda kv: kv[1].get("SQ_WAVE_CYCLES", 0.0))
g = c.get
out = {
    "kernel": "band_attn_f8_kernel<bf16>" if "band_attn_f8" in name else
              ("band_attn_m16_kernel<bf16>" if "band_attn_m16" in name else None) or next((f"{k}<bf16,128>" for k in ("band_attn_pp2q_kernel", "band_attn_w4_kernel", "band_attn_pp2_kernel", "band_attn_kernel") if k in name), name),
    "kernel_symbol": name,
    "source": sys.argv[4] if len(sys.argv) > 4 else
              f"tools/gpu_pmc.sh {tag}: separate rocprofv3 --pmc passes of `bench.py --steps 1 --warmup 1 --no-profiler`, one launch each",
    "FETCH_SIZE_KB": g("FETCH_SIZE"),
    "WRITE_SIZE_KB": g("WRITE_SIZE"),
    "traffic_bytes_per_launch": (2.0 * g("FETCH_SIZE", 0.0) + g("WRITE_SIZE", 0.0)) * 1024.0,
    "note": "FETCH_SIZE x2: gfx950 correction (MI355X_MICROARCH.md).  The counter sits between L2 and the fabric: reads served by the "
            "256 MiB Infinity Cache are included, so this is an upper bound of the HBM bytes.",
    "l2_hit_rate": (g("TCC_HIT_sum", 0.0) / max(1.0, g("TCC_HIT_sum", 0.0) + g("TCC_MISS_sum", 0.0))) if g("TCC_HIT_sum") else None,
    # busy cycles summed over the 1024 SIMDs / elapsed cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
    "mfma_busy_frac": (g("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / max(1.0, g("GRBM_GUI_ACTIVE", 8.0) / 8.0))
                      if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("GRBM_GUI_ACTIVE") else None,
    "per_mfma": {k: round(g(c, 0.0) / max(1.0, g("SQ_INSTS_MFMA", 1.0)), 3) for k, c in
                 (("valu_incl_mfma", "SQ_INSTS_VALU"), ("salu", "SQ_INSTS_SALU"), ("lds", "SQ_INSTS_LDS"), ("vmem_rd", "SQ_INSTS_VMEM_RD"))}
                if g("SQ_INSTS_MFMA") else None,
    "counters": c,
}
print(json.dumps(out, indent=1))
