#!/bin/bash
# closing pass of round 6 on the final code, one box: GPU suite, default bench line, rocprofv3 kernel trace, PMC passes (headline kernel, SVG2),
# the complete variable-block grid, fabric traffic of the headline launch with v in place / o token-major against contiguous tensors, and both
# denoise steps with the transpose copies switched back on (SVG_STEP_TOKEN_MAJOR_IO=0) beside the bench line's own step blocks
tag=${1:-r06v}; O=gpurun_out/$tag; mkdir -p $O
bash tools/gpu_round_pass.sh $tag tests bench trace pmc svg2pmc
bash tools/gpu_fullgrid.sh $tag
for mode in strided contiguous; do
  PMC_CMD="python tools/ab_strided.py pmc-$mode" PMC_ORDER="4 5" PMC_PASS_TIMEOUT=200 bash tools/gpu_pmc.sh ${tag}_io_$mode > $O/pmc_io_$mode.txt 2>&1
  grep -A 12 "band_attn_m16" gpurun_out/pmc_${tag}_io_$mode/summary.txt 2>/dev/null | head -14
done
SVG_STEP_TOKEN_MAJOR_IO=0 timeout 600 python bench_step.py --model hy720p --steps 2 --warmup 1 > $O/step_hy_copies.json 2> $O/step_hy_copies.err; echo "step hy (copies) rc=$?"
SVG_STEP_TOKEN_MAJOR_IO=0 timeout 600 python bench_step.py --model wan720p --steps 2 --warmup 1 > $O/step_wan_copies.json 2> $O/step_wan_copies.err; echo "step wan (copies) rc=$?"
python - <<PY
import json
for f in ("step_hy_copies","step_wan_copies"):
    try:
        d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d.get("token_major_io"), d.get("denoise_steps_per_s"), d.get("sparse_step",{}).get("ms"), d.get("sparse_step",{}).get("step_breakdown_ms"))
    except Exception as e: print(f, "ERR", e)
PY
