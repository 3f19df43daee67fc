#!/bin/bash
# Round 5, GPU call zv: PMC passes over the online profiler's kernel (HBM bytes fetched, L2 hits, instruction mix) through the torch-free harness
tag=${1:-r05zv}
PMC_KERNEL=profile16 PMC_CMD="tools/native_harness --geom hy720p --profiler --warm 1 --reps 1" PMC_PASS_TIMEOUT=40 PMC_ORDER="4 5 1 2" bash tools/gpu_pmc.sh $tag 2>&1 | tail -40
cat gpurun_out/pmc_$tag/passes.txt
