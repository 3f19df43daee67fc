#!/bin/bash
# Round 5, GPU call zs: Wan block glue with scale / shift from LDS: the glue tests, the parts' timing, bench_hbm rows
tag=${1:-r05zs}; O=gpurun_out/$tag; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_glue.py tests/test_gpu_triton_golden.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4 | tee $O/pytest_glue.txt
bash tools/history/r05/gpu_r05zr.sh $tag 2>&1 | tail -4
timeout 300 python bench_hbm.py --reps 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for n,r in d['kernels'].items(): print(n, r['ms'], r['GBs'], r['frac_of_8TBs'])
print('copy', d['torch_copy_this_box'])" | tee $O/hbm.txt
