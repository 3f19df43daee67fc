#!/bin/bash
# svg_kmeans_loop_strided: the video tokens of q / k read in place by the Lloyd loop (HunyuanVideo SVG2: no contiguous copies of q[:, :, :V], k[:, :, :V])
tag=${1:-r06k2}; O=gpurun_out/$tag; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_strided.py tests/test_gpu_kernels.py tests/test_gpu_fullsize_svg2.py tests/test_gpu_processors.py tests/test_gpu_reference_calls.py tests/test_gpu_bench_contract.py -q -m gpu -x -k "kmeans or svg2 or sap or SAP or strided or bench_svg2 or extras" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
grep -v amdgpu.ids $O/pytest.txt | tail -5
for w in hy720p wan720p; do timeout 300 python bench_svg2.py --workload $w --steps 4 --warmup 2 2> $O/svg2_$w.err | tail -n 1 > $O/svg2_$w.json; python -c "
import json; d=json.load(open('$O/svg2_$w.json')); print('$w', d['ms'], d['kmeans_init_50it_ms'], d.get('io_layout_ab'))"; done
