#!/bin/bash
O=gpurun_out/r03f; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize_svg2.py tests/test_gpu_fp8.py tests/test_gpu_bsr.py tests/test_gpu_processors.py tests/test_gpu_bench_contract.py -q -m gpu -k "varblock or svg2 or fp8 or bsr or processor or bench or i2v" > $O/pytest_vb.txt 2>&1; tail -30 $O/pytest_vb.txt
for i in 1 2; do
  for v in 3 6; do
    timeout 300 python bench_svg2.py --steps 4 --warmup 2 --variant $v 2>>$O/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('svg2 variant $v', d['ms'], d['attention_tflops_algorithmic'], d['spot_rows_rel_l2_vs_torch_fp32'])"
  done
done 2>&1 | tee $O/ab_svg2.txt
timeout 300 python bench_svg2.py --steps 4 --warmup 2 --fp8 2>>$O/ab.err | tail -1 > $O/svg2_fp8.json; cut -c1-500 $O/svg2_fp8.json
timeout 300 python bench_svg2.py --workload hy720p --steps 3 --warmup 1 2>>$O/ab.err | tail -1 > $O/svg2_hy720p.json; cut -c1-500 $O/svg2_hy720p.json
cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt -o svg2 -- python $GRAFT_REPO_ROOT/bench_svg2.py --steps 3 --warmup 1 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -1) $O/svg2_kernel_trace.txt; grep -E "varblock|kernel  " $O/svg2_kernel_trace.txt | cut -c1-150
tail -5 $O/ab.err
