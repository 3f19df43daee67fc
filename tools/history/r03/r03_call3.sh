#!/bin/bash
O=gpurun_out/r03c; mkdir -p $O
L=$PWD/sparse-videogen_amd/lib
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "varblock or band_attention" -x > $O/pytest_kernels.txt 2>&1; tail -3 $O/pytest_kernels.txt
B="python bench.py --steps 6 --warmup 2 --no-cpu --no-profiler --no-dense --no-svg2 --no-step --no-ab"
for i in 1 2 3; do
  for t in cur nd; do
    [ "$t" = "cur" ] && f=libsvgattn.so || f=libsvgattn_$t.so
    for extra in "" "--prescaled"; do
      SVG_ATTN_LIB=$L/$f timeout 300 $B $extra 2>>$O/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t $extra', d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('prescaled_rel_l2_vs_default_kernel'), d['clock']['sclk_mhz_median'])"
    done
  done
done 2>&1 | tee $O/ab_pp2.txt
SVG_ATTN_LIB=$L/libsvgattn_abl.so timeout 300 python tools/pp_trace.py 15616 0,3,8 > $O/pp_trace.txt 2>&1; cat $O/pp_trace.txt
for v in 3 6; do
  timeout 300 python bench_svg2.py --steps 4 --warmup 2 --variant $v 2>>$O/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('svg2 variant $v', d['ms'], d['attention_tflops_algorithmic'], d['spot_rows_rel_l2_vs_torch_fp32'])"
done 2>&1 | tee $O/ab_svg2.txt
cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt -o svg2 -- python $GRAFT_REPO_ROOT/bench_svg2.py --steps 3 --warmup 1 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -1) $O/svg2_kernel_trace.txt; head -12 $O/svg2_kernel_trace.txt | cut -c1-160
