#!/bin/bash
# round 3, GPU call 1: parity suite on the new code, same-box A/Bs of the two-phase body (carried operands 0 / 4 / 8, pre-scaled q)
# and of the SVG2 launch orders, then the default bench line.   gpurun --timeout 1500 -- 'bash tools/r03_call1.sh'
O=gpurun_out/r03a; mkdir -p $O
L=$PWD/sparse-videogen_amd/lib
{ ls /sys/class/drm/; for c in /sys/class/drm/card*/device; do echo "== $c"; ls $c | tr '\n' ' '; ls $c/hwmon/*/ 2>/dev/null | tr '\n' ' '; cat $c/pp_dpm_sclk 2>/dev/null; cat $c/hwmon/*/freq1_input 2>/dev/null; done; rocm-smi --showclocks; python -c "import amdsmi; print('amdsmi ok')"; } > $O/clock_probe.txt 2>&1
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -5 $O/pytest_gpu.txt
B="python bench.py --steps 6 --warmup 2 --no-cpu --no-profiler --no-dense --no-svg2 --no-step --no-ab"
for i in 1 2; do
  for t in cur c0 c8; do
    [ "$t" = "cur" ] && f=libsvgattn.so || f=libsvgattn_$t.so
    for extra in "" "--prescaled"; do
      SVG_ATTN_LIB=$L/$f timeout 300 $B $extra 2>>$O/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t $extra', d['roofline']['kernel_ms'], d.get('prescaled_rel_l2_vs_default_kernel'), d.get('clock'))"
    done
  done
done 2>&1 | tee $O/ab_pp2.txt
for i in 1 2; do
  for v in 3 6; do
    timeout 300 python bench_svg2.py --steps 4 --warmup 2 --variant $v 2>>$O/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('svg2 variant $v', d['ms'], d['attention_tflops_algorithmic'], d['spot_rows_rel_l2_vs_torch_fp32'])"
  done
done 2>&1 | tee $O/ab_svg2.txt
timeout 300 python bench_svg2.py --steps 4 --warmup 2 --fp8 2>>$O/ab.err | tail -1 > $O/svg2_fp8.json; cut -c1-600 $O/svg2_fp8.json
timeout 300 python bench_svg2.py --workload hy720p --steps 3 --warmup 1 2>>$O/ab.err | tail -1 > $O/svg2_hy720p.json; cut -c1-600 $O/svg2_hy720p.json
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
