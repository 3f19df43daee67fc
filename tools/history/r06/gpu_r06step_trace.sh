#!/bin/bash
# rocprofv3 --kernel-trace --stats over one measured denoise step of each stack (bench_step.py, 1 warm-up + 1 timed step of the sparse and the dense
# kind): which kernels the step time is made of.  Only the summaries leave the box (the trace databases are deleted).
tag=${1:-r06zzz}; O=gpurun_out/${tag}_step_trace; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for m in wan720p hy720p; do
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/kt_$m -o $m -- python $R/bench_step.py --model $m --steps 1 --warmup 1 > $R/$O/step_$m.json 2> $R/$O/step_$m.err)
  echo "$m rc=$?"
  python3 tools/rocprof_summary.py $(find $O/kt_$m -name "*.db" | head -1) $O/step_${m}_kernel_trace.txt
  head -14 $O/step_${m}_kernel_trace.txt | cut -c1-190
  rm -rf $O/kt_$m
done
