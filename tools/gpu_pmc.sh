#!/bin/bash
# PMC counter passes for the dominant kernel (run on the GPU box through gpurun).  Each pass is its own rocprofv3 run
# with --kernel-trace only (gpurun refuses --pmc combined with sys/hip traces).  usage: tools/gpu_pmc.sh <tag> [bench args]
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
mkdir -p $out
# PMC_CMD: the profiled command (default: the bench's headline launch; tools/native_harness for a torch-free pass that costs seconds),
# PMC_PASS_TIMEOUT: seconds per pass, PMC_ORDER: which sets, in which order (a short call takes the traffic sets first: "4 5 1 3 2 6").
B=${PMC_CMD:-"python bench.py --steps 1 --warmup 1 --no-cpu --no-dense --no-profiler --no-ab $@"}
[ -z "$PMC_CMD" ] && rocprofv3 -L > $out/counters_list.txt 2>&1
sets=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY"
 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE"
 "FETCH_SIZE"
 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum" )
for i in ${PMC_ORDER:-1 2 3 4 5 6}; do
  set=${sets[$((i-1))]}
  t0=$(date +%s)
  timeout ${PMC_PASS_TIMEOUT:-300} rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o p$i -- $B > $out/p$i.log 2>&1
  echo "pass $i rc=$? $(( $(date +%s) - t0 )) s : $set" >> $out/passes.txt
done
python tools/pmc_summary.py $out
