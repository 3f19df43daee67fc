#!/bin/bash
# One GPU call that produces a round's evidence on ONE box: the GPU suite, the default bench line, the same command under rocprofv3
# --kernel-trace --stats (summary -> profiles), the torch-free harness pass with the PMC sets of the headline kernel, the SVG2 PMC passes.
#   gpurun --timeout 3000 -- 'bash tools/gpu_round_pass.sh <tag> [what: tests bench trace pmc svg2pmc]'
tag=${1:-r06z}; shift
what=${*:-"tests bench trace pmc svg2pmc"}
O=gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
has() { case " $what " in *" $1 "*) return 0;; *) return 1;; esac; }
if has tests; then
  t0=$(date +%s)
  timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$? $(( $(date +%s) - t0 )) s" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
fi
if has bench; then
  t1=$(date +%s)
  timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? $(( $(date +%s) - t1 )) s"; tail -2 $O/bench.err
  python3 - "$O" <<'PY'
import json, sys
O = sys.argv[1]
try:
    d = json.loads(open(f"{O}/bench.json").read().strip().splitlines()[-1])
except Exception as e:
    print("no bench line", e); sys.exit(0)
for k in ("value", "ms_per_step", "roofline", "clock", "output_checksum"): print(k, d.get(k))
print("ab", {k: v for k, v in d.get("same_box_ab", {}).items() if k != "what"})
for key in ("svg2_wan720p", "svg2_hy720p"):
    b = d.get(key, {}); print(key, b.get("ms"), b.get("kmeans_init_50it_ms"), b.get("attention_tflops_algorithmic"), b.get("error"))
print("other", d.get("svg1_other_models"))
for key in ("denoise_step_hy720p", "denoise_step_wan720p_svg2"):
    b = d.get(key, {}); print(key, {k: v for k, v in b.items() if "per_s" in k or k == "speedup_sparse_vs_dense_step"}, (b.get("sparse_step") or {}).get("step_breakdown_ms"), b.get("error"))
for n, r in (d.get("hbm_kernels", {}).get("kernels") or {}).items(): print("hbm", n, r["ms"], r["GBs"], r["frac_of_8TBs"])
PY
fi
if has trace; then
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o $tag -- python $R/bench.py --steps 5 --warmup 2 --no-cpu --no-dense --no-svg2 --no-step --no-ab --no-hbm > $R/$O/bench_under_rocprof.json 2>/dev/null)
  python3 tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -1) $O/bench_kernel_trace.txt; head -6 $O/bench_kernel_trace.txt | cut -c1-170
fi
if has pmc; then bash tools/gpu_native_pass.sh ${tag} 2>&1 | head -8; fi
if has svg2pmc; then bash tools/gpu_pmc_svg2.sh ${tag}_svg2 2>&1 | grep "varblock_attn" | cut -c1-400; fi
