#!/bin/bash
# The 1 -> 8 GPU curve of BASELINE.json configs[3] in one command, for the first minutes on an 8-GPU MI355X node (no node was available to any
# round so far: NOT RUN ON HARDWARE; its pieces are exercised at N = 2 on one GPU over gloo by tests/test_gpu_bench_contract.py).
#   bash tools/scale_run.sh [out_dir] [Ns, default "1 2 4 8"]
# For every N: `bench.py --gpus N` (the SVG1 layer-call, head-sharded with the overlapped all-gather), `bench_step.py --gpus N` (the
# token-sharded HunyuanVideo SVG1 denoise step) and `bench_step.py --model wan720p --gpus N` (the Wan 2.1 SVG2 denoise step, configs[2]) under
# torch.distributed.run over RCCL; then checks
#   * exchange.rccl_ranks_seen == N (every rank of the communicator answered an all-reduce),
#   * output_checksum(N) == output_checksum(1): the gathered layer-call output is the one-rank output bit for bit (inputs are seeded per global head),
#   * no waiter timed out / no fallback to chunk launches (reported, not fatal),
# and prints one JSON line per N: {"n_gpus", "attn_tflops", "ms_per_step", "denoise_steps_per_s", "scaling_efficiency_vs_1", ...}.
set -u
OUT=${1:-gpurun_out/scale}; NS=${2:-"1 2 4 8"}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
PORT=29600
for N in $NS; do
  PORT=$((PORT + 2))
  if [ "$N" = 1 ]; then
    python bench.py --gpus 1 --steps 10 --warmup 3 --no-svg2 --no-hbm > "$OUT/bench_$N.json" 2> "$OUT/bench_$N.err"
    python bench_step.py --steps 3 --warmup 1 > "$OUT/step_$N.json" 2> "$OUT/step_$N.err"
    python bench_step.py --model wan720p --steps 2 --warmup 1 > "$OUT/step_wan_$N.json" 2> "$OUT/step_wan_$N.err"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus "$N" --steps 10 --warmup 3 \
      > "$OUT/bench_$N.json" 2> "$OUT/bench_$N.err"
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((PORT + 1)) bench_step.py --gpus "$N" --steps 3 --warmup 1 \
      > "$OUT/step_$N.json" 2> "$OUT/step_$N.err"
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((PORT + 50)) bench_step.py --model wan720p --gpus "$N" --steps 2 --warmup 1 \
      > "$OUT/step_wan_$N.json" 2> "$OUT/step_wan_$N.err"
  fi
  echo "N=$N bench rc=$? (logs: $OUT/bench_$N.err, $OUT/step_$N.err)" >&2
done
python3 - "$OUT" $NS <<'PY'
import json, sys
out, ns = sys.argv[1], [int(x) for x in sys.argv[2:]]
def last_json(path):
    try:
        lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except OSError:
        return None
base = None
bad = 0
for n in ns:
    b, s = last_json(f"{out}/bench_{n}.json"), last_json(f"{out}/step_{n}.json")
    if b is None:
        print(json.dumps({"n_gpus": n, "error": f"no bench line, see {out}/bench_{n}.err"})); bad += 1; continue
    row = {"n_gpus": n, "attn_tflops": b["value"], "ms_per_step": b["ms_per_step"], "output_checksum": b.get("output_checksum")}
    if n == 1:
        base = b
    ex = b.get("exchange") or {}
    if n > 1:
        row["rccl_ranks_seen"] = ex.get("rccl_ranks_seen")
        row["fallback_to_chunk_launches"] = ex.get("fallback_to_chunk_launches")
        row["waiter_timeouts_in_timed_steps"] = ex.get("waiter_timeouts_in_timed_steps")
        if ex.get("rccl_ranks_seen") != n:
            row["error"] = f"rccl_ranks_seen {ex.get('rccl_ranks_seen')} != {n}"; bad += 1
        if base is not None:
            row["scaling_efficiency_vs_1"] = round(b["value"] / (n * base["value"]), 4)
            row["output_equals_one_rank"] = b.get("output_checksum") == base.get("output_checksum")
            if not row["output_equals_one_rank"]:
                row["error"] = "gathered output differs from the one-rank run"; bad += 1
    sd = (s or {})
    row["denoise_steps_per_s"] = sd.get("denoise_steps_per_s", (b.get("denoise_step_hy720p") or {}).get("denoise_steps_per_s"))
    sw = last_json(f"{out}/step_wan_{n}.json") or {}
    row["denoise_steps_per_s_wan720p_svg2"] = sw.get("denoise_steps_per_s", (b.get("denoise_step_wan720p_svg2") or {}).get("denoise_steps_per_s"))
    print(json.dumps(row))
sys.exit(1 if bad else 0)
PY
