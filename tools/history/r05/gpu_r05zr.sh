#!/bin/bash
# Round 5, GPU call zr: the Wan block glue split into its parts (LayerNorm only, modulate only, both) at [75600, 5120] bf16; profiler A/B bits after the reference bias
tag=${1:-r05zr}; O=gpurun_out/$tag; mkdir -p $O
for g in hy720p cog15; do timeout 60 tools/native_harness --lib sparse-videogen_amd/lib/libsvgattn.so --geom $g --profiler --reps 10 > $O/prof_$g.json 2>/dev/null; python3 -c "
import json; d=json.load(open('$O/prof_$g.json')); print('$g', d['ms_mean'], d['mse_sum'], d['mse_bits'])"; done
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/glue_parts.txt
import sys, torch
sys.path.insert(0, "sparse-videogen_amd")
from svg import _native as nat
nat.load()
dev = torch.device("cuda")
S, hid = 75600, 5120
hs = torch.randn(1, S, hid, device=dev, dtype=torch.bfloat16)
sc, sh = (torch.randn(1, 1, hid, device=dev) * 0.2 for _ in range(2))
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
el = S * hid
for name, fn in (("layernorm+modulate", lambda: nat.layernorm_modulate_forward(hs, None, None, sc, sh, 1e-6)),
                 ("layernorm only", lambda: nat.layernorm_forward(hs, None, None, 1e-6, out_dtype=torch.bfloat16)),
                 ("modulate only", lambda: nat.modulate_shift_forward(hs, sc, sh, out_dtype=torch.bfloat16)),
                 ("copy", lambda: hs.clone())):
    ms = t(fn)
    print(f"{name:20s} {ms:.4f} ms  {4 * el / ms / 1e6:.0f} GB/s")
PY
