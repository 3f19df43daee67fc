"""tools/native_harness on the GPU box: the band entry points driven through the C ABI from a torch-free C++ / HIP program with its OWN
input generator and fp32 restatement of the masked attention (mask predicate of include/svg_attn.h, placement rule of svg_perm_desc_t) —
a second checker beside the Python oracle, sharing no code with it.  The harness exits non-zero when its spot rows miss the bound it
holds them to (3e-3 bf16, 1e-3 fp16, 1e-2 on the opt-in pre-scaled entry); production-size runs of it are under profiles/ (r04zu_*)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def harness():
    sys.path.insert(0, str(ROOT))
    import __graft_entry__ as entry

    # (tools/native_harness is git-ignored: build() compiles it in the build container and the GPU-box snapshot carries it; anywhere else it is
    #  compiled from tools/native_harness.hip here, and again whenever the source or the C-ABI header is newer)
    return entry.build_native_harness()


CASES = [
    ["--geom", "small"],                                        # HunyuanVideo layout (text last, pads), head_dim 128, default schedule
    ["--geom", "small", "--variant", "2", "--dtype", "f16"],
    ["--geom", "small", "--variant", "3", "--flags", "one"],
    ["--geom", "small", "--prescaled"],
    ["--geom", "small64"],                                      # CogVideoX layout (text first), head_dim 64: four waves per SIMD
    ["--geom", "small64", "--dtype", "f16", "--flags", "one"],
    ["--geom", "small64", "--variant", "3"],
    ["--geom", "small64", "--switch", "0"],                     # svg_band_attention_switch, sparse side
    ["--geom", "small64", "--switch", "1", "--flags", "one"],   # ... dense side: the placement flags must be ignored
    ["--geom", "cog480p", "--heads", "4"],                      # production sequence length of CogVideoX-v1 480p, four heads
    ["--geom", "small", "--fill", "zero"],                      # all-zero operands (the schedule-only ceiling runs of profiles/r05a_*): output exactly 0
    ["--geom", "small", "--band", "256"],                       # band override (tools/history/r05/gpu_r05a.sh band sweep)
]


@pytest.mark.parametrize("args", CASES, ids=lambda a: " ".join(a).replace("--", ""))
def test_native_harness_spot_rows(harness, args):
    r = subprocess.run([str(harness), "--lib", str(ROOT / "sparse-videogen_amd" / "lib" / "libsvgattn.so"), "--warm", "1", "--reps", "2",
                        "--check", "12", *args], capture_output=True, text=True, timeout=120, cwd=str(ROOT))
    assert r.returncode == 0, (r.stdout[-600:], r.stderr[-600:])
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["spot_rows"] >= 24
    assert d["rel_l2"] <= (1e-3 if d["dtype"] == "f16" else (1e-2 if d["prescaled"] else 3e-3)), d


def test_native_harness_switch_equals_plain_call(harness):
    """flag 0 of svg_band_attention_switch is the plain call: identical output bits at head_dim 64, where both are band_attn_pp2_kernel's
    body behind a flag (profiles/r04zp_native_switch_d64.txt).  (At head_dim 128 the two are different instantiations of the 16x16x32 body
    and agree to rounding: tests/test_gpu_kernels.py::test_band_attention_device_switch.)"""
    def checksum(*args):
        r = subprocess.run([str(harness), "--warm", "0", "--reps", "1", "--check", "0", *args], capture_output=True, text=True, timeout=120,
                           cwd=str(ROOT))
        assert r.returncode == 0, (r.stdout[-600:], r.stderr[-600:])
        return json.loads(r.stdout.strip().splitlines()[-1])["o_checksum"]

    for extra in ((), ("--dtype", "f16", "--flags", "one")):
        assert checksum("--geom", "small64", *extra) == checksum("--geom", "small64", "--switch", "0", *extra), extra


def test_native_harness_profiler_mode(harness):
    """--profiler: svg_sample_mse through the C ABI without torch (the instrument of profiles/r05k_*): finite mse values, a plausible time"""
    r = subprocess.run([str(harness), "--lib", str(ROOT / "sparse-videogen_amd" / "lib" / "libsvgattn.so"), "--geom", "small", "--profiler",
                        "--warm", "1", "--reps", "2"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["ms_mean"] > 0 and d["mse_sum"] > 0 and d["mse0"][0] == d["mse0"][0]


def test_native_svg2_small():
    """tools/native_svg2: the SVG2 layer-call (k-means loops, block map, variable-block attention) through the C ABI without torch, one stream
    and two: spot rows against its own fp32 restatement (exit code), identical outputs and block maps in both modes."""
    sys.path.insert(0, str(ROOT))
    import __graft_entry__ as entry

    exe = entry.build_native_svg2()
    outs = []
    for extra in ([], ["--two-streams"]):
        r = subprocess.run([str(exe), "--lib", str(ROOT / "sparse-videogen_amd" / "lib" / "libsvgattn.so"), "--geom", "small", *extra],
                           capture_output=True, text=True, timeout=300, cwd=ROOT)
        assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0]["rel_l2"] <= 4e-3 and outs[0]["spot_rows"] > 0
    assert outs[0]["o_checksum"] == outs[1]["o_checksum"] and outs[0]["map_checksum"] == outs[1]["map_checksum"]
