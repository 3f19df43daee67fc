#!/usr/bin/env python3
"""Build libsvgattn.so (the C-ABI HIP library) for gfx950, in-tree.

    python sparse-videogen_amd/build.py [--force] [--asm] [--ablations]

Every csrc/*.hip is compiled with hipcc --offload-arch=gfx950 to an object (in parallel), then linked into
sparse-videogen_amd/lib/libsvgattn.so.  hipcc cross-compiles, so this works on a machine without a GPU.
Objects are rebuilt only when a source or header is newer (or with --force).
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
OBJ = ROOT / "build"
LIB = ROOT / "lib" / "libsvgattn.so"
INCLUDE = ROOT.parent / "include"

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = [
    f"--offload-arch={ARCH}",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffast-math",
    "-fno-finite-math-only",  # -inf is used as the masked score; keep inf/nan semantics
    "-Wall",
    "-Wno-unused-function",
    "-Wno-unused-variable",
    "-Wno-unused-but-set-variable",
    "-Wno-unknown-pragmas",
    f"-I{INCLUDE}",
]


# Files whose arithmetic has to reproduce torch's rounding points bit for bit: IEEE semantics, no fma contraction, no
# folding of double -> float -> half conversions.
STRICT_FP = {"prologue.hip", "glue.hip", "dynmap.hip"}
# Files whose gfx950 assembly is always kept (build/<stem>-hip-amdgcn-amd-amdhsa-gfx950.s): their MFMAs are inline asm, which
# hipcc's hazard recognizer does not see — tools/asm_hazards.py audits the listing (tests/test_w4_asm_audit.py).
# (attention.hip / attention_f8.hip: the first contraction step of a tile of the fp8 bodies, csrc/attn_f8.h mfma_qk_first)
KEEP_ASM = {"attention_w4.hip", "attention.hip", "attention_f8.hip"}


def _newest_header() -> float:
    hs = list(CSRC.glob("*.h")) + list(INCLUDE.glob("*.h"))
    return max(h.stat().st_mtime for h in hs)


def _compile(src: Path, force: bool, asm: bool, ablations: bool = False, tag: str = "") -> tuple[Path, str]:
    obj = OBJ / (src.stem + (f".{tag}" if tag else "") + (".abl.o" if ablations else ".o"))
    stamp = max(src.stat().st_mtime, _newest_header())
    if not force and obj.exists() and obj.stat().st_mtime >= stamp:
        return obj, ""
    flags = list(FLAGS) + os.environ.get("SVG_EXTRA_HIPCC_FLAGS", "").split()   # (experiments: -DSVG_... switches of the kernels)
    if src.name in STRICT_FP:
        flags = [f for f in flags if f not in ("-ffast-math", "-fno-finite-math-only")] + ["-fno-fast-math", "-ffp-contract=off"]
    if ablations:
        flags.append("-DSVG_ABLATIONS")
    cmd = [HIPCC, *flags, "-c", str(src), "-o", str(obj)]
    # (the kept listing is named after the source: only the PRODUCT build writes it — an ablation or tagged A/B build would overwrite
    #  the file the audit tests read with the listing of a different binary)
    if asm or (src.name in KEEP_ASM and not ablations and not tag):
        cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=str(OBJ))
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
    return obj, (r.stdout + r.stderr)


def _summarise(logs: str) -> str:
    """Condense -Rpass-analysis=kernel-resource-usage remarks to one line per kernel; keep warnings verbatim."""
    import re

    out, cur = [], {}
    for line in logs.splitlines():
        m = re.search(r"remark: .*?: +(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|"
                      r"VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
        if m:
            k, v = m.group(1), m.group(2)
            if k == "Function Name":
                cur = {"name": v}
            else:
                cur[k] = v
                if k.startswith("LDS"):
                    out.append("  {name}: vgpr={VGPRs} agpr={AGPRs} sgpr={TotalSGPRs} scratch={s} spill={sp} occ={o}".format(
                        name=cur.get("name", "?")[:90], VGPRs=cur.get("VGPRs"), AGPRs=cur.get("AGPRs"),
                        TotalSGPRs=cur.get("TotalSGPRs"), s=cur.get("ScratchSize [bytes/lane]"),
                        sp=cur.get("VGPRs Spill"), o=cur.get("Occupancy [waves/SIMD]")))
        elif "remark:" not in line and line.strip():
            out.append(line)
    return "\n".join(out)


def build(force: bool = False, asm: bool = False, verbose: bool = True, ablations: bool = False, tag: str = "") -> Path:
    """ablations=True builds lib/libsvgattn_abl.so with -DSVG_ABLATIONS: the traced / timing-ablation kernels of the diagnostics
    tools (tools/pp_trace.py, tools/wg_timeline.py; select it with SVG_ATTN_LIB).  The product library never contains them."""
    OBJ.mkdir(exist_ok=True)
    LIB.parent.mkdir(exist_ok=True)
    lib = LIB.with_name("libsvgattn_abl.so") if ablations else LIB
    if tag:   # comparison build for same-box A/B runs (tools/ab_*.sh; SVG_EXTRA_HIPCC_FLAGS carries its -DSVG_... switches)
        lib = LIB.with_name(f"libsvgattn_{tag}.so")
    srcs = sorted(CSRC.glob("*.hip"))
    if not srcs:
        raise RuntimeError("no HIP sources found")
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force, asm, ablations, tag), srcs))
    objs = [o for o, _ in results]
    logs = "".join(l for _, l in results)
    if verbose and logs.strip():
        print(_summarise(logs))
    newest_obj = max(o.stat().st_mtime for o in objs)
    if force or not lib.exists() or lib.stat().st_mtime < newest_obj:
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(lib), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"linked {lib} ({lib.stat().st_size / 1e6:.1f} MB)")
    return lib


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--asm", action="store_true", help="keep .s files and print register usage")
    ap.add_argument("--ablations", action="store_true", help="diagnostics library lib/libsvgattn_abl.so (-DSVG_ABLATIONS)")
    ap.add_argument("--tag", default="", help="comparison build lib/libsvgattn_<tag>.so (flags from SVG_EXTRA_HIPCC_FLAGS)")
    a = ap.parse_args()
    try:
        build(force=a.force, asm=a.asm, ablations=a.ablations, tag=a.tag)
    except RuntimeError as e:
        print(e, file=sys.stderr)
        sys.exit(1)
