#!/usr/bin/env python3
"""Per basic block of a kernel in the kept gfx950 assembly (build.py --asm): instruction-class histogram of every block that
holds MFMAs — MFMA / VALU / v_accvgpr moves / LDS / VMEM / SALU / waits / scratch — i.e. the issue budget per MFMA gap of a
software-pipelined body (csrc/attn_w4.h wants <= ~5 non-MFMA instructions per MFMA).

    python tools/asm_blocks.py <kernel substring> [file.s] [--dump N]     (--dump N prints block N)
"""
import collections
import re
import sys
from pathlib import Path

BUILD = Path(__file__).resolve().parent.parent / "sparse-videogen_amd" / "build"


def classify(op: str) -> str:
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "acc"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op in ("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_setprio") or op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu"
    return "other"


def main():
    argv = sys.argv[1:]
    dump = None
    if "--dump" in argv:
        i = argv.index("--dump")
        dump = int(argv[i + 1])
        del argv[i:i + 2]
    pat = argv[0] if argv else "w4_kernel"
    files = [Path(argv[1])] if len(argv) > 1 else sorted(BUILD.glob("*-hip-amdgcn-amd-amdhsa-gfx950.s"))
    for f in files:
        s = f.read_text()
        for m in re.finditer(r"^(_Z\w+):\s*; @", s, re.M):
            name = m.group(1)
            if pat not in name:
                continue
            end = s.index(".Lfunc_end", m.end())
            blocks, cur, label = [], [], "entry"
            for raw in s[m.end():end].split("\n"):
                l = raw.strip()
                if not l or l.startswith(";") or l.startswith("."):
                    if l.startswith(".LBB"):
                        blocks.append((label, cur))
                        label, cur = l.split(":")[0], []
                    continue
                cur.append(l)
                if l.startswith(("s_cbranch", "s_branch")):      # a branch ends the block: what follows is the fall-through path
                    blocks.append((label, cur))
                    label, cur = label + "+", []
            blocks.append((label, cur))
            print(name[:110])
            tot = collections.Counter()
            for i, (lab, ins) in enumerate(blocks):
                c = collections.Counter(classify(x.split()[0]) for x in ins)
                tot.update(c)
                if c["mfma"] >= 4:
                    fill = sum(v for k, v in c.items() if k != "mfma")
                    ops = collections.Counter(x.split()[0] for x in ins)
                    print(f"  block {i:3d} {lab:10s} n={len(ins):4d} mfma={c['mfma']:3d} valu={c['valu']:3d} acc={c['acc']:3d} lds={c['lds']:3d} "
                          f"vmem={c['vmem']:2d} salu={c['salu']:3d} wait={c['wait']:3d} scratch={c['scratch']:3d}  fill/mfma={fill / c['mfma']:.2f}"
                          f"  exp={ops['v_exp_f32']} cvt={ops['v_cvt_pk_bf16_f32'] + ops['v_cvt_pk_f16_f32'] + ops['v_cvt_pkrtz_f16_f32']} mov={ops['v_mov_b32'] + ops['v_mov_b64']} nop={ops['s_nop']}")
                if dump == i:
                    print("\n".join("      " + x for x in ins))
            print("  total:", dict(tot))


if __name__ == "__main__":
    main()
