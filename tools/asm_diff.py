#!/usr/bin/env python3
"""Which kernels of two sets of kept gfx950 listings (build/*.s) differ in their instruction stream?  A refactoring that only removes
dead switches must leave every surviving kernel's listing identical (labels and comments aside).
    python tools/asm_diff.py <dir with the old *gfx950.s> [dir with the new ones, default sparse-videogen_amd/build]"""
import glob
import os
import re
import sys
from pathlib import Path


def funcs(path):
    out, name, body = {}, None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        if line.startswith(".Lfunc_end"):
            out[name] = "\n".join(body)
            name = None
            continue
        t = re.sub(r";.*", "", line).strip()
        if not t or t.startswith("."):
            continue
        body.append(re.sub(r"\.LBB\d+_\d+", "LBB", t))
    return out


def main():
    old = sys.argv[1]
    new = sys.argv[2] if len(sys.argv) > 2 else str(Path(__file__).resolve().parent.parent / "sparse-videogen_amd" / "build")
    bad = 0
    for f in sorted(glob.glob(old + "/*gfx950.s")):
        g = os.path.join(new, os.path.basename(f))
        if not os.path.exists(g):
            print("missing", g)
            continue
        a, b = funcs(f), funcs(g)
        diff = [k for k in a if k in b and a[k] != b[k]]
        gone = [k for k in a if k not in b]
        added = [k for k in b if k not in a]
        print(f"{os.path.basename(f)[:28]:28s} kernels {len(a):3d} -> {len(b):3d}  identical {len([k for k in a if k in b]) - len(diff):3d}  changed {len(diff)}  "
              f"removed {len(gone)}  added {len(added)}")
        for k in diff:
            print("   CHANGED", k[:110])
        for k in gone:
            print("   removed", k[:110])
        for k in added:
            print("   added  ", k[:110])
        bad += len(diff)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
