#!/usr/bin/env python3
"""Inspect the gfx950 assembly kept by `build.py --asm`: per kernel, list MFMA count and every vmcnt wait
between the first and last MFMA (a vmcnt(0) there means the KV prefetch is serialised behind compute)."""
import collections
import re
import sys
from pathlib import Path

BUILD = Path(__file__).resolve().parent.parent / "sparse-videogen_amd" / "build"


def kernels(s):
    for m in re.finditer(r"^(_Z\w+):\s*; @", s, re.M):
        name = m.group(1)
        end = s.index(".Lfunc_end", m.end())
        yield name, [l.strip() for l in s[m.end():end].split("\n") if l.strip() and not l.strip().startswith(";")]


def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else "attn"
    for f in sorted(BUILD.glob("*-hip-amdgcn-amd-amdhsa-gfx950.s")):
        s = f.read_text()
        for name, lines in kernels(s):
            if pat not in name:
                continue
            mf = [n for n, l in enumerate(lines) if l.startswith("v_mfma")]
            cnt = collections.Counter(l.split()[0] for l in lines if not l.endswith(":"))
            print(f"{name[:100]}\n   instrs={len(lines)} mfma={len(mf)} tr={cnt['ds_read_b64_tr_b16']} b128={cnt['ds_read_b128']} "
                  f"gload={cnt['global_load_dwordx4']} saveexec={cnt['s_and_saveexec_b64']} scratch={cnt['scratch_load_dword']}")
            if mf:
                waits = [(n - mf[0], l) for n, l in enumerate(lines) if "vmcnt" in l and mf[0] - 5 <= n]
                print("   vmcnt waits from first mfma:", waits)


if __name__ == "__main__":
    main()
