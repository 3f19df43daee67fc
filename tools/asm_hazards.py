#!/usr/bin/env python3
"""Audit the kept gfx950 assembly (build.py --asm) for MFMA hazards hipcc cannot see: the MFMAs of csrc/attn_w4.h are inline
asm, so the compiler's hazard recognizer pads nothing around them (cdna_hip_programming.md §5.7 item 2).  Checked per kernel:

  W->M   a VALU / v_accvgpr write of a register that an MFMA reads as A, B or C operand fewer than 2 wait states earlier
         (e.g. a compiler spill reload placed directly in front of an asm MFMA)
  M->R   a non-MFMA instruction reading (or overwriting) the VGPR / AGPR destination of an MFMA fewer than `--mfma-states`
         (default 18) wait states later, an MFMA that accumulates into the same tuple excepted

  T->A   an instruction INSIDE an asm statement reading the result of a transcendental (v_exp_f32, v_rcp_f32 ...) issued fewer than
         2 wait states earlier: the transcendental unit delivers some lane groups late and hipcc pads this hazard only for
         instructions it knows (seen on gfx950: stale operands in every other quad of lanes)
  AGPR   a scratch_* instruction, or a v_accvgpr_* instruction OUTSIDE an asm statement that touches a0 .. a223: those registers are
         owned by the asm statements (csrc/attn_w4_agpr.inc).  hipcc parking values of its own in a224 .. a255 is reported as a
         count ("spill moves": harmless for the owned registers; a reload directly in front of an asm MFMA shows up as W->M)

A wait state is one issued instruction; `s_nop N` counts N + 1.  Branches end a window (conservatively clean).
    python tools/asm_hazards.py [kernel substring] [file.s]      exit code 1 if a hazard is found
"""
import re
import sys
from pathlib import Path

BUILD = Path(__file__).resolve().parent.parent / "sparse-videogen_amd" / "build"
OWNED_AGPRS = 224   # a0 .. a223: O^T, Q, row sums (tools/gen_w4_agpr.py)
REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(2) is not None:
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(1), r) for r in range(int(m.group(3)), int(m.group(4)) + 1))
    return out


def split_ops(line):
    op, _, rest = line.partition(" ")
    rest = rest.split(";")[0]
    return op, [x.strip() for x in rest.split(",")]


def states(line):
    op, ops = split_ops(line)
    if op == "s_nop":
        return int(ops[0], 0) + 1
    return 1


def writes(line):
    """registers a vector instruction writes (first operand; stores / compares / DMA write none)"""
    op, ops = split_ops(line)
    if not (op.startswith("v_") or op.startswith("ds_read") or op.startswith("global_load") or op.startswith("scratch_load")
            or op.startswith("buffer_load")):
        return set()
    if op.startswith("v_cmp") or op.startswith("global_load_lds"):
        return set()
    w = regs(ops[0])
    if op.startswith("v_permlane") or op.startswith("v_swap"):
        w |= regs(ops[1])
    return w


def reads(line):
    op, ops = split_ops(line)
    if op.startswith(("v_", "ds_", "global_", "scratch_", "buffer_")):
        src = ops[1:] if not (op.startswith("v_cmp") or "store" in op or op.startswith("global_load_lds") or op.startswith("ds_write")) else ops
        return set().union(*[regs(x) for x in src]) if src else set()
    return set()


def audit(name, lines, mfma_states, only=None):
    """only: set of line indices to audit (the MFMAs inside asm statements), None = every MFMA"""
    bad = []
    n = len(lines)
    for i, l in enumerate(lines):
        if not l.startswith("v_mfma") or (only is not None and i not in only):
            continue
        op, ops = split_ops(l)
        dst, srcs = regs(ops[0]), set().union(*[regs(x) for x in ops[1:6]])   # A, B, C (+ the two block-scale words of v_mfma_scale_*)
        # W->M: look back 2 wait states
        st, j = 0, i - 1
        while j >= 0 and st < 2:
            p = lines[j]
            if p.startswith(("s_cbranch", "s_branch", "s_endpgm")) or p.endswith(":"):
                break
            if (p.startswith("v_") and not p.startswith("v_mfma")) and (writes(p) & srcs):
                bad.append((i, "W->M", p, l))
            st += states(p)
            j -= 1
        # M->R: look ahead
        st, j = 0, i + 1
        while j < n and st < mfma_states:
            p = lines[j]
            if p.startswith(("s_cbranch", "s_branch", "s_endpgm")) or p.endswith(":"):
                break
            if p.startswith("v_mfma"):
                pop, pops = split_ops(p)
                if regs(pops[0]) == dst and regs(pops[3]) == dst:   # accumulate chain on the same tuple: no wait needed
                    st += 8   # an MFMA in between holds the pipe for 8 passes
                    j += 1
                    continue
                if (regs(pops[1]) | regs(pops[2]) | regs(pops[3]) | regs(pops[0])) & dst:
                    bad.append((i, "M->R", p, l))
                st += 8
                j += 1
                continue
            if (reads(p) | writes(p)) & dst:
                bad.append((i, "M->R", p, l))
            st += states(p)
            j += 1
    return bad


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    mfma_states = 18
    for a in sys.argv[1:]:
        if a.startswith("--mfma-states="):
            mfma_states = int(a.split("=")[1])
    pat = argv[0] if argv else "w4"
    # --asm-mfma: kernels with VGPR-destination MFMAs too, and only the MFMAs written as inline asm are audited (csrc/attn_f8.h:
    # the first contraction step of a tile; every other MFMA there is a builtin hipcc pads itself)
    asm_only = "--asm-mfma" in sys.argv[1:]
    files = [Path(argv[1])] if len(argv) > 1 else sorted(BUILD.glob("*-hip-amdgcn-amd-amdhsa-gfx950.s"))
    total = 0
    for f in files:
        s = f.read_text()
        for m in re.finditer(r"^(_Z\w+):\s*; @", s, re.M):
            name = m.group(1)
            if pat not in name:
                continue
            end = s.index(".Lfunc_end", m.end())
            lines, in_asm, foreign, spill_moves, asm_lines = [], False, [], 0, set()
            for raw in s[m.end():end].split("\n"):
                l = raw.strip()
                if l.startswith(";;#ASMSTART"):
                    in_asm = True
                elif l.startswith(";;#ASMEND"):
                    in_asm = False
                if not l or l.startswith(";") or (l.startswith(".") and not l.startswith(".LBB")):
                    continue
                asm_lines.add(len(lines)) if in_asm else None
                if not in_asm and l.startswith("scratch_"):
                    foreign.append(l)
                elif not in_asm and l.startswith("v_accvgpr"):
                    if any(k == "a" and r < OWNED_AGPRS for k, r in regs(l)):
                        foreign.append(l)
                    else:
                        spill_moves += 1
                lines.append(l)
            if asm_only:
                only = {i for i in asm_lines if lines[i].startswith("v_mfma")}
                if not only:
                    continue
                bad = audit(name, lines, mfma_states, only)
            else:
                if not any(l.startswith("v_mfma") and "a[" in l.split(",")[0] for l in lines):
                    continue   # kernels whose MFMAs are compiler builtins: hipcc pads those itself
                bad = audit(name, lines, mfma_states)
            TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
            for i, l in enumerate(lines):
                if l.startswith(TRANS):
                    w, st, j = writes(l), 0, i + 1
                    while j < len(lines) and st < 2:
                        pl = lines[j]
                        if pl.startswith(("s_cbranch", "s_branch")) or pl.endswith(":"):
                            break
                        if j in asm_lines and pl.startswith("v_") and (reads(pl) & w):
                            bad.append((i, "T->A", pl, l))
                        st += states(pl)
                        j += 1
            bad += [(-1, "AGPR", l, "outside an asm statement") for l in foreign]
            n_mfma = len(only) if asm_only else sum(l.startswith('v_mfma') for l in lines)
            print(f"{name[:100]}: {n_mfma} {'asm ' if asm_only else ''}MFMAs, {len(bad)} hazards, {spill_moves} spill moves")
            for i, kind, p, l in bad[:12]:
                print(f"   {kind} line {i}: `{p}`  vs  `{l}`")
            total += len(bad)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
