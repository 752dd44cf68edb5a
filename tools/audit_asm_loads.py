"""Audit of hand-issued LDS loads in a hipcc .s file (cdna_hip_programming.md, "What hipcc does not do", item 1).

An inline-asm `ds_read*` is invisible to hipcc's wait-count bookkeeping: its destination registers count as written when the
statement ends, so under register pressure the compiler may copy, spill or reuse them BEFORE the data has landed.  The kernels
here tie every such destination to a hand-written `s_waitcnt lgkmcnt(..)` statement through "+v" operands; this script checks
the generated code: between an asm load and the next asm wait no other instruction may read or write its destination.

    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -I <csrc> -o x.s <file>.hip && python tools/audit_asm_loads.py x.s
"""
import re
import sys


def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def main(path):
    pending = {}          # register -> line of the asm load that targets it
    in_asm, bad, n_loads = False, 0, 0
    for ln, raw in enumerate(open(path), 1):
        line = raw.split(";")[0].strip() if not raw.strip().startswith(";;#") else raw.strip()
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not line or line.endswith(":") or line.startswith("."):
            if line.endswith(":") and not line.startswith(".LBB"):
                pending.clear()       # a new function
            continue
        op, _, args = line.partition(" ")
        if in_asm:
            if op.startswith("ds_read"):
                n_loads += 1
                for r in regs(args.split(",")[0]):
                    pending[r] = ln
            elif op == "s_waitcnt" and "lgkmcnt" in args:
                pending.clear()       # (every hand-written wait in these kernels covers all outstanding asm loads or names a count that does)
            continue
        touched = regs(args) & set(pending)
        if touched:
            bad += 1
            print(f"{path}:{ln}: `{line}` touches v{sorted(touched)} loaded by the asm ds_read at line {pending[min(touched)]} "
                  "before its wait")
    print(f"{path}: {n_loads} hand-issued LDS loads, {bad} violations")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(max(main(p) for p in sys.argv[1:]))
