#!/usr/bin/env python3
"""Build-time check for kernels_fused_r.hip's hand-placed MFMAs.

Its matrix instructions are assembly statements, which the compiler's hazard recogniser does not look into: a vector
instruction (a register-file move, v_accvgpr_write/read, or anything else) that writes an operand of such an MFMA needs
two wait states before it.  The kernel is arranged so that none is ever placed there; this script reads the generated
ISA (hipcc --save-temps) and fails if the arrangement broke: for every MFMA inside an ASMSTART/ASMEND pair it takes the
two instructions in front and refuses a VALU destination that overlaps one of the MFMA's source registers."""
import re
import sys


def regs(tok):
    """'v[4:7]' / 'a12' -> set of ('v', n)."""
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r"([va])(\d+)", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


def check(path):
    prev, in_asm, bad, seen = [], False, [], 0
    for ln, raw in enumerate(open(path), 1):
        line = raw.strip()
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not line or line.startswith(";") or line.startswith(".") or line.endswith(":"):
            continue
        op, _, rest = line.partition(" ")
        ops = [t.strip() for t in rest.split(";")[0].split(",")]
        if in_asm and op.startswith("v_mfma"):
            seen += 1
            src = set().union(*[regs(t.split()[0]) for t in ops[1:] if t])
            for back, (pop, pops, pln) in enumerate(reversed(prev[-2:]), 1):
                if pop.startswith("s_nop"):
                    break                                 # an explicit wait in between: fine
                if pop.startswith("v_") and not pop.startswith("v_mfma") and pops and regs(pops[0].split()[0]) & src:
                    bad.append((ln, line, pln, pop + " " + ", ".join(pops)))
        prev.append((op, ops, ln))
        prev = prev[-4:]
    return seen, bad


if __name__ == "__main__":
    total = 0
    for p in sys.argv[1:]:
        seen, bad = check(p)
        total += seen
        for ln, line, pln, pline in bad:
            print(f"{p}:{ln}: {line}\n    is fed by line {pln}: {pline}")
        if bad:
            sys.exit(1)
    if total == 0:
        sys.exit("no assembly-statement MFMAs found: wrong file?")
    print(f"check_mfma_hazards: {total} hand-placed MFMAs, no vector write within two instructions of an operand")
