#!/usr/bin/env python3
"""Build-time check on the ISA of every kernel file (hipcc --save-temps; csrc/Makefile runs it on each *.hip).

1. Packed fp32 arithmetic must not take the HIGH register of src1 into its LOW half (op_sel:[x,1,...] on v_pk_fma_f32 /
   v_pk_mul_f32 / v_pk_add_f32).  On MI355X that form intermittently returns the low half WITHOUT its product (or sum) in lanes
   48-63 while another wave of the SIMD executes matrix instructions -- measured in isolation by tools/ubench/pkfma_opsel.hip
   (up to 3 % of the instructions executed; the same selection on src0 or src2, op_sel_hi, and no selection never fail), found as
   the cause of round 4's run-to-run differences of the staggered wide GEMM by tools/debug/wide_trace.py (MEASUREMENTS R5.1).
   The compiler makes the form by itself when it packs two scalar multiply-adds whose common factor is the odd element of a
   loaded vector; writing that factor as the FIRST multiplicand puts the selection on src0.

2. For kernels that issue LDS-DMA through inline assembly (kernels_wide.hip: the compiler sees neither the LDS write nor the
   outstanding load): M0 may appear only inside those assembly statements, and every s_barrier in a function with such a
   statement must have an s_waitcnt vmcnt(0) in front of it in its basic block."""
import re
import sys

PK = re.compile(r"^\s*v_pk_(fma|mul|add)_f32\b(.*)$")


def check(path):
    bad = []
    func, in_asm, has_dma, since_label = None, False, {}, []
    pending_barriers, m0_outside = [], []
    n_pk = 0
    for ln, raw in enumerate(open(path), 1):
        line = raw.split(";")[0].rstrip() if not raw.lstrip().startswith(";;#") else raw.strip()
        s = line.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s:
            continue
        m = re.match(r"^([A-Za-z_$][\w$.]*):", s)
        if m and not s.startswith(".L"):
            func = m.group(1)
            since_label = []
            continue
        if s.startswith(".L") and s.endswith(":"):
            since_label = []
            continue
        if s.startswith("."):
            continue
        mm = PK.match(s)
        if mm:
            n_pk += 1
        # (measured on the packed fp32 instructions; refused on every instruction that selects operand halves -- v_fma_mix_*,
        # packed f16, dot products -- since none of the kernels needs the form)
        sel = re.search(r"\bop_sel:\[([01]),([01])", s)
        if sel and sel.group(2) == "1" and s.startswith("v_"):
            bad.append("%s:%d: %s   <- src1's high half into the low half of the result" % (path, ln, s))
        if re.search(r"\bm0\b", s) and not in_asm:
            m0_outside.append((func, "%s:%d: %s   <- M0 outside the LDS-DMA assembly statements of this function" % (path, ln, s)))
        if in_asm and re.search(r"\blds\b", s) and s.startswith(("buffer_load", "global_load")):
            has_dma[func] = True
        if s.startswith("s_barrier"):
            waited = any(re.match(r"s_waitcnt\b.*vmcnt\(0\)", t) for t in since_label)
            pending_barriers.append((func, ln, waited or not has_dma.get(func)))     # (barriers in front of the function's first DMA: the prologue's)
        if s.startswith(("s_cbranch", "s_branch")):
            since_label = []
        else:
            since_label.append(s)
    bad += [msg for func, msg in m0_outside if has_dma.get(func)]     # (the compiler's own LDS-DMA builtin manages M0 itself: other functions)
    for func, ln, waited in pending_barriers:
        if has_dma.get(func) and not waited:
            bad.append("%s:%d: s_barrier in %s without s_waitcnt vmcnt(0) in front of it in its block (LDS-DMA by assembly in this function)" % (path, ln, func))
    return bad, n_pk, sum(1 for f in has_dma)


if __name__ == "__main__":
    rc = 0
    for p in sys.argv[1:]:
        bad, n_pk, n_dma = check(p)
        for b in bad:
            print(b)
        if bad:
            rc = 1
        else:
            print("check_pk_opsel: %s: %d packed fp32 instructions, none selects src1's high register into the low half%s" % (
                p.rsplit("/", 1)[-1], n_pk, "; %d functions with LDS-DMA by assembly: every barrier behind vmcnt(0), M0 theirs alone" % n_dma if n_dma else ""))
    sys.exit(rc)
