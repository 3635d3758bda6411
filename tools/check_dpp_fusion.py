#!/usr/bin/env python3
"""Build-time check of kernels_bdft.hip's ISA: the sliding sums' lane shifts must arrive INSIDE their additions (v_add_f32_dpp /
v_sub_f32_dpp), not as moves of their own -- what the SLP vectoriser, a contraction of the blocks' products into the sums, or a
re-association of `(lead + shifted) + shifted` silently brings back (8 % of the kernel's time: MEASUREMENTS R5.4).
    tools/check_dpp_fusion.py file.s [max row-shift moves per kernel = 120] [min shifted additions per kernel = 150]"""
import re, sys

def check(path, max_moves=120, min_fused=150):
    cur, counts = None, {}
    for line in open(path):
        if line.startswith("_Z") and ":" in line and "@" in line:
            cur = line.split(":")[0]
            counts[cur] = [0, 0]
        elif cur:
            if "v_mov_b32_dpp" in line and ("row_shr" in line or "row_shl" in line):
                counts[cur][0] += 1
            elif re.search(r"\bv_(add|sub|subrev)_f32_dpp\b", line):
                counts[cur][1] += 1
    bad = ["%s: %d row-shift moves (at most %d), %d shifted additions (at least %d)" % (k, m, max_moves, f, min_fused)
           for k, (m, f) in counts.items() if "bdft_net_kernel" in k and (m > max_moves or f < min_fused)]
    return counts, bad

if __name__ == "__main__":
    a = sys.argv[1:]
    counts, bad = check(a[0], int(a[1]) if len(a) > 1 else 120, int(a[2]) if len(a) > 2 else 150)
    if not any("bdft_net_kernel" in k for k in counts):
        sys.exit("check_dpp_fusion: no bdft_net_kernel in " + a[0])
    if bad:
        sys.exit("check_dpp_fusion: the sliding sums' shifts are moves again:\n  " + "\n  ".join(bad))
    print("check_dpp_fusion: %d kernels, shifts folded into their additions" % sum("bdft_net_kernel" in k for k in counts))
