#!/usr/bin/env python3
"""tools/arms_summary.py <log of tools/kstats_arms.sh> -- one line per arm: the SpMV kernels' averages and the bench line."""
import re
import sys
for l in open(sys.argv[1]):
    n = l.split()[0] if l.strip() else ""
    ks = re.findall(r"(spmv_tiled_phase1|spmv_tiled_phase2s|spmv_csr_kernel|bits_\w+|tiled_mark_dead) ([\d.]+) us", l)
    m3 = re.search(r"\| (bench.*)$", l)
    if ks:
        print(f"{n:12s}", "  ".join(f"{k.replace('spmv_tiled_', '')} {v}" for k, v in ks), "|", m3.group(1) if m3 else "")
