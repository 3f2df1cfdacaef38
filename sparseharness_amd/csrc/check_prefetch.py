#!/usr/bin/env python3
"""check_prefetch.py <engine.s> -- static check of the hand-scheduled loader of spmv_tiled_phase2s.

The loader issues its loads through inline asm and waits for them with a hand-counted `s_waitcnt vmcnt(8)`: the
compiler sees the destination registers as valid right behind the load.  That is only sound while no instruction
READS a destination register between the load and the wait that covers it (a copy made by the register allocator
would copy stale bits).  This script walks every spmv_tiled_phase2s kernel in the ISA text: asm-issued global loads
enter a FIFO (loads return in issue order), an asm `s_waitcnt vmcnt(N)` retires all but the youngest N, a compiler
`s_waitcnt vmcnt(M)` likewise, and any other instruction that names a register still in flight is reported.
Labels do not reset the FIFO: the loader's loop body is laid out in issue order (prologue, then the unrolled steps),
and the back edge re-enters with the same two steps in flight; a register read that is only reachable through a
path on which the load was NOT issued would be a false alarm -- none occurs today, the check prints what it finds."""
import re
import sys

text = open(sys.argv[1]).read().splitlines()
reg_re = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(operand_text):
    out = set()
    for m in reg_re.finditer(operand_text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


bad, kernels = [], 0
i = 0
while i < len(text):
    line = text[i]
    if line.startswith("_ZN2sh18spmv_tiled_phase2s") and line.rstrip().endswith(":") or (line.startswith("_ZN2sh18spmv_tiled_phase2s") and ": ;" in line):
        kernels += 1
        name = line.split(":")[0]
        fifo = []          # list of sets of destination registers, oldest first
        in_asm = False
        i += 1
        while i < len(text) and not text[i].startswith(".Lfunc_end"):
            l = text[i].strip()
            i += 1
            if l.startswith(";;#ASMSTART") or l.startswith("; ;#ASMSTART") or "#ASMSTART" in l:
                in_asm = True
                continue
            if "#ASMEND" in l:
                in_asm = False
                continue
            if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
                continue
            op = l.split()[0]
            rest = l[len(op):].split(";")[0]
            if in_asm and op.startswith("global_load_dword"):
                dst = rest.split(",")[0]
                fifo.append(regs_of(dst))
                # the address operands are read at issue: they must not be in flight themselves
                used = regs_of(",".join(rest.split(",")[1:]))
            elif op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", rest)
                if m:
                    n = int(m.group(1))
                    # compiler-visible loads / stores also count in vmcnt: the asm FIFO can only be SHORTER than assumed
                    # after a compiler wait, never longer -- retire down to n in both cases
                    while len(fifo) > n:
                        fifo.pop(0)
                continue
            else:
                used = regs_of(rest)
            flying = set().union(*fifo[:-1]) if (in_asm and op.startswith("global_load_dword") and fifo) else (set().union(*fifo) if fifo else set())
            hit = used & flying
            if hit:
                bad.append((name[:60], i, l, sorted(hit)))
        continue
    i += 1

if not kernels:
    sys.exit("no spmv_tiled_phase2s kernel found in the ISA text")
for b in bad[:20]:
    print("prefetch check FAILED: %s line %d: `%s` touches in-flight v%s" % (b[0], b[1], b[2], b[3]))
print(f"{kernels} phase-2 kernels checked, {len(bad)} reads of registers with a load in flight")
sys.exit(1 if bad else 0)
