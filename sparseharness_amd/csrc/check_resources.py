#!/usr/bin/env python3
"""Reads hipcc's -Rpass-analysis=kernel-resource-usage remarks and fails if a kernel of the tiled plan uses scratch
or spills VGPRs or SGPRs (see the asm-check target of the Makefile)."""
import re
import sys

text = open(sys.argv[1]).read()
bad, seen = [], 0
for blk in text.split("remark: Function Name: ")[1:]:
    name = blk.split()[0]
    if "spmv_tiled" not in name:
        continue
    seen += 1
    scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", blk).group(1))
    spill = int(re.search(r"VGPRs Spill: (\d+)", blk).group(1))
    vgprs = int(re.search(r"VGPRs: (\d+)", blk).group(1))
    sspill = int(re.search(r"SGPRs Spill: (\d+)", blk).group(1))
    # 1024-thread workgroups: 128 VGPRs per lane is all there is.  The experimental fused kernel may carry a
    # scavenger slot (a few bytes of private segment that no instruction touches: checked in the .s once); the
    # default kernels must have none.
    # (a spilled SGPR lives in a lane of a VGPR the hand-scheduled loaders might otherwise count on, and costs
    # v_writelane / v_readlane traffic wherever it is used)
    if spill or sspill or vgprs > 128 or scratch:
        bad.append((name, scratch, spill, vgprs, sspill))
if not seen:
    sys.exit("no spmv_tiled kernels found in the resource-usage remarks")
for b in bad:
    print("resource check FAILED: %s scratch=%d vgpr_spill=%d vgprs=%d sgpr_spill=%d" % b)
print(f"{seen} tiled kernels checked, {len(bad)} offenders")
sys.exit(1 if bad else 0)
