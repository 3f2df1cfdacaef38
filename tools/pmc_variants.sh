#!/bin/bash
# tools/pmc_variants.sh "<counters>" -- the same counter pass for the in-tree engine and every build under sparseharness_amd/variants
for lib in sparseharness_amd/libsparseharness_hip.so sparseharness_amd/variants/*.so; do
  name=$(basename $lib .so); echo "== $name"
  SH_LIB=$GRAFT_REPO_ROOT/$lib ./tools/pmc_sq.sh pmcv_$name "$1" 2>/dev/null | grep -A12 "phase2s"
done
