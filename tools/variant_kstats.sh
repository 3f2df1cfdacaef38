#!/bin/bash
# per-kernel average times for each engine build under sparseharness_amd/variants (timing-only ablations)
for lib in sparseharness_amd/libsparseharness_hip.so sparseharness_amd/variants/*.so; do
  name=$(basename $lib .so); echo "== $name"
  SH_LIB=$GRAFT_REPO_ROOT/$lib ./tools/kstats.sh var_$name 2>/dev/null | grep "sh::"
done
