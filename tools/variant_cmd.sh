#!/bin/bash
# tools/variant_cmd.sh <python script> [args] -- run a script against the in-tree engine and every build under sparseharness_amd/variants
for lib in sparseharness_amd/libsparseharness_hip.so sparseharness_amd/variants/*.so; do
  echo "== $(basename $lib .so)"; SH_LIB=$PWD/$lib python "$@" 2>/dev/null | tail -1 | cut -c1-900
done
