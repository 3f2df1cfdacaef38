#!/bin/bash
# tools/variant_bench.sh -- run bench.py against alternative in-tree builds of the engine (SH_LIB)
# and print ms/step + per-kernel averages from a rocprofv3 kernel trace of each.
out=gpurun_out/variants; mkdir -p $out
for lib in sparseharness_amd/libsparseharness_hip.so sparseharness_amd/variants/*.so; do
  name=$(basename $lib .so)
  SH_LIB=$PWD/$lib python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ablation "$@" 2>/dev/null > $out/$name.json
  python - "$name" "$out/$name.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(f"{sys.argv[1]:28s} ms/step {d['ms_per_step']:.4f}  frac {d['roofline']['frac']:.4f}  parity_bad {d['parity']['mismatches_rel_1e-5']}")
PY
done
