#!/bin/bash
# tools/env_bench.sh "NAME=VAL ..." ... -- bench.py under different engine env knobs
mkdir -p gpurun_out/envb
i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ablation $BENCH_ARGS 2>/dev/null > gpurun_out/envb/$i.json
  python - "$envs" gpurun_out/envb/$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(f"{sys.argv[1]:40s} ms/step {d['ms_per_step']:.4f}  frac {d['roofline']['frac']:.4f}  bad {d['parity']['mismatches_rel_1e-5']} upload {d['upload_seconds']}")
PY
done
