#!/bin/bash
# tools/env_shard.sh "NAME=VAL ..." ... -- tools/shard_bench.py (N = 1 and 8) under different engine env knobs
for envs in "$@"; do
  echo "== $envs"; env $envs python tools/shard_bench.py 1 8 2>/dev/null
done
