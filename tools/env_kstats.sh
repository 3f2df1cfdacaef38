#!/bin/bash
# tools/env_kstats.sh "NAME=VAL ..." ... -- per-kernel averages (rocprofv3) of a short bench run under engine env knobs
i=0
for envs in "$@"; do
  i=$((i+1)); echo "== $envs"
  export $envs
  ./tools/kstats.sh envk_$i 2>/dev/null | grep "sh::"
  for kv in $envs; do unset ${kv%%=*}; done
done
