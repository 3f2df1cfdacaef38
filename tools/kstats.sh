#!/bin/bash
# tools/kstats.sh <outdir> [bench args] -- rocprofv3 kernel-trace stats of a short bench run, summary to stdout
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"; [ -n "$1" ] || { echo "usage: $0 <outdir> ..." >&2; exit 2; }
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
rm -rf "$out"; mkdir -p "$out"; cd /tmp; export TMPDIR=/tmp SH_PLACEMENT_TRIES=${SH_PLACEMENT_TRIES:-1}   # (one placement: the averages hold the timed launches only)   # a fresh directory per run: the summary below can only see this run
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ablation "$@" > $out/bench.json 2> $out/err.log
python3 - $out <<'PY'
import csv, glob, sys
import os
for r in csv.DictReader(open(max(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"), key=os.path.getmtime))):
    print(f"{r['Name'][:48]:48s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:10.1f}")
PY
