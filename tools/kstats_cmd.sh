#!/bin/bash
# tools/kstats_cmd.sh <outdir> <python script> [args] -- rocprofv3 kernel-trace stats of any script in this repo
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"; [ -n "$1" ] || { echo "usage: $0 <outdir> ..." >&2; exit 2; }
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
rm -rf "$out"; mkdir -p "$out"; cd /tmp; export TMPDIR=/tmp
script=$GRAFT_REPO_ROOT/$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $script "$@" > $out/stdout.txt 2> $out/err.log
python3 - $out <<'PY'
import csv, glob, sys
import os
for r in csv.DictReader(open(max(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"), key=os.path.getmtime))):
    print(f"{r['Name'][:48]:48s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f}")
PY
