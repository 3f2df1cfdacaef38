#!/bin/bash
# tools/pmc_sq.sh <outdir> "<counters>" [bench args] -- SQ counters per kernel (one pass)
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"; [ -n "$1" ] || { echo "usage: $0 <outdir> ..." >&2; exit 2; }
out=$GRAFT_REPO_ROOT/gpurun_out/$1; ctrs="$2"; shift; shift
rm -rf "$out"; mkdir -p "$out"; cd /tmp; export TMPDIR=/tmp   # a fresh directory per run
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ablation "$@" > $out/bench.json 2> $out/err.log
python3 - $out <<'PY'
import csv, glob, os, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")
if not f:
    print("no counters; stderr tail:"); print(open(sys.argv[1] + "/err.log").read()[-1500:]); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(max(f, key=os.path.getmtime))):
    if "sh::spmv" in r["Kernel_Name"] or "sh::bits" in r["Kernel_Name"]:   # (the timed kernels, not the upload's layout builders)
        acc[r["Kernel_Name"].split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print(f"   {c:28s} {sum(v)/len(v):16.0f}")
PY
