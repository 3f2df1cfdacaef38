#!/bin/bash
# tools/pmc.sh <outdir> [bench args] -- HBM traffic counters of the bench kernels, one counter per pass
# (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots").
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/$ctr -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ablation "$@" > $out/$ctr.json 2> $out/$ctr.err
done
python3 - $out <<'PY'
import csv, glob, sys, collections
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{sys.argv[1]}/{ctr}/*/*counter_collection.csv")
    if not f:
        print(ctr, "no counter file", glob.glob(f"{sys.argv[1]}/{ctr}/*/*")); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == ctr:
            acc[r["Kernel_Name"][:44]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{ctr:10s} {k:44s} n={len(v):3d} avg={sum(v)/len(v):14.1f} (KB units per rocprof)")
PY
