#!/bin/bash
# tools/kstats_arms.sh <outdir> "name|LIB|ENV=VAL ENV=VAL" ... -- per-kernel rocprofv3 averages of a short bench.py run for
# several engine builds / knob settings on one box (LIB: path under sparseharness_amd/, or "head").  bench args: $BENCH_ARGS
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"; [ -n "$1" ] || { echo "usage: $0 <outdir> ..." >&2; exit 2; }
root=$GRAFT_REPO_ROOT; outroot=$root/gpurun_out/$1; shift
rm -rf "$outroot"; mkdir -p "$outroot"; cd /tmp; export TMPDIR=/tmp SH_PLACEMENT_TRIES=${SH_PLACEMENT_TRIES:-1}   # (one placement: the averages hold the timed launches only)
for arm in "$@"; do
  IFS='|' read -r name lib envs <<< "$arm"
  out=$outroot/$name; mkdir -p $out
  (
    if [ "$lib" != "head" ]; then export SH_LIB=$root/sparseharness_amd/$lib SH_LIB_PARTIAL=1; fi
    for kv in $envs; do export "$kv"; done
    rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ablation $BENCH_ARGS > $out/bench.json 2> $out/err.log
  )
  python3 - $out "$name" <<'PY'
import csv, glob, json, os, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")
if not f:
    print(sys.argv[2], "no kernel stats; stderr tail:", open(sys.argv[1] + "/err.log").read()[-400:]); sys.exit(0)
rows = [r for r in csv.DictReader(open(max(f, key=os.path.getmtime))) if "sh::" in r["Name"]]
try:
    b = json.loads([l for l in open(sys.argv[1] + "/bench.json") if l.startswith("{")][0])
    extra = f"bench ms/step {b['ms_per_step']:.4f} frac {b['roofline']['frac']:.4f} bad {b['parity']['mismatches_rel_1e-5']}"
except Exception as e:   # noqa: BLE001
    extra = f"(no bench line: {e})"
print(f"{sys.argv[2]:18s} " + "  ".join(f"{r['Name'].split('sh::')[1].split('<')[0]} {float(r['AverageNs'])/1e3:.1f} us" for r in rows) + "  | " + extra)
PY
done
