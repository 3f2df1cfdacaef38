#!/bin/bash
# tools/pmc.sh <outdir> [bench args] -- HBM traffic counters of the bench kernels, one counter per pass
# (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Writes <outdir>/traffic.json: one record {workload, layout, n_gpus, bytes, ...} for profiles/measured_traffic.json
# (bench.py echoes it as roofline.traffic only for exactly that workload and device layout).
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"; [ -n "$1" ] || { echo "usage: $0 <outdir> ..." >&2; exit 2; }
name=$1; out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
rm -rf "$out"; mkdir -p "$out"; cd /tmp; export TMPDIR=/tmp   # a fresh directory per run
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/$ctr -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ablation "$@" > $out/$ctr.json 2> $out/$ctr.err
done
python3 - $out $name <<'PY'
import csv, glob, json, os, sys, collections
out, name = sys.argv[1], sys.argv[2]
tot = {}
per = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/{ctr}/*/*counter_collection.csv")
    if not f:
        print(ctr, "no counter file", glob.glob(f"{out}/{ctr}/*/*")); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(max(f, key=os.path.getmtime))):
        if r["Counter_Name"] == ctr and "spmv" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{ctr:10s} {k:60s} n={len(v):3d} avg={sum(v)/len(v):14.1f} (KB units per rocprof)")
        per[f"{ctr} {k}"] = round(sum(v) / len(v), 1)
    tot[ctr] = sum(sum(v) / len(v) for v in acc.values())   # per SpMV: one dispatch of each kernel
if len(tot) == 2:
    b = json.loads([l for l in open(f"{out}/FETCH_SIZE.json") if l.startswith("{")][0])
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE reports 1/2 of wide coalesced reads on gfx950 -> x2; WRITE_SIZE exact; rocprofv3 units are KB
    nbytes = int(tot["FETCH_SIZE"] * 1024 * 2 + tot["WRITE_SIZE"] * 1024)
    rec = {"workload": b["config"]["workload"].split(":")[0], "layout": b["plan"]["layout"], "n_gpus": b["n_gpus"], "bytes": nbytes,
           "fetch_kb_x2_corrected": round(tot["FETCH_SIZE"] * 2, 1), "write_kb": round(tot["WRITE_SIZE"], 1),
           "algorithmic_bytes": b["roofline"]["algorithmic_bytes_per_launch"], "per_kernel_avg_kb": per,
           "source": f"separate rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc.sh), gpurun_out/{name}"}
    json.dump(rec, open(f"{out}/traffic.json", "w"), indent=1)
    print(f"HBM traffic per SpMV = {nbytes} B = x{nbytes / rec['algorithmic_bytes']:.3f} algorithmic  ({rec['layout']})")
PY
