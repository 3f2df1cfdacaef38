set -x
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
./tools/kstats.sh r04_kstats_final > gpurun_out/r04_kstats_final.log 2>&1
./tools/pmc.sh r04_pmc_powerlaw > gpurun_out/r04_pmc_powerlaw.log 2>&1
SH_VALCODE=off ./tools/pmc.sh r04_pmc_powerlaw_raw > gpurun_out/r04_pmc_powerlaw_raw.log 2>&1
./tools/pmc.sh r04_pmc_rmat23 --workload rmat-23 > gpurun_out/r04_pmc_rmat23.log 2>&1
python bench.py --workload rmat-23 --steps 20 --warmup 3 --no-cpu-baseline --no-ablation > gpurun_out/r04_bench_rmat23.json 2> gpurun_out/r04_bench_rmat23.err
python bench.py --workload rmat-23-unpermuted --steps 20 --warmup 3 --no-cpu-baseline --no-ablation > gpurun_out/r04_bench_rmat23_unpermuted.json 2>> gpurun_out/r04_bench_rmat23.err
python bench.py --real-values --steps 20 --warmup 3 --no-cpu-baseline --no-ablation > gpurun_out/r04_bench_real_values.json 2>> gpurun_out/r04_bench_rmat23.err
python bench.py --workload scircuit-like --steps 200 --warmup 20 --no-cpu-baseline --no-ablation > gpurun_out/r04_bench_scircuit.json 2>> gpurun_out/r04_bench_rmat23.err
tail -3 gpurun_out/r04_kstats_final.log; tail -2 gpurun_out/r04_pmc_powerlaw.log; tail -1 gpurun_out/r04_pmc_powerlaw_raw.log; tail -1 gpurun_out/r04_pmc_rmat23.log
python3 - <<'PY'
import json
for f in ("r04_bench_default","r04_bench_rmat23","r04_bench_rmat23_unpermuted","r04_bench_real_values","r04_bench_scircuit"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][0])
        print(f, d["ms_per_step"], d["roofline"]["frac"], d.get("frac_raw_values"), d.get("frac_dict16_values"), d["parity"]["mismatches_rel_1e-5"], d["upload_seconds"], (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "failed", e)
PY
