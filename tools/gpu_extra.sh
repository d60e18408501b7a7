#!/bin/bash
# Extra measurements: stress (cfg5 slice) bench + kernel stats, yelp64 kernel stats, module-API lines.
# Usage (through gpurun): bash tools/gpu_extra.sh TAG
set -o pipefail
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 500 python bench.py --workload stress --steps 5 --warmup 2 > gpurun_out/${TAG}_bench_stress.json 2> gpurun_out/${TAG}_stress.err || { echo STRESS FAILED; tail -20 gpurun_out/${TAG}_stress.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof_stress --output-format csv -- python3 bench.py --workload stress --steps 3 --warmup 1 --no-cpu-baseline --no-kernels > gpurun_out/${TAG}_prof_stress.log 2>&1 || { echo PROF stress FAILED; tail -5 gpurun_out/${TAG}_prof_stress.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof_yelp64 --output-format csv -- python3 bench.py --workload yelp64 --steps 20 --warmup 5 --no-cpu-baseline --no-kernels > gpurun_out/${TAG}_prof_yelp64.log 2>&1 || { echo PROF yelp64 FAILED; tail -5 gpurun_out/${TAG}_prof_yelp64.log; exit 1; }
timeout -k 10 200 python bench.py --workload yelp64 --api module --steps 50 --warmup 10 --no-cpu-baseline --no-kernels > gpurun_out/${TAG}_bench_yelp64_module.json 2>> gpurun_out/${TAG}_stress.err || { echo yelp64 module FAILED; exit 1; }
timeout -k 10 200 python bench.py --api module --steps 20 --warmup 5 --no-cpu-baseline --no-kernels > gpurun_out/${TAG}_bench_module.json 2>> gpurun_out/${TAG}_stress.err || { echo cfg3 module FAILED; exit 1; }
python - <<PY
import json,glob,csv
for f in ("${TAG}_bench_stress.json","${TAG}_bench_yelp64_module.json","${TAG}_bench_module.json"):
    j=json.loads(open("gpurun_out/"+f).read().strip().splitlines()[-1])
    print(f, "value=%.0f sent/s ms/step=%.3f unfrozen=%.3f ms whole-step mfma=%.3f cpu=%s" % (j["value"], j["ms_per_step"], j["unfrozen"]["ms_per_step"], j["whole_step"]["mfma_frac"], (j["cpu_baseline"] or {}).get("value")))
    for k in j["kernels"] or []:
        print("    %-72s %8.1f us  %s" % (k["kernel"][:72], k["avg_launch_us"], {x: k[x] for x in ("TFLOPs", "mfma_frac", "GBps", "hbm_frac", "us_per_time_step") if x in k}))
for w in ("stress","yelp64"):
    fs=glob.glob("gpurun_out/${TAG}_prof_%s/*/*kernel_stats.csv" % w)
    print("---", w)
    for r in list(csv.DictReader(open(fs[0])))[:14]:
        print("%-60s calls=%5s avg_us=%8.2f pct=%s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
