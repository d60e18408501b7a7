#!/bin/bash
# One GPU iteration: parity tests -> bench (cfg3 + yelp64) -> rocprofv3 kernel stats.
# Usage (through gpurun): bash tools/gpu_cycle.sh TAG [notest|testonly]
set -o pipefail
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
if [ "${2:-}" != "notest" ]; then
  timeout -k 10 900 python -u -m pytest tests -m gpu -x -q -s --timeout 240 --timeout-method thread > gpurun_out/${TAG}_test.log 2>&1
  rc=$?; tail -3 gpurun_out/${TAG}_test.log
  if [ $rc -ne 0 ]; then echo "TESTS FAILED rc=$rc"; grep -E "Error|error|FAILED|assert" gpurun_out/${TAG}_test.log | head -30; exit $rc; fi
fi
if [ "${2:-}" == "testonly" ]; then exit 0; fi
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { echo BENCH FAILED; tail -20 gpurun_out/${TAG}_bench.err; exit 1; }
timeout -k 10 200 python bench.py --workload yelp64 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/${TAG}_bench_yelp64.json 2>> gpurun_out/${TAG}_bench.err || { echo BENCH yelp64 FAILED; tail -20 gpurun_out/${TAG}_bench.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernels > gpurun_out/${TAG}_prof.log 2>&1 || { echo PROF FAILED; tail -5 gpurun_out/${TAG}_prof.log; exit 1; }
python - <<PY
import json,glob,csv
for f in ("gpurun_out/${TAG}_bench.json","gpurun_out/${TAG}_bench_yelp64.json"):
    j=json.loads(open(f).read().strip().splitlines()[-1])
    r=j["roofline"]
    print(f.split("/")[-1], "value=%.0f sent/s ms/step=%.3f unfrozen=%.3f ms | roofline %s %s frac=%.3f | cpu=%s" % (
        j["value"], j["ms_per_step"], j["unfrozen"]["ms_per_step"], r["kernel"][:40], r["bound"], r["frac"], (j["cpu_baseline"] or {}).get("value")))
    for k in j["kernels"]:
        print("    %-72s %8.1f us  %s" % (k["kernel"][:72], k["avg_launch_us"], {x: k[x] for x in ("TFLOPs", "mfma_frac", "GBps", "hbm_frac", "us_per_time_step") if x in k}))
fs=glob.glob("gpurun_out/${TAG}_prof/*/*kernel_stats.csv")
rows=list(csv.DictReader(open(fs[0])))
for r in rows[:14]:
    print("%-60s calls=%5s avg_us=%8.2f pct=%s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
