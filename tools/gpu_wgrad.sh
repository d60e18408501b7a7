#!/bin/bash
# Param-grad parity tests + per-kernel times of the unfrozen step.  Usage: bash tools/gpu_wgrad.sh TAG
set -o pipefail
TAG=${1:-w}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q -s --timeout 180 --timeout-method thread -k "param_grads or module_dropin" > gpurun_out/${TAG}_test.log 2>&1
rc=$?; tail -2 gpurun_out/${TAG}_test.log
if [ $rc -ne 0 ]; then echo "TESTS FAILED rc=$rc"; grep -E "Error|error|FAILED|assert" gpurun_out/${TAG}_test.log | head -30; exit $rc; fi
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_prof.log 2>&1 || { echo PROF FAILED; tail -5 gpurun_out/${TAG}_prof.log; exit 1; }
python - <<PY
import json,glob,csv
j=[json.loads(l) for l in open("gpurun_out/${TAG}_prof.log") if l.startswith("{")][-1]
print("frozen %.0f sent/s %.3f ms | unfrozen %s" % (j["value"], j["ms_per_step"], j["unfrozen"]))
rows=list(csv.DictReader(open(glob.glob("gpurun_out/${TAG}_prof/*/*kernel_stats.csv")[0])))
for r in rows[:16]:
    if "sad::" in r["Name"]:
        print("%-60s calls=%5s avg_us=%8.2f pct=%s" % (r["Name"].split("sad::")[1][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
