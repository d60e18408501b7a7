#!/bin/bash
# usage: bash tools/pmc.sh TAG "COUNTERS..." -- kbench args
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; CNT=$2; shift 3
timeout -s KILL 120 rocprofv3 --pmc $CNT -d gpurun_out/pmc_$TAG --output-format csv -- python tools/kbench.py --iters 3 "$@" > gpurun_out/pmc_$TAG.log 2>&1 || { echo PMC FAIL; tail -5 gpurun_out/pmc_$TAG.log; exit 1; }
python - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_$TAG/*/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"]
    if "sad::" not in k: continue
    agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k)
    for c,vals in v.items(): print("    %-28s avg=%.4g n=%d" % (c, sum(vals)/len(vals), len(vals)))
PY
