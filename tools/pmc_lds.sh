#!/bin/bash
# LDS bank conflicts and LDS activity of every libsad kernel in the bench command (unfrozen steps included), one --pmc pass.
# usage (on the GPU box): bash tools/pmc_lds.sh > gpurun_out/r03_pmc_lds.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES -d gpurun_out/pmc_lds --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernels > gpurun_out/pmc_lds.log 2>&1 || { echo PMC FAIL; tail -5 gpurun_out/pmc_lds.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = sorted(glob.glob("gpurun_out/pmc_lds/*/*counter_collection.csv"))[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "sad::k_" not in k:
        continue
    agg[k.split("sad::")[1].split("(")[0][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-46s %14s %14s %10s" % ("kernel", "LDS_IDX_ACTIVE", "BANK_CONFLICT", "conflict %"))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_LDS_IDX_ACTIVE", [0]))):
    a = sum(v.get("SQ_LDS_IDX_ACTIVE", [0])) / max(1, len(v.get("SQ_LDS_IDX_ACTIVE", [0])))
    c = sum(v.get("SQ_LDS_BANK_CONFLICT", [0])) / max(1, len(v.get("SQ_LDS_BANK_CONFLICT", [0])))
    if a > 0:
        print("%-46s %14.0f %14.0f %9.1f%%" % (k, a, c, 100.0 * c / a))
PY
