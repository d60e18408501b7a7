#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for d in 0 1 2 4 7; do
  SAD_DBG=$d timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/dbg_$d --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/dbg_$d.log 2>&1 || { echo fail $d; tail -3 gpurun_out/dbg_$d.log; exit 1; }
  python - <<PY
import glob,csv
rows=list(csv.DictReader(open(glob.glob("gpurun_out/dbg_$d/*/*kernel_stats.csv")[0])))
for r in rows:
    if "k_lstm_fwd" in r["Name"]: print("dbg=$d lstm_fwd avg_us=%.2f" % (float(r["AverageNs"])/1e3))
PY
done
