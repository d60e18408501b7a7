#!/bin/bash
# HBM traffic of the bench command per kernel (separate --pmc passes, as MI355X_MICROARCH.md prescribes).
# Writes gpurun_out/pmc_bench_traffic.json (copy to profiles/ to have bench.py report roofline.traffic).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c -d gpurun_out/pmcb_$c --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmcb_$c.log 2>&1 || { echo PMC FAIL $c; tail -5 gpurun_out/pmcb_$c.log; exit 1; }
done
python - <<'PY'
import csv,glob,collections,json
out=collections.defaultdict(dict)
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob("gpurun_out/pmcb_%s/*/*counter_collection.csv" % c)[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "sad::k_" not in k or r["Counter_Name"]!=c: continue
        agg[k.split("sad::")[1].split("(")[0]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): out[k][c+"_KB_avg"]=sum(v)/len(v); out[k]["launches"]=len(v)
for k,v in out.items():
    # gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> double it (MI355X_MICROARCH.md, HBM)
    v["hbm_bytes_per_launch"]=int((2*v.get("FETCH_SIZE_KB_avg",0)+v.get("WRITE_SIZE_KB_avg",0))*1024)
json.dump(out, open("gpurun_out/pmc_bench_traffic.json","w"), indent=1, sort_keys=True)
for k,v in sorted(out.items(), key=lambda kv:-kv[1]["hbm_bytes_per_launch"])[:8]: print(k[:50], v)
PY
