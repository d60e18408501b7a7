#!/bin/bash
# PMC counters of the bench command per kernel: separate --pmc passes, as MI355X_MICROARCH.md prescribes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass; SQ/GRBM in a third).  Writes
# gpurun_out/pmc_bench_traffic.json with the source id of the build it measured (bench.source_id());
# copy it to profiles/ to have bench.py report roofline.traffic — bench refuses a file from another build.
# usage (on the GPU box): bash tools/pmc_bench.sh [bench args, default: cfg3]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="${*:---workload cfg3}"
pass() {  # name, counters
  timeout -s KILL 240 rocprofv3 --pmc $2 -d gpurun_out/pmcb_$1 --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernels $ARGS > gpurun_out/pmcb_$1.log 2>&1 || { echo PMC FAIL $1; tail -5 gpurun_out/pmcb_$1.log; exit 1; }
}
pass FETCH "FETCH_SIZE" && pass WRITE "WRITE_SIZE" && pass SQ "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" || exit 1
python3 - <<'PY'
import csv, glob, collections, json, sys
sys.path.insert(0, ".")
import bench
out = collections.defaultdict(dict)
for tag, names in (("FETCH", ["FETCH_SIZE"]), ("WRITE", ["WRITE_SIZE"]),
                   ("SQ", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"])):
    f = glob.glob("gpurun_out/pmcb_%s/*/*counter_collection.csv" % tag)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "sad::k_" not in k or r["Counter_Name"] not in names:
            continue
        agg[k.split("sad::")[1].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            out[k][c + ("_KB_avg" if c.endswith("_SIZE") else "_avg")] = sum(v) / len(v)
            out[k]["launches"] = len(v)
for k, v in out.items():
    # gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> double it (MI355X_MICROARCH.md, HBM)
    v["hbm_bytes_per_launch"] = int((2 * v.get("FETCH_SIZE_KB_avg", 0) + v.get("WRITE_SIZE_KB_avg", 0)) * 1024)
    g = v.get("GRBM_GUI_ACTIVE_avg", 0)
    if g:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over all 1024 SIMDs (256 CUs x 4)
        v["mfma_busy_frac"] = round(v.get("SQ_VALU_MFMA_BUSY_CYCLES_avg", 0) / (g / 8 * 1024), 4)
out["_source_id"] = bench.source_id()
out["_command"] = "rocprofv3 --pmc <pass> -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernels " + " ".join(sys.argv[1:])
json.dump(out, open("gpurun_out/pmc_bench_traffic.json", "w"), indent=1, sort_keys=True)
rows = [(k, v) for k, v in out.items() if not k.startswith("_")]
for k, v in sorted(rows, key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
    print("%-46s hbm %8.1f MB  mfma_busy %s" % (k[:46], v["hbm_bytes_per_launch"] / 1e6, v.get("mfma_busy_frac")))
PY
