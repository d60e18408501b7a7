#!/bin/bash
# A/B library variants: per-kernel averages from rocprofv3 for each .so under _native/variants
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for so in non-parallel-text-style-transfer-using-self-attn-discriminator_amd/_native/variants/*.so; do
  n=$(basename $so .so)
  SAD_LIB=$PWD/$so timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/ab_$n --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/ab_$n.log 2>&1 || { echo fail $n; tail -3 gpurun_out/ab_$n.log; exit 1; }
  python - <<PY
import glob,csv
rows=list(csv.DictReader(open(glob.glob("gpurun_out/ab_$n/*/*kernel_stats.csv")[0])))
out=[]
for r in rows:
    if "sad::k_" in r["Name"] and float(r["Percentage"])>1.5:
        out.append("%s=%.1f" % (r["Name"].split("sad::")[1].split("<")[0], float(r["AverageNs"])/1e3))
print("$n", " ".join(out))
PY
done
