#!/bin/bash
# Same-box A/B of library builds (box-to-box noise of the step time is +-1 %, of the recurrences +-3 %: variants are only
# comparable inside one gpurun call).  Variant libraries are built here as tools/abl/libsad_<name>.so (e.g. the previous
# commit's sources, or the current ones with a -D switch) and selected through SAD_LIB; "new" is the in-tree library.
#   LIBS="old new" [BARGS="--workload yelp64"] bash tools/ab_bench.sh        (through gpurun)
# Prints ms per step, unfrozen ms per step and the per-kernel launch times of two alternating runs per variant.
set -e
export TMPDIR=/tmp
LIBS="${LIBS:-old new}"
mkdir -p gpurun_out
for i in 1 2; do
  for lib in $LIBS; do
    if [ $lib = new ]; then unset SAD_LIB; else export SAD_LIB=$PWD/tools/abl/libsad_$lib.so; fi
    timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline ${BARGS:-} > gpurun_out/ab_${lib}_$i.json 2> gpurun_out/ab_${lib}_$i.err
    python3 - <<PY
import json
d=json.loads(open("gpurun_out/ab_${lib}_$i.json").read().strip().splitlines()[-1])
k={x["kernel"][:12]:x["avg_launch_us"] for x in d["kernels"]}
print("$lib $i", d["ms_per_step"], d["unfrozen"]["ms_per_step"], k)
PY
  done
done
