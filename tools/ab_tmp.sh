set -e
export TMPDIR=/tmp
LIBS="${LIBS:-old new}"
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
