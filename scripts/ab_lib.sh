#!/bin/bash
# A/B of two builds of the library on one box: bench lines (serial and pipelined frames) for c2 / c3 / c5.
# usage: scripts/ab_lib.sh outdir libA.so libB.so
out=$1; shift
mkdir -p $out
for rep in 1 2; do
for lib in "$@"; do
  tag=$(basename $lib .so)
  for w in c2 c3 c5; do
    for mode in no-pipeline pipeline; do
      LBM_AMD_LIB=$PWD/$lib timeout -k 10 200 python bench.py --workload $w --steps 60 --no-cpu-baseline --no-profile --$mode > $out/${tag}_${w}_${mode}_$rep.json 2>/dev/null || exit 1
      python - <<PY
import json
d=json.load(open("$out/${tag}_${w}_${mode}_$rep.json"))
print("$tag $w $mode rep$rep ms/step %.3f p50 %.3f" % (d["ms_per_step"], d["p50_latency_ms"]))
PY
    done
  done
done
done
