# A/B of the LDS-staged msda kernel (debug key 22) on c3 / c5, both checkpoint presets.  usage: scripts/msda_ab.sh outdir
out=${1:-gpurun_out/msda_ab}; mkdir -p $out
for preset in peaked flat; do for w in c3 c5; do for k22 in 0 3; do
  LBM_DEBUG_SET=22:$k22 timeout -k 10 200 python bench.py --workload $w --preset $preset --steps 20 --warmup 3 --no-cpu-baseline --allow-knobs > $out/${w}_${preset}_$k22.json 2>$out/err.txt || { tail -5 $out/err.txt; exit 1; }
  python - <<PY
import json
d=json.load(open("$out/${w}_${preset}_$k22.json"))
ms=[k for k in d.get("kernels",[]) if k["name"].startswith("msda")]
print("$w $preset window=$k22: serial %.3f pipelined %.3f ms/step; " % (d["serial_frames"]["ms_per_step"], d["frame_pipeline"]["ms_per_step"]), [(k["name"],k["launches_per_step"],k["avg_us"]) for k in ms])
PY
done; done; done
