# SQ wave-cycle counters of the convolution micro-benchmark (conv_patch against the tiled kernel).  usage (GPU box): scripts/profile_conv_sq.sh
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/convsq; mkdir -p $O; export TMPDIR=/tmp
cd /tmp; rm -rf $O/sq
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o m -- python3 $R/scripts/conv_patch_bench.py > $O/run.log 2> $O/run.err
cd $R
python3 scripts/pmc_sq.py $(find $O/sq -name "*counter_collection.csv" | head -1) $O/conv_sq.json > $O/conv_sq.txt
find $O/sq -name "*.csv" -delete
cat $O/conv_sq.txt
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/convsq/conv_sq.json"))["kernels"]
for k,v in d.items():
    if "conv" in k:
        gui=v["GRBM_GUI_ACTIVE_per_launch"]/8
        print(k, "mfma util %.1f %%" % (100*v["SQ_VALU_MFMA_BUSY_CYCLES_per_launch"]/(1024*gui)), "active kcyc %.0f" % (gui/1e3))
PY
