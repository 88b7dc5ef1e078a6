# LDS counters of the convolution micro-benchmark.  usage (GPU box): scripts/profile_conv_lds.sh
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/convsq; mkdir -p $O; export TMPDIR=/tmp
cd /tmp; rm -rf $O/lds
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/lds -o m -- python3 $R/scripts/conv_patch_bench.py > $O/run2.log 2> $O/run2.err
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/convsq/lds/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"]
    if "conv" not in k: continue
    k = "conv_patch" if "conv_patch" in k else "conv_ss"
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    if row["Counter_Name"] == "SQ_WAVES": n[k] += 1
for k, c in acc.items():
    print(k, "launches", n[k], {a: round(b / n[k]) for a, b in c.items()})
PY
find $O/lds -name "*.csv" -delete
