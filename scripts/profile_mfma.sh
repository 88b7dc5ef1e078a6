# MFMA utilisation per kernel (rocprofv3 PMC, its own pass) for the serial C3 and C5 loops.  usage (on the GPU box): scripts/profile_mfma.sh r03_v3
set -e
TAG=${1:-r03_v3}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for WL in c3 c5 c2; do
  case $WL in c2) PST=10;; c3) PST=3;; c5) PST=4;; esac
  cd /tmp; rm -rf $O/pm_$WL
  timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pm_$WL -o m -- python3 $R/bench.py --workload $WL --no-cpu-baseline --no-profile --no-pipeline --no-extra-workloads --steps $PST --warmup 1 > $O/pm_$WL.json 2> $O/pm_$WL.err
  cd $R
  python3 scripts/pmc_mfma.py $(find $O/pm_$WL -name "*counter_collection.csv" | head -1) $O/${TAG}_${WL}_mfma_util.json > $O/mfma_$WL.txt
  find $O/pm_$WL -name "*.csv" -delete
  echo $WL; cat $O/mfma_$WL.txt
done
