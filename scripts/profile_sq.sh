# Wave-cycle breakdown per kernel (rocprofv3 PMC, SQ counters, its own pass) for the serial C3 and C5 loops.
# usage (on the GPU box): scripts/profile_sq.sh r03_v4
set -e
TAG=${1:-r03_v4}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for WL in c3 c5; do
  case $WL in c3) PST=3;; c5) PST=4;; esac
  cd /tmp; rm -rf $O/sq_$WL
  timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/sq_$WL -o m -- python3 $R/bench.py --workload $WL --no-cpu-baseline --no-profile --no-pipeline --no-extra-workloads --steps $PST --warmup 1 > $O/sq_$WL.json 2> $O/sq_$WL.err
  cd $R
  python3 scripts/pmc_sq.py $(find $O/sq_$WL -name "*counter_collection.csv" | head -1) $O/${TAG}_${WL}_sq.json > $O/${TAG}_${WL}_sq.txt
  find $O/sq_$WL -name "*.csv" -delete
  echo $WL; cat $O/${TAG}_${WL}_sq.txt
done
