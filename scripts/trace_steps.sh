set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tr; mkdir -p $O; export TMPDIR=/tmp
for WL in c3 c5; do
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/$WL -o t -- python3 $R/bench.py --workload $WL --no-cpu-baseline --no-profile --no-pipeline --no-extra-workloads --steps 6 --warmup 2 > $O/$WL.json 2> $O/$WL.err
cd $R
python3 scripts/step_trace.py $(find $O/$WL -name "*kernel_trace.csv" | head -1) > $O/${WL}_step_trace.txt
find $O/$WL -name "*.csv" -size +20M -delete
done
tail -2 $O/c3_step_trace.txt
