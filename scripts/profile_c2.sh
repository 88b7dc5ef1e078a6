# Round profile of the C2 workload: rocprofv3 kernel stats (serial frames only / default run with both frame loops), PMC
# traffic of every kernel (two separate --pmc passes), kernel stats of the serial C3 and C5 loops, and the bench line.  usage (on the GPU box): scripts/profile_c2.sh r02_v1
set -e
TAG=${1:-r02_v1}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf $O/ser $O/pip $O/pf $O/pw
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $O/ser -o s -- python3 $R/bench.py --no-cpu-baseline --no-profile --no-pipeline --no-extra-workloads --steps 150 --warmup 10 > $O/ser.json 2> $O/ser.err
echo serial done
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $O/pip -o p -- python3 $R/bench.py --no-cpu-baseline --no-profile --no-extra-workloads --steps 150 --warmup 10 > $O/pip.json 2> $O/pip.err
echo both loops done
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -o f -- python3 $R/bench.py --no-cpu-baseline --no-profile --no-pipeline --no-extra-workloads --steps 10 --warmup 2 > $O/pf.json 2> $O/pf.err
echo fetch done
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -o w -- python3 $R/bench.py --no-cpu-baseline --no-profile --no-pipeline --no-extra-workloads --steps 10 --warmup 2 > $O/pw.json 2> $O/pw.err
echo write done
cd $R
F=$(find $O/pf -name "*counter_collection.csv" | head -1); W=$(find $O/pw -name "*counter_collection.csv" | head -1)
python3 scripts/pmc_traffic.py $F $W profiles/${TAG}_c2_pmc.json > $O/pmc.txt
cp profiles/${TAG}_c2_pmc.json $O/
python3 scripts/rocpd_summary.py $(find $O/ser -name "*.db" | head -1) $O/${TAG}_c2_kernel_stats_serial.csv
python3 scripts/rocpd_summary.py $(find $O/pip -name "*.db" | head -1) $O/${TAG}_c2_kernel_stats_both_loops.csv
for WL in c3 c5; do
  cd /tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/$WL -o s -- python3 $R/bench.py --workload $WL --no-cpu-baseline --no-profile --no-pipeline --no-extra-workloads --steps 20 --warmup 3 > $O/$WL.json 2> $O/$WL.err
  cd $R
  python3 scripts/rocpd_summary.py $(find $O/$WL -name "*.db" | head -1) $O/${TAG}_${WL}_kernel_stats_serial.csv
  echo $WL done
done
find $O -name "*.db" -delete          # (gpurun copies back at most 64 MiB)
find $O/pf $O/pw -name "*.csv" -size +2M -delete
timeout -k 10 400 python3 bench.py > $O/${TAG}_c2_bench.json 2> $O/bench.err
tail -c 400 $O/${TAG}_c2_bench.json
