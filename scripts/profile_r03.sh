# Round-3 profile: rocprofv3 kernel stats of the serial C2 / C3 / C5 loops, PMC traffic (two separate --pmc passes each) for C2,
# C3 and C5, and the bench line.  usage (on the GPU box): scripts/profile_r03.sh r03_v1 [nobench]
set -e
TAG=${1:-r03_v1}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for WL in c2 c3 c5; do
  case $WL in c2) ST=100; PST=10;; c3) ST=12; PST=3;; c5) ST=20; PST=4;; esac
  cd /tmp
  rm -rf $O/$WL $O/pf_$WL $O/pw_$WL
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/$WL -o s -- python3 $R/bench.py --workload $WL --no-cpu-baseline --no-profile --no-pipeline --no-extra-workloads --steps $ST --warmup 3 > $O/$WL.json 2> $O/$WL.err
  echo $WL kernel stats done
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf_$WL -o f -- python3 $R/bench.py --workload $WL --no-cpu-baseline --no-profile --no-pipeline --no-extra-workloads --steps $PST --warmup 1 > $O/pf_$WL.json 2> $O/pf_$WL.err
  echo $WL fetch done
  timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw_$WL -o w -- python3 $R/bench.py --workload $WL --no-cpu-baseline --no-profile --no-pipeline --no-extra-workloads --steps $PST --warmup 1 > $O/pw_$WL.json 2> $O/pw_$WL.err
  echo $WL write done
  cd $R
  F=$(find $O/pf_$WL -name "*counter_collection.csv" | head -1); W=$(find $O/pw_$WL -name "*counter_collection.csv" | head -1)
  python3 scripts/pmc_traffic.py $F $W $O/${TAG}_${WL}_pmc.json > $O/pmc_$WL.txt
  python3 scripts/rocpd_summary.py $(find $O/$WL -name "*.db" | head -1) $O/${TAG}_${WL}_kernel_stats_serial.csv
  find $O -name "*.db" -delete          # (gpurun copies back at most 64 MiB)
  find $O/pf_$WL $O/pw_$WL -name "*.csv" -delete
done
if [ "$2" != "nobench" ]; then
  cp $O/${TAG}_c*_pmc.json profiles/      # so that the bench line can quote this very run's counters
  timeout -k 10 500 python3 bench.py > $O/${TAG}_c2_bench.json 2> $O/bench.err
  tail -c 600 $O/${TAG}_c2_bench.json
fi
