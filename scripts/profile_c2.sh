set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $O/v7_ser $O/v7_pip $O/v7_pf $O/v7_pw
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $O/v7_ser -o s -- python3 $R/bench.py --no-cpu-baseline --no-profile --no-pipeline --steps 150 --warmup 10 > $O/v7_ser.json 2> $O/v7_ser.err
echo serial done
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $O/v7_pip -o p -- python3 $R/bench.py --no-cpu-baseline --no-profile --pipeline --steps 150 --warmup 10 > $O/v7_pip.json 2> $O/v7_pip.err
echo pipelined done
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/v7_pf -o f -- python3 $R/bench.py --no-cpu-baseline --no-profile --no-pipeline --steps 10 --warmup 2 > $O/v7_pf.json 2> $O/v7_pf.err
echo fetch done
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/v7_pw -o w -- python3 $R/bench.py --no-cpu-baseline --no-profile --no-pipeline --steps 10 --warmup 2 > $O/v7_pw.json 2> $O/v7_pw.err
echo write done
cd $R
F=$(find $O/v7_pf -name "*counter_collection.csv" | head -1); W=$(find $O/v7_pw -name "*counter_collection.csv" | head -1)
python3 scripts/pmc_traffic.py $F $W profiles/r01_v7_c2_pmc.json > $O/v7_pmc.txt
cp profiles/r01_v7_c2_pmc.json $O/
python3 scripts/rocpd_summary.py $(find $O/v7_ser -name "*.db" | head -1) $O/r01_v7_c2_kernel_stats_serial.csv
python3 scripts/rocpd_summary.py $(find $O/v7_pip -name "*.db" | head -1) $O/r01_v7_c2_kernel_stats_pipelined.csv
find $O/v7_ser $O/v7_pip $O/v7_pf $O/v7_pw -name "*.db" -size +20M -delete
find $O/v7_pf $O/v7_pw -name "*.csv" -size +20M -delete
timeout -k 10 400 python3 bench.py > $O/r01_v7_c2_bench.json 2> $O/v7_bench.err
tail -c 600 $O/r01_v7_c2_bench.json
