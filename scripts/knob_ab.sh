# A/B of debug knobs on the serial loop of a workload.  usage (GPU box): scripts/knob_ab.sh <workload> <steps> "<knobs or ->" ...
set -e
WL=$1; ST=$2; shift 2
O=gpurun_out/ab; mkdir -p $O
for K in "$@"; do
  if [ "$K" = "-" ]; then
    timeout -k 10 200 python3 bench.py --workload $WL --no-cpu-baseline --no-profile --no-pipeline --no-extra-workloads --steps $ST --warmup 3 > $O/x.json 2> $O/x.err
  else
    LBM_DEBUG_SET=$K timeout -k 10 200 python3 bench.py --workload $WL --allow-knobs --no-cpu-baseline --no-profile --no-pipeline --no-extra-workloads --steps $ST --warmup 3 > $O/x.json 2> $O/x.err
  fi
  python3 -c "import json,sys; d=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); print('$WL knobs $K: ms_per_step', d['ms_per_step'])"
done
