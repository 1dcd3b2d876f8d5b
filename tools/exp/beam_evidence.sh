#!/bin/bash
# re-collects config 3's rocprofv3 kernel stats, the beam kernel's PMC passes and the driver's line
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-beam_ev}; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg3 -o s -- python3 bench.py --config 3 --steps 30 --warmup 5 --cpu-budget 0 --extras off > $O/bench_cfg3_rocprof.json 2> $O/err.log || exit 1
echo "stats done"
bash tools/pmc_score.sh ${1:-beam_ev}_beam beam_converged --config 3 > $O/pmc.log 2>&1 || exit 1
echo "pmc done"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2>> $O/err.log || exit 1
find $O gpurun_out/pmc_${1:-beam_ev}_* -name "*.csv" -size +2M -delete
echo "all done"
