#!/bin/bash
# evidence for the BORDER form of k_cloud_score: its test, rocprofv3 kernel stats of config 5, the PMC passes, the driver's line
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-border_ev}; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_cloud.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg5 -o s -- python3 bench.py --config 5 --steps 10 --warmup 2 --cpu-budget 0 --extras off > $O/bench_cfg5_rocprof.json 2> $O/err.log || exit 1
echo "stats done"
bash tools/pmc_score.sh ${1:-border_ev}_cloud3d cloud3d_converged --config 5 > $O/pmc.log 2>&1 || exit 1
echo "pmc done"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2>> $O/err.log || exit 1
find $O gpurun_out/pmc_${1:-border_ev}_* -name "*.csv" -size +2M -delete
echo "all done"
