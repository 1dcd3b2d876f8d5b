#!/bin/bash
# One bench line per single-GPU BASELINE config (1, 2, 3, 5), each with roofline + cpu_baseline, plus the
# self-launched two-rank rehearsal.  usage (on the GPU box): bash tools/bench_configs.sh <tag>
set -o pipefail
TAG=${1:-cfg}
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
O=gpurun_out/$TAG
mkdir -p $O
python3 bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err || exit 1
echo "cfg2 done"
python3 bench.py --config 1 --steps 1000 --warmup 100 > $O/bench_cfg1.json 2> $O/bench_cfg1.err || exit 1
echo "cfg1 done"
python3 bench.py --config 3 --steps 30 --warmup 5 > $O/bench_cfg3.json 2> $O/bench_cfg3.err || exit 1
echo "cfg3 done"
python3 bench.py --config 5 --steps 10 --warmup 2 > $O/bench_cfg5.json 2> $O/bench_cfg5.err || exit 1
echo "cfg5 done"
BPF_BENCH_REHEARSAL=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 100 --warmup 5 --cpu-budget 0 > $O/bench_selflaunch_2ranks_1gpu.json 2> $O/bench_selflaunch.err || exit 1
echo "self-launch done"
