#!/bin/bash
# Rehearsal of bench.py's N > 1 flow on a one-GPU box: N ranks share cuda:0 (gloo for the host-side exchanges,
# the mailbox between the processes for the path's own).  The timings mean nothing (the ranks share one GPU).
# usage: bash tools/rehearse_multi.sh [N=2] [steps=100] [extra bench.py arguments ...]
N=${1:-2}
STEPS=${2:-100}
shift; shift
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
for X in auto collective; do
  # no external launcher: bench.py --gpus N starts its own ranks when WORLD_SIZE is unset
  BPF_BENCH_REHEARSAL=1 BPF_SHARD_EXCHANGE=$X timeout -k 10 300 python3 bench.py --gpus $N --steps $STEPS --warmup 5 \
    --cpu-budget 0 "$@" || exit 1
done
