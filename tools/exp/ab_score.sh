#!/bin/bash
# A/B of two builds of the engine on one box: headline bench lines alternating between BPF_LIB builds.
# usage (GPU box): bash tools/exp/ab_score.sh <tag> <base.so> [extra bench flags]
set -o pipefail
TAG=${1:-ab}; BASE=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/$TAG; mkdir -p $O
for round in 1 2; do
  for v in base new; do
    if [ $v = base ]; then export BPF_LIB=$PWD/$BASE; else unset BPF_LIB; fi
    python3 bench.py --steps 300 --warmup 50 --cpu-budget 0 --extras off --host-path off "$@" > $O/${v}_$round.json 2> $O/${v}_$round.err || exit 1
    python3 - $O/${v}_$round.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "ms_per_step %.4f" % d["ms_per_step"], "kernel_ms %.4f" % d["roofline"]["kernel_ms"], d["config"].get("cloud"), d["config"]["workload"][:40])
PY
  done
done
