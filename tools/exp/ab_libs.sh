#!/bin/bash
# Headline bench lines of several builds of the engine on one box, in turn, twice over (A/B/C ... A/B/C).
# usage (GPU box): bash tools/exp/ab_libs.sh <tag> "<bench flags>" <lib.so | product> ...
set -o pipefail
TAG=$1; FLAGS=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/$TAG; mkdir -p $O
for round in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = product ]; then unset BPF_LIB; else export BPF_LIB=$PWD/$lib; fi
    name=$(basename $lib .so)
    python3 bench.py --steps 300 --warmup 50 --cpu-budget 0 --extras off --host-path off $FLAGS > $O/${name}_$round.json 2> $O/${name}_$round.err || exit 1
    python3 - $O/${name}_$round.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-28s ms_per_step %.4f kernel_ms %.4f %s" % (sys.argv[2], d["ms_per_step"], d["roofline"]["kernel_ms"], d["config"].get("cloud")))
PY
  done
done | tee $O/summary.txt
