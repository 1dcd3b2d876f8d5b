#!/bin/bash
# the gather-locality probe (tools/ta_probe.py) on several builds of the engine, on one box
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-probe}; mkdir -p $O; shift
for lib in "$@"; do
  echo "== $lib"
  if [ "$lib" = product ]; then unset BPF_LIB; else export BPF_LIB=$PWD/$lib; fi
  python3 tools/ta_probe.py 2>&1 | grep -v Warning || exit 1
done | tee $O/ta.log
