#!/bin/bash
# the 2 x 4-piece LUT layout (-DBPF_LUT_P24=1) against the 8 x 8 tiles: parity subset, A/B bench lines, PMC accesses per read
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-p24}; mkdir -p $O
BPF_LIB=$PWD/_exp/libbadger_pf_hip_p24.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_oracle.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
bash tools/exp/ab_libs.sh ${1:-p24}_ab "" _exp/libbadger_pf_hip_base.so _exp/libbadger_pf_hip_p24.so || exit 1
bash tools/exp/ab_libs.sh ${1:-p24}_abs "--cloud spread --steps 100 --warmup 10" _exp/libbadger_pf_hip_base.so _exp/libbadger_pf_hip_p24.so || exit 1
BPF_LIB=$PWD/_exp/libbadger_pf_hip_p24.so bash tools/exp/ta_pmc.sh ${1:-p24}_pmc
