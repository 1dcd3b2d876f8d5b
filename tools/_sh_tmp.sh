cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sh
timeout -k 10 500 python -m pytest tests/test_gpu_sharded.py -x -q > gpurun_out/sh/pytest.log 2>&1 || exit 1
for x in auto collective auto; do
BPF_SHARD_EXCHANGE=$x BPF_FORCE_SHARDED=1 python3 bench.py --steps 300 --warmup 10 --cpu-budget 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$x', d['ms_per_step'], d['config']['shard_exchange'], d['roofline']['achieved'])" || exit 1
done > gpurun_out/sh/b2.log 2>&1
