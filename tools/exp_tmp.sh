cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1; rc=$?; tail -5 gpurun_out/t_all.log; [ $rc -eq 0 ] || exit $rc
BPF_BENCH_REHEARSAL=1 timeout -k 10 300 python3 bench.py --gpus 4 --steps 100 --warmup 5 --cpu-budget 0 2>gpurun_out/b4.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4 ranks', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms_per_step'].items()}, d['config']['resampled_to'], d['config']['kld_leaf_count'], d['config']['shard_exchange'])"
