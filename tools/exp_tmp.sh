cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BPF_BENCH_REHEARSAL=1 timeout -k 10 400 python3 bench.py --gpus 5 --steps 50 --warmup 5 --cpu-budget 0 2>gpurun_out/b6.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('5 ranks', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms_per_step'].items()}, d['config']['resampled_to'], d['config']['kld_leaf_count'], d['config']['shard_exchange'], d['n_gpus'])" || tail -20 gpurun_out/b6.err
BPF_BENCH_REHEARSAL=1 BPF_SHARD_EXCHANGE=collective timeout -k 10 400 python3 bench.py --gpus 3 --steps 30 --warmup 5 --cpu-budget 0 2>gpurun_out/b3c.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('3 ranks collective', round(d['ms_per_step'],4), d['config']['resampled_to'], d['config']['kld_leaf_count'], d['config']['shard_exchange'], d['n_gpus'])" || tail -20 gpurun_out/b3c.err
