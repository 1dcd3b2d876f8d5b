#!/bin/bash
# Collects the per-round evidence kept under profiles/: bench lines, rocprofv3 kernel stats, PMC passes.
# usage (on the GPU box): bash tools/collect_profiles.sh r01c
set -o pipefail
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
python3 bench.py > $O/bench_lf_converged.json 2> $O/bench_lf_converged.err || exit 1
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_lf -o s -- python3 bench.py --cpu-budget 0 > $O/bench_lf_converged_rocprof.json 2> $O/stats_lf.err || exit 1
echo "stats done"
python3 bench.py --model gompertz --resampler systematic --cpu-budget 0 > $O/bench_gompertz_systematic.json 2>> $O/err.log || exit 1
python3 bench.py --cloud spread --steps 100 --warmup 10 --cpu-budget 0 > $O/bench_lf_spread.json 2>> $O/err.log || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_beam -o s -- python3 bench.py --model beam --steps 30 --warmup 5 --cpu-budget 0 > $O/bench_beam.json 2>> $O/err.log || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cloud3d -o s -- python3 bench.py --model cloud3d --steps 10 --warmup 2 --cpu-budget 0 > $O/bench_cloud3d.json 2>> $O/err.log || exit 1
BPF_FORCE_SHARDED=1 python3 bench.py --cpu-budget 0 > $O/bench_lf_sharded_world1.json 2>> $O/err.log || exit 1
BPF_FORCE_SHARDED=1 BPF_SHARD_EXCHANGE=collective python3 bench.py --cpu-budget 0 > $O/bench_lf_sharded_world1_collective.json 2>> $O/err.log || exit 1
# the N = 2 flow with both ranks on this one GPU (code-path rehearsal: the time is two engines sharing a GPU)
bash tools/rehearse_multi.sh 2 100 > $O/bench_lf_rehearsal_2ranks_1gpu.jsonl 2>> $O/err.log || exit 1
echo "models done"
bash tools/pmc_score.sh $TAG || exit 1
find $O -name "*.csv" -size +2M -delete
echo "all done"
