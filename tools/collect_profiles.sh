#!/bin/bash
# Collects the per-round evidence kept under profiles/: one bench line per single-GPU BASELINE config (with roofline
# and cpu_baseline), rocprofv3 kernel stats of the same commands, PMC passes for the three scoring kernels.
# usage (on the GPU box): bash tools/collect_profiles.sh r02
set -o pipefail
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
python3 bench.py > $O/bench_cfg2_lf_converged.json 2> $O/err.log || exit 1
python3 bench.py --config 1 --steps 1000 --warmup 100 > $O/bench_cfg1.json 2>> $O/err.log || exit 1
python3 bench.py --config 3 --steps 30 --warmup 5 > $O/bench_cfg3_beam.json 2>> $O/err.log || exit 1
python3 bench.py --config 5 --steps 10 --warmup 2 > $O/bench_cfg5_cloud3d.json 2>> $O/err.log || exit 1
echo "bench lines done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg2 -o s -- python3 bench.py --cpu-budget 0 > $O/bench_cfg2_rocprof.json 2>> $O/err.log || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg3 -o s -- python3 bench.py --config 3 --steps 30 --warmup 5 --cpu-budget 0 > $O/bench_cfg3_rocprof.json 2>> $O/err.log || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg5 -o s -- python3 bench.py --config 5 --steps 10 --warmup 2 --cpu-budget 0 > $O/bench_cfg5_rocprof.json 2>> $O/err.log || exit 1
echo "stats done"
python3 bench.py --model gompertz --resampler systematic --cpu-budget 0 > $O/bench_gompertz_systematic.json 2>> $O/err.log || exit 1
python3 bench.py --cloud spread --steps 100 --warmup 10 --cpu-budget 0 > $O/bench_lf_spread.json 2>> $O/err.log || exit 1
python3 bench.py --config 4 --steps 100 --warmup 10 --cpu-budget 0 > $O/bench_cfg4_one_gpu_125k.json 2>> $O/err.log || exit 1
python3 bench.py --particles 1000000 --steps 50 --warmup 10 --cpu-budget 0 > $O/bench_lf_1M_one_gpu.json 2>> $O/err.log || exit 1
BPF_FORCE_SHARDED=1 python3 bench.py --cpu-budget 0 > $O/bench_lf_sharded_world1.json 2>> $O/err.log || exit 1
BPF_BENCH_REHEARSAL=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 100 --warmup 5 --cpu-budget 0 > $O/bench_selflaunch_2ranks_1gpu.json 2>> $O/err.log || exit 1
echo "variants done"
bash tools/pmc_score.sh ${TAG}_lf lf_converged || exit 1
bash tools/pmc_score.sh ${TAG}_beam beam_converged --config 3 || exit 1
bash tools/pmc_score.sh ${TAG}_cloud3d cloud3d_converged --config 5 || exit 1
find $O gpurun_out/pmc_${TAG}_* -name "*.csv" -size +2M -delete
echo "all done"
