#!/bin/bash
# Collects the per-round evidence kept under profiles/: the driver's own command (one line with every BASELINE config:
# other_configs, strong_scaling), rocprofv3 kernel stats of configs 2 / 3 / 5, the variants, the two-rank rehearsal of
# the N > 1 line, the host-buffer seam's timeline, PMC passes for the scoring kernels of every workload bench.py
# attaches a roofline to.
# usage (on the GPU box): bash tools/collect_profiles.sh r03 [part ...]   parts: bench stats variants pmc (default all)
set -o pipefail
TAG=$1; shift
PARTS=${@:-bench stats variants pmc}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has bench; then
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/err.log || exit 1
  python3 bench.py > $O/bench_cfg2_lf_converged.json 2>> $O/err.log || exit 1
  echo "bench lines done"
fi
if has stats; then
  for C in 2 3 5; do
    S=300; W=50; [ $C = 3 ] && S=30 && W=5; [ $C = 5 ] && S=10 && W=2
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg$C -o s -- python3 bench.py --config $C --steps $S --warmup $W --cpu-budget 0 --extras off > $O/bench_cfg${C}_rocprof.json 2>> $O/err.log || exit 1
  done
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_spread -o s -- python3 bench.py --cloud spread --steps 100 --warmup 10 --cpu-budget 0 --extras off > $O/bench_lf_spread_rocprof.json 2>> $O/err.log || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/seam_trace -- python3 tools/seam_probe2.py >> $O/err.log 2>&1 || exit 1
  python3 tools/trace_tail.py $O/seam_trace 7 > $O/seam_timeline.txt || exit 1
  echo "stats done"
fi
if has variants; then
  python3 bench.py --config 1 --steps 1000 --warmup 100 > $O/bench_cfg1.json 2>> $O/err.log || exit 1
  python3 bench.py --model gompertz --resampler systematic --cpu-budget 0 > $O/bench_gompertz_systematic.json 2>> $O/err.log || exit 1
  python3 bench.py --cloud spread --steps 100 --warmup 10 --cpu-budget 0 > $O/bench_lf_spread.json 2>> $O/err.log || exit 1
  python3 bench.py --config 4 --steps 100 --warmup 10 --cpu-budget 0 > $O/bench_cfg4_one_gpu_125k.json 2>> $O/err.log || exit 1
  python3 bench.py --lut exact-edt --cpu-budget 0 > $O/bench_cfg2_exact_edt_lut.json 2>> $O/err.log || exit 1
  BPF_FORCE_SHARDED=1 python3 bench.py --cpu-budget 0 --extras off > $O/bench_lf_sharded_world1.json 2>> $O/err.log || exit 1
  BPF_BENCH_REHEARSAL=1 timeout -k 10 400 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_rehearsal_2ranks_1gpu.json 2>> $O/err.log || exit 1
  python3 tools/seam_probe.py > $O/seam_probe.txt 2>> $O/err.log || exit 1
  echo "variants done"
fi
if has pmc; then
  bash tools/pmc_score.sh ${TAG}_lf lf_converged || exit 1
  bash tools/pmc_score.sh ${TAG}_beam beam_converged --config 3 || exit 1
  bash tools/pmc_score.sh ${TAG}_cloud3d cloud3d_converged --config 5 || exit 1
  bash tools/pmc_score.sh ${TAG}_lf_spread lf_spread --cloud spread || exit 1
  bash tools/pmc_score.sh ${TAG}_lf_125k lf_converged_125000 --config 4 || exit 1
  bash tools/pmc_score.sh ${TAG}_lf_1M lf_converged_1000000 --particles 1000000 || exit 1
  bash tools/pmc_score.sh ${TAG}_lf_cfg1 lf_converged_5000 --config 1 || exit 1
fi
find $O gpurun_out/pmc_${TAG}_* -name "*.csv" -size +2M -delete
echo "all done"
