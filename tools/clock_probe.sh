#!/bin/bash
# samples the shader clock while the bench loop runs (diagnostic: is the scoring kernel clock / power limited?)
cd $GRAFT_REPO_ROOT
python bench.py --steps 20000 --warmup 5 --cpu-budget 0 > gpurun_out/clock_bench.json 2>/dev/null &
BP=$!
sleep 8
for i in $(seq 1 12); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | head -4 | tr '\n' ' '
  echo
  sleep 0.3
done
wait $BP
python -c "
import json; d=json.load(open('gpurun_out/clock_bench.json')); print(d['ms_per_step'], d['roofline']['kernel_ms'])"
