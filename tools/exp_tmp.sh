cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_next_rows.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_fullsize.py tests/test_gpu_motion.py -x -q > gpurun_out/t_f.log 2>&1; rc=$?; tail -5 gpurun_out/t_f.log; [ $rc -eq 0 ] || exit $rc
python3 bench.py --cpu-budget 0 2>gpurun_out/b2.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain', round(d['ms_per_step'],4), d['host_buffer_path'])"
python3 bench.py --cloud spread --steps 100 --warmup 10 --cpu-budget 0 2>gpurun_out/sp2.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spread', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms_per_step'].items()}, d['host_buffer_path'])"
python3 tools/time_next_rows.py 2>/dev/null | tail -12
