cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1; rc=$?; tail -4 gpurun_out/t_all.log; [ $rc -eq 0 ] || exit $rc
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), '%.3e'%d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['prewarm'], d['cpu_baseline']['value'])"
