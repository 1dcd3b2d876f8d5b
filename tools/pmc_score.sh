#!/bin/bash
# PMC passes for the scoring kernel of one bench configuration (separate passes, kernel-trace only; see
# MI355X_MICROARCH.md), a summary of the means, and the entry of profiles/pmc_traffic.json that bench.py reads.
# usage (on the GPU box): bash tools/pmc_score.sh <tag> <json-key> [bench args ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; KEY=$2; shift; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
ARGS="$@"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM" \
           "SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_LDS" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p -- python3 bench.py --steps 5 --warmup 2 --prewarm 3 --prewarm-seconds 0 --cpu-budget 0 --extras off --host-path off $ARGS > $OUT/p$i.out 2>&1 || echo "pass $i failed"
done
python3 - "$OUT" "$TAG" "$KEY" $ARGS <<'PY'
import csv, collections, glob, json, os, sys
out, tag, key = sys.argv[1:4]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_score" in k or "k_cloud_score" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    # the kernel with the most dispatches first (pmc_json.py reads the first): a spread cloud's first update still
    # runs the index-order form, the host-buffer path has forms of its own
    for k, d in sorted(agg.items(), key=lambda kv: -max(len(v) for v in kv[1].values())):
        print(k, file=fh); print(k)
        for c, v in sorted(d.items()):
            line = "  %-34s n=%d mean=%.5g" % (c, len(v), sum(v) / len(v))
            print(line, file=fh); print(line)
PY
# the entry bench.py reads (tools/pmc_json.py): particles / beams of the configuration from the bench line of pass 1
read N B <<< $(python3 - "$OUT" <<'PY'
import json, sys
n, b = 100000, 1081
for line in open(sys.argv[1] + "/p1.out", errors="replace"):
    if line.lstrip().startswith('{"metric"'):
        d = json.loads(line)
        n = d["config"]["particles_per_gpu"]
        w = d["config"]["workload"]
        b = 65536 if "point cloud" in w or "1024-point" in w else (181 if "181" in w else 1081)
print(n, b)
PY
)
python3 tools/pmc_json.py $OUT/summary.txt $KEY $N $B profiles/${TAG}_pmc_summary.txt gpurun_out/pmc_traffic_$KEY.json
