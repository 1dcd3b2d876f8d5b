#!/bin/bash
# L1 accesses / L2 requests per vector read of k_score_field for the set in index order and in a sorted order
# (tools/order_probe.py), optionally with an experiment build (BPF_LIB).
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-order_pmc}; mkdir -p $O
i=0
for key in "as generated" "0.01, then 0.2 m"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD --output-format csv -d $O/o$i -o p -- python3 tools/order_probe.py converged "$key" > $O/o$i.out 2>&1 || { echo "pass $i failed"; tail -5 $O/o$i.out; }
done
python3 - $O <<'PY' | tee $O/summary.txt
import csv, glob, sys, collections
for i, name in ((1, "index order"), (2, "heading buckets of 0.01 rad, then 0.2 m tiles")):
    agg = collections.defaultdict(list)
    for f in glob.glob(sys.argv[1] + "/o%d/**/p_counter_collection.csv" % i, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_score_field" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    if m:
        rd = m["SQ_INSTS_VMEM_RD"]
        print("%-48s L1 accesses per read %.1f, L2 requests per read %.2f" % (name, m["TCP_TOTAL_CACHE_ACCESSES_sum"] / rd, m["TCP_TCC_READ_REQ_sum"] / rd))
PY
