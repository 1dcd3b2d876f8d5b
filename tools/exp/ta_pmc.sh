#!/bin/bash
# L1 accesses per gather of k_score_field for the four scans of tools/ta_probe.py: one rocprofv3 --pmc pass each
# (kernel-trace only), means over the scoring launches.
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-ta_pmc}; mkdir -p $O
for scan in real same arc short; do
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD --output-format csv -d $O/$scan -o p -- python3 tools/ta_probe.py $scan > $O/$scan.out 2>&1 || { echo "pass $scan failed"; tail -5 $O/$scan.out; }
done
python3 - $O <<'PY' | tee $O/summary.txt
import csv, glob, sys, collections
for scan in ("real", "same", "arc", "short"):
    agg = collections.defaultdict(list)
    for f in glob.glob(sys.argv[1] + "/" + scan + "/**/p_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_score_field" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    if m:
        rd = m.get("SQ_INSTS_VMEM_RD", float("nan"))
        print("%-6s launches %d  VMEM_RD %.4g  L1 accesses %.4g (%.1f per read)  L2 requests %.4g (%.2f per read)" % (
            scan, len(next(iter(agg.values()))), rd, m.get("TCP_TOTAL_CACHE_ACCESSES_sum", float("nan")),
            m.get("TCP_TOTAL_CACHE_ACCESSES_sum", float("nan")) / rd, m.get("TCP_TCC_READ_REQ_sum", float("nan")),
            m.get("TCP_TCC_READ_REQ_sum", float("nan")) / rd))
    else:
        print(scan, "no counters")
PY
