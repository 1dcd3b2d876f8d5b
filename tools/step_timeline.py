"""Gaps between the kernels of one headline step, from a rocprofv3 --kernel-trace csv (on the GPU box):
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python3 bench.py --steps 40 --warmup 10 --cpu-budget 0
   python3 tools/step_timeline.py gpurun_out/tl
Prints, per kernel of the step, its mean duration and the mean idle time in front of it."""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0]))
rows.sort()
# steps start at k_copy4 (the restore)
idx = [i for i, r in enumerate(rows) if r[2].endswith("k_copy4")]
steps = [rows[a:b] for a, b in zip(idx[:-1], idx[1:])]
# keep the timed-region shape: the most common kernel sequence
shape = collections.Counter(tuple(k for _, _, k in s) for s in steps).most_common(1)[0][0]
sel = [(i, s) for i, s in enumerate(steps) if tuple(k for _, _, k in s) == shape]
print("steps with the common shape: %d of %d; shape: %s" % (len(sel), len(steps), " -> ".join(shape)))
n = len(shape)
dur = [0.0] * n
gap = [0.0] * n
cnt = 0
for i, s in sel:
    if i == 0:
        continue
    prev_end = steps[i - 1][-1][1]
    for k, (st, en, name) in enumerate(s):
        dur[k] += en - st
        gap[k] += st - (prev_end if k == 0 else s[k - 1][1])
    cnt += 1
tot = 0.0
for k in range(n):
    print("%-28s gap before %7.2f us   duration %7.2f us" % (shape[k][-28:], gap[k] / cnt / 1e3, dur[k] / cnt / 1e3))
    tot += (gap[k] + dur[k]) / cnt / 1e3
print("step = %.2f us" % tot)
