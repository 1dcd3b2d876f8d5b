# last events of a rocprofv3 --kernel-trace --memory-copy-trace run, on one time axis (us)
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
k = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
mm = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
m = list(csv.DictReader(open(mm[0]))) if mm else []
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48]) for r in k] + \
     [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r["Direction"]) for r in m]
ev.sort()
tail = ev[-n:]
t0 = tail[0][0]
for s, e, name in tail:
    print("%9.1f %9.1f  (%6.1f)  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, name))
