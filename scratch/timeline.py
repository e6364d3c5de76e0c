"""Kernels and copies of the last W ms of a rocprofv3 --kernel-trace --memory-copy-trace run, by start time.
usage: python scratch/timeline.py <dir> [W ms = 100] [min duration ms = 0.05]"""
import csv, glob, sys
d = sys.argv[1]; W = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0; mind = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
ev = []
for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44] + " q" + r.get("Queue_Id", "?")))
mt = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
if mt:
    for r in csv.DictReader(open(mt[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "")))
end = max(e[1] for e in ev)
ev = sorted(e for e in ev if e[0] >= end - W * 1e6 and (e[1] - e[0]) >= mind * 1e6)
base = ev[0][0]
for s, e, n in ev:
    print(f"{(s - base) / 1e6:9.3f} .. {(e - base) / 1e6:9.3f}  {(e - s) / 1e6:7.3f} ms  {n}")
