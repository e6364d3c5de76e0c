"""Timeline of the last bgp_load_bam_device of scratch/e2e_file.py under rocprofv3 --kernel-trace --memory-copy-trace:
the k_inflate dispatches and the uploads of the compressed file, ms from the first of them.
usage: python scratch/inflate_timeline.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv>"""
import csv, glob, sys
d = sys.argv[1]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
mt = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
ks = []
for r in csv.DictReader(open(kt)):
    if "k_inflate" in r["Kernel_Name"]: ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)))
ms = []
if mt:
    for r in csv.DictReader(open(mt[0])):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        ms.append((s, e, r.get("Direction", "")))
ks.sort()
# the last load: the k_inflate dispatches behind the last gap of more than 50 ms
win = int(sys.argv[2]) if len(sys.argv) > 2 else 400          # ms in front of the last dispatch's end
ks = [k for k in ks if k[0] >= ks[-1][1] - win * 1_000_000]
t0 = ks[-1][1] - win * 1_000_000
up = [m for m in ms if m[0] >= t0 and m[0] <= ks[-1][1] and "HOST_TO_DEVICE" in m[2].upper() and m[1] - m[0] > 200_000]
base = min([k[0] for k in ks] + [m[0] for m in up])
print("uploads (> 0.2 ms):", len(up))
for m in up: print(f"  up   {(m[0] - base) / 1e6:8.2f} .. {(m[1] - base) / 1e6:8.2f}  ({(m[1] - m[0]) / 1e6:.2f} ms)")
print("k_inflate dispatches:", len(ks))
for k in ks: print(f"  inf  {(k[0] - base) / 1e6:8.2f} .. {(k[1] - base) / 1e6:8.2f}  ({(k[1] - k[0]) / 1e6:.2f} ms, grid {k[2]})")
