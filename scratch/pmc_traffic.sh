#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_traffic
rm -rf $OUT; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/$C.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for c in ("FETCH_SIZE","WRITE_SIZE"):
    for f in glob.glob(f"/root/repo/gpurun_out/pmc_traffic/{c}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0,0])
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c: continue
            k = r["Kernel_Name"]
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
        for k,(v,n) in sorted(acc.items(), key=lambda kv:-kv[1][0])[:12]:
            print(c, f"{v/n:.4e} per dispatch x{n}", k[:70])
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
