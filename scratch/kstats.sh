#!/bin/bash
# usage (on the GPU box): scratch/kstats.sh <tag> [bench args...]: rocprofv3 --kernel-trace --stats of a short bench run, the per-kernel table into gpurun_out/<tag>_kstats.txt
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/ks_$TAG
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$TAG -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-checks --no-md --no-e2e "$@" > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
F=$(find /tmp/ks_$TAG -name '*kernel_stats.csv' | head -1)
python - "$F" > $R/gpurun_out/${TAG}_kstats.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"]); n = re.sub(r"\(.*", "", n)
    print(f'{n[:60]:60s} calls {int(r["Calls"]):6d}  avg_us {float(r["AverageNs"])/1e3:10.1f}  total_ms {float(r["TotalDurationNs"])/1e6:9.2f}')
PY
cat $R/gpurun_out/${TAG}_kstats.txt | head -40
