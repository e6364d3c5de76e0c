#!/bin/bash
# per-kernel time of one bench step (rocprofv3 kernel stats over 5 steps + 1 warm-up + set-up), sorted
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-checks > /tmp/kt.json 2>/tmp/kt.err
grep "^{" /tmp/kt.json | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['stages_ms'])"
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:45]:
    n=re.sub(r"\(anonymous namespace\)::","",r["Name"]); n=re.sub(r"\(.*","",n)[:60]
    print(f"{n:62s} calls {int(r['Calls']):4d} total {float(r['TotalDurationNs'])/1e6:9.3f} ms avg {float(r['AverageNs'])/1e3:10.1f} us")
PY
