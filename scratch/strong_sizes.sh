#!/bin/bash
# what a rank of the strong-scaling run sees: cfg3 at 1/2, 1/4, 1/8 of the reads on one GPU (fixed per-step costs show here)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for n in 100000000 50000000 25000000 12500000; do
BGP_BENCH_FORCE_COMM=1 timeout 300 python bench.py --reads $n --steps 20 --warmup 3 --no-cpu-baseline --no-md --no-e2e 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print($n, d['value'], d['ms_per_step'], 'sum of kernels', round(sum(v['ms_per_launch']*v['launches_per_step'] for v in d['kernels'].values()),3), d['stages_ms'])"
done
