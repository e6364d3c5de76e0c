#!/bin/bash
# runs bench.py (short) once per library variant and prints the K4 kernel times; usage: scratch/ab_run.sh v1 v2 ... (names of scratch/*.bin; "base" = the in-tree build)
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  if [ "$v" = base ]; then unset BIODGPU_SO; else export BIODGPU_SO=$R/scratch/$v.bin; fi
  timeout 150 python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$v', 'step', d['ms_per_step'], ' '.join('%s=%.2f'%(n,k[n]['ms_per_launch']) for n in k if n.startswith(('k4','k3'))))"
done
