#!/bin/bash
# K4's floor with the rows route: the shipped library against ablations (wrong columns): 1 = bases from a cache-resident stretch, 16 = bases as dense as an arena of packed bases, 3 = 1 + no column stores
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for V in shipped abl16 abl1 abl3; do
  if [ $V = shipped ]; then unset BIODGPU_SO; else export BIODGPU_SO=$R/scratch/$V.bin; fi
  python bench.py --no-cpu-baseline --no-checks --no-md --no-e2e --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', d['ms_per_step'], {k: round(v,2) for k,v in d['stages_ms'].items()}, {k: round(v['ms_per_launch'],3) for k,v in d['kernels'].items() if k.startswith('k4')})"
done
