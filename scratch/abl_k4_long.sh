#!/bin/bash
# k4_acc2<LONG> on cfg5 by parts (wrong columns): shipped | without the accumulate (loads + bookkeeping) | without dealing and accumulate (the ops' scan alone)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for V in "" abl32 abl64; do
  so=$R/biod_amd/libbiodgpu.so; [ -n "$V" ] && so=$R/scratch/$V.bin
  BIODGPU_SO=$so timeout 300 python bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline --no-checks --no-md --no-e2e 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${V:-shipped}', d['ms_per_step'], {k: round(v['ms_per_launch'],3) for k,v in d['kernels'].items() if k.startswith('k4_')})"
done
