#!/bin/bash
# k4_plan under its ablation builds (timing only: the columns are wrong)
cd /root/repo
for v in "" 1 2 4 7; do
  so=""; [ -n "$v" ] && so="BIODGPU_SO=scratch/pabl$v.bin"
  env $so python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-checks 2>/dev/null | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('[abl $v]', d['ms_per_step'], 'plan', d['kernels']['k4_plan']['ms_per_launch'], 'exec', d['kernels']['k4_tile']['ms_per_launch'])"
done
