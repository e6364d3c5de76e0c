#!/bin/bash
# k12_lds by parts on one box (wrong arenas): shipped | no SEQ arena | no QUAL arena | the window load and the rows alone
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2; do for V in base k12_noseq k12_noqual k12_rowsonly; do
  BIODGPU_SO=$R/scratch/$V.bin timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-checks --no-e2e --no-md 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', d['ms_per_step'], 'k12', round(d['kernels']['k12_lds']['ms_per_launch'],3))"
done; done
