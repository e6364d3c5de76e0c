#!/bin/bash
# k4_acc2 variants on one box: shipped | non-temporal column stores | two batches of loads in flight | no depth histogram (wrong histogram) | the flush alone (wrong columns) | the flush alone without the histogram | the flush alone without its column stores
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2; do for V in ${VARIANTS:-base ntcols depth2 abl128 abl8 abl136 abl10}; do
  BIODGPU_SO=$R/scratch/$V.bin timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-checks --no-e2e --no-md 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', d['ms_per_step'], 'k12', round(d['kernels']['k12_lds']['ms_per_launch'],3), 'k4_acc2', round(d['kernels']['k4_acc2']['ms_per_launch'],3), 'K4', round(d['stages_ms']['K4 pileup columns'],3))"
done; done
