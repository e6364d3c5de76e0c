#!/bin/bash
# k12_lds variants on ONE box: HEAD (run-time tests for the K4 products) | templated | templated, plain row stores | round 4's decode.hip in today's tree
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2; do for V in ${VARIANTS:-head tmpl tmpl_nt0 r04dec}; do
  BIODGPU_SO=$R/scratch/$V.bin python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-checks --no-e2e 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', d['ms_per_step'], 'k12', round(d['kernels']['k12_lds']['ms_per_launch'],3), 'K0', round(d['stages_ms']['K0 framing'],3),'K4', round(d['stages_ms']['K4 pileup columns'],3), 'decode-only k12', d['decode_only']['decode_kernels_ms'], 'md', d['md_path']['k6_ms'])"
done; done
