#!/bin/bash
# k12_lds with and without what it writes for K4, same box: rows (default) | rows + side array | side array only | neither (decode-only leg of the same run)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2; do for V in "X=1" "BGP_K12_CIG4=1" "BGP_K4_ROWS=0"; do
  env $V python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-checks --no-e2e 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', d['ms_per_step'], 'k12', round(d['kernels']['k12_lds']['ms_per_launch'],3), 'K4', round(d['stages_ms']['K4 pileup columns'],3), 'decode-only k12', d['decode_only']['decode_kernels_ms'], 'md', d['md_path']['k6_ms'])"
done; done
