#!/bin/bash
# usage: ab_k4.sh label [ENV=..]...  -> one line: label, k4_tile ms, K4 stage ms, step ms
cd ${GRAFT_REPO_ROOT:-/root/repo}
label=$1; shift
env "$@" python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-checks --no-md 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$label', 'k4_tile', k['k4_tile']['ms_per_launch'], 'plan', k['k4_plan']['ms_per_launch'], 'stage', d['stages_ms']['K4 pileup columns'], 'step', d['ms_per_step'])"
