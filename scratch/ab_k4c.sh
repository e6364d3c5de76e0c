#!/bin/bash
# usage: ab_k4c.sh label config [ENV=..]...
cd ${GRAFT_REPO_ROOT:-/root/repo}
label=$1; cfg=$2; shift; shift
env "$@" python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-checks --no-md 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$label', '$cfg', {n: v['ms_per_launch'] for n, v in k.items() if n.startswith('k4')}, 'stage', d['stages_ms']['K4 pileup columns'], 'step', d['ms_per_step'])"
