#!/bin/bash
# usage: md_ab.sh label config [ENV=..]...   -- the use_md_tag leg of bench.py: K6's kernels, ms
cd ${GRAFT_REPO_ROOT:-/root/repo}
label=$1; cfg=$2; shift; shift
env "$@" python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); m=d['md_path']
print('$label', '$cfg', m['kernels_ms'], 'sum', round(sum(m['kernels_ms'].values()), 2))"
