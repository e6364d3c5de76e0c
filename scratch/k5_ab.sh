#!/bin/bash
# usage: k5_ab.sh label config [ENV=..]...   -- the K5 leg of bench.py (per-read entries of the first windows), kernels' ms
cd ${GRAFT_REPO_ROOT:-/root/repo}
label=$1; cfg=$2; shift; shift
env "$@" python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['entries_path']
print('$label', '$cfg', e['kernels_ms'], 'k5', e['k5_ms'], 'frac', e['frac'], 'entries', e['entries'])"
