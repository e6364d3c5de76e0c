#!/bin/bash
cd /root/repo
for n in 12500000 25000000; do for g in 2048 4096 8192 16384 32768; do
BGP_G_BLOCKS=$g timeout 300 python bench.py --reads $n --steps 20 --warmup 3 --no-cpu-baseline --no-checks --no-md 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print($n, $g, d['ms_per_step'], {k:v['ms_per_launch'] for k,v in d['kernels'].items() if k.startswith('k4')})"
done; done
