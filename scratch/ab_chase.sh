#!/bin/bash
cd /root/repo
run() { BIODGPU_SO=$1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-checks --no-md 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$2', d['value'], d['ms_per_step'], {k:round(v['ms_per_launch'],3) for k,v in d['kernels'].items() if k[:2] in ('k0','k1')})"; }
for i in 1 2 3; do run $PWD/scratch/headlib.bin head; run $PWD/biod_amd/libbiodgpu.so new; done
