#!/bin/bash
# A/B of the fused K1+K2 kernel (BGP_FUSE_MODE=1) at different occupancy caps against the default K1 then K2
cd /root/repo
run() { BIODGPU_SO=$1 BGP_FUSE_MODE=$2 timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$3', d['value'], d['ms_per_step'], {k:v['ms_per_launch'] for k,v in d['kernels'].items() if k[:2] in ('k1','k2')}, d['roofline_decode_8d']['frac'], d.get('checks_full_size',{}).get('ok'))"; }
run $PWD/biod_amd/libbiodgpu.so 2 default
run $PWD/biod_amd/libbiodgpu.so 1 fused_w4
for v in "$@"; do run $PWD/scratch/$v.bin 1 $v; done
