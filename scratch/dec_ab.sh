#!/bin/bash
# A/B of decode variants (scratch/<name>.bin built by build_variant_dec.sh) against the product build, cfg3 full size
cd /root/repo
run() { BIODGPU_SO=$1 timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$2', d['value'], d['ms_per_step'], {k:round(v['ms_per_launch'],3) for k,v in d['kernels'].items() if k[:2] in ('k0','k1')}, d.get('checks_full_size',{}).get('ok'))"; }
run $PWD/biod_amd/libbiodgpu.so default
for v in "$@"; do run $PWD/scratch/$v.bin $v; done
run $PWD/biod_amd/libbiodgpu.so default_again
