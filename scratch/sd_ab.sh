#!/bin/bash
# A/B of single-pass decode variants: scratch/sd_ab.sh name1 name2 ...   ("base" = the in-tree library)
cd /root/repo
for v in "$@"; do
  so=$PWD/scratch/$v.bin; [ "$v" = base ] && so=$PWD/biod_amd/libbiodgpu.so
  BIODGPU_SO=$so timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/tmp/err_$v.log | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$v', d['value'], d['ms_per_step'], {k:v['ms_per_launch'] for k,v in d['kernels'].items() if k[:2] in ('k0','k1','k2')}, d['roofline_decode_8d']['frac'], d.get('checks_full_size',{}).get('ok'))"
  grep sd_prof /tmp/err_$v.log | tail -1
done
