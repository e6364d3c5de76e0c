#!/bin/bash
cd /root/repo
for f in "" "-DBGP_K4_W=2048 -DBGP_K4_T=512 -DBGP_K4_HB=1024"; do
  rm -f biod_amd/_obj/pileup.hip.o
  make -C biod_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-gpu-rdc $f" > /dev/null 2>&1
  for o in 0 1; do
  BGP_OVERLAP=$o python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('[$f] overlap=$o', d['value'], d['ms_per_step'], {k:v['ms_per_launch'] for k,v in d['kernels'].items() if k in ('k2_unpack','k3_keys_scan','k4_tile')})"
  done
done
rm -f biod_amd/_obj/pileup.hip.o
