#!/bin/bash
# usage: k2_ab.sh "<flags A>" "<flags B>" ... : K2 / step time per build variant (100 M reads)
cd /root/repo
for f in "$@"; do
  rm -f biod_amd/_obj/decode.hip.o biod_amd/_obj/pileup.hip.o
  make -C biod_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-gpu-rdc $f" > /dev/null 2>&1
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('[$f]', d['value'], d['ms_per_step'], 'k2', d['kernels']['k2_unpack']['ms_per_launch'], 'k4', d['kernels']['k4_tile']['ms_per_launch'])"
done
rm -f biod_amd/_obj/decode.hip.o biod_amd/_obj/pileup.hip.o
