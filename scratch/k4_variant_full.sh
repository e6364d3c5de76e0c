#!/bin/bash
cd /root/repo
rm -f biod_amd/_obj/pileup.hip.o
make -C biod_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-gpu-rdc $1" > /dev/null 2>&1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for cfg in cfg3 cfg2 cfg5; do
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --config $cfg 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$cfg', d['value'], d['ms_per_step'], 'k4', d['kernels']['k4_tile']['ms_per_launch'])"
done
rm -f biod_amd/_obj/pileup.hip.o
