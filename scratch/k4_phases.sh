#!/bin/bash
cd /root/repo
rm -f biod_amd/_obj/pileup.hip.o
make -C biod_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-gpu-rdc -DBGP_K4_PHASES $1" > /dev/null 2>&1
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --reads ${2:-50000000} 2>&1 | grep -E "k4 phases|k4 tiles|k4 occ|metric" | cut -c1-400 | tail -5
rm -f biod_amd/_obj/pileup.hip.o
