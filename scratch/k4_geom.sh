#!/bin/bash
# A/B of K4 tile geometries: builds pileup.hip variants on the GPU box (hipcc is there) and runs the bench line for each
cd /root/repo
run() {
  rm -f biod_amd/_obj/pileup.hip.o
  make -C biod_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-gpu-rdc $1" > /dev/null 2>&1
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline --reads ${2:-50000000} 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['ms_per_step'], 'k4', d['kernels']['k4_tile']['ms_per_launch'])"
}
run ""
run "-DBGP_K4_W=1024 -DBGP_K4_T=256 -DBGP_K4_HB=512"
run "-DBGP_K4_W=1024 -DBGP_K4_T=512 -DBGP_K4_HB=512"
run "-DBGP_K4_W=2048 -DBGP_K4_T=256 -DBGP_K4_HB=1024"
run "-DBGP_K4_W=4096 -DBGP_K4_T=1024 -DBGP_K4_HB=1024"
rm -f biod_amd/_obj/pileup.hip.o
