#!/bin/bash
cd /root/repo
for f in "-DBGP_DEEP_WINDOW=0" ""; do
  rm -f biod_amd/_obj/pileup.hip.o
  make -C biod_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-gpu-rdc $f" > /dev/null 2>&1
  echo "[$f]"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "scale_properties or synthetic_parity or deep" 2>&1 | grep -E "passed|failed|^E   " | head -6
done
rm -f biod_amd/_obj/pileup.hip.o
