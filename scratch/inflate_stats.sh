#!/bin/bash
# match-distance statistics of k_inflate on the fixtures and on a generator file (build with -DBGP_INFLATE_STATS)
cd /root/repo
rm -f biod_amd/_obj/inflate.hip.o
make -C biod_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-gpu-rdc -DBGP_INFLATE_STATS" > /dev/null 2>&1
python - <<'PY'
import sys; sys.path.insert(0, '/root/repo')
from biod_amd.pileup import Context
for f in ["ex1_header.bam", "bins.bam", "mg1655_chunk.bam", "illu_20_chunk.bam"]:
    print(f, flush=True); c = Context(0); c.load_bam_device("/root/repo/tests/golden/data/" + f)
PY
python scratch/e2e_file.py 2000000 2>&1 | grep "inflate stats" | head -1
rm -f biod_amd/_obj/inflate.hip.o
