#!/bin/bash
# usage: inflate_ab.sh "<flags A>" "<flags B>" ... : k_inflate time per build variant (10 M-read generator file)
cd /root/repo
python scratch/e2e_file.py 10000000 > /dev/null 2>&1
for f in "$@"; do
  rm -f biod_amd/_obj/inflate.hip.o
  make -C biod_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-gpu-rdc $f" > /dev/null 2>&1
  python - <<PY
import sys; sys.path.insert(0, '/root/repo')
from biod_amd.pileup import Context
c = Context(0); c.timing_enable(True)
for rep in range(2):
    c.reset(); c.timing_reset()
    try:
        c.load_bam_device("/tmp/e2e_file.bam")
    except Exception as e:
        print("  (load raised:", str(e)[:80], ")")
    k = c.timing_get().get("k_inflate")
print("[$f] k_inflate %.2f ms" % k["total_ms"])
PY
done
rm -f biod_amd/_obj/inflate.hip.o
