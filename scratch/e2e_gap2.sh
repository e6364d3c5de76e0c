#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for V in 0 1; do
  if [ $V = 1 ]; then export E2E_IMPORT_TORCH=1; fi
  echo "== torch first: $V"
  BGP_VERBOSE=1 timeout 600 python scratch/e2e_file.py 2>&1 | grep "torch imported\|load_bam_device\|device inflate" | tail -7
  python - <<'PY'
import os
if os.environ.get("E2E_IMPORT_TORCH"):
    import torch; torch.cuda.init()
import ctypes; ctypes.CDLL(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "biod_amd/libbiodgpu.so"))
print([l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l][::6])
PY
done
