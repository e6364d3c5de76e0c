#!/bin/bash
# A/B of pileup variants on cfg5 (scratch/<name>.bin) + the long-read parity tests with each
cd /root/repo
run() { BIODGPU_SO=$1 timeout 300 python bench.py --config cfg5 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$2', d['value'], d['ms_per_step'], {k:round(v['ms_per_launch'],3) for k,v in d['kernels'].items() if k[:2] in ('k4',)}, d.get('checks_full_size',{}).get('ok'))"; }
run $PWD/biod_amd/libbiodgpu.so default
for v in "$@"; do run $PWD/scratch/$v.bin $v; BIODGPU_SO=$PWD/scratch/$v.bin timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_interior.py -m gpu -x -q -k "long or accumulate or cfg5 or interior" 2>&1 | grep -E "passed|failed" | tail -1; done
