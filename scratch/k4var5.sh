#!/bin/bash
cd /root/repo
cp biod_amd/csrc/pileup.hip /tmp/pileup.hip.orig
sed -i "$1" biod_amd/csrc/pileup.hip
make -C biod_amd/csrc -j8 2>&1 | grep -E "error" -A3
python bench.py --config cfg5 --reads 100000 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', 'k4', d['kernels']['k4_tile']['ms_per_launch'])"
cp /tmp/pileup.hip.orig biod_amd/csrc/pileup.hip
