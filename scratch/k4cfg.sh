#!/bin/bash
# usage: k4cfg.sh W T LCAP  -> rebuild with those constants and bench
cd /root/repo
cp biod_amd/csrc/pileup.hip /tmp/pileup.hip.orig
sed -i "s/constexpr int W = [0-9]*;/constexpr int W = $1;/; s/constexpr int K4_T = [0-9]*;/constexpr int K4_T = $2;/; s/constexpr int LCAP = [0-9]*;/constexpr int LCAP = $3;/" biod_amd/csrc/pileup.hip
make -C biod_amd/csrc -j8 2>&1 | grep -E "error" -A3
python bench.py --reads 20000000 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('W=$1 T=$2 LCAP=$3', 'step', d['ms_per_step'], 'k4', d['kernels']['k4_tile']['ms_per_launch'], 'k1', d['kernels']['k1_emit']['ms_per_launch'], 'k0g', d['kernels']['k0_guess']['ms_per_launch'])"
cp /tmp/pileup.hip.orig biod_amd/csrc/pileup.hip
