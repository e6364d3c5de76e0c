#!/bin/bash
# One fuzz campaign on the GPU box: every differential fuzzer against the oracle on fresh seeds; summary lines into gpurun_out/fuzz_campaign.txt
# usage: scratch/fuzz_campaign.sh <seed prefix, e.g. 61>
P=${1:-61}
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
OUT=gpurun_out/fuzz_campaign.txt
SHA=$(python -c "import bench; print(bench.kernels_sha())")
echo "# Fuzz campaign, kernels_sha $SHA, MI355X, seeds ${P}10000 .. ${P}90000; every round compares the device path with the oracle bit for bit" > $OUT
run() { name=$1; shift; ( timeout 900 python "$@" 2>&1 | grep -E "MISMATCH|STREAM|REGION|rounds" | tail -4 | sed "s/^/$name: /" ) >> $OUT; }
run fuzz_gpu scratch/fuzz_gpu.py 400 ${P}10000
run fuzz_rows scratch/fuzz_rows.py 250 ${P}15000
run fuzz_tiny scratch/fuzz_tiny_batches.py 400 ${P}20000
run fuzz_chunks scratch/fuzz_chunks.py 400 ${P}30000
run fuzz_batch scratch/fuzz_batch_api.py 200 ${P}40000
run fuzz_long scratch/fuzz_long_reads.py 250 ${P}50000
run fuzz_inflate scratch/fuzz_inflate.py 300 ${P}60000
run fuzz_maq scratch/fuzz_maq.py 40 ${P}70000
run fuzz_loader scratch/fuzz_loader.py 60 ${P}80000
run fuzz_stream scratch/fuzz_stream.py 300 ${P}90000
run fuzz_region scratch/fuzz_region_device.py 300 ${P}95000
cat $OUT
