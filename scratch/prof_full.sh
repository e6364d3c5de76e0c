#!/bin/bash
mkdir -p /root/repo/gpurun_out
cd /root/repo
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
cat gpurun_out/bench_full.json
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_r1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_r1 -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /root/repo/gpurun_out/bench_prof.json 2> /root/repo/gpurun_out/bench_prof.err
find /root/repo/gpurun_out/prof_r1 -name "*kernel_stats.csv" -exec cat {} \;
find /root/repo/gpurun_out/prof_r1 -name "*kernel_trace.csv" -exec rm {} \;
