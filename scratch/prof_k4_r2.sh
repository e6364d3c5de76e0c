#!/bin/bash
# rocprofv3 kernel stats of a short bench run (cfg3), summary copied to gpurun_out/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/prof -o k4 --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > $R/gpurun_out/prof_bench.log 2>&1
f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/k4_kernel_stats.csv
head -16 $R/gpurun_out/k4_kernel_stats.csv | cut -c1-220
