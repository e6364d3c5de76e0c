#!/bin/bash
# rocprofv3 kernel stats of the MD reference_base path (k6_prep / k6_chain / k6_fill) on cfg3 and cfg5 shards
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/prof_md
rm -rf $OUT && mkdir -p $OUT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg3 -- python /root/repo/scratch/md_scale.py cfg3 ${1:-50000000} > $OUT/cfg3.log 2> $OUT/cfg3.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg5 -- python /root/repo/scratch/md_scale.py cfg5 ${2:-100000} > $OUT/cfg5.log 2> $OUT/cfg5.err
tail -n 5 $OUT/cfg3.log; tail -n 5 $OUT/cfg5.log
find $OUT -name "*kernel_stats.csv" | head
find $OUT -name "*kernel_trace.csv" -delete
