#!/bin/bash
# BASELINE config 4, ready to press on an 8-GPU node: the 1 / 2 / 4 / 8-GPU strong-scaling curve of the 100 M-read BAM split by contig
# (bench.py, generated on each rank's device), then the FILE form (one indexed BAM, ranks cut by the .bai, device-inflated region loads).
# One JSON line per N under gpurun_out/cfg4/.  (CFG4_READS / CFG4_FILE_READS: smaller inputs, for a rehearsal of this script on a box with fewer GPUs -- the lines then say REHEARSAL.)  Nothing here is a measurement until it has run on 8 GPUs: "unmeasured on hardware" until then.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/cfg4
export HSA_ENABLE_IPC_MODE_LEGACY=0
for N in 1 2 4 8; do
  timeout 900 python bench.py --gpus $N ${CFG4_READS:+--reads $CFG4_READS} --steps 10 --warmup 3 --no-cpu-baseline --parity --no-e2e 2> gpurun_out/cfg4/bench_n$N.err | grep "^{" | tail -1 > gpurun_out/cfg4/bench_n$N.json
  python -c "import json; d=json.load(open('gpurun_out/cfg4/bench_n$N.json')); print('N=$N', d['value'], 'Mreads/s', d['ms_per_step'], 'ms', d.get('reduce'), 'hist', d.get('hist_reduce_equals_sum_of_ranks'), 'parity', d['parity_interior']['all_equal'])"
done
for N in 1 2 4 8; do
  timeout 1800 python bench_cfg4_file.py --gpus $N --reads ${CFG4_FILE_READS:-100000000} 2> gpurun_out/cfg4/file_n$N.err | grep "^{" | tail -1 > gpurun_out/cfg4/file_n$N.json
  python -c "import json; d=json.load(open('gpurun_out/cfg4/file_n$N.json')); print('file N=$N', d['Mreads_per_s'], 'Mreads/s', d['best_s'], 's', 'REHEARSAL' if d['REHEARSAL'] else '', 'equals whole file', d['equals_whole_file'])"
done
