#!/bin/bash
# why is file_e2e inside bench.py slower than scratch/e2e_file.py?  the load's own breakdown (BGP_VERBOSE) in both, and the bench with torch's CPU threads capped
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/gap
for V in default omp1; do
  if [ $V = omp1 ]; then export OMP_NUM_THREADS=1 MKL_NUM_THREADS=1; fi
  BGP_VERBOSE=1 timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-checks --no-md > gpurun_out/gap/bench_$V.json 2> gpurun_out/gap/bench_$V.err
  python - gpurun_out/gap/bench_$V.json $V <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); f = d["e2e"]["file_e2e"]; print(sys.argv[2], "file_e2e", f["Mreads_per_s"], f["best_s"], f["runs_s"], "device_e2e", d["e2e"]["device_e2e"]["Mreads_per_s"])
PY
  grep "load_bam_device\|batch [0-9] at" gpurun_out/gap/bench_$V.err | tail -14
done
unset OMP_NUM_THREADS MKL_NUM_THREADS
BGP_VERBOSE=1 timeout 600 python scratch/e2e_file.py 2>&1 | grep "load_bam_device\|file-E2E" | tail -8
