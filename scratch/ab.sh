#!/bin/bash
# usage (GPU box): scratch/ab.sh "<env A>" "<env B>" [bench args]: the bench's stage and kernel timers, A against B, twice each
A=$1; B=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
for rep in 1 2; do for V in A B; do
  if [ $V = A ]; then E="$A"; else E="$B"; fi
  env $E python bench.py --no-cpu-baseline --no-checks --no-md --no-e2e --steps 10 --warmup 2 "$@" > gpurun_out/ab_$V$rep.json 2> gpurun_out/ab_$V$rep.err
  python - gpurun_out/ab_$V$rep.json "$V$rep [$E]" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(sys.argv[2], d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in d["stages_ms"].items()}, {k: round(v["ms_per_launch"], 3) for k, v in d["kernels"].items()})
except Exception as e: print(sys.argv[2], "ERR", e, open(sys.argv[1].replace(".json", ".err")).read()[-500:])
PY
done; done
