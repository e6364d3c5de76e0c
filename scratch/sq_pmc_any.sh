#!/bin/bash
# SQ counters of the kernels whose names match $1 during `python $2...` (default: bench.py with its md / entries legs).  usage: sq_pmc_any.sh k5_ [tag] [-- cmd...]
R=${GRAFT_REPO_ROOT:-/root/repo}
PAT=$1; TAG=${2:-any}
OUT=$R/gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python $R/bench.py ${BENCH_ARGS:-} --steps 1 --warmup 0 --no-cpu-baseline --no-checks > $OUT/$tag.log 2>&1
done
PAT=$PAT OUT=$OUT python - <<'PY'
import csv, glob, collections, re, os
OUT = os.environ["OUT"]; PAT = os.environ["PAT"]
acc = collections.defaultdict(dict); nd = collections.defaultdict(dict)
for f in sorted(glob.glob(OUT + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")).strip().replace(",", ";")
        acc[k][r["Counter_Name"]] = acc[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        nd[k][r["Counter_Name"]] = nd[k].get(r["Counter_Name"], 0) + 1
names = sorted({c for v in acc.values() for c in v})
print("kernel,dispatches," + ",".join(names))
for k in sorted(acc):
    if PAT in k:
        print(k + "," + str(max(nd[k].values())) + "," + ",".join(f"{acc[k].get(c, 0):.0f}" for c in names))
PY
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
