#!/bin/bash
# SQ + HBM counters of the K4 kernels of one bench step (cfg3, 100 M reads).  usage: sq_pmc_k4.sh [tag]  (BIODGPU_SO / BGP_* pass through)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-k4}
OUT=$R/gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python $R/bench.py ${BENCH_ARGS:-} --steps 1 --warmup 0 --no-cpu-baseline --no-checks --no-md --no-e2e > $OUT/$tag.log 2>&1
done
python - <<PY
import csv, glob, collections, re, os
OUT = "$OUT"
acc = collections.defaultdict(dict); nd = collections.defaultdict(dict)
for f in sorted(glob.glob(OUT + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")).strip().replace(",", ";")
        acc[k][r["Counter_Name"]] = acc[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        nd[k][r["Counter_Name"]] = nd[k].get(r["Counter_Name"], 0) + 1
names = sorted({c for v in acc.values() for c in v})
with open(OUT + "/summary.csv", "w") as w:
    w.write("# rocprofv3 --kernel-trace --pmc <set per pass>, bench.py --steps 1 --warmup 0 (the set-up flush + one step: kernels of the step dispatch twice where the set-up runs them too), cfg3 100 M reads; MEAN per dispatch\n")
    w.write("kernel,dispatches," + ",".join(names) + "\n")
    for k in sorted(acc):
        if any(x in k for x in ("k0_", "k1", "k2_", "k3_", "k4_", "k12")):
            w.write(k + "," + str(max(nd[k].values())) + "," + ",".join(f"{acc[k].get(c, 0) / max(nd[k].get(c, 1), 1):.0f}" for c in names) + "\n")
print(open(OUT + "/summary.csv").read())
PY
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
