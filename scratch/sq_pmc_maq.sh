#!/bin/bash
# SQ counters of the per-window consumers (K5 entries, K8 MAQ) over a 10 M-read cfg3 shard
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/sqmaq; rm -rf $OUT; mkdir -p $OUT
for set in "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python $R/scratch/maq_scale.py cfg3 ${1:-4000000} > $OUT/$tag.log 2>&1
done
python - <<'PY'
import csv, glob, collections, re, os
OUT = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "sqmaq")
acc = collections.defaultdict(dict)
for f in sorted(glob.glob(OUT + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")).strip().replace(",", ";")
        acc[k][r["Counter_Name"]] = acc[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
names = sorted({c for v in acc.values() for c in v})
for k in sorted(acc):
    if k.startswith(("k5_", "k8_")):
        print(k); [print("   ", c, f"{acc[k].get(c, 0):.0f}") for c in names]
PY
rm -rf $OUT/SQ_*
