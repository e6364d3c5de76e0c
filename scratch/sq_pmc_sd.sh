#!/bin/bash
# SQ counters of the single-pass decode kernel (smaller run: 20 M reads)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/sq_pmc_sd; rm -rf $OUT; mkdir -p $OUT
for set in "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  BIODGPU_SO=${SO:-$R/biod_amd/libbiodgpu.so} timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-checks --reads 20000000 > $OUT/$tag.log 2>&1
done
python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(dict)
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")).strip().replace(",", ";")
        acc[k][r["Counter_Name"]] = acc[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k in sorted(acc):
    if "stream_decode" in k or "k2_unpack" in k:
        print(k)
        for c in sorted(acc[k]): print("   ", c, f"{acc[k][c]:.3e}")
PY
