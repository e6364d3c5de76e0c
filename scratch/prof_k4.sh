#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_k4
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/a -- python /root/repo/bench.py --reads 20000000 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM --output-format csv -d $OUT/b -- python /root/repo/bench.py --reads 20000000 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/b.log 2>&1
find $OUT -name "*.csv" | head -20
python - <<'PY'
import csv, glob, collections
for sub in ("a","b"):
    for f in glob.glob(f"/root/repo/gpurun_out/pmc_k4/{sub}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            if "k4_tile" in k or "k2_unpack" in k or "k1_emit" in k:
                print(sub, k, {n: f"{x:.3e}" for n, x in v.items()})
PY
