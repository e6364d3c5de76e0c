#!/bin/bash
# SQ counters of k_inflate on the 10 M-read generator file (what keeps the waves from issuing)
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/inflate_pmc; rm -rf $OUT; mkdir -p $OUT
python /root/repo/scratch/e2e_file.py 10000000 > /dev/null 2>&1
cat > /tmp/one_load.py <<'PY'
import sys; sys.path.insert(0, '/root/repo')
from biod_amd.pileup import Context
c = Context(0); c.load_bam_device("/tmp/e2e_file.bam")
PY
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $OUT/sq_counters.txt
for set in "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_SALU SQ_IFETCH SQ_WAIT_IFETCH SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python /tmp/one_load.py > $OUT/$tag.log 2>&1
done
python - <<'PY'
import csv, glob, collections
acc = collections.OrderedDict()
for f in sorted(glob.glob("/root/repo/gpurun_out/inflate_pmc/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "k_inflate" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, v in acc.items():
    print(f"{k},{v:.0f}")
PY
