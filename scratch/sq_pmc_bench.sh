#!/bin/bash
# SQ counters per kernel of one bench step (cfg3, 100 M reads): instruction mix and wait states of K1 / K2 / K4
cd /tmp && export TMPDIR=/tmp
OUT=${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/sq_pmc; rm -rf $OUT; mkdir -p $OUT
for set in "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/$tag.log 2>&1
done
python - <<'PY'
import csv, glob, collections, re, os
OUT = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "sq_pmc")
acc = collections.defaultdict(dict)
for f in sorted(glob.glob(OUT + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")).strip().replace(",", ";")
        acc[k][r["Counter_Name"]] = acc[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
names = sorted({c for v in acc.values() for c in v})
with open(OUT + "/r02_pmc_sq_per_kernel_cfg3_100M.csv", "w") as w:
    w.write("# rocprofv3 --kernel-trace --pmc <4 SQ counters per pass>, bench.py --steps 1 --warmup 0, cfg3 100 M reads; sums over a kernel's dispatches in the step\n")
    w.write("kernel," + ",".join(names) + "\n")
    for k in sorted(acc):
        if any(x in k for x in ("k0_", "k1", "k2_", "k3_", "k4_", "k12", "k8_")):
            w.write(k + "," + ",".join(f"{acc[k].get(c, 0):.0f}" for c in names) + "\n")
print(open(OUT + "/r02_pmc_sq_per_kernel_cfg3_100M.csv").read())
PY
