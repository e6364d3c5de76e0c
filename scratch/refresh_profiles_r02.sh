#!/bin/bash
# Round 2: regenerates everything under profiles/ that bench.py and DESIGN.md quote (run on the GPU box through gpurun):
#   bench lines for cfg3 (with cpu_baseline), cfg2, cfg5; rocprofv3 --kernel-trace --stats of the cfg3 bench command;
#   the two PMC passes (FETCH_SIZE, WRITE_SIZE -- separately, as MI355X_MICROARCH.md prescribes) folded into one CSV whose
#   header carries the hash of the kernel sources (bench.py quotes the traffic only while that hash matches);
#   device-E2E through batched staging.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/refresh
rm -rf $OUT; mkdir -p $OUT
cd $R
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-checks > $OUT/$C.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, re, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "refresh")
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            k = re.sub(r"\(.*", "", k).strip().replace(",", ";")      # template arguments stay: k4_tile has several instantiations
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    vals[c] = acc
with open(f"{out}/r02_pmc_hbm_traffic_cfg3_100M.csv", "w") as w:
    w.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 0, cfg3 100M reads\n")
    w.write("# units: KB per dispatch (mean over the kernel's dispatches: the run holds the set-up flush and one step); gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section)\n")
    w.write(f"# kernels_sha={bench.kernels_sha()}\n")
    w.write("kernel,FETCH_SIZE_KB_per_dispatch,WRITE_SIZE_KB_per_dispatch,dispatches,hbm_bytes_per_dispatch_corrected(2*FETCH+WRITE)*1024\n")
    for k in sorted(set(vals["FETCH_SIZE"]) | set(vals["WRITE_SIZE"])):
        f, nf = vals["FETCH_SIZE"].get(k, [0.0, 0]); wv, nw = vals["WRITE_SIZE"].get(k, [0.0, 0])
        fk, wk = f / max(nf, 1), wv / max(nw, 1)
        w.write(f"{k},{fk:.0f},{wk:.0f},{max(nf, nw, 1)},{int((2 * fk + wk) * 1024)}\n")
print(open(f"{out}/r02_pmc_hbm_traffic_cfg3_100M.csv").read())
PY
cp $OUT/r02_pmc_hbm_traffic_cfg3_100M.csv $R/profiles/r02_pmc_hbm_traffic_cfg3_100M.csv      # (on the box: the bench lines below quote it)
timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/r02_bench_cfg3_100M.json 2> $OUT/bench_cfg3.err
timeout 600 python bench.py --steps 5 --warmup 2 --config cfg2 --no-cpu-baseline > $OUT/r02_bench_cfg2_10M.json 2> $OUT/bench_cfg2.err
timeout 600 python bench.py --steps 5 --warmup 2 --config cfg5 --no-cpu-baseline > $OUT/r02_bench_cfg5_200k.json 2> $OUT/bench_cfg5.err
BGP_BENCH_FORCE_COMM=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2> $OUT/bench_comm.err | grep "^{" | tail -1 > $OUT/r02_bench_cfg3_100M_one_rank_communicator.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/r02_bench_under_rocprofv3_cfg3_100M.json 2> $OUT/stats.err
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/r02_rocprofv3_kernel_stats_cfg3_100M.csv \;
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats5 -- python $R/bench.py --steps 3 --warmup 1 --config cfg5 --no-cpu-baseline > /dev/null 2> $OUT/stats5.err
find $OUT/stats5 -name "*kernel_stats.csv" -exec cp {} $OUT/r02_rocprofv3_kernel_stats_cfg5_200k.csv \;
cd $R
timeout 600 python scratch/e2e_batched.py 10000000 > $OUT/r02_device_e2e_batched_10M.log 2>&1
BGP_VERBOSE=1 timeout 600 python scratch/e2e_file.py 10000000 2>&1 | grep -v "framing round" > $OUT/r02_file_e2e_10M.log
rm -rf $OUT/stats $OUT/stats5 $OUT/FETCH_SIZE $OUT/WRITE_SIZE
cut -c1-600 $OUT/r02_bench_cfg3_100M.json; cat $OUT/r02_device_e2e_batched_10M.log | tail -4
ls -la $OUT
