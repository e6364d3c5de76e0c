#!/bin/bash
# Regenerates everything under profiles/ that bench.py and DESIGN.md quote (run on the GPU box through gpurun):
#   bench lines for cfg3 (with cpu_baseline), cfg2, cfg5; rocprofv3 --kernel-trace --stats of the cfg3 bench command;
#   the two PMC passes (FETCH_SIZE, WRITE_SIZE -- separately, as MI355X_MICROARCH.md prescribes) folded into one CSV.
OUT=/root/repo/gpurun_out/refresh
rm -rf $OUT; mkdir -p $OUT
cd /root/repo
timeout 900 python bench.py --steps 5 --warmup 2 > $OUT/r01_bench_cfg3_100M.json 2> $OUT/bench_cfg3.err
timeout 600 python bench.py --steps 5 --warmup 2 --config cfg2 --no-cpu-baseline > $OUT/r01_bench_cfg2_10M.json 2> $OUT/bench_cfg2.err
timeout 600 python bench.py --steps 5 --warmup 2 --config cfg5 --no-cpu-baseline > $OUT/r01_bench_cfg5_200k.json 2> $OUT/bench_cfg5.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/r01_bench_under_rocprofv3_cfg3_100M.json 2> $OUT/stats.err
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/r01_rocprofv3_kernel_stats_cfg3_100M.csv \;
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/$C.log 2>&1
done
python - <<'PY'
import csv, glob, collections, re
out = "/root/repo/gpurun_out/refresh"
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            k = re.sub(r"\(.*", "", k).strip().replace(",", ";")      # template arguments stay: k4_tile has two instantiations per step
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    vals[c] = acc
with open(f"{out}/r01_pmc_hbm_traffic_cfg3_100M.csv", "w") as w:
    w.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 0, cfg3 100M reads\n")
    w.write("# units: KB per dispatch as reported; gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section)\n")
    w.write("kernel,FETCH_SIZE_KB_per_dispatch,WRITE_SIZE_KB_per_dispatch,dispatches,hbm_bytes_per_dispatch_corrected(2*FETCH+WRITE)*1024\n")
    for k in sorted(set(vals["FETCH_SIZE"]) | set(vals["WRITE_SIZE"])):
        f, nf = vals["FETCH_SIZE"].get(k, [0.0, 0]); wv, nw = vals["WRITE_SIZE"].get(k, [0.0, 0])
        n = max(nf, nw, 1)
        fk, wk = f / max(nf, 1), wv / max(nw, 1)
        w.write(f"{k},{fk:.0f},{wk:.0f},{n},{int((2 * fk + wk) * 1024)}\n")
print(open(f"{out}/r01_pmc_hbm_traffic_cfg3_100M.csv").read())
PY
# file-E2E with both loaders (host zlib / k_inflate), and the same run under rocprofv3 for k_inflate's launch time
cd /root/repo
BGP_VERBOSE=1 timeout 600 python scratch/e2e_file.py 10000000 2>&1 | grep -v "framing round" > $OUT/r01_file_e2e_10M.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e2e -- python /root/repo/scratch/e2e_file.py 10000000 > /dev/null 2>&1
find $OUT/e2e -name "*kernel_stats.csv" -exec cp {} $OUT/r01_rocprofv3_kernel_stats_file_e2e_10M.csv \;
timeout 300 python /root/repo/scratch/bai_scale.py 4000000 > $OUT/r01_create_index_4M.log 2>&1
rm -rf $OUT/stats $OUT/FETCH_SIZE $OUT/WRITE_SIZE $OUT/e2e
cat $OUT/r01_bench_cfg3_100M.json | cut -c1-1200
ls -la $OUT
