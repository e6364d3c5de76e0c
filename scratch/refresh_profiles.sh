#!/bin/bash
# Regenerates everything under profiles/ that bench.py and DESIGN.md quote (run on the GPU box through gpurun):
#   usage: scratch/refresh_profiles.sh r05
#   bench lines for cfg3 (with cpu_baseline and parity_interior), cfg2, cfg5; rocprofv3 --kernel-trace --stats of the cfg3 and cfg5
#   bench commands; the PMC passes (FETCH_SIZE, WRITE_SIZE -- separately, as MI355X_MICROARCH.md prescribes) folded into one CSV per
#   configuration whose header carries the read count and the hash of the kernel sources (bench.py quotes the traffic only while both
#   match); SQ counters per kernel of the cfg3 step; device-E2E through batched staging; file-E2E.
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/refresh
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
pmc() {   # config reads
  cd /tmp
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$1_$C -- python $R/bench.py --config $1 --steps 1 --warmup 0 --no-cpu-baseline --no-checks --no-md --no-e2e > $OUT/$1_$C.log 2>&1
  done
  cd $R
  CFG=$1 READS=$2 TAG=$TAG python - <<'PY'
import csv, glob, collections, re, os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import bench
cfg, reads, tag = os.environ["CFG"], os.environ["READS"], os.environ["TAG"]
out = os.path.join(R, "gpurun_out", "refresh")
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{out}/{cfg}_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            k = re.sub(r"\(.*", "", k).strip().replace(",", ";")
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    vals[c] = acc
path = f"{out}/{tag}_pmc_hbm_traffic_{cfg}.csv"
with open(path, "w") as w:
    w.write(f"# rocprofv3 --kernel-trace --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes), bench.py --config {cfg} --steps 1 --warmup 0\n")
    w.write("# units: KB per dispatch (mean over the kernel's dispatches: the run holds the set-up flush and one step); gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section)\n")
    w.write(f"# reads={reads}\n# kernels_sha={bench.kernels_sha()}\n")
    w.write("kernel,FETCH_SIZE_KB_per_dispatch,WRITE_SIZE_KB_per_dispatch,dispatches,hbm_bytes_per_dispatch_corrected(2*FETCH+WRITE)*1024\n")
    for k in sorted(set(vals["FETCH_SIZE"]) | set(vals["WRITE_SIZE"])):
        f, nf = vals["FETCH_SIZE"].get(k, [0.0, 0]); wv, nw = vals["WRITE_SIZE"].get(k, [0.0, 0])
        fk, wk = f / max(nf, 1), wv / max(nw, 1)
        w.write(f"{k},{fk:.0f},{wk:.0f},{max(nf, nw, 1)},{int((2 * fk + wk) * 1024)}\n")
print(open(path).read())
PY
  cp $OUT/${TAG}_pmc_hbm_traffic_$1.csv $R/profiles/      # (on the box: the bench lines below quote it)
  rm -rf $OUT/$1_FETCH_SIZE $OUT/$1_WRITE_SIZE
}
pmc cfg3 100000000
pmc cfg5 200000
pmc cfg2 10000000
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_cfg3_100M.json 2> $OUT/bench_cfg3.err
timeout 600 python bench.py --steps 10 --warmup 3 --config cfg2 > $OUT/${TAG}_bench_cfg2_10M.json 2> $OUT/bench_cfg2.err
timeout 600 python bench.py --steps 10 --warmup 3 --config cfg5 > $OUT/${TAG}_bench_cfg5_200k.json 2> $OUT/bench_cfg5.err
BGP_BENCH_FORCE_COMM=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --parity --no-e2e 2> $OUT/bench_comm.err | grep "^{" | tail -1 > $OUT/${TAG}_bench_cfg3_100M_one_rank_communicator.json
# N = 2 REHEARSED on the one GPU (two ranks, real kernels on both shards, histogram summed over gloo by the script): not a scaling measurement
timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --parity --no-md 2> $OUT/bench_two.err | grep "^{" | tail -1 > $OUT/${TAG}_bench_cfg3_100M_two_ranks_rehearsal_one_gpu.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-md --no-e2e > $OUT/${TAG}_bench_under_rocprofv3_cfg3_100M.json 2> $OUT/stats.err
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_rocprofv3_kernel_stats_cfg3_100M.csv \;
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats5 -- python $R/bench.py --steps 3 --warmup 1 --config cfg5 --no-cpu-baseline --no-md --no-e2e > /dev/null 2> $OUT/stats5.err
find $OUT/stats5 -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_rocprofv3_kernel_stats_cfg5_200k.csv \;
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/statsmd -- python $R/scratch/md_scale.py cfg3 100000000 > $OUT/${TAG}_md_path_cfg3_100M.log 2> $OUT/statsmd.err
find $OUT/statsmd -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_rocprofv3_md_path_cfg3_100M.csv \;
rm -rf $OUT/stats $OUT/stats5 $OUT/statsmd
cd $R
scratch/sq_pmc_k4.sh refresh_sq > /dev/null 2>&1; cp $R/gpurun_out/pmc_refresh_sq/summary.csv $OUT/${TAG}_pmc_sq_per_kernel_cfg3_100M.csv
timeout 600 python scratch/e2e_batched.py 10000000 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_device_e2e_batched_10M.log
BGP_VERBOSE=1 timeout 600 python scratch/e2e_file.py 10000000 2>&1 | grep -v "framing round\|amdgpu.ids" > $OUT/${TAG}_file_e2e_10M.log
{ echo "# cfg3 at 1, 1/2, 1/4, 1/8 of its reads on ONE GPU with the N > 1 code path (BGP_BENCH_FORCE_COMM=1: process group, bgp_hist_reduce in the step): what a rank of the strong-scaling run sees; reads, Mreads/s, ms per step, sum of kernel times, stages"; scratch/strong_sizes.sh 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; } > $OUT/${TAG}_strong_scaling_shard_sizes.log
cut -c1-700 $OUT/${TAG}_bench_cfg3_100M.json; tail -6 $OUT/${TAG}_device_e2e_batched_10M.log
ls -la $OUT
