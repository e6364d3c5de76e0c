#!/bin/bash
# only the two PMC passes of refresh_profiles_r02.sh (FETCH_SIZE, WRITE_SIZE) and the folded CSV; then one bench line that quotes it
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/refresh
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/$C
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-checks > $OUT/$C.log 2>&1
done
cd $R
sed -n '/^python - <<.PY.$/,/^PY$/p' $R/scratch/refresh_profiles_r02.sh | sed '1d;$d' > /tmp/fold.py
python /tmp/fold.py > /dev/null
cp $OUT/r02_pmc_hbm_traffic_cfg3_100M.csv $R/profiles/r02_pmc_hbm_traffic_cfg3_100M.csv
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE
timeout 900 python bench.py --steps 10 --warmup 3 2>/dev/null | grep "^{" > $OUT/r02_bench_cfg3_100M.json
head -3 $OUT/r02_pmc_hbm_traffic_cfg3_100M.csv | tail -1; cut -c1-400 $OUT/r02_bench_cfg3_100M.json
