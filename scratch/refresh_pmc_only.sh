#!/bin/bash
# only the two PMC passes of refresh_profiles.sh
OUT=/root/repo/gpurun_out/refresh
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/$C
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/$C.log 2>&1
done
sed -n '/^python - <<.PY.$/,/^PY$/p' /root/repo/scratch/refresh_profiles.sh | sed '1d;$d' > /tmp/fold.py
python /tmp/fold.py
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE
