cd /root/repo
python scratch/e2e_file.py 10000000 > /dev/null 2>&1
for t in 4 8 12 16 24; do
BGP_READ_THREADS=$t BGP_VERBOSE=1 python - <<PY 2>&1 | grep load_bam_device | tail -2 | sed "s/^/T=$t /" | cut -c1-200
import sys; sys.path.insert(0,'/root/repo')
from biod_amd.pileup import Context
c=Context(0)
for r in range(3):
    c.reset(); c.load_bam_device("/tmp/e2e_file.bam")
PY
done
