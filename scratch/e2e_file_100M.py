"""The metric's own input as a FILE, once: cfg3's 100 M-read coordinate-sorted BAM (29 GB of records, ~16 GB BGZF-compressed) written to
local disk slice by slice, then end to end through the library: bgp_load_bam_device (file -> HBM compressed -> k_inflate) -> bgp_flush_decode
(K0 + k12_lds) -> bgp_pileup_run (K3 + K4).  The reference's path starts at the file (inputstream.d:386-447, block.d:127-216).
Usage (GPU box): python scratch/e2e_file_100M.py [n_reads] [path]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from concurrent.futures import ThreadPoolExecutor
from biod_amd import _lib as L
from biod_amd import bamio
from biod_amd.pileup import Context
from biod_amd.synth import host_stream_np, scaled

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
path = sys.argv[2] if len(sys.argv) > 2 else "/tmp/cfg3_100M.bam"
cfg = scaled("cfg3", n)
os.system("df -h /tmp | tail -1; free -g | head -2")
t0 = time.time()
SL = 2_000_000
inflated = 0
with open(path, "wb") as f, ThreadPoolExecutor(32) as ex:
    f.write(bamio.bgzf_block(bamio.bam_header_bytes([(f"chr{i + 1}", 1 << 28) for i in range(max(1, cfg.n_ref))])))
    carry = b""
    for first in range(0, n, SL):
        st = carry + host_stream_np(cfg, first, min(SL, n - first)).tobytes()
        whole = len(st) - len(st) % 64000 if first + SL < n else len(st)
        b = memoryview(st)
        for blk in ex.map(lambda at: bamio.bgzf_block(bytes(b[at:min(at + 64000, whole)])), range(0, whole, 64000)):
            f.write(blk)
        inflated += whole
        carry = st[whole:]
    f.write(bamio.BGZF_EOF)
fsz = os.path.getsize(path)
print(f"{n} reads, {inflated / 1e9:.2f} GB of records, file {fsz / 1e9:.2f} GB, written in {time.time() - t0:.0f} s", flush=True)
ctx = Context(0)
for rep in range(3):
    ctx.reset()
    t0 = time.perf_counter(); ctx.load_bam_device(path); t1 = time.perf_counter()
    ctx.flush(decode=L.BGP_FETCH_ALL); t2 = time.perf_counter()
    info = ctx.pileup_run(); t3 = time.perf_counter()
    print(f"rep {rep}: load (read + upload + device inflate) {t1 - t0:.3f} s = {fsz / (t1 - t0) / 1e9:.1f} GB/s of file, {inflated / (t1 - t0) / 1e9:.1f} GB/s inflated; "
          f"decode (K0 + k12_lds) {t2 - t1:.3f} s; pileup (K3 + K4) {t3 - t2:.3f} s -> {n / (t3 - t0) / 1e6:.1f} Mreads/s file-E2E; "
          f"records {ctx.num_records()}, columns {info.n_columns}, sum of coverage {info.sum_coverage}, max {info.max_coverage}", flush=True)
assert ctx.num_records() == n
ctx.close()
os.unlink(path)
