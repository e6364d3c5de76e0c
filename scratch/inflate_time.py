"""k_inflate alone on a generator file (one launch: BGP_INFLATE_PIPELINE=0), tolerant of a variant library that inflates wrong bytes.
usage: [BIODGPU_SO=scratch/x.bin] python scratch/inflate_time.py [n_reads]"""
import os, sys, time
os.environ["BGP_INFLATE_PIPELINE"] = "0"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from concurrent.futures import ThreadPoolExecutor
from biod_amd import bamio, _lib as L
from biod_amd.pileup import Context
from biod_amd.synth import host_stream_np, scaled
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
path = "/tmp/e2e_file.bam"
if not os.path.exists(path):
    cfg = scaled("cfg3", n)
    b = memoryview(host_stream_np(cfg))
    with open(path, "wb") as f, ThreadPoolExecutor(32) as ex:
        f.write(bamio.bgzf_block(bamio.bam_header_bytes([(f"chr{i + 1}", 1 << 28) for i in range(max(1, cfg.n_ref))])))
        for blk in ex.map(lambda at: bamio.bgzf_block(bytes(b[at:at + 64000])), range(0, len(b), 64000)):
            f.write(blk)
        f.write(bamio.BGZF_EOF)
ctx = Context(0)
ctx.timing_enable(True)
for rep in range(4):
    ctx.reset(); ctx.timing_reset()
    ok = True
    try:
        ctx.load_bam_device(path)
    except L.BiodGpuError as e:
        ok = False
    k = ctx.timing_get().get("k_inflate")
    print(os.environ.get("BIODGPU_SO", "shipped"), "rep", rep, "ok" if ok else "refused (expected of an ablation)", f"k_inflate {k['total_ms']:.2f} ms" if k else "no timing", flush=True)
