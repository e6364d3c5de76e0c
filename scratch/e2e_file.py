"""File-E2E phases on the GPU box: generator stream -> BGZF file -> bgp_load_bam (+BGP_VERBOSE split) -> flush -> pileup.
Usage: BGP_VERBOSE=1 python scratch/e2e_file.py [n_reads]"""
import os, sys, time
if os.environ.get("E2E_IMPORT_TORCH"):      # torch first: its bundled HIP 7.0 runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7) is then the one libbiodgpu.so binds to
    import torch; torch.cuda.init(); print("torch imported:", torch.version.hip, flush=True)
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from concurrent.futures import ThreadPoolExecutor
from biod_amd import bamio
from biod_amd.pileup import Context
from biod_amd.synth import host_stream_np, scaled

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
cfg = scaled("cfg3", n)
st = host_stream_np(cfg)
path = "/tmp/e2e_file.bam"
t = time.time()
b = memoryview(st)
with open(path, "wb") as f, ThreadPoolExecutor(32) as ex:
    f.write(bamio.bgzf_block(bamio.bam_header_bytes([(f"chr{i + 1}", 1 << 28) for i in range(max(1, cfg.n_ref))])))
    for blk in ex.map(lambda at: bamio.bgzf_block(bytes(b[at:at + 64000])), range(0, len(b), 64000)):
        f.write(blk)
    f.write(bamio.BGZF_EOF)
print(f"{n} reads, stream {len(b) / 1e9:.2f} GB, file {os.path.getsize(path) / 1e9:.2f} GB, written in {time.time() - t:.1f} s", flush=True)
ctx = Context(0)
ctx.timing_enable(True)
for rep in range(6):
    ctx.reset(); ctx.timing_reset()
    dev = rep >= 3
    t0 = time.perf_counter(); (ctx.load_bam_device(path) if dev else ctx.load_bam(path, 0)); t1 = time.perf_counter()
    ctx.flush(); t2 = time.perf_counter()
    info = ctx.pileup_run(); t3 = time.perf_counter()
    if dev:
        k = ctx.timing_get().get("k_inflate")
        if k:                                       # (only timed when the loader falls back to one launch)
            print(f"   k_inflate {k['total_ms']:.2f} ms -> {k['algorithmic_bytes'] / k['total_ms'] / 1e6:.1f} GB/s of inflated bytes", flush=True)
    print(f"rep {rep} ({'device' if dev else 'host'} inflate): load {t1 - t0:.3f} s, flush (H2D + framing) {t2 - t1:.3f} s, pileup {t3 - t2:.3f} s -> {n / (t3 - t0) / 1e6:.1f} Mreads/s file-E2E, columns {info.n_columns}", flush=True)
