"""BGP_VERBOSE timeline of one streamed pass over a generator file (usage: python scratch/stream_timeline.py [n_reads] [batch_MiB])"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from concurrent.futures import ThreadPoolExecutor
from biod_amd import _lib as L, bamio
from biod_amd.pileup import Context
from biod_amd.synth import host_stream_np, scaled
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
batch = (int(sys.argv[2]) if len(sys.argv) > 2 else 256) << 20
cfg = scaled("cfg3", n); path = "/tmp/tl.bam"
st = host_stream_np(cfg).tobytes(); b = memoryview(st)
with open(path, "wb") as f, ThreadPoolExecutor(32) as ex:
    f.write(bamio.bgzf_block(bamio.bam_header_bytes([(f"chr{i + 1}", 1 << 28) for i in range(max(1, cfg.n_ref))])))
    for blk in ex.map(lambda at: bamio.bgzf_block(bytes(b[at:at + 64000])), range(0, len(st), 64000)):
        f.write(blk)
    f.write(bamio.BGZF_EOF)
lib = L.lib(); ctx = Context(0); h = ctx.handle
def run():
    t0 = time.perf_counter()
    L.check(lib.bgp_stream_bam_begin(h, path.encode(), batch, 0, 1 << 22), h)
    cw = L.Columns(); nw = 0
    while L.check(lib.bgp_batch_next_window(h, C.byref(cw)), h) == L.BGP_OK:
        nw += 1
        if os.environ.get("BGP_VERBOSE"): print(f"[consumer] {1e3 * (time.perf_counter() - t0):8.2f} ms: window {nw}, {cw.n_positions} positions", file=sys.stderr)
    dt = time.perf_counter() - t0; ctx.reset(); return dt
run()
os.environ["BGP_VERBOSE"] = "1"
ctx.close(); ctx = Context(0); h = ctx.handle
run()     # (warm-up of this ctx)
print("---- timed pass ----", file=sys.stderr)
ctx.timing_enable(True); ctx.timing_reset()
dt = run()
print(f"{n / dt / 1e6:.1f} Mreads/s; kernel timers: " + ", ".join(f"{k} {v['total_ms']:.1f} ms / {v['launches']}" for k, v in sorted(ctx.timing_get().items(), key=lambda kv: -kv[1]['total_ms'])[:12]), file=sys.stderr)
