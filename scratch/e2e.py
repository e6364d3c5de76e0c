"""Device-E2E and file-E2E rates (not the headline metric): host buffers -> libbiodgpu -> host results."""
import sys, time, os, json
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from biod_amd import _lib as L, bamio
from biod_amd.pileup import Context
from biod_amd.synth import scaled, host_stream_np
import oracle_lib as O
N = 4_000_000
cfg = scaled("cfg3", N)
st = host_stream_np(cfg)
print("stream MB", len(st)/1e6, flush=True)
ctx = Context(0)
res = {}
for rep in range(3):
    ctx.reset()
    t0=time.perf_counter(); ctx.push_inflated(st.tobytes() if rep==0 else b); t1=time.perf_counter() if rep else None
    b = st.tobytes() if rep==0 else b
    t0=time.perf_counter(); ctx.reset(); ctx.push_inflated(b); tp=time.perf_counter()-t0
    t0=time.perf_counter(); ctx.flush(); tf=time.perf_counter()-t0
    t0=time.perf_counter(); info = ctx.pileup_run(); tpl=time.perf_counter()-t0
    t0=time.perf_counter()
    ncols=0
    for w in ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, skip_zero_coverage=False, window_positions=1<<22): ncols += len(w.columns)
    tw=time.perf_counter()-t0
    print(f"rep{rep}: push(memcpy to pinned) {tp:.3f}s flush(H2D+K0+K1) {tf:.3f}s pileup {tpl:.3f}s columns D2H {tw:.3f}s  -> device-E2E {N/(tf+tpl+tw)/1e6:.1f} Mreads/s (excl. staging memcpy), {N/(tp+tf+tpl+tw)/1e6:.1f} incl.", flush=True)
# file E2E
offs = O.frame(b)
recs = (b[int(a):int(c)] for a, c in zip(offs, list(offs[1:]) + [len(b)]))
path = "/tmp/e2e.bam"
t0=time.perf_counter(); bamio.write_bam(path, [(f"chr{i+1}", cfg.ref_len) for i in range(cfg.n_ref)], recs); print("wrote BAM", os.path.getsize(path)/1e6, "MB in", round(time.perf_counter()-t0,1), "s", flush=True)
for nt in (0,):
    for rep in range(2):
        ctx.reset()
        t0=time.perf_counter(); ctx.load_bam(path, nt); tl=time.perf_counter()-t0
        t0=time.perf_counter(); ctx.flush(); info=ctx.pileup_run(); tr=time.perf_counter()-t0
        print(f"file-E2E threads={nt or os.cpu_count()}: load+inflate {tl:.3f}s flush+pileup {tr:.3f}s -> {N/(tl+tr)/1e6:.1f} Mreads/s", flush=True)
