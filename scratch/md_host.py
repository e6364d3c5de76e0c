import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from biod_amd import _lib as L
from biod_amd.pileup import Context
from biod_amd.synth import scaled
import ctypes as C
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
ctx = Context(0)
ctx.reset(); ctx.synth_device(scaled('cfg3', n)); ctx.flush()
info = ctx.pileup_run()
for wp in (1 << 20, 1024):
    for rep in range(3):
        ctx.timing_enable(True); ctx.timing_reset()
        t0 = time.time()
        gen = ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, use_md_tag=True, skip_zero_coverage=True, window_positions=wp)
        w = next(gen); t1 = time.time()
        w2 = next(gen); t2 = time.time()
        gen.close()
        tm = ctx.timing_get()
        print(wp, 'rep', rep, 'begin+first %.1f ms, second %.2f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), 'kernels %.1f ms' % sum(v['total_ms'] for k, v in tm.items() if k.startswith('k6')), flush=True)
