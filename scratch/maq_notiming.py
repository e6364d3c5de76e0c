import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from biod_amd import _lib as L
from biod_amd.pileup import Context
from biod_amd.synth import scaled
ctx = Context(0)
ctx.reset(); ctx.synth_device(scaled('cfg3', 10_000_000)); ctx.flush()
info = ctx.pileup_run()
for timing in (True, False, False):
    for wp in (1 << 18, 1 << 20, 1 << 22):
        ctx.timing_enable(timing); ctx.timing_reset()
        t0 = time.time(); n = 0
        for w in ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, window_positions=wp, maq=(0.17, 0.03, 13), copy=False):
            n += 1
        print('timing', timing, 'window', wp, n, 'windows: %.3f s' % (time.time() - t0), flush=True)
