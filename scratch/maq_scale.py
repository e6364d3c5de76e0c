"""Time the per-window consumers (K5 entries, K8 MAQ calls) over a whole synthetic shard.  usage: python scratch/maq_scale.py [cfg] [n_reads] [window]"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from biod_amd import _lib as L
from biod_amd.pileup import Context
from biod_amd.synth import scaled
model = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
wp = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 18
ctx = Context(0)
ctx.reset(); ctx.synth_device(scaled(model, n)); ctx.flush()
info = ctx.pileup_run()
print('records', ctx.num_records(), 'positions', info.n_positions, 'sum coverage', info.sum_coverage, flush=True)
for mode in ('columns', 'maq', 'entries'):
    ctx.timing_enable(True); ctx.timing_reset()
    t0 = time.time(); nw = 0; npos = 0
    kw = dict(window_positions=wp)
    if mode == 'maq': kw['maq'] = (0.17, 0.03, 13)
    if mode == 'entries': kw['want_entries'] = True
    for w in ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, copy=False, **kw):
        nw += 1; npos += len(w.columns)
    dt = time.time() - t0
    tm = ctx.timing_get()
    print(mode, '%d windows, %d positions in %.3f s = %.1f Mpositions/s' % (nw, npos, dt, npos / dt / 1e6), {k: round(v['total_ms'], 1) for k, v in tm.items() if v['total_ms'] > 1}, flush=True)
