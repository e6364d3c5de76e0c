"""Time the MD reference_base path (k6_prep / k6_chain / k6_fill) on a synthetic shard.
usage: python scratch/md_scale.py [cfg3|cfg5] [n_reads]"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from biod_amd import _lib as L
from biod_amd.pileup import Context
from biod_amd.synth import scaled

model = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
cfg = scaled(model, n)
ctx = Context(0)
ctx.reset(); ctx.synth_device(cfg); ctx.flush()
info = ctx.pileup_run()
print('records', ctx.num_records(), 'runs', len(info.runs), 'positions', info.n_positions, flush=True)
for rep in range(3):
    ctx.timing_enable(True); ctx.timing_reset()
    t0 = time.time()
    gen = ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, use_md_tag=True, skip_zero_coverage=True, window_positions=1 << 20)
    w = next(gen)
    t1 = time.time()
    gen.close()
    tm = ctx.timing_get()
    print('rep', rep, 'begin+first window %.1f ms' % ((t1 - t0) * 1e3), {k: (round(v['total_ms'], 3), v['launches']) for k, v in tm.items() if k.startswith('k6')}, flush=True)
