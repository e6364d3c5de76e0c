import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from biod_amd import _lib as L, bamio
from biod_amd.pileup import Context
ctx = Context(0)
_, st = bamio.read_bam('/root/repo/tests/golden/data/b7_295_chunk.bam')
ctx.reset(); ctx.push_inflated(st); ctx.flush(); ctx.pileup_run()
want = {}
for w in ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, window_positions=500, maq=(0.17, 0.03, 13)):
    for k in np.nonzero(w.columns["coverage"])[0]:
        want[(w.ref_id, int(w.first_position) + int(k))] = w.maq[k].copy()
got = {}
pieces = [st[o:o + 7001] for o in range(0, len(st), 7001)]
for w in ctx.batched_windows(pieces, batch_bytes=40000, window_positions=333, maq=(0.17, 0.03, 13)):
    for k in np.nonzero(w.columns["coverage"])[0]:
        got[(w.ref_id, int(w.first_position) + int(k))] = (w.maq[k].copy(), int(w.columns["coverage"][k]))
print(len(want), len(got))
bad = [k for k in want if k not in got or got[k][0].tobytes() != want[k].tobytes()]
print(len(bad), bad[:10])
for k in bad[:8]:
    print(k, want[k], got.get(k))
