import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import oracle_lib as O
import test_gpu_parity as T
from biod_amd import _lib as L
from biod_amd.pileup import Context
from biod_amd.synth import scaled, host_stream

cfg = scaled('cfg3', 1_000_000)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
st = host_stream(cfg, 0, n)
ctx = Context(0)
T.stage(ctx, st)
ora = O.pileup(st, use_md=True, skip_zero=True)
exp = T._synth_ref_chars(cfg.seed, 0, ora.pos)
print('oracle vs generator mismatches', int((ora.refbase != exp).sum()), 'of', len(exp))
got = []
for w in ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, use_md_tag=True, skip_zero_coverage=True, window_positions=1 << 20):
    keep = w.columns['coverage'] > 0
    got.append(w.reference_base[keep])
got = np.concatenate(got)
bad = np.nonzero(got != ora.refbase)[0]
print('device vs oracle mismatches', len(bad), 'first', [(int(ora.pos[k]), chr(got[k]), chr(ora.refbase[k])) for k in bad[:6]])
offs = O.frame(st)
for k in range(0, 70):
    f = O.fields_at(st, int(offs[k]))
    if k < 3 or 28 <= k <= 34 or 58 <= k <= 66:
        print(k, f.pos, f.end_pos)
