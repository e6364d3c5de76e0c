import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
import oracle_lib as O
import test_gpu_parity as T
from biod_amd import _lib as L
from biod_amd.pileup import Context
seed = int(sys.argv[1]); k = seed - 50000
ctx = Context(0)
rng = np.random.default_rng(seed)
st = T._random_stream(rng, int(rng.integers(50, 3000)), quirks=bool(k % 2))
try:
    O.pileup(st); rng.integers(100, 9000)
except O.OracleError:
    pass
cfg = T._MD_CFGS[k % len(T._MD_CFGS)]
st = T._md_stream(rng, int(rng.integers(50, 1500)), **cfg)
print('cfg', cfg, 'bytes', len(st))
T.stage(ctx, st)
for skip in (True, False):
    w = int(rng.integers(50, 5000))
    for chain in ('seq', 'par'):
        os.environ['BGP_MD_CHAIN'] = chain
        try:
            T._check_refbase_one(ctx, st, L.BGP_PILEUP_COLUMNS, skip, 0, O.U64_MAX, w)
            print(skip, chain, 'ok')
        except AssertionError as e:
            print(skip, chain, 'FAIL', str(e)[:200])
offs = O.frame(st)
recs = [O.fields_at(st, int(o)) for o in offs]
# print the reads of the failing reference near the first failing column
for i, f in enumerate(recs):
    if f.ref_id == int(sys.argv[2]) and f.pos < int(sys.argv[3]) + 80:
        try: d = O.dna_read(st, int(offs[i]))
        except Exception: d = '?'
        print(i, f.ref_id, f.pos, f.end_pos, len(d))
