import sys, time, json, subprocess
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from biod_amd import _lib as L
from biod_amd.pileup import Context
from biod_amd.synth import scaled, host_stream
import oracle_lib as O
ctx = Context(0)
# parity on a small long-read sample
cfg = scaled("cfg5", 300)
st = host_stream(cfg)
ctx.reset(); ctx.synth_device(cfg); ctx.flush()
t=time.time(); info = ctx.pileup_run(); print("small pileup s", time.time()-t)
ora = O.pileup(st)
print("small:", (info.n_columns, info.sum_coverage, info.max_coverage), (ora.n_columns, ora.sum_coverage, ora.max_coverage))
assert (info.n_columns, info.sum_coverage, info.max_coverage) == (ora.n_columns, ora.sum_coverage, ora.max_coverage)
# scale
for n in (20000, 200000):
    cfg = scaled("cfg5", n)
    ctx.reset(); t=time.time(); nb = ctx.synth_device(cfg); tg=time.time()-t
    t=time.time(); ctx.flush(); tf=time.time()-t
    ctx.timing_enable(True); ctx.timing_reset()
    t=time.time(); ctx.decode_device(L.BGP_FETCH_ALL); td=0
    t=time.time(); info = ctx.pileup_run(); tp=time.time()-t
    dec = ctx.decode_fetch(0)
    cov = (dec.rec["end_pos"]-dec.rec["pos"]).astype(np.int64)
    print({k: round(v["total_ms"],3) for k,v in ctx.timing_get().items()})
    print(n, "bytes", nb, "gen", round(tg,3), "flush", round(tf,4), "decode", round(td,4), "pileup", round(tp,4), "cols", info.n_columns, "sumcov ok", info.sum_coverage == int(cov.sum()), "max", info.max_coverage, flush=True)
