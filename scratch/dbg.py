import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from biod_amd.pileup import Context
from biod_amd.synth import scaled, make_cfg
import time
ctx = Context(0)
for n in (50_000, 400_000):
    cfg = scaled("cfg3", n)
    ctx.reset(); ctx.synth_device(cfg)
    t=time.time(); ctx.flush(); print(n, 'flush s', time.time()-t, ctx.num_records(), flush=True)
