import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import ctypes as C
from biod_amd import _lib as L
from biod_amd.pileup import Context
from biod_amd.synth import scaled
ctx = Context(0)
cfg = scaled("cfg3", 4_000_000)
ctx.reset(); ctx.synth_device(cfg); ctx.flush(); info = ctx.pileup_run()
print("positions", info.n_positions, "sumcov", info.sum_coverage)
for win in (65536, 1<<20):
    ctx.timing_enable(True); ctx.timing_reset()
    L.check(L.lib().bgp_pileup_begin(ctx.handle, L.BGP_PILEUP_COLUMNS, 0, 0, (1<<64)-1, 1, 1, win))
    t=time.time(); n=0; ne=0
    cw = L.Columns()
    while True:
        rc = L.check(L.lib().bgp_pileup_next_window(ctx.handle, C.byref(cw)))
        if rc == L.BGP_EOF: break
        n += cw.n_positions
        ne += C.cast(cw.entry_off, C.POINTER(C.c_uint64))[cw.n_positions]
        if n >= 4_000_000: break
    dt=time.time()-t
    tm = ctx.timing_get()
    print("window", win, "positions", n, "entries", ne, "wall s", round(dt,3), "entries/s", round(ne/dt/1e6,1), "M", {k: round(v["total_ms"],2) for k,v in tm.items()})
