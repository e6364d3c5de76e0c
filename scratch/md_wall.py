"""Where the wall time of bgp_pileup_begin(use_md_tag = 1) up to the first window goes: every timed region beside the wall clock.
usage: python scratch/md_wall.py [n_reads]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from biod_amd import _lib as L
from biod_amd.pileup import Context
from biod_amd.synth import scaled
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
ctx = Context(0)
ctx.synth_device(scaled("cfg3", n)); ctx.flush()
for rep in range(4):
    ctx.pileup_run()
    ctx.timing_enable(True); ctx.timing_reset()
    import ctypes as C
    t0 = time.perf_counter()
    L.check(L.lib().bgp_pileup_begin(ctx.handle, L.BGP_PILEUP_COLUMNS, 1, 0, 2**64 - 1, 1, 0, 1 << 20), ctx.handle)
    t1 = time.perf_counter()
    cw = L.Columns()
    L.check(L.lib().bgp_pileup_next_window(ctx.handle, C.byref(cw)), ctx.handle)
    t2 = time.perf_counter()
    L.check(L.lib().bgp_pileup_next_window(ctx.handle, C.byref(cw)), ctx.handle)
    t3 = time.perf_counter()
    tm = ctx.timing_get(); ctx.timing_enable(False)
    print(f"rep {rep}: bgp_pileup_begin {1e3 * (t1 - t0):.2f} ms, first window {1e3 * (t2 - t1):.2f} ms, second {1e3 * (t3 - t2):.2f} ms; regions:",
          {k: round(v['total_ms'], 3) for k, v in tm.items() if v['launches']}, flush=True)
