import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch  # noqa
from biod_amd import _lib as L
from biod_amd.pileup import Context
from biod_amd.synth import scaled
ctx = Context(0)
ctx.synth_device(scaled("cfg3", 10_000_000)); ctx.flush(decode=L.BGP_FETCH_ALL); ctx.pileup_run()
lib, h = L.lib(), ctx.handle
for pause in (0.0, 0.001, 0.002):
    L.check(lib.bgp_pileup_begin(h, L.BGP_PILEUP_COLUMNS, 0, 0, 2**64 - 1, 1, 0, 1 << 20), h)
    cw = L.Columns(); calls = []
    while True:
        t0 = time.perf_counter()
        rc = L.check(lib.bgp_pileup_next_window(h, C.byref(cw)), h)
        calls.append(time.perf_counter() - t0)
        if rc != L.BGP_OK: break
        if pause: 
            t1 = time.perf_counter()
            while time.perf_counter() - t1 < pause: pass
    print(f"pause {pause*1e3:.0f} ms: call durations (us) first 8: {[round(x*1e6) for x in calls[:8]]}, median {np.median(calls)*1e6:.0f}")
