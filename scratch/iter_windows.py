"""How fast does a consumer get the columns of a resident result, window by window (bgp_pileup_begin / bgp_pileup_next_window)?
Raw ABI calls; 10 M reads of cfg3's model = 50 M positions = 1.6 GB of columns."""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch  # noqa: F401
from biod_amd import _lib as L
from biod_amd.pileup import Context
from biod_amd.synth import scaled

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ctx = Context(0)
ctx.synth_device(scaled("cfg3", n)); ctx.flush(decode=L.BGP_FETCH_ALL)
info = ctx.pileup_run()
lib, h = L.lib(), ctx.handle
for window in (65536, 1 << 20, 1 << 22):
    for touch in (False, True):
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            L.check(lib.bgp_pileup_begin(h, L.BGP_PILEUP_COLUMNS, 0, 0, 2**64 - 1, 1, 0, window), h)
            cw = L.Columns(); npos = 0; nw = 0; s = 0
            while True:
                rc = L.check(lib.bgp_pileup_next_window(h, C.byref(cw)), h)
                if rc != L.BGP_OK:
                    break
                npos += cw.n_positions; nw += 1
                if touch:
                    s += int(np.ctypeslib.as_array(C.cast(cw.columns, C.POINTER(C.c_uint32)), shape=(cw.n_positions, 8))[:, 0].sum(dtype=np.int64))
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        print(f"window {window:8d}: {nw:6d} windows, {npos} positions, {'summing coverage' if touch else 'taking the windows only'}: {best * 1e3:8.1f} ms = {npos * 32 / best / 1e9:5.1f} GB/s of columns, {best / max(nw, 1) * 1e6:7.1f} us a window", flush=True)
