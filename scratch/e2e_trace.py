"""Timeline of the calling thread through batched staging (acquire / commit and next_window in turn): every call with its start, its
duration and its return code, to see where a batch's 7-8 ms go."""
import ctypes as C, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, "/root/repo")
import torch  # noqa: F401
from biod_amd import _lib as L
from biod_amd.pileup import Context
from biod_amd.synth import host_stream_np, scaled
n, batch, P = 10_000_000, 256 << 20, 16
st = host_stream_np(scaled("cfg3", n))
ctx = Context(0); lib, h = L.lib(), ctx.handle
pool = ThreadPoolExecutor(P)
log = []
def call(name, f, *a):
    t = time.perf_counter(); rc = L.check(f(*a), h); log.append((t, time.perf_counter() - t, name, rc)); return rc
for rep in range(2):
    log.clear()
    t0 = time.perf_counter()
    L.check(lib.bgp_batch_begin(h, batch, 0, 1 << 22), h)
    off, futs, m = 0, None, 0
    cw = L.Columns()
    while off < len(st) or futs is not None:
        progressed = False
        if futs is None and off < len(st):
            ptr, cap = C.c_void_p(), C.c_size_t()
            if call("acquire", lib.bgp_batch_acquire, h, C.byref(ptr), C.byref(cap)) == L.BGP_OK:
                m = min(cap.value, len(st) - off)
                dst = np.ctypeslib.as_array(C.cast(ptr.value, C.POINTER(C.c_uint8)), shape=(m,)); src = st[off:off + m]; step = -(-m // P)
                tf = time.perf_counter()
                futs = [pool.submit(np.copyto, dst[k * step:min(m, (k + 1) * step)], src[k * step:min(m, (k + 1) * step)]) for k in range(P)]
                progressed = True
        if futs is not None and all(f.done() for f in futs):
            log.append((tf, time.perf_counter() - tf, "fill(done seen)", 0))
            call("commit", lib.bgp_batch_commit, h, m); off += m; futs = None
            continue
        if call("next_window", lib.bgp_batch_next_window, h, C.byref(cw)) == L.BGP_OK:
            progressed = True
        if not progressed and futs is not None:
            for f in futs: f.result()
    call("end", lib.bgp_batch_end, h)
    while call("next_window", lib.bgp_batch_next_window, h, C.byref(cw)) == L.BGP_OK: pass
    print(f"rep {rep}: {time.perf_counter() - t0:.3f} s")
for t, d, name, rc in log:
    if d > 0.0002 or name != "next_window":
        print(f"{(t - t0) * 1e3:8.2f} ms  +{d * 1e3:6.2f}  {name:16s} rc={rc}")
