"""Device-E2E through batched staging (NOT the headline: the input crosses PCIe): a synthetic record stream in host memory ->
pinned staging -> HBM -> K0..K4 per batch -> final columns handed out window by window (pinned host memory, read in place).
Three ways of getting the bytes into the pinned buffers, each with and without a consumer that reads every column:
  push     bgp_batch_push copies the caller's bytes (its own threads, BGP_PUSH_THREADS)
  acquire  bgp_batch_acquire / _commit: the producer writes straight into the pinned buffer (here: P threads copying from the
           array, standing in for P inflate workers that would otherwise write into buffers of their own)
Raw ctypes calls: the Python facade's copies would be the slowest thing measured."""
import ctypes as C
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, "/root/repo")
if not os.environ.get("BGP_E2E_ONLY"):
    import torch  # noqa: F401  (first: its HIP runtime)
from biod_amd import _lib as L
from biod_amd.pileup import Context
from biod_amd.synth import host_stream_np, scaled

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256 << 20
P = int(sys.argv[3]) if len(sys.argv) > 3 else 16
st = host_stream_np(scaled("cfg3", n))
ctx = Context(0)
lib, h = L.lib(), ctx.handle
piece = 64 << 20
pool = ThreadPoolExecutor(P)


def drain(stats, touch):
    cw = L.Columns()
    while True:
        rc = L.check(lib.bgp_batch_next_window(h, C.byref(cw)), h)
        if rc != L.BGP_OK:
            return rc
        stats[0] += cw.n_positions
        if touch:
            cov = np.ctypeslib.as_array(C.cast(cw.columns, C.POINTER(C.c_uint32)), shape=(cw.n_positions, 8))[:, 0]
            stats[1] += int(np.count_nonzero(cov)); stats[2] += int(cov.sum(dtype=np.int64))


def fill(dst_ptr, src, m):          # P threads, each its slice (np.copyto releases the GIL)
    dst = np.ctypeslib.as_array(C.cast(dst_ptr, C.POINTER(C.c_uint8)), shape=(m,))
    step = -(-m // P)
    list(pool.map(lambda k: np.copyto(dst[k * step:min(m, (k + 1) * step)], src[k * step:min(m, (k + 1) * step)]), range(P)))


def run(mode, touch):
    t0 = time.perf_counter()
    stats = [0, 0, 0]
    L.check(lib.bgp_batch_begin(h, batch, 0, 1 << 22), h)
    off, taken = 0, C.c_size_t()
    while off < len(st):
        if mode == "push":
            m = min(piece, len(st) - off)
            rc = L.check(lib.bgp_batch_push(h, C.c_void_p(st.ctypes.data + off), m, C.byref(taken)), h)
            off += taken.value
        else:
            ptr, cap = C.c_void_p(), C.c_size_t()
            rc = L.check(lib.bgp_batch_acquire(h, C.byref(ptr), C.byref(cap)), h)
            if rc == L.BGP_OK:
                m = min(cap.value, len(st) - off)
                fill(ptr.value, st[off:off + m], m)
                L.check(lib.bgp_batch_commit(h, m), h)
                off += m
        if rc == L.BGP_EAGAIN:
            drain(stats, touch)
    L.check(lib.bgp_batch_end(h), h)
    drain(stats, touch)
    dt = time.perf_counter() - t0
    return dt, stats


def run_overlapped(touch):
    """acquire / commit with the fill running on the producer threads WHILE this thread hands out the windows of the batches
    already on the device: what an inflater pool writing into the pinned buffer does to a consumer."""
    t0 = time.perf_counter()
    stats = [0, 0, 0]
    L.check(lib.bgp_batch_begin(h, batch, 0, 1 << 22), h)
    off = 0
    step = None
    while off < len(st):
        ptr, cap = C.c_void_p(), C.c_size_t()
        rc = L.check(lib.bgp_batch_acquire(h, C.byref(ptr), C.byref(cap)), h)
        if rc == L.BGP_EAGAIN:
            drain(stats, touch)
            continue
        m = min(cap.value, len(st) - off)
        dst = np.ctypeslib.as_array(C.cast(ptr.value, C.POINTER(C.c_uint8)), shape=(m,))
        src = st[off:off + m]
        step = -(-m // P)
        futs = [pool.submit(np.copyto, dst[k * step:min(m, (k + 1) * step)], src[k * step:min(m, (k + 1) * step)]) for k in range(P)]
        drain(stats, touch)                      # the windows of what is on the device already, while the producers fill
        for f in futs:
            f.result()
        L.check(lib.bgp_batch_commit(h, m), h)
        off += m
    L.check(lib.bgp_batch_end(h), h)
    drain(stats, touch)
    return time.perf_counter() - t0, stats


def take_one(stats, touch):
    cw = L.Columns()
    rc = L.check(lib.bgp_batch_next_window(h, C.byref(cw)), h)
    if rc == L.BGP_OK:
        stats[0] += cw.n_positions
        if touch:
            cov = np.ctypeslib.as_array(C.cast(cw.columns, C.POINTER(C.c_uint32)), shape=(cw.n_positions, 8))[:, 0]
            stats[1] += int(np.count_nonzero(cov)); stats[2] += int(cov.sum(dtype=np.int64))
    return rc


def run_interleaved(touch):
    """acquire / commit and the hand-out INTERLEAVED, a window at a time: as soon as a window of batch i + 1 has come (batch i's device
    buffer is free from then on) the next buffer is acquired -- which ships the full one -- so that the copy to the device of batch i + 2
    runs beside the kernels and the copies back of batch i + 1.  (Draining every window first, as run() does, leaves the two directions
    of the link taking turns.)"""
    t0 = time.perf_counter()
    stats = [0, 0, 0]
    L.check(lib.bgp_batch_begin(h, batch, 0, 1 << 22), h)
    off, futs, m = 0, None, 0
    while off < len(st) or futs is not None:
        progressed = False
        if futs is None and off < len(st):
            ptr, cap = C.c_void_p(), C.c_size_t()
            rc = L.check(lib.bgp_batch_acquire(h, C.byref(ptr), C.byref(cap)), h)
            if rc == L.BGP_OK:
                m = min(cap.value, len(st) - off)
                dst = np.ctypeslib.as_array(C.cast(ptr.value, C.POINTER(C.c_uint8)), shape=(m,))
                src = st[off:off + m]
                step = -(-m // P)
                futs = [pool.submit(np.copyto, dst[k * step:min(m, (k + 1) * step)], src[k * step:min(m, (k + 1) * step)]) for k in range(P)]
                progressed = True
        if futs is not None and all(f.done() for f in futs):
            for f in futs:
                f.result()
            L.check(lib.bgp_batch_commit(h, m), h)
            off += m; futs = None
            continue
        if take_one(stats, touch) == L.BGP_OK:
            progressed = True
        if not progressed and futs is not None:
            for f in futs:
                f.result()
    L.check(lib.bgp_batch_end(h), h)
    drain(stats, touch)
    return time.perf_counter() - t0, stats


print(f"{n} reads of cfg3's model, {len(st) / 1e9:.2f} GB inflated, batches of {batch >> 20} MiB ({len(st) / batch:.1f}), producer threads {P}")
if os.environ.get("BGP_E2E_ONLY") == "interleaved":      # (for a trace: the in-turn loop alone)
    for rep in range(3):
        dt, stats = run_interleaved(False)
        print(f"in turn, windows only: {dt:.3f} s = {len(st) / dt / 1e9:.1f} GB/s of input", flush=True)
    pool.shutdown(wait=True)
    ctx.close()
    sys.exit(0)
last = None
for mode in ("push", "acquire"):
    for touch in (True, False):
        best = None
        for rep in range(3):
            dt, stats = run(mode, touch)
            best = dt if best is None or dt < best else best
            if touch:
                last = stats
        print(f"{mode:8s} consumer {'reads every column' if touch else 'takes the windows only'}: best of 3 {best:.3f} s = {n / best / 1e6:.1f} Mreads/s device-E2E, "
              f"{len(st) / best / 1e9:.1f} GB/s of input ({stats[0]} positions = {stats[0] * 32 / 1e9:.2f} GB of columns back)", flush=True)
for touch in (True, False):
    best = None
    for rep in range(3):
        dt, stats = run_overlapped(touch)
        best = dt if best is None or dt < best else best
    assert stats[0] == last[0] and (not touch or tuple(stats) == tuple(last)), "overlapped != serial"
    print(f"acquire, fill overlapped with the hand-out, consumer {'reads every column' if touch else 'takes the windows only'}: best of 3 {best:.3f} s = {n / best / 1e6:.1f} Mreads/s device-E2E, "
          f"{len(st) / best / 1e9:.1f} GB/s of input", flush=True)
for touch in (True, False):
    best = None
    for rep in range(3):
        dt, stats = run_interleaved(touch)
        best = dt if best is None or dt < best else best
    assert stats[0] == last[0] and (not touch or tuple(stats) == tuple(last)), "interleaved != serial"
    print(f"acquire, a window and a buffer in turn, consumer {'reads every column' if touch else 'takes the windows only'}: best of 3 {best:.3f} s = {n / best / 1e6:.1f} Mreads/s device-E2E, "
          f"{len(st) / best / 1e9:.1f} GB/s of input", flush=True)
ctx.reset(); ctx.push_inflated(st.tobytes()); ctx.flush()
info = ctx.pileup_run()
assert (info.n_positions, info.n_columns, info.sum_coverage) == tuple(last), "batched != one shot"
print("batched == one shot (positions, columns, sum of coverage)")

# The critical path of a batch, from HIP events on the streams the work runs on (bgp_timing_*): the copy to HBM, the carry in front
# of it, the kernels, the windows' copies back -- and what the host spends filling the pinned buffer (wall clock around the fill).
ctx.timing_enable(True); ctx.timing_reset()
t_fill = [0.0]
_fill = fill
def fill(dst_ptr, src, m):
    t = time.perf_counter(); _fill(dst_ptr, src, m); t_fill[0] += time.perf_counter() - t
dt, stats = run("acquire", False)
tm = ctx.timing_get()
ctx.timing_enable(False)
nb = max(1, tm.get("h2d_batch", {}).get("launches", 1))
grp = {"h2d_batch": ("h2d_batch",), "d2d_carry": ("d2d_carry",), "K0 framing": ("k0_",), "K1 record table": ("k1_",), "K3 keys": ("k3_",), "K4 columns": ("k4_",), "d2h windows": ("d2h_",)}
print(f"per batch ({nb} batches of {batch >> 20} MiB, acquire, consumer takes the windows only; wall {dt:.3f} s = {dt / nb * 1e3:.1f} ms a batch):")
print(f"   host fills the pinned buffer ({P} threads): {t_fill[0] / nb * 1e3:.1f} ms a batch = {len(st) / t_fill[0] / 1e9:.1f} GB/s")
for name, pre in grp.items():
    ms = sum(v["total_ms"] for k, v in tm.items() if k.startswith(pre))
    by = sum(v["algorithmic_bytes"] for k, v in tm.items() if k.startswith(pre))
    extra = f" = {by / (ms * 1e-3) / 1e9:.1f} GB/s" if name in ("h2d_batch", "d2d_carry", "d2h windows") and ms > 0 else ""
    print(f"   {name:18s} {ms / nb:7.2f} ms a batch{extra}")

# ... and where the calling thread's wall time goes (the same run, untimed kernels): inside bgp_batch_acquire (ships a full buffer, waits for
# the pinned buffer about to be refilled), the fill, bgp_batch_commit, bgp_batch_next_window
_orig = {k: getattr(lib, k) for k in ("bgp_batch_acquire", "bgp_batch_commit", "bgp_batch_next_window", "bgp_batch_end")}
wall = {k: 0.0 for k in _orig}
class _Timed:
    def __init__(self, name): self.name, self.f = name, _orig[name]
    def __call__(self, *a):
        t = time.perf_counter(); r = self.f(*a); wall[self.name] += time.perf_counter() - t; return r
class _Lib:
    def __getattr__(self, k): return _Timed(k) if k in _orig else getattr(L.lib(), k)
lib = _Lib()
t_fill[0] = 0.0
dt, stats = run("acquire", False)
print(f"calling thread, {dt:.3f} s: fill {t_fill[0]:.3f} s, " + ", ".join(f"{k[10:]} {v:.3f} s" for k, v in wall.items()))
