"""Device-E2E through batched staging (NOT the headline: the input crosses PCIe): a synthetic record stream pushed from
host memory in 64 MiB pieces -> bgp_batch_push -> final columns handed out window by window (read in place, pinned).
Raw ctypes calls: the Python facade's copies would be the slowest thing measured."""
import ctypes as C
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch  # noqa: F401  (first: its HIP runtime)
from biod_amd import _lib as L
from biod_amd.pileup import Context
from biod_amd.synth import host_stream_np, scaled

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256 << 20
st = host_stream_np(scaled("cfg3", n))
ctx = Context(0)
lib, h = L.lib(), ctx.handle
piece = 64 << 20


def drain(stats):
    cw = L.Columns()
    while True:
        rc = L.check(lib.bgp_batch_next_window(h, C.byref(cw)), h)
        if rc != L.BGP_OK:
            return rc
        cov = np.ctypeslib.as_array(C.cast(cw.columns, C.POINTER(C.c_uint32)), shape=(cw.n_positions, 8))[:, 0]
        stats[0] += cw.n_positions; stats[1] += int(np.count_nonzero(cov)); stats[2] += int(cov.sum(dtype=np.int64))


for rep in range(3):
    t0 = time.perf_counter()
    stats = [0, 0, 0]
    L.check(lib.bgp_batch_begin(h, batch, 0, 1 << 22), h)
    off, taken = 0, C.c_size_t()
    while off < len(st):
        m = min(piece, len(st) - off)
        rc = L.check(lib.bgp_batch_push(h, C.c_void_p(st.ctypes.data + off), m, C.byref(taken)), h)
        off += taken.value
        if rc == L.BGP_EAGAIN:
            drain(stats)
    L.check(lib.bgp_batch_end(h), h)
    drain(stats)
    dt = time.perf_counter() - t0
    print(f"run {rep}: {n} reads, {len(st) / 1e9:.2f} GB inflated in {len(st) / batch:.1f} batches of {batch >> 20} MiB: {dt:.3f} s = {n / dt / 1e6:.1f} Mreads/s device-E2E "
          f"({len(st) / dt / 1e9:.1f} GB/s of input; {stats[0]} positions, {stats[1]} columns, sum coverage {stats[2]}; columns copied to the host: {stats[0] * 32 / 1e9:.2f} GB)")
ctx.reset(); ctx.push_inflated(st.tobytes()); ctx.flush()
info = ctx.pileup_run()
assert (info.n_positions, info.n_columns, info.sum_coverage) == tuple(stats), "batched != one shot"
print("batched == one shot (positions, columns, sum of coverage)")
