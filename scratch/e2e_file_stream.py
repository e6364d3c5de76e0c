"""The metric's own input as a FILE through BOUNDED device memory: cfg3's 100 M-read coordinate-sorted BAM (29 GB of records, ~16 GB
BGZF-compressed) written to local disk slice by slice, then (a) the one-shot load (bgp_load_bam_device: the whole file and the whole
inflated stream in HBM) and (b) bgp_stream_bam_begin + bgp_batch_next_window under a cap on the library's device memory
(BGP_HBM_CAP's worth, set through bgp_set_device_memory_cap): reader threads -> ring of compressed chunks -> k_inflate -> batched pass.
One CHECKED streamed pass sums every window's columns on the host (positions, covered columns, summed coverage, maximum) and holds
them against the one-shot pass's totals; the TIMED passes take every window and read nothing (the link and the kernels, not numpy).
Usage (GPU box): python scratch/e2e_file_stream.py [n_reads] [cap_GB] [path]"""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from concurrent.futures import ThreadPoolExecutor
import numpy as np
from biod_amd import _lib as L
from biod_amd import bamio
from biod_amd.pileup import Context
from biod_amd.synth import host_stream_np, scaled

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
cap_gb = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
path = sys.argv[3] if len(sys.argv) > 3 else "/tmp/cfg3_stream.bam"
cfg = scaled("cfg3", n)
os.system("df -h /tmp | tail -1; free -g | head -2")
t0 = time.time()
SL = 2_000_000
inflated = 0
with open(path, "wb") as f, ThreadPoolExecutor(32) as ex:
    f.write(bamio.bgzf_block(bamio.bam_header_bytes([(f"chr{i + 1}", 1 << 28) for i in range(max(1, cfg.n_ref))])))
    carry = b""
    for first in range(0, n, SL):
        st = carry + host_stream_np(cfg, first, min(SL, n - first)).tobytes()
        whole = len(st) - len(st) % 64000 if first + SL < n else len(st)
        b = memoryview(st)
        for blk in ex.map(lambda at: bamio.bgzf_block(bytes(b[at:min(at + 64000, whole)])), range(0, whole, 64000)):
            f.write(blk)
        inflated += whole
        carry = st[whole:]
    f.write(bamio.BGZF_EOF)
fsz = os.path.getsize(path)
print(f"{n} reads, {inflated / 1e9:.2f} GB of records, file {fsz / 1e9:.2f} GB, written in {time.time() - t0:.0f} s", flush=True)
lib = L.lib()


def devmem(reset=False):
    cur, peak = C.c_uint64(), C.c_uint64()
    lib.bgp_device_memory(C.byref(cur), C.byref(peak), int(reset))
    return cur.value, peak.value


ctx = Context(0)
one = None
for rep in range(2):
    ctx.reset()
    t0 = time.perf_counter(); ctx.load_bam_device(path); t1 = time.perf_counter()
    ctx.flush(); t2 = time.perf_counter()
    info = ctx.pileup_run(); t3 = time.perf_counter()
    one = (ctx.num_records(), info.n_positions, info.n_columns, info.sum_coverage, info.max_coverage)
    print(f"one-shot rep {rep}: load {t1 - t0:.3f} s ({inflated / (t1 - t0) / 1e9:.1f} GB/s inflated), framing + table {t2 - t1:.3f} s, pileup {t3 - t2:.3f} s -> {n / (t3 - t0) / 1e6:.1f} Mreads/s; "
          f"device memory {devmem()[0] / 2**30:.1f} GiB; records {one[0]}, positions {one[1]}, columns {one[2]}, sum of coverage {one[3]}, max {one[4]}", flush=True)
ctx.close()
assert devmem()[0] == 0

cap = int(cap_gb * (1 << 30))
lib.bgp_set_device_memory_cap(cap)
ctx = Context(0)
h = ctx.handle


def streamed(batch, window, check, compact=False):
    lib.bgp_set_compact_columns(h, int(compact))
    devmem(reset=True)
    t0 = time.perf_counter()
    L.check(lib.bgp_stream_bam_begin(h, path.encode(), batch, 0, window), h)
    npos = ncol = scov = mx = nwin = 0
    t_first = None
    cw = L.Columns()
    while True:
        rc = L.check(lib.bgp_batch_next_window(h, C.byref(cw)), h)
        if rc != L.BGP_OK:
            break
        if t_first is None:
            t_first = time.perf_counter() - t0
        nwin += 1; npos += cw.n_positions
        if check:
            cov = (L.as_array(cw.columns8, cw.n_positions, L.COLUMN8_DTYPE) if cw.columns8 else L.as_array(cw.columns, cw.n_positions, L.COLUMN_DTYPE))["coverage"]
            ncol += int(np.count_nonzero(cov)); scov += int(cov.sum(dtype=np.uint64)); mx = max(mx, int(cov.max()) if len(cov) else 0)
    base = C.c_uint64(); L.check(lib.bgp_batch_ordinal_base(h, C.byref(base)), h)
    nrec = base.value + ctx.num_records()
    dt = time.perf_counter() - t0
    ctx.reset()
    return dt, t_first, nwin, nrec, npos, ncol, scov, mx, devmem()[1]


dt, tf, nwin, nrec, npos, ncol, scov, mx, peak = streamed(0, 1 << 22, True, True)
assert (nrec, npos, ncol, scov, mx) == one, ((nrec, npos, ncol, scov, mx), one)
dt, tf, nwin, nrec, npos, ncol, scov, mx, peak = streamed(0, 1 << 22, True)
print(f"streamed, CHECKED pass (numpy sums every window; in both transports): {dt:.2f} s; records {nrec}, positions {npos}, columns {ncol}, sum of coverage {scov}, max {mx}; peak device memory {peak / 2**30:.2f} GiB (cap {cap_gb} GiB)", flush=True)
assert (nrec, npos, ncol, scov, mx) == one, ((nrec, npos, ncol, scov, mx), one)
print("streamed == one-shot: records, positions, covered columns, summed coverage, maximum")
for batch, compact in ((1 << 30, False), (256 << 20, True), (1 << 30, True), (3 << 29, True)):
    for rep in range(2):
        dt, tf, nwin, nrec, npos, *_, peak = streamed(batch, 1 << 22, False, compact)
        assert (nrec, npos) == one[:2]
        print(f"streamed, batch {batch >> 20} MiB, {'compact' if compact else '32-byte'} columns, rep {rep}: {dt:.3f} s -> {n / dt / 1e6:.1f} Mreads/s file-E2E ({fsz / dt / 1e9:.1f} GB/s of file, {inflated / dt / 1e9:.1f} GB/s inflated), first window after {tf * 1e3:.0f} ms, {nwin} windows, peak device memory {peak / 2**30:.2f} GiB of a {cap_gb} GiB cap", flush=True)
ctx.close()
os.unlink(path)
