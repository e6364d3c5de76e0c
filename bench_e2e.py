"""bench.py's end-to-end legs (SURVEY 8(d) ii and iii), measured in the bench's own process AFTER its timed region and reported
beside the headline, never as `value`: the input crosses PCIe here (and, for the file, is inflated first).

  device_e2e   inflated records in host memory -> pinned staging -> HBM -> K0..K4 per batch -> the windows of columns handed
               out (bgp_batch_acquire / _commit / _next_window: what BioD's BgzfInputStream hands BamReadRange, inputstream.d:386-447,
               taken a batch at a time).  The consumer takes every window and reads nothing: the link and the kernels, not Python.
  file_e2e     a BGZF-compressed BAM file on disk -> HBM compressed -> k_inflate -> K0..K4 (bgp_load_bam_device + bgp_flush +
               bgp_pileup_run): block.d:127-216 and everything behind it on the device.

Raw ctypes calls: the Python facade's copies would be the slowest thing measured.
"""
import ctypes as C
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from biod_amd import _lib as L
from biod_amd import bamio


def device_e2e(ctx, st, n_reads, batch_bytes=256 << 20, producers=16, reps=3, compact=False):
    """compact: the windows cross the link as 8 bytes a position (bgp_set_compact_columns) -- the columns going back were the critical path."""
    lib, h = L.lib(), ctx.handle
    L.check(lib.bgp_set_compact_columns(h, int(compact)), h)
    pool = ThreadPoolExecutor(producers)

    def take_one(stats):
        cw = L.Columns()
        rc = L.check(lib.bgp_batch_next_window(h, C.byref(cw)), h)
        if rc == L.BGP_OK:
            stats[0] += cw.n_positions
        return rc

    def run():
        """a window and a buffer in turn: as soon as a window of batch i + 1 has come the next pinned buffer is acquired (which ships the
        full one), so the copy in of batch i + 2 runs beside the kernels and the copies back of batch i + 1"""
        t0 = time.perf_counter()
        stats = [0]
        L.check(lib.bgp_batch_begin(h, batch_bytes, 0, 1 << 22), h)
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
                    step = -(-m // producers)
                    futs = [pool.submit(np.copyto, dst[k * step:min(m, (k + 1) * step)], src[k * step:min(m, (k + 1) * step)]) for k in range(producers)]
                    progressed = True
            if futs is not None and all(f.done() for f in futs):
                for f in futs:
                    f.result()
                L.check(lib.bgp_batch_commit(h, m), h)
                off += m; futs = None
                continue
            if take_one(stats) == L.BGP_OK:
                progressed = True
            if not progressed and futs is not None:
                for f in futs:
                    f.result()
        L.check(lib.bgp_batch_end(h), h)
        while take_one(stats) == L.BGP_OK:
            pass
        return time.perf_counter() - t0, stats[0]

    ctx.reset()
    times, positions = [], 0
    for _ in range(reps):
        dt, positions = run()
        times.append(dt)
        ctx.reset()
    # the critical path of a batch from HIP events on the streams the work runs on (bgp_timing_*)
    ctx.timing_enable(True); ctx.timing_reset()
    dt_t, _ = run()
    tm = ctx.timing_get()
    ctx.timing_enable(False)
    ctx.reset()
    pool.shutdown(wait=True)
    L.check(lib.bgp_set_compact_columns(h, 0), h)
    nb = max(1, tm.get("h2d_batch", {}).get("launches", 1))
    per = lambda pre: round(sum(v["total_ms"] for k, v in tm.items() if k.startswith(pre)) / nb, 3)      # noqa: E731
    best = min(times)
    return {"what": "inflated records in host memory -> bgp_batch_acquire/commit -> HBM -> K0..K4 per batch -> windows of columns handed out (a window and a buffer in turn; the consumer takes the windows, reads nothing)",
            "reads": int(n_reads), "inflated_bytes": int(len(st)), "batch_bytes": int(batch_bytes), "producer_threads": producers,
            "Mreads_per_s": round(n_reads / best / 1e6, 1), "input_GBps": round(len(st) / best / 1e9, 1), "best_s": round(best, 4), "runs_s": [round(t, 4) for t in times],
            "columns_back_bytes": int(positions) * (8 if compact else 32), "columns_layout": "8 x u8 a position (bgp_set_compact_columns; every window of this stream fits)" if compact else "bgp_column, 32 bytes a position",
            "per_batch_ms_hip_events": {"batches": nb, "h2d_batch": per("h2d_batch"), "d2d_carry": per("d2d_carry"), "K0 framing": per("k0_"), "K1 record table": per("k1_"), "K3 keys": per("k3_"),
                                        "K4 columns": per("k4_"), "d2h windows": per("d2h"), "wall": round(dt_t / nb * 1e3, 3)}}


def device_e2e_pinned(ctx, st, n_reads, batch_bytes=256 << 20, reps=3, compact=True):
    """SURVEY 8(d) (ii) as it is worded -- "device E2E from pinned host inflated buffers incl. H2D / D2H": the inflated records lie in page-locked
    memory already (where an inflater that owns its buffers keeps them) and go to the device from there (bgp_batch_push_pinned), no staging
    copy inside the timed region; windows out in compact transport.  A buffer and a window in turn, as device_e2e."""
    import torch
    lib, h = L.lib(), ctx.handle
    pinned = torch.from_numpy(st).pin_memory()                       # (set-up: the inflater's own buffers)
    base = pinned.data_ptr()
    L.check(lib.bgp_set_compact_columns(h, int(compact)), h)

    def run():
        t0 = time.perf_counter()
        L.check(lib.bgp_batch_begin(h, batch_bytes, 0, 1 << 22), h)
        npos = 0
        cw = L.Columns()
        taken = C.c_size_t()
        L.check(lib.bgp_batch_push_pinned(h, C.c_void_p(base), len(st), C.byref(taken)), h)      # all of it: the library ships a piece whenever a device buffer falls free
        assert taken.value == len(st)
        L.check(lib.bgp_batch_end(h), h)
        while L.check(lib.bgp_batch_next_window(h, C.byref(cw)), h) == L.BGP_OK:
            npos += cw.n_positions
        return time.perf_counter() - t0, npos

    ctx.reset()
    times, positions = [], 0
    try:
        for _ in range(reps):
            dt, positions = run()
            times.append(dt)
            ctx.reset()
        ctx.timing_enable(True); ctx.timing_reset()
        dt_t, _ = run()
        tm = ctx.timing_get()
        ctx.timing_enable(False)
        ctx.reset()
    finally:
        L.check(lib.bgp_set_compact_columns(h, 0), h)
    nb = max(1, tm.get("h2d_batch", {}).get("launches", 1))
    per = lambda pre: round(sum(v["total_ms"] for k, v in tm.items() if k.startswith(pre)) / nb, 3)      # noqa: E731
    best = min(times)
    return {"what": "inflated records in PAGE-LOCKED host memory -> bgp_batch_push_pinned (to the device from where they lie) -> K0..K4 per batch -> windows of columns handed out, 8 bytes a position "
                    "(SURVEY 8(d) ii: 'from pinned host inflated buffers')",
            "reads": int(n_reads), "inflated_bytes": int(len(st)), "batch_bytes": int(batch_bytes), "positions": int(positions),
            "Mreads_per_s": round(n_reads / best / 1e6, 1), "input_GBps": round(len(st) / best / 1e9, 1), "best_s": round(best, 4), "runs_s": [round(t, 4) for t in times],
            "per_batch_ms_hip_events": {"batches": nb, "h2d_batch": per("h2d_batch"), "K0 framing": per("k0_"), "K1 record table": per("k1_"), "K3 keys": per("k3_"), "K4 columns": per("k4_"),
                                        "d2h windows": per("d2h"), "wall": round(dt_t / nb * 1e3, 3)}}


def write_bam(path, st, n_ref, threads=32, block=64000):
    """The stream as a BGZF file (writer.d:260: blocks of up to 64 KB, the EOF block at the end)."""
    b = memoryview(st)
    with open(path, "wb") as f, ThreadPoolExecutor(threads) as ex:
        f.write(bamio.bgzf_block(bamio.bam_header_bytes([(f"chr{i + 1}", 1 << 28) for i in range(max(1, n_ref))])))
        for blk in ex.map(lambda at: bamio.bgzf_block(bytes(b[at:at + block])), range(0, len(b), block)):
            f.write(blk)
        f.write(bamio.BGZF_EOF)
    return os.path.getsize(path)


def file_e2e(ctx, st, n_reads, n_ref, path="/tmp/bench_e2e.bam", reps=8):
    t0 = time.perf_counter()
    fsz = write_bam(path, st, n_ref)
    t_write = time.perf_counter() - t0
    rows = []
    info = None
    for rep in range(reps):
        ctx.reset()
        ctx.timing_enable(rep == reps - 1); ctx.timing_reset()
        t0 = time.perf_counter(); ctx.load_bam_device(path); t1 = time.perf_counter()
        ctx.flush(); t2 = time.perf_counter()
        info = ctx.pileup_run(); t3 = time.perf_counter()
        rows.append((t3 - t0, t1 - t0, t2 - t1, t3 - t2))
    ctx.timing_enable(False)
    ctx.reset()
    try:
        os.unlink(path)
    except OSError:
        pass
    warm = sorted(rows[1:]) if len(rows) > 1 else rows      # (a ctx's first load also makes its streams and pins its ring)
    best = warm[0]
    return {"what": "BGZF BAM file -> HBM compressed -> k_inflate -> K0 framing, record table, K3, K4 (bgp_load_bam_device + bgp_flush + bgp_pileup_run)",
            "reads": int(n_reads), "file_bytes": int(fsz), "inflated_bytes": int(len(st)), "columns": int(info.n_columns),
            "Mreads_per_s": round(n_reads / best[0] / 1e6, 1), "file_GBps": round(fsz / best[0] / 1e9, 2), "inflated_GBps": round(len(st) / best[0] / 1e9, 1),
            "best_s": {"total": round(best[0], 4), "load (read + upload + inflate)": round(best[1], 4), "flush (framing + table)": round(best[2], 4), "pileup": round(best[3], 4)},
            "first_load_s": round(rows[0][0], 4), "runs_s": [round(r[0], 4) for r in rows], "file_written_s": round(t_write, 1)}


def file_e2e_streamed(ctx, st, n_reads, n_ref, path="/tmp/bench_e2e_s.bam", batch_bytes=512 << 20, cap_bytes=8 << 30, reps=4):
    """The same file through BOUNDED device memory: bgp_stream_bam_begin (reader threads -> compressed ring in HBM -> k_inflate -> a batch's
    kernels) + bgp_batch_next_window until BGP_EOF, every window of columns handed to the host (compact transport), under a cap on the
    library's device memory of `cap_bytes` beyond what the process holds when the leg begins (the bench's own resident result).  One
    checked pass sums the windows' coverage on the host and holds it against the one-shot pass over the same file."""
    import ctypes as C
    import numpy as np
    lib, h = L.lib(), ctx.handle
    fsz = write_bam(path, st, n_ref)
    ctx.reset()
    ctx.load_bam_device(path); ctx.flush()
    info = ctx.pileup_run()
    one = (int(ctx.num_records()), int(info.n_positions), int(info.n_columns), int(info.sum_coverage))
    ctx.reset()
    cur, peak = C.c_uint64(), C.c_uint64()
    lib.bgp_device_memory(C.byref(cur), C.byref(peak), 1)
    base = cur.value
    lib.bgp_set_device_memory_cap(base + cap_bytes)
    rows, got, peak_b = [], None, 0
    try:
        L.check(lib.bgp_set_compact_columns(h, 1), h)
        for rep in range(reps + 1):
            check = rep == 0
            lib.bgp_device_memory(C.byref(cur), C.byref(peak), 1)
            t0 = time.perf_counter()
            L.check(lib.bgp_stream_bam_begin(h, path.encode(), batch_bytes, 0, 1 << 22), h)
            npos = ncol = scov = 0; t_first = None
            cw = L.Columns()
            while L.check(lib.bgp_batch_next_window(h, C.byref(cw)), h) == L.BGP_OK:
                if t_first is None:
                    t_first = time.perf_counter() - t0
                npos += cw.n_positions
                if check:
                    cov = (L.as_array(cw.columns8, cw.n_positions, L.COLUMN8_DTYPE, False) if cw.columns8 else L.as_array(cw.columns, cw.n_positions, L.COLUMN_DTYPE, False))["coverage"]
                    ncol += int(np.count_nonzero(cov)); scov += int(cov.sum(dtype=np.uint64))
            ob = C.c_uint64(); L.check(lib.bgp_batch_ordinal_base(h, C.byref(ob)), h)
            nrec = int(ob.value) + int(ctx.num_records())
            dt = time.perf_counter() - t0
            lib.bgp_device_memory(C.byref(cur), C.byref(peak), 0)
            peak_b = max(peak_b, peak.value - base)
            ctx.reset()
            if check:
                got = (nrec, npos, ncol, scov)
            else:
                rows.append((dt, t_first))
    finally:
        L.check(lib.bgp_set_compact_columns(h, 0), h)
        lib.bgp_set_device_memory_cap(0)
        try:
            os.unlink(path)
        except OSError:
            pass
    best = min(rows)
    return {"what": "BGZF BAM file -> reader threads -> ring of compressed chunks in HBM -> k_inflate into a batch's buffer -> K0 framing, record table, K3, K4 per batch -> every window of columns "
                    "handed to the host, 8 bytes a position (bgp_stream_bam_begin + bgp_batch_next_window): device memory bounded whatever the file's size",
            "reads": int(n_reads), "file_bytes": int(fsz), "inflated_bytes": int(len(st)), "batch_bytes": int(batch_bytes),
            "Mreads_per_s": round(n_reads / best[0] / 1e6, 1), "file_GBps": round(fsz / best[0] / 1e9, 2), "inflated_GBps": round(len(st) / best[0] / 1e9, 1),
            "best_s": round(best[0], 4), "first_window_after_s": round(best[1], 4), "runs_s": [round(r[0], 4) for r in rows],
            "device_memory_cap_bytes": int(cap_bytes), "peak_device_memory_bytes": int(peak_b), "within_cap": bool(peak_b <= cap_bytes),
            "equals_one_shot": {"equal": got == one, "records": got[0], "positions": got[1], "covered_columns": got[2], "sum_coverage": got[3]},
            "full_size": "the 100 M-read file (15.9 GB -> 29 GB of records): profiles/r06_file_e2e_100M_streamed.log"}
