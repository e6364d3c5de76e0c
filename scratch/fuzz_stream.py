"""The file-fed pass (bgp_stream_bam_begin) under random files and settings: BGZF blocks of random sizes (records straddle them, some blocks
tiny, some stored, an empty block here and there), random batch sizes from two blocks up, random windows, compact or 32-byte transport,
use_md_tag or not, entries or not -- columns (and reference bases, and entries' totals) == the stream staged at once on the same ctx.
Every few rounds the pass is abandoned half way and the ctx reused.  usage: python scratch/fuzz_stream.py [rounds] [seed0]"""
import ctypes as C, os, sys, time, zlib
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, R + "/tests"); sys.path.insert(0, R + "/oracle")
import numpy as np
import test_gpu_parity as T
from biod_amd import _lib as L, bamio
from biod_amd.pileup import Context
from biod_amd.synth import host_stream
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
ctx = Context(0); lib, h = L.lib(), ctx.handle
bad = ran = 0; t0 = time.time()
path = "/tmp/fuzz_stream.bam"
for k in range(n):
    rng = np.random.default_rng(seed0 + k)
    with open("/tmp/fuzz_stream.seed", "w") as sf: sf.write(str(seed0 + k))      # (a round that never returns is found by its seed)
    cfg = T.make_cfg(n_reads=int(rng.integers(300, 30000)), n_ref=int(rng.integers(1, 4)), ref_len=int(rng.integers(20_000, 200_000)), read_len=int(rng.integers(50, 151)), profile=int(rng.integers(0, 2)), seed=int(rng.integers(1, 1 << 30)))
    try:
        st = host_stream(cfg)
    except L.BiodGpuError:
        continue
    if rng.random() < 0.35:                                # a pile deeper than a byte holds, somewhere: that window crosses the link in the 32-byte layout
        import oracle_lib as O
        offs = [int(x) for x in O.frame(st)] + [len(st)]
        k0 = int(rng.integers(0, len(offs) - 1))
        rec = st[offs[k0]:offs[k0 + 1]]
        st = st[:offs[k0 + 1]] + rec * int(rng.integers(260, 700)) + st[offs[k0 + 1]:]
    payload = bamio.bam_header_bytes([(f"chr{i + 1}", 1 << 28) for i in range(cfg.n_ref)], text="@HD\tVN:1.0\n" * int(rng.integers(0, 3000))) + st
    mode = int(rng.integers(0, 3))
    with open(path, "wb") as f:
        at = 0
        while at < len(payload):
            m = int(rng.integers(1, 400)) if (mode == 0 and rng.random() < 0.1) else int(rng.integers(200, 65000))
            chunk = payload[at:at + m]; at += m
            if rng.random() < 0.03:
                f.write(bamio.BGZF_EOF)                    # (an empty block in the middle of the file: skipped, inputstream.d:447-450)
            if mode == 2 and rng.random() < 0.02:
                f.write(bamio.BGZF_EOF * int(rng.integers(1000, 40000)))      # (a stretch of container that holds nothing: with small chunks it fills the ring of compressed chunks before a batch's worth of bytes has been scanned)
            level = int(rng.choice([0, 1, 6, 9]))
            co = zlib.compressobj(level, zlib.DEFLATED, -15)
            cdata = co.compress(chunk) + co.flush()
            f.write(bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0]) + (len(cdata) + 25).to_bytes(2, "little") + cdata + (zlib.crc32(chunk) & 0xFFFFFFFF).to_bytes(4, "little") + len(chunk).to_bytes(4, "little"))
        if rng.random() < 0.8:
            f.write(bamio.BGZF_EOF)
    # the ring of compressed chunks at test size: chunks of 64 KiB .. 1 MiB and one to eight readers, so that a file of a few MB goes round it many times
    # (slots reused, batches cut because the ring is full); now and then the library's own 8 MiB
    if rng.random() < 0.8: os.environ["BGP_STREAM_CHUNK"] = str(int(rng.choice([65536, 100_000, 262144, 1 << 20])))
    else: os.environ.pop("BGP_STREAM_CHUNK", None)
    os.environ["BGP_READ_THREADS"] = str(int(rng.choice([1, 2, 3, 8])))
    use_md = bool(rng.random() < 0.4); entries = bool(rng.random() < 0.3); compact = bool(rng.random() < 0.5)
    batch = int(rng.choice([128 << 10, int(rng.integers(128 << 10, 1 << 20)), int(rng.integers(1 << 20, 8 << 20)), 0]))
    window = int(rng.choice([0, int(rng.integers(100, 5000)), 65536]))
    T.stage(ctx, st)
    info = ctx.pileup_run()
    want = T.device_columns(ctx, info)
    want_rb = None
    try:
        if use_md:
            want_rb = np.concatenate([w.reference_base for w in ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, use_md_tag=True, window_positions=1 << 20)] or [np.zeros(0, np.uint8)])
    except L.BiodGpuError:
        use_md = False
    want_ne = None
    if entries:
        want_ne = sum(int(w.entry_off[-1]) for w in ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, want_entries=True, window_positions=1 << 20))
    try:
        if rng.random() < 0.5 and len(want):               # makePileup clipped at a random end_at, windows of a random size: compact == wide (the last window is shorter than the one copied ahead)
            run0 = info.runs[0]
            lo_p, hi_p = int(run0["start_pos"]), int(run0["end_pos"])
            end_at = int(rng.integers(lo_p + 1, hi_p + 2)); start_from = int(rng.integers(lo_p, end_at))
            wsz = int(rng.integers(50, 3000))
            def clipped(comp):
                L.check(lib.bgp_set_compact_columns(h, int(comp)), h)
                L.check(lib.bgp_pileup_begin(h, L.BGP_MAKE_PILEUP, 0, start_from, end_at, int(rng.integers(0, 2)) if False else 1, 0, wsz), h)
                out = []; cw = L.Columns()
                while L.check(lib.bgp_pileup_next_window(h, C.byref(cw)), h) == L.BGP_OK:
                    out.append((int(cw.first_position), L.window_columns(cw)))
                return out
            a_, b_ = clipped(False), clipped(True)
            assert len(a_) == len(b_) and all(x[0] == y[0] and np.array_equal(x[1], y[1]) for x, y in zip(a_, b_)), "makePileup clipped: compact != wide"
        L.check(lib.bgp_set_compact_columns(h, int(compact)), h)
        if rng.random() < 0.15:                            # a pass abandoned after a window or two
            L.check(lib.bgp_stream_bam_begin(h, path.encode(), batch, 0, window), h)
            cw = L.Columns()
            for _ in range(int(rng.integers(0, 3))):
                if L.check(lib.bgp_batch_next_window(h, C.byref(cw)), h) != L.BGP_OK: break
            ctx.reset()
        L.check(lib.bgp_stream_bam_begin(h, path.encode(), batch, int(entries), window), h)
        if use_md: L.check(lib.bgp_batch_set_md(h, 1, 1), h)
        cols, rbs, ne = [], [], 0
        cw = L.Columns()
        while L.check(lib.bgp_batch_next_window(h, C.byref(cw)), h) == L.BGP_OK:
            cols.append(L.window_columns(cw))
            if use_md: rbs.append(L.as_array(cw.reference_base, cw.n_positions, "u1"))
            if entries: ne += int(L.as_array(cw.entry_off, cw.n_positions + 1, "<u8")[-1])
        got = np.concatenate(cols) if cols else np.zeros(0, L.COLUMN_DTYPE)
        assert len(got) == len(want), (len(got), len(want))
        for fld in L.COLUMN_DTYPE.names:
            assert np.array_equal(got[fld], want[fld]), fld
        if use_md:
            rb = np.concatenate(rbs) if rbs else np.zeros(0, np.uint8)
            m = want["coverage"] > 0
            assert np.array_equal(rb[m], want_rb[m]), "reference_base"
        if entries:
            assert ne == want_ne, (ne, want_ne)
        ran += 1
    except Exception as e:
        bad += 1; print("STREAM seed", seed0 + k, "batch", batch, "window", window, "md", use_md, "entries", entries, "compact", compact, "len", len(st), type(e).__name__, str(e)[:300])
    finally:
        lib.bgp_set_compact_columns(h, 0)
        ctx.reset()
print(n, "rounds,", ran, "ran,", bad, "failures,", round(time.time() - t0, 1), "s")
