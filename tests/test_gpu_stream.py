"""A BAM FILE streamed through the device in bounded memory (bgp_stream_bam_begin, biod_amd/csrc/stream.cpp): reader threads ->
ring of compressed chunks in HBM -> k_inflate into a batch's device buffer -> the batched pass of batch.cpp -> windows.  The shape
of BgzfInputStream's ring of blocks in flight under a pulling consumer (bio/core/bgzf/inputstream.d:386-447,455-483; block.d:127-216).
Columns, per-read entries and reference bases must be those of the one-shot load and of the oracle, whatever the batch size; the
library's device memory must not grow with the file (asserted under a cap smaller than the file's inflated size)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
from biod_amd import _lib as L
from biod_amd import bamio
from biod_amd.pileup import Context
from biod_amd.synth import host_stream, host_stream_np, make_cfg, scaled

from test_gpu_parity import FIXTURES, _assert_columns_equal_oracle, _covered, _write_bgzf, device_columns

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def device_memory(reset_peak=False):
    cur, peak = C.c_uint64(), C.c_uint64()
    L.lib().bgp_device_memory(C.byref(cur), C.byref(peak), int(reset_peak))
    return int(cur.value), int(peak.value)


def write_bam_fast(path, refs, st, block=60000):
    from concurrent.futures import ThreadPoolExecutor
    b = memoryview(st)
    with open(path, "wb") as f, ThreadPoolExecutor(8) as ex:
        f.write(bamio.bgzf_block(bamio.bam_header_bytes(refs)))
        for blk in ex.map(lambda at: bamio.bgzf_block(bytes(b[at:at + block])), range(0, len(b), block)):
            f.write(blk)
        f.write(bamio.BGZF_EOF)


@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("batch", [128 << 10, 0])
def test_stream_fixtures(ctx, data_dir, name, batch):
    """Every fixture of the reference's test/data the other loaders are held to: columns == oracle, the header's references known
    after the begin call; the smallest batches (two BGZF blocks) and the default."""
    path = os.path.join(data_dir, name)
    hdr, st = bamio.read_bam(path)
    ref, pos, cols, _, npos = _covered(ctx.stream_bam_windows(path, batch))
    assert ctx.references() == [(n, l) for n, l in hdr.refs]
    ora = O.pileup(st)
    if name != "bins.bam":
        _assert_columns_equal_oracle(ref, pos, cols, ora)
    else:                                               # empty SEQ: coverage only (test_bins_bam_coverage_parity)
        assert np.array_equal(pos, ora.pos.astype(np.int64)) and np.array_equal(cols["coverage"], ora.cov) and np.array_equal(cols["n_starting_here"], ora.nstart)
    ctx.reset()


@pytest.mark.parametrize("name", ["corrupted_zlib_archive.bam", "duplicated_block_size.bam", "no_block_size.bam", "wrong_bc_subfield_length.bam", "wrong_extra_gzip_length.bam"])
def test_stream_negative_fixtures(ctx, data_dir, name):
    """test/unittests.d:132-142: every malformed container is refused on this path too (at the begin call when the damage is in the
    header's blocks, else at the window whose batch holds the block)."""
    with pytest.raises(L.BiodGpuError):
        for _ in ctx.stream_bam_windows(os.path.join(data_dir, name), 128 << 10):
            pass
    ctx.reset()


@pytest.mark.parametrize("batch", [128 << 10, 300_000, 1 << 20, 0])
def test_stream_small_batches_entries_and_reference_bases(ctx, tmp_path, batch):
    """Blocks of random sizes (records straddle them), batches of two to a few dozen blocks, N-skip reads reaching across batches:
    columns, per-read entries (stream-wide ordinals) and use_md_tag's reference bases == the oracle's."""
    cfg = make_cfg(n_reads=12_000, n_ref=3, ref_len=30_000, read_len=100, profile=1, seed=batch % 97 + 5)
    st = host_stream(cfg)
    path = str(tmp_path / "s.bam")
    _write_bgzf(path, bamio.bam_header_bytes([(f"chr{i + 1}", 1 << 28) for i in range(cfg.n_ref)]) + st, np.random.default_rng(batch % 89))
    ora = O.pileup(st, want_entries=True)
    ref, pos, cols, ents, _ = _covered(ctx.stream_bam_windows(path, batch, want_entries=True, window_positions=3000), want_entries=True)
    _assert_columns_equal_oracle(ref, pos, cols, ora)
    assert len(ents) == ora.n_columns
    for k in range(0, ora.n_columns, 11):
        oe = ora.entries[int(ora.entry_off[k]):int(ora.entry_off[k + 1])]
        assert np.array_equal(ents[k]["read_ordinal"], oe["read_ordinal"].astype(np.uint32)), k
        assert np.array_equal(ents[k]["query_offset"], oe["query_offset"]) and np.array_equal(ents[k]["base"], oe["base"]) and np.array_equal(ents[k]["qual"], oe["qual"])
    ora_md = O.pileup(st, use_md=True)
    got = np.concatenate([w.reference_base[w.columns["coverage"] > 0] for w in ctx.stream_bam_windows(path, batch, use_md_tag=True, window_positions=5000)])
    assert np.array_equal(got, ora_md.refbase)
    ctx.reset()


def test_stream_state_rules_and_reuse(ctx, data_dir, tmp_path):
    """Nobody pushes into a pass a file feeds; a pass abandoned half way (reset) leaves a ctx that loads and streams again; an empty
    BAM (header only) has no windows; a missing file and a file that is no BGZF are refused at the begin call."""
    path = os.path.join(data_dir, "ex1_header.bam")
    L.check(L.lib().bgp_stream_bam_begin(ctx._h, path.encode(), 128 << 10, 0, 0), ctx._h)
    for call in (lambda: ctx.batch_push(b"abcd"), ctx.batch_acquire, ctx.batch_end):
        with pytest.raises(L.BiodGpuError) as e:
            call()
        assert e.value.code == L.BGP_ESTATE
    first = next(ctx.batch_drain(1))
    assert len(first.columns) > 0
    ctx.reset()                                            # readers stopped, rings given back
    _, st = bamio.read_bam(path)
    ctx.push_inflated(st); ctx.flush()
    info = ctx.pileup_run()
    ref, pos, cols, _, npos = _covered(ctx.stream_bam_windows(path, 200_000))
    assert npos == info.n_positions and len(pos) == info.n_columns
    empty = str(tmp_path / "empty.bam")
    bamio.write_bam(empty, [("chr1", 1000)], [])
    assert list(ctx.stream_bam_windows(empty)) == []
    assert ctx.references() == [("chr1", 1000)]
    for bad in (str(tmp_path / "nope.bam"), __file__):
        with pytest.raises(L.BiodGpuError):
            list(ctx.stream_bam_windows(bad))
    ctx.reset()


def test_stream_damage_is_refused_like_the_one_shot_load(ctx, tmp_path):
    """One flipped bit anywhere in a block's CDATA, CRC or ISIZE: refused exactly when the host loader refuses it."""
    cfg = scaled("cfg2", 60_000)
    st = host_stream(cfg)
    path = str(tmp_path / "s.bam")
    rng = np.random.default_rng(11)
    _write_bgzf(path, bamio.bam_header_bytes([(f"chr{i + 1}", 1 << 28) for i in range(max(1, cfg.n_ref))]) + st, rng)
    raw = bytearray(open(path, "rb").read())
    for trial in range(10):
        bad = bytearray(raw)
        at = int(rng.integers(200, len(raw) - 40))
        bad[at] ^= 1 << int(rng.integers(0, 8))
        p2 = str(tmp_path / f"bad{trial}.bam")
        open(p2, "wb").write(bytes(bad))
        host_ok = True
        try:
            h = Context(0); h.load_bam(p2, 2); h.close()
        except L.BiodGpuError:
            host_ok = False
        dev_ok = True
        try:
            for _ in ctx.stream_bam_windows(p2, 1 << 20):
                pass
        except L.BiodGpuError:
            dev_ok = False
        ctx.reset()
        assert dev_ok == host_ok, (trial, at)


def test_stream_file_larger_than_the_memory_cap(tmp_path):
    """2.4 M reads of cfg3's model (0.7 GB inflated, a file of 13 read chunks) streamed in 32 MiB batches under a cap of 400 MB on the
    library's device memory -- less than the inflated stream alone: the one-shot device load is refused by that cap, the streamed
    pass runs, its peak is far below the cap's worth of a whole-file load, and columns, coverage histogram and read count equal the
    one-shot result (without the cap)."""
    cfg = scaled("cfg3", 2_400_000)
    st = host_stream_np(cfg)
    path = str(tmp_path / "big.bam")
    write_bam_fast(path, [(f"chr{i + 1}", 1 << 28) for i in range(max(1, cfg.n_ref))], st)
    assert len(st) > 600 << 20 and os.path.getsize(path) > 5 * (8 << 20)
    one = Context(0)
    one.load_bam_device(path); one.flush()
    info = one.pileup_run()
    dense = device_columns(one, info)
    nz = np.nonzero(dense["coverage"])[0]
    n_reads = one.num_records()
    one.close()
    base = device_memory()[0]                              # (what other contexts of the process hold: the accounting is process-wide)
    cap = base + (400 << 20)
    L.lib().bgp_set_device_memory_cap(cap)
    try:
        c = Context(0)
        with pytest.raises(L.BiodGpuError) as e:
            c.load_bam_device(path)
        assert e.value.code == L.BGP_ENOMEM
        c.close()                                          # (bgp_reset keeps allocations for reuse -- the compressed file's 0.4 GB among them)
        assert device_memory()[0] == base
        c = Context(0)
        device_memory(reset_peak=True)
        pos, cols, npos, n_rec = [], [], 0, 0
        L.check(L.lib().bgp_stream_bam_begin(c._h, path.encode(), 32 << 20, 0, 1 << 20), c._h)
        last_base = -1
        for w in c.batch_drain():
            k = np.nonzero(w.columns["coverage"])[0]
            pos.append(w.first_position + k.astype(np.int64)); cols.append(w.columns[k]); npos += len(w.columns)
            ob = c.batch_ordinal_base()
            if ob != last_base:
                last_base = ob; n_rec = ob + c.num_records()
        peak = device_memory()[1]
        assert peak <= cap
        print(f"streamed pass: peak device memory {(peak - base) / 2**20:.0f} MiB for a stream of {len(st) / 2**20:.0f} MiB")
        c.close()
        assert device_memory()[0] == base                  # (rings, batch buffers, tables: all given back)
    finally:
        L.lib().bgp_set_device_memory_cap(0)
    pos = np.concatenate(pos); cols = np.concatenate(cols)
    assert n_rec == n_reads == cfg.n_reads
    assert npos == info.n_positions and len(pos) == info.n_columns == len(nz)
    for f in L.COLUMN_DTYPE.names:
        assert np.array_equal(cols[f], dense[f][nz]), f


# ---- compact transport of the windows (bgp_set_compact_columns): a byte a count over the link, the same numbers for the consumer ----

def _raw_windows(ctx, begin):
    """(columns as COLUMN_DTYPE, came compact?) of every window of an iteration begun by `begin`"""
    begin()
    out = []
    while True:
        cw = L.Columns()
        rc = L.check(L.lib().bgp_pileup_next_window(ctx._h, C.byref(cw)), ctx._h)
        if rc == L.BGP_EOF:
            return out
        out.append((cw.ref_id, cw.first_position, L.window_columns(cw), bool(cw.columns8) and not cw.columns))


@pytest.mark.parametrize("name", ["ex1_header.bam", "b7_295_chunk.bam", "mg1655_chunk.bam"])
@pytest.mark.parametrize("window", [700, 0])
def test_compact_columns_same_numbers(ctx, data_dir, name, window):
    """Every window of pileupColumns and of makePileup with skip_zero_coverage = false (uncovered stretches are windows of zeros) in both
    transports: the same positions and counts; the compact one really is columns8 wherever no position is deeper than 255 (windows are
    copied ahead on a second stream: both buffers take part)."""
    _, st = bamio.read_bam(os.path.join(data_dir, name))
    ctx.reset(); ctx.push_inflated(st); ctx.flush()
    for begin in (lambda: L.check(L.lib().bgp_pileup_begin(ctx._h, L.BGP_PILEUP_COLUMNS, 0, 0, 2**64 - 1, 1, 0, window), ctx._h),
                  lambda: L.check(L.lib().bgp_pileup_begin(ctx._h, L.BGP_MAKE_PILEUP, 0, 0, 2**64 - 1, 0, 0, window), ctx._h)):
        ctx.set_compact_columns(False)
        wide = _raw_windows(ctx, begin)
        ctx.set_compact_columns(True)
        comp = _raw_windows(ctx, begin)
        ctx.set_compact_columns(False)
        assert len(wide) == len(comp) > 0 and not any(w[3] for w in wide)
        for a, b in zip(wide, comp):
            assert a[:2] == b[:2] and np.array_equal(a[2], b[2])
            assert b[3] == (int(a[2]["coverage"].max(initial=0)) <= 255)


def test_compact_columns_deep_window_comes_wide(ctx):
    """A pile of 300 reads on one position among ordinary windows: the window that holds it crosses the link in the 32-byte layout, its
    neighbours compact; entries and reference bases ride along unchanged; the batched and the file-fed pass take the setting too."""
    recs = [bamio.pack_record(f"a{k}", "ACGTACGTAC", [(10, "M")], ref_id=0, pos=100 + k) for k in range(50)]
    recs += [bamio.pack_record(f"d{k}", "ACGTACGTAC", [(10, "M")], ref_id=0, pos=5000, tags=bamio.tag_z("MD", "10")) for k in range(300)]
    recs += [bamio.pack_record(f"b{k}", "ACGTACGTAC", [(10, "M")], ref_id=0, pos=9000 + k) for k in range(50)]
    st = b"".join(recs)
    ora = O.pileup(st, want_entries=True)
    ctx.reset(); ctx.push_inflated(st); ctx.flush()
    ctx.set_compact_columns(True)
    try:
        ws = _raw_windows(ctx, lambda: L.check(L.lib().bgp_pileup_begin(ctx._h, L.BGP_PILEUP_COLUMNS, 0, 0, 2**64 - 1, 1, 1, 1000), ctx._h))
        kinds = [w[3] for w in ws]
        assert False in kinds and True in kinds            # the deep window wide, others compact
        pos = np.concatenate([w[1] + np.nonzero(w[2]["coverage"])[0] for w in ws]); cols = np.concatenate([w[2][w[2]["coverage"] > 0] for w in ws])
        _assert_columns_equal_oracle(np.zeros(len(pos), np.int64), pos.astype(np.int64), cols, ora)
        # through the facade (widened), with entries: the same as the oracle's
        ref, pos2, cols2, ents, _ = _covered(ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, want_entries=True, window_positions=1000), want_entries=True)
        _assert_columns_equal_oracle(ref, pos2, cols2, ora)
        assert sum(len(e) for e in ents) == int(ora.entry_off[-1])
        # batched staging under the same setting
        ref, pos3, cols3, _, _ = _covered(ctx.batched_windows([st[o:o + 3001] for o in range(0, len(st), 3001)], 8192, window_positions=1000))
        _assert_columns_equal_oracle(ref, pos3, cols3, ora)
    finally:
        ctx.set_compact_columns(False)
        ctx.reset()


def test_compact_columns_window_shorter_than_the_one_copied_ahead(ctx):
    """makePileup(..., end_at) ends inside the window that was copied ahead: the window handed out is shorter than what was packed, and
    the flag that says "a count beyond 255" lies behind the PACKED positions.  A pile of 300 reads inside the short window, nothing at the
    position where a flag would have been looked for: the window must come wide (or compact with the right counts), never wrapped."""
    recs = [bamio.pack_record(f"a{k}", "ACGTACGTAC", [(10, "M")], ref_id=0, pos=10 * k) for k in range(400)]            # coverage 1 over [0, 4000)
    recs += [bamio.pack_record(f"d{k}", "ACGTACGTAC", [(10, "M")], ref_id=0, pos=5000) for k in range(300)]
    recs += [bamio.pack_record(f"b{k}", "ACGTACGTAC", [(10, "M")], ref_id=0, pos=6000 + 10 * k) for k in range(300)]
    st = b"".join(recs)
    ora = O.pileup(st, mode=O.MODE_MAKE_PILEUP, start_from=0, end_at=5500, skip_zero=True)
    ctx.reset(); ctx.push_inflated(st); ctx.flush()
    ctx.set_compact_columns(True)
    try:
        ws = _raw_windows(ctx, lambda: L.check(L.lib().bgp_pileup_begin(ctx._h, L.BGP_MAKE_PILEUP, 0, 0, 5500, 1, 0, 1000), ctx._h))
        assert [len(w[2]) for w in ws][-1] == 500
        pos = np.concatenate([w[1] + np.nonzero(w[2]["coverage"])[0] for w in ws]); cols = np.concatenate([w[2][w[2]["coverage"] > 0] for w in ws])
        _assert_columns_equal_oracle(np.zeros(len(pos), np.int64), pos.astype(np.int64), cols, ora)
        assert int(cols["coverage"].max()) == 300
    finally:
        ctx.set_compact_columns(False)
        ctx.reset()


def test_compact_columns_streamed_file(ctx, tmp_path):
    """The file-fed pass with compact transport: 150 k reads of cfg3's model, windows of 64 k positions (dozens of them, copied ahead),
    columns == the wide transport's."""
    cfg = scaled("cfg3", 150_000)
    st = host_stream(cfg)
    path = str(tmp_path / "c.bam")
    write_bam_fast(path, [(f"chr{i + 1}", 1 << 28) for i in range(max(1, cfg.n_ref))], st)
    wide = _covered(ctx.stream_bam_windows(path, 8 << 20, window_positions=65536))
    ctx.set_compact_columns(True)
    try:
        n8 = 0
        L.check(L.lib().bgp_stream_bam_begin(ctx._h, path.encode(), 8 << 20, 0, 65536), ctx._h)
        pos, cols = [], []
        while True:
            cw = L.Columns()
            if L.check(L.lib().bgp_batch_next_window(ctx._h, C.byref(cw)), ctx._h) != L.BGP_OK:
                break
            n8 += bool(cw.columns8)
            w = L.window_columns(cw); k = np.nonzero(w["coverage"])[0]
            pos.append(cw.first_position + k.astype(np.int64)); cols.append(w[k])
        assert n8 > 10
        assert np.array_equal(np.concatenate(pos), wide[1]) and np.array_equal(np.concatenate(cols), wide[2])
    finally:
        ctx.set_compact_columns(False)
        ctx.reset()


# ---- bam[ref][beg .. end] with the chunks' blocks inflated on the device (bgp_load_bam_region_device) ----

@pytest.mark.parametrize("name", ["ex1_header.bam", "tags.bam", "bins.bam"])
def test_region_load_device_matches_host_loader_and_oracle(ctx, data_dir, name):
    """getChunks + StreamChunksSupplier + BamReadFilter (randomaccessmanager.d:209-244,366-462; inputstream.d:254-330) with
    decompressBgzfBlock on the device: the staged bytes, the records kept and their virtual offsets are the host loader's, the
    records the oracle's restatement of the index-driven load; the argument checks of reference.d:76-78 hold here too."""
    import bai_oracle as B
    from biod_amd.pileup import BaiFile
    from test_gpu_parity import _check_subset_decode, _expected_region
    path = os.path.join(data_dir, name)
    hdr, st = bamio.read_bam(path)
    bai = BaiFile(path)
    obai = B.parse_bai(path)
    rng = np.random.default_rng(29)
    host = Context(0)
    for rid, (rname, rlen) in enumerate(hdr.refs):
        for beg, end in [(0, max(rlen, 1)), (0, 1), (10, 11)] + [(int(a), int(a) + int(b)) for a, b in zip(rng.integers(0, max(rlen, 2), 5), rng.integers(1, 3000, 5))]:
            cst, _ = B.region_reads(path, obai, rid, beg, end)
            _, exp_index = _expected_region(cst, rid, beg, end)
            ctx.reset(); ctx.load_bam_region_device(path, bai, rid, beg, end)
            host.reset(); host.load_bam_region(path, bai, rid, beg, end, 2)
            ctx.flush(); host.flush()
            assert ctx.staged_bytes().tobytes() == host.staged_bytes().tobytes() == cst
            _check_subset_decode(ctx.decode_fetch(), exp_index)
            n = ctx.num_records()
            assert n == host.num_records()
            for i in sorted({0, n // 2, n - 1} if n else set()):
                assert ctx.record_voffset(i) == host.record_voffset(i)
            if n:
                a, b = ctx.pileup_run(), host.pileup_run()
                assert (a.n_columns, a.sum_coverage, a.max_coverage) == (b.n_columns, b.sum_coverage, b.max_coverage)
    host.close()
    with pytest.raises(L.BiodGpuError):
        ctx.reset(); ctx.load_bam_region_device(path, bai, 0, 5, 5)                # enforce(start < end), reference.d:76
    with pytest.raises(L.BiodGpuError):
        ctx.reset(); ctx.load_bam_region_device(path, bai, len(hdr.refs), 0, 5)    # checkRefId, randomaccessmanager.d:205-207
    ctx.reset()


def test_region_load_device_large_file_and_file_shards(ctx, tmp_path):
    """A 300 k-read file of three contigs indexed on the device (createIndex), then cut into file shards by its .bai (shard.py): every
    shard's region loaded both ways gives the same bytes; regions that begin and end inside BGZF blocks (cut blocks at both ends)."""
    from biod_amd.pileup import BaiFile
    cfg = make_cfg(n_reads=300_000, n_ref=3, ref_len=400_000, read_len=150, profile=1, seed=77)
    st = host_stream(cfg)
    path = str(tmp_path / "r.bam")
    write_bam_fast(path, [(f"chr{i + 1}", 1 << 28) for i in range(cfg.n_ref)], st)
    ctx.reset(); ctx.load_bam(path, 4); ctx.flush()
    open(path + ".bai", "wb").write(ctx.bai_build())
    bai = BaiFile(path)
    host = Context(0)
    for rid, beg, end in [(0, 0, 400_000), (1, 123_456, 234_567), (2, 399_000, 400_000), (1, 50_000, 50_001), (2, 0, 1)]:
        ctx.reset(); ctx.load_bam_region_device(path, bai, rid, beg, end); ctx.flush()
        host.reset(); host.load_bam_region(path, bai, rid, beg, end, 4); host.flush()
        assert ctx.num_records() == host.num_records() and (ctx.num_records() > 0 or end - beg < 2000)
        assert ctx.staged_bytes().tobytes() == host.staged_bytes().tobytes()
        if not ctx.num_records():
            continue
        a, b = ctx.pileup_run(), host.pileup_run()
        assert (a.n_columns, a.sum_coverage, a.max_coverage) == (b.n_columns, b.sum_coverage, b.max_coverage)
    host.close(); ctx.reset()


def test_stream_ctx_destroyed_with_an_open_pass(data_dir, tmp_path):
    """bgp_destroy with a file-fed pass open (an exception in the caller, a facade's destructor): the reader threads are stopped, the
    inflate and upload streams drained and only then destroyed, everything given back -- and the process goes on to stream the file again."""
    cfg = scaled("cfg3", 300_000)
    st = host_stream(cfg)
    path = str(tmp_path / "d.bam")
    write_bam_fast(path, [(f"chr{i + 1}", 1 << 28) for i in range(max(1, cfg.n_ref))], st)
    base = device_memory()[0]
    for take in (0, 1, 3):
        c = Context(0)
        L.check(L.lib().bgp_stream_bam_begin(c._h, path.encode(), 4 << 20, 0, 65536), c._h)
        for _ in c.batch_drain(take) if take else ():
            pass
        c.close()
        assert device_memory()[0] == base
    c = Context(0)
    n = sum(len(w.columns) for w in c.stream_bam_windows(path, 4 << 20, window_positions=65536))
    c.reset(); c.load_bam_device(path); c.flush()
    assert n == c.pileup_run().n_positions
    c.close()


# ---- bgp_batch_push_pinned: bytes that already lie in page-locked host memory go to the device from where they are ----

def _pinned_copy(data: bytes):
    hip = C.CDLL("libamdhip64.so")
    p = C.c_void_p()
    assert hip.hipHostMalloc(C.byref(p), C.c_size_t(max(len(data), 1)), C.c_uint(0)) == 0
    C.memmove(p.value, data, len(data))
    return hip, p


@pytest.mark.parametrize("name,batch", [("ex1_header.bam", 150_000), ("b7_295_chunk.bam", 40_000)])
def test_batch_push_pinned_equals_push(ctx, data_dir, name, batch):
    """SURVEY 8(d) (ii), "from pinned host inflated buffers": pushes from page-locked memory (no staging copy) in pieces of any size, mixed
    with ordinary pushes -- the oracle's columns; bytes that are not page-locked are refused."""
    _, st = bamio.read_bam(os.path.join(data_dir, name))
    ora = O.pileup(st)
    hip, p = _pinned_copy(st)
    try:
        lib, h = L.lib(), ctx._h
        for piece in (len(st), 3 * batch + 17, 9_999):
            L.check(lib.bgp_batch_begin(h, batch, 0, 500), h)
            wins, off, k = [], 0, 0
            while off < len(st):
                n = min(piece, len(st) - off)
                taken = C.c_size_t()
                if k % 3 == 2:                               # now and then the ordinary way: what was pushed first is piled up first
                    buf = (C.c_char * n).from_buffer_copy(st[off:off + n])
                    rc = L.check(lib.bgp_batch_push(h, buf, n, C.byref(taken)), h)
                else:
                    rc = L.check(lib.bgp_batch_push_pinned(h, C.c_void_p(p.value + off), n, C.byref(taken)), h)
                off += taken.value; k += 1
                if rc == L.BGP_EAGAIN:                       # (an ordinary push behind page-locked bytes that still wait: windows first)
                    wins += list(ctx.batch_drain(1))
                    assert wins
            L.check(lib.bgp_batch_end(h), h)
            wins += list(ctx.batch_drain())
            ref, pos, cols, _, _ = _covered(wins)
            _assert_columns_equal_oracle(ref, pos, cols, ora)
            ctx.reset()
        L.check(lib.bgp_batch_begin(h, batch, 0, 500), h)
        pageable = (C.c_char * 1000).from_buffer_copy(st[:1000])
        taken = C.c_size_t()
        with pytest.raises(L.BiodGpuError) as e:
            L.check(lib.bgp_batch_push_pinned(h, pageable, 1000, C.byref(taken)), h)
        assert e.value.code == L.BGP_EINVAL
        ctx.reset()
    finally:
        hip.hipHostFree(p)


@pytest.mark.timeout(300, method="thread")           # (a regression here is a wait inside the library: only the thread method ends it)
def test_stream_file_of_tiny_and_empty_blocks_longer_than_the_ring(ctx, tmp_path, monkeypatch):
    """The ring of compressed chunks is sized for blocks no larger than what they inflate to.  A file that is mostly container -- runs of
    empty BGZF blocks (28 bytes for nothing: what `cat a.bam b.bam` leaves in the middle, a million times over) and blocks of a few
    records -- fills it long before a batch's worth of bytes has been scanned; the readers then wait for slots only a batch gives back.
    The pass must cut its batch at what the ring holds (and give back the slots of chunks that held nothing to inflate): columns == the
    oracle's, in time."""
    cfg = make_cfg(n_reads=3000, n_ref=2, ref_len=40_000, read_len=100, profile=0, seed=77)
    st = host_stream(cfg)
    offs, o = [], 0
    while o < len(st):
        offs.append(o); o += 4 + int.from_bytes(st[o:o + 4], "little")
    offs.append(len(st))
    path = str(tmp_path / "tiny_blocks.bam")
    empties = bamio.BGZF_EOF * 800_000                          # 22 MB of blocks that hold nothing
    with open(path, "wb") as f:
        f.write(bamio.bgzf_block(bamio.bam_header_bytes([("chr1", 40_000), ("chr2", 40_000)])))
        n = len(offs) - 1
        for part, (a, b) in enumerate(((0, n // 3), (n // 3, 2 * n // 3), (2 * n // 3, n))):
            for i in range(a, b, 3):                            # three records a block, cut mid-record now and then
                lo, hi = offs[i], offs[min(i + 3, b)]
                cut = lo + (hi - lo) // 2 if i % 5 == 0 else hi
                f.write(bamio.bgzf_block(st[lo:cut]))
                if cut < hi: f.write(bamio.bgzf_block(st[cut:hi]))
            if part < 2: f.write(empties)
        f.write(bamio.BGZF_EOF)
    assert os.path.getsize(path) > 5 * (8 << 20)                # longer than the ring of the smallest pass (five chunks with one reader)
    monkeypatch.setenv("BGP_READ_THREADS", "1")
    ora = O.pileup(st)
    for batch in (128 << 10, 1 << 20):
        ref, pos, cols, _, _ = _covered(ctx.stream_bam_windows(path, batch))
        _assert_columns_equal_oracle(ref, pos, cols, ora)
        ctx.reset()
    monkeypatch.delenv("BGP_READ_THREADS")
    ref, pos, cols, _, _ = _covered(ctx.stream_bam_windows(path, 256 << 10))
    _assert_columns_equal_oracle(ref, pos, cols, ora)
    ctx.reset()


def test_a_closed_ctx_has_given_back_its_device_memory(data_dir, tmp_path):
    """bgp_destroy releases every device buffer a ctx ever made (the count bgp_device_memory keeps must come back to where it stood): after
    short reads, long reads with use_md_tag (k6_prep_long's list, the checkpoints, the long providers' lists), entries, compact windows, a
    device-inflated file, a streamed file and a region load through the device."""
    from biod_amd.pileup import BaiFile
    before = device_memory()[0]
    c = Context(0)
    c.synth_device(scaled("cfg5", 300)); c.flush(); c.pileup_run()
    for w in c.pileup_windows(L.BGP_PILEUP_COLUMNS, use_md_tag=True, want_entries=True, window_positions=1 << 16):
        pass
    c.reset()
    path = os.path.join(data_dir, "ex1_header.bam")
    c.load_bam_device(path); c.flush(); c.pileup_run()
    L.check(L.lib().bgp_set_compact_columns(c._h, 1), c._h)
    for w in c.pileup_windows(L.BGP_PILEUP_COLUMNS, window_positions=500):
        pass
    c.reset()
    for w in c.stream_bam_windows(path, 128 << 10):
        pass
    c.reset()
    c.load_bam_region_device(path, BaiFile(path), 0, 100, 900); c.flush(); c.pileup_run()
    assert device_memory()[0] > before
    c.close()
    assert device_memory()[0] == before
