"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle, bit for bit.
Run on the GPU box with `pytest -m gpu`."""
import ctypes as C
import os
import struct
import re

import numpy as np
import pytest

import oracle_lib as O
from biod_amd import _lib as L
from biod_amd import bamio
from biod_amd.pileup import BamReader, Context, RecordStream, makePileup, pileupColumns
from biod_amd.synth import host_stream, host_stream_np, make_cfg, scaled

pytestmark = pytest.mark.gpu

L_ULONG_MAX = (1 << 64) - 1
FIXTURES = ["ex1_header.bam", "illu_20_chunk.bam", "ion_20_chunk.bam", "b7_295_chunk.bam", "mg1655_chunk.bam", "tags.bam", "bins.bam"]


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def stage(ctx, st, n_ref=0):
    ctx.reset()
    if n_ref:
        ctx.set_n_ref(n_ref)
    ctx.push_inflated(st)
    ctx.flush()
    return ctx


class _OraView:
    """The oracle's columns that lie inside a run (with skip_zero_coverage = false the oracle also walks the uncovered stretches
    between two runs of a reference: zero columns the device keeps no storage for)."""

    def __init__(self, ora, keep):
        self.cov, self.nstart, self.counts = ora.cov[keep], ora.nstart[keep], ora.counts[keep]


def dense_from_oracle(ora, runs, skip_zero=True):
    """Oracle columns scattered into the dense per-run layout the device produces."""
    n_pos = int(sum(int(r["end_pos"]) - int(r["start_pos"]) for r in runs))
    dense = np.zeros(n_pos, L.COLUMN_DTYPE)
    # (a reference is several runs where wide stretches of it are uncovered: a column belongs to the last run that starts at or before it)
    rkey = np.array([(int(r["ref_id"]) + 1) << 40 | int(r["start_pos"]) for r in runs], dtype=np.int64)
    ckey = ((ora.ref.astype(np.int64) + 1) << 40) | ora.pos.astype(np.int64)
    j = np.searchsorted(rkey, ckey, side="right") - 1
    inside = ora.pos.astype(np.int64) < runs["end_pos"][j].astype(np.int64) if len(ckey) else np.zeros(0, bool)
    assert len(ckey) == 0 or ((j >= 0).all() and (runs["ref_id"][j] == ora.ref).all() and (ora.cov[~inside] == 0).all())
    idx = (runs["column_base"][j].astype(np.int64) - runs["start_pos"][j].astype(np.int64) + ora.pos.astype(np.int64))[inside]
    ora = _OraView(ora, inside)
    dense["coverage"][idx] = ora.cov
    dense["n_starting_here"][idx] = ora.nstart
    for j, f in enumerate(("a", "c", "g", "t", "other", "del")):
        dense[f][idx] = ora.counts[:, j]
    return dense


def device_columns(ctx, info):
    """Copy the dense column array out of HBM through the window iterator."""
    out = []
    for w in ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, skip_zero_coverage=True, window_positions=1 << 20):      # (windows are dense inside a run either way; with
        out.append(w.columns)                                                                                      # skip_zero_coverage = false the uncovered stretches BETWEEN runs would come too)
    return np.concatenate(out) if out else np.zeros(0, L.COLUMN_DTYPE)


def check_decode(ctx, st):
    dec = ctx.decode_fetch()
    soa = O.decode_soa(st)
    assert len(dec) == len(soa.rec_off)
    assert np.array_equal(dec.rec["rec_off"], soa.rec_off)
    for f in ("ref_id", "pos", "end_pos", "bin_mq_nl", "flag_nc", "l_seq"):
        assert np.array_equal(dec.rec[f], getattr(soa, f)), f
    assert np.array_equal(dec.seq_off, soa.seq_off) and np.array_equal(dec.cigar_off, soa.cigar_off)
    assert np.array_equal(dec.seq, soa.seq), "seq"
    assert np.array_equal(dec.qual, soa.qual), "qual"
    assert np.array_equal(dec.cigar, soa.cigar), "cigar"
    return dec, soa


def check_pileup(ctx, st, entries=True, window=4096):
    info = ctx.pileup_run()
    ora = O.pileup(st, want_entries=entries)
    assert info.soft_status == (L.BGP_ESEQ if ora.status == O.ORC_ESEQ else 0)
    assert (info.n_columns, info.sum_coverage, info.max_coverage) == (ora.n_columns, ora.sum_coverage, ora.max_coverage)
    dense = device_columns(ctx, info)
    exp = dense_from_oracle(ora, info.runs)
    assert len(dense) == len(exp) == info.n_positions
    for f in L.COLUMN_DTYPE.names:
        assert np.array_equal(dense[f], exp[f]), f
    assert (dense["coverage"] == dense["a"] + dense["c"] + dense["g"] + dense["t"] + dense["other"] + dense["del"]).all()
    assert int(dense["n_starting_here"].sum()) == info.n_reads_used
    if entries:
        k = 0
        for w in ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, want_entries=True, window_positions=window):
            nz = np.nonzero(w.columns["coverage"])[0]
            for j in nz:
                e = w.entries[int(w.entry_off[j]):int(w.entry_off[j + 1])]
                oe = ora.entries[int(ora.entry_off[k]):int(ora.entry_off[k + 1])]
                assert int(ora.pos[k]) == w.first_position + j and int(ora.ref[k]) == w.ref_id
                assert np.array_equal(e["read_ordinal"], oe["read_ordinal"].astype(np.uint32)), (k, "ordinal")
                assert np.array_equal(e["cigar_op_index"], oe["cigar_op_index"].astype(np.uint16)), (k, "op index")
                assert np.array_equal(e["cigar_op_offset"], oe["cigar_op_offset"]), (k, "op offset")
                assert np.array_equal(e["query_offset"], oe["query_offset"]), (k, "query offset")
                assert np.array_equal(e["base"], oe["base"]) and np.array_equal(e["qual"], oe["qual"]), (k, "base/qual")
                k += 1
        assert k == ora.n_columns
    return info, ora


# ------------------------------------------------------------------------------------------ fixtures of the reference

@pytest.mark.parametrize("name", FIXTURES)
def test_fixture_decode_parity(ctx, data_dir, name):
    hdr, st = bamio.read_bam(os.path.join(data_dir, name))
    stage(ctx, st, len(hdr.refs))
    check_decode(ctx, st)


@pytest.mark.parametrize("name", [f for f in FIXTURES if f != "bins.bam"])
def test_fixture_pileup_parity(ctx, data_dir, name):
    hdr, st = bamio.read_bam(os.path.join(data_dir, name))
    stage(ctx, st, len(hdr.refs))
    check_pileup(ctx, st, entries=True)


def test_bins_bam_coverage_parity(ctx, data_dir):
    """Reads with CIGARs but empty SEQ: BioD would throw on current_base; coverage is still defined."""
    hdr, st = bamio.read_bam(os.path.join(data_dir, "bins.bam"))
    stage(ctx, st, len(hdr.refs))
    info, ora = check_pileup(ctx, st, entries=False)
    assert info.soft_status == L.BGP_ESEQ and info.n_columns == 1114626


def test_load_bam_matches_python_inflate(ctx, data_dir):
    path = os.path.join(data_dir, "mg1655_chunk.bam")       # has records spanning BGZF blocks
    hdr, st = bamio.read_bam(path)
    ctx.reset()
    ctx.load_bam(path, 3)
    assert ctx.references() == hdr.refs
    ctx.flush()
    check_decode(ctx, st)


@pytest.mark.parametrize("name", ["corrupted_zlib_archive.bam", "duplicated_block_size.bam", "no_block_size.bam",
                                  "wrong_bc_subfield_length.bam", "wrong_extra_gzip_length.bam"])
def test_Z1_negative_bgzf_through_abi(ctx, data_dir, name):
    ctx.reset()
    with pytest.raises(L.BiodGpuError):
        ctx.load_bam(os.path.join(data_dir, name), 2)


# ------------------------------------------------------------------------------------------ KATs through the reference-shaped API

def test_P1_through_api(ctx):
    from test_oracle_kats import P1_CIGARS, p1_stream
    st = p1_stream()
    reads = RecordStream(st, ctx)
    pile = makePileup(reads, False, 796, 849, False)
    cols = list(pile)
    assert cols[0].position == 796 and cols[-1].position == 848 and pile.ref_id == 0
    by_pos = {c.position: c for c in cols}
    assert by_pos[796].bases == "CCCCCCAC" and by_pos[805].bases == "TCCCCCCCC" and by_pos[806].bases == "AAAAAAAGA"
    assert by_pos[821].bases == "AAGG-AA" and by_pos[826].bases == "CCCCCC"
    c = by_pos[810]; assert c.reads[c.coverage - 2].cigar_after[0].type == "D"
    c = by_pos[817]; assert c.reads[c.coverage - 2].cigar_before[-1].type == "I"
    c = by_pos[821]; assert c.reads[c.coverage - 3].cigar_operation.type == "D" and c.reads[c.coverage - 3].current_base_quality == 255
    full = {c.position: c for c in makePileup(RecordStream(st, ctx), False, 0, (1 << 64) - 1, False)}
    assert len(full) == 113 and full[849].bases == "TAT"
    assert all(by_pos[p].coverage == full[p].coverage for p in by_pos)
    assert [r.name for r in full[758].reads] == ["r0"]


def test_P3_P4_through_api(ctx, data_dir):
    bam = BamReader(os.path.join(data_dir, "ex1_header.bam"), ctx=ctx)
    assert [n for n, _ in bam.reference_sequences] == ["chr1", "chr2"]
    cols = list(pileupColumns(bam.reads, want_entries=False))
    refs = np.array([c.ref_id for c in cols])
    assert int((refs == 0).sum()) == 1470 and int((refs == 1).sum()) == 1567 and (np.diff(refs) >= 0).all()
    assert cols[0].position == 99 and cols[1470].position == 0
    only = list(makePileup(bam.reads, want_entries=False))
    assert len(only) == 1470 and all(c.ref_id == 0 for c in only)


def test_P5_reads_starting_here(ctx, data_dir):
    _, st = bamio.read_bam(os.path.join(data_dir, "illu_20_chunk.bam"))
    starting = []
    for col in makePileup(RecordStream(st, ctx)):
        starting += [r.read_ordinal for r in col.reads_starting_here]
    assert starting == list(range(29))


def test_make_pileup_clipping_matches_oracle(ctx, data_dir):
    _, st = bamio.read_bam(os.path.join(data_dir, "ex1_header.bam"))
    for (a, b, skip) in [(0, (1 << 64) - 1, True), (500, 700, True), (500, 700, False), (1569, 5000, True), (1600, 1700, False), (3000, 4000, True)]:
        ora = O.pileup(st, O.MODE_MAKE_PILEUP, start_from=a, end_at=b, skip_zero=skip)
        pile = makePileup(RecordStream(st, ctx), False, a, b, skip, want_entries=False)
        cols = list(pile)
        assert [c.position for c in cols] == [int(p) for p in ora.pos], (a, b, skip)
        assert [c.coverage for c in cols] == [int(c) for c in ora.cov]
        assert [c.ref_id for c in cols] == [int(r) for r in ora.ref]


def test_zero_coverage_gaps(ctx):
    recs = [bamio.pack_record("a", "ACGTACGTAC", [(10, "M")], ref_id=0, pos=10), bamio.pack_record("b", "ACGTACGTAC", [(10, "M")], ref_id=0, pos=100),
            bamio.pack_record("c", "ACGTA", [(5, "M")], ref_id=2, pos=7)]
    st = b"".join(recs)
    for skip in (True, False):
        ora = O.pileup(st, O.MODE_PILEUP_COLUMNS, skip_zero=skip)
        cols = list(pileupColumns(RecordStream(st, ctx), False, skip, want_entries=False))
        assert [(c.ref_id, c.position, c.coverage) for c in cols] == [(int(r), int(p), int(c)) for r, p, c in zip(ora.ref, ora.pos, ora.cov)]
    ora = O.pileup(st, O.MODE_MAKE_PILEUP, start_from=50, skip_zero=False)      # sweep starts at the first surviving read
    cols = list(makePileup(RecordStream(st, ctx), False, 50, (1 << 64) - 1, False, want_entries=False))
    assert [c.position for c in cols] == [int(p) for p in ora.pos] == list(range(100, 110))


# ------------------------------------------------------------------------------------------ edge cases of the record format

def _shift_positions(st, shift_of_read):
    """The stream with read i moved shift_of_read(i) positions to the right (MD tags and everything else stay valid)."""
    import struct
    out = bytearray(st)
    for i, off in enumerate(O.frame(st)):
        off = int(off)
        (pos,) = struct.unpack_from("<i", out, off + 8)
        struct.pack_into("<i", out, off + 8, pos + shift_of_read(i))
    return bytes(out)


def test_sparse_reads_two_billion_positions_apart(ctx):
    """BioD streams such input (skip_zero_coverage jumps via initNewReference, pileup.d:388-392); dense columns over the whole
    span would be 64 GB.  A reference is cut into runs at uncovered stretches of 65 536 positions and more."""
    recs = [bamio.pack_record("a", "ACGTACGTAC", [(10, "M")], ref_id=0, pos=10), bamio.pack_record("b", "ACGTTTGTAC", [(4, "M"), (3, "D"), (6, "M")], ref_id=0, pos=1000),
            bamio.pack_record("c", "ACGTA", [(5, "M")], ref_id=0, pos=1004), bamio.pack_record("d", "ACGTA", [(5, "M")], ref_id=1, pos=7)]
    st = _shift_positions(b"".join(recs), lambda i: 1_999_999_000 if i in (1, 2) else 0)      # (beyond what a BAI bin can say, so not through pack_record)
    stage(ctx, st)
    info, ora = check_pileup(ctx, st, entries=True)
    assert len(info.runs) == 3 and info.n_positions < 100
    cols = list(makePileup(RecordStream(st, ctx), False, 0, L_ULONG_MAX, True))
    assert [c.position for c in cols] == list(range(10, 20)) + list(range(2_000_000_000, 2_000_000_013))
    assert [c.position for c in makePileup(RecordStream(st, ctx), False, 15, 2_000_000_002, True)] == list(range(15, 20)) + [2_000_000_000, 2_000_000_001]
    # skip_zero_coverage = false walks the stretch: only a clipped look at it is affordable (BioD would pop two billion columns)
    pile = makePileup(RecordStream(st, ctx), False, 15, 70_000, False, want_entries=False)
    cols = list(pile)
    assert [c.position for c in cols] == list(range(15, 70_000)) and all(c.coverage == 0 for c in cols[5:]) and cols[0].coverage == 1


@pytest.mark.parametrize("use_md", [False, True])
def test_amplicon_islands_match_oracle(ctx, use_md):
    """Islands of 30 reads on a 250 Mbp contig, 250 kbp apart (1000 of them), and 40 islands 80 kbp apart walked with
    skip_zero_coverage = false: columns, entries, reference_base against the oracle."""
    base = host_stream(make_cfg(n_reads=30_000, n_ref=1, ref_len=200_000, read_len=100, profile=1, seed=5))
    st = _shift_positions(base, lambda i: 249_000 * (i // 30))
    stage(ctx, st)
    info, ora = check_pileup(ctx, st, entries=False)
    assert len(info.runs) == 1000 and info.n_positions < 8_000_000
    if use_md:
        _check_refbase(ctx, st, skip_zero=True, window=100_000)
    small = host_stream(make_cfg(n_reads=1200, n_ref=2, ref_len=8_000, read_len=100, profile=1, seed=6))
    st2 = _shift_positions(small, lambda i: 80_000 * ((i % 600) // 30))
    stage(ctx, st2)
    info, _ = check_pileup(ctx, st2, entries=True, window=5000)
    assert len(info.runs) == 40
    for skip in (True, False):
        if use_md:                                   # (profile 1 has N skips: where BioD's chunk drifts off and its popFront throws, so must this)
            _check_refbase(ctx, st2, skip_zero=skip, window=30_000)
            continue
        ora = O.pileup(st2, skip_zero=skip)
        got_pos, got_cov = [], []
        for w in ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, skip_zero_coverage=skip, window_positions=30_000):
            keep = w.columns["coverage"] > 0 if skip else np.ones(len(w.columns), bool)
            got_pos.append((np.int64(w.ref_id) << 40) + w.first_position + np.nonzero(keep)[0]); got_cov.append(w.columns["coverage"][keep])
        assert np.array_equal(np.concatenate(got_pos), (ora.ref.astype(np.int64) << 40) + ora.pos.astype(np.int64)), skip
        assert np.array_equal(np.concatenate(got_cov), ora.cov)
    # M-only reads with consistent MD tags: reference_base through the uncovered stretches, both modes
    clean = host_stream(make_cfg(n_reads=1200, n_ref=2, ref_len=8_000, read_len=100, profile=0, seed=8))
    st3 = _shift_positions(clean, lambda i: 80_000 * ((i % 600) // 30))
    stage(ctx, st3)
    if use_md:
        for skip in (True, False):
            assert _check_refbase(ctx, st3, skip_zero=skip, window=30_000) is not None


def test_empty_and_ragged_streams(ctx):
    rec = bamio.pack_record("r", "ACGT", [(4, "M")], ref_id=0, pos=5)
    for st, n in [(b"", 0), (b"\x01\x02", 0), (rec, 1), (rec + b"\x07", 1), (rec + rec + b"\x00\x00\x00", 2)]:
        stage(ctx, st)
        assert ctx.num_records() == n
        info = ctx.pileup_run()
        assert info.n_columns == 4 * (n > 0)
    with pytest.raises(L.BiodGpuError) as e:
        stage(ctx, rec + rec[:-1])
    assert e.value.code == L.BGP_ETRUNC
    bad = bytearray(rec); bad[0:4] = (0xFFFFFFF0).to_bytes(4, "little")
    with pytest.raises(L.BiodGpuError) as e:
        stage(ctx, bytes(bad))
    assert e.value.code == L.BGP_EINVAL
    bad = bytearray(rec); bad[20:24] = (5000).to_bytes(4, "little")       # l_seq larger than the record
    with pytest.raises(L.BiodGpuError) as e:
        stage(ctx, bytes(bad) + rec)
    assert e.value.code == L.BGP_EINVAL


def test_error_statuses_match_oracle(ctx):
    lead_n = bamio.pack_record("r", "ACGTACGTAC", [(5, "N"), (10, "M")], ref_id=0, pos=10)
    stage(ctx, lead_n)
    with pytest.raises(L.BiodGpuError) as e:
        ctx.pileup_run()
    assert e.value.code == L.BGP_ECIGAR
    a = bamio.pack_record("a", "ACGT", [(4, "M")], ref_id=0, pos=50)
    b = bamio.pack_record("b", "ACGT", [(4, "M")], ref_id=0, pos=10)
    stage(ctx, a + b)
    with pytest.raises(L.BiodGpuError) as e:
        ctx.pileup_run()
    assert e.value.code == L.BGP_EUNSORTED


def test_odd_cigars(ctx):
    """Zero-length ops, P/H, =/X, trailing soft clips, unmapped reads in the middle, reads that are all clip."""
    recs = [
        bamio.pack_record("z0", "ACGTACGT", [(2, "S"), (0, "M"), (3, "M"), (0, "D"), (2, "I"), (1, "M")], ref_id=0, pos=3),
        bamio.pack_record("z1", "ACGTAC", [(1, "H"), (2, "="), (1, "P"), (2, "X"), (3, "D"), (2, "M"), (4, "H")], ref_id=0, pos=3),
        bamio.pack_record("un", "ACGT", [(4, "M")], ref_id=0, pos=4, flag=4),
        bamio.pack_record("cl", "ACGT", [(4, "S")], ref_id=0, pos=5),
        bamio.pack_record("z2", "NNRYACGT", [(8, "M")], ref_id=0, pos=6),
        bamio.pack_record("z3", "ACG", [(1, "M"), (7, "N"), (2, "M")], ref_id=1, pos=0),
    ]
    st = b"".join(recs)
    stage(ctx, st)
    check_decode(ctx, st)
    check_pileup(ctx, st, entries=True, window=3)


# ------------------------------------------------------------------------------------------ synthetic configurations

@pytest.mark.parametrize("profile,read_len,n_reads,ref_len", [(0, 150, 20000, 120_000), (1, 150, 20000, 150_000), (1, 100, 10000, 50_000),
                                                              (0, 250, 8000, 80_000), (1, (170, 255), 8000, 80_000), (1, (240, 300), 6000, 80_000),
                                                              (2, (2000, 9000), 600, 400_000)])
def test_synthetic_parity(ctx, profile, read_len, n_reads, ref_len):
    cfg = make_cfg(n_reads=n_reads, n_ref=2, ref_len=ref_len, read_len=read_len, profile=profile, seed=100 + profile)
    st = host_stream(cfg)
    ctx.reset()
    nbytes = ctx.synth_device(cfg)                      # generated by the gfx950 kernel straight into HBM
    assert nbytes == len(st)
    ctx.flush()
    dec, soa = check_decode(ctx, st)                    # also proves device bytes == host bytes for every decoded field
    check_pileup(ctx, st, entries=(profile != 2), window=8192)
    if profile == 2:                                    # long reads: entries on a few windows only
        ora = O.pileup(st, want_entries=True)
        it = ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, want_entries=True, window_positions=2000)
        k = 0
        for wi, w in enumerate(it):
            nz = np.nonzero(w.columns["coverage"])[0]
            if wi % 37 == 0:
                for jj, j in enumerate(nz):
                    e = w.entries[int(w.entry_off[j]):int(w.entry_off[j + 1])]
                    oe = ora.entries[int(ora.entry_off[k + jj]):int(ora.entry_off[k + jj + 1])]
                    assert np.array_equal(e["read_ordinal"], oe["read_ordinal"].astype(np.uint32))
                    assert np.array_equal(e["query_offset"], oe["query_offset"]) and np.array_equal(e["base"], oe["base"])
                    assert np.array_equal(e["cigar_op_index"], oe["cigar_op_index"].astype(np.uint16)) and np.array_equal(e["qual"], oe["qual"])
            k += len(nz)


def test_device_stream_bytes_equal_host(ctx):
    import torch
    cfg = make_cfg(n_reads=5000, n_ref=2, ref_len=60_000, read_len=150, profile=1, seed=5)
    host = host_stream_np(cfg)
    ctx.reset()
    n = ctx.synth_device(cfg)
    ctx.flush()
    r = ctx.decode_device(0)
    dev = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(C.c_void_p(dev.data_ptr()), C.c_void_p(r.stream), C.c_size_t(n), 3) == 0
    assert np.array_equal(dev.cpu().numpy(), host)


def test_attach_torch_tensor(ctx):
    """Caller-owned HBM: a torch uint8 tensor handed over by pointer (zero copy)."""
    import torch
    cfg = make_cfg(n_reads=3000, n_ref=1, ref_len=40_000, read_len=150, profile=1, seed=9)
    host = host_stream_np(cfg)
    t = torch.from_numpy(host).to("cuda:0")
    torch.cuda.synchronize()
    ctx.reset()
    ctx.attach_device_stream(t.data_ptr(), t.numel(), keepalive=t)
    ctx.flush()
    check_decode(ctx, host.tobytes())
    check_pileup(ctx, host.tobytes(), entries=False)


def test_scale_properties(ctx):
    """Size-independent properties at a size the oracle would take minutes for: 2 M reads of cfg3's model."""
    cfg = scaled("cfg3", 2_000_000)
    ctx.reset()
    ctx.synth_device(cfg)
    ctx.flush()
    assert ctx.num_records() == 2_000_000
    info = ctx.pileup_run()
    dec = ctx.decode_fetch(0)
    covered = (dec.rec["end_pos"] - dec.rec["pos"]).astype(np.int64)
    assert info.n_reads_used == int((covered > 0).sum()) == 2_000_000
    assert info.sum_coverage == int(covered.sum())                       # every read contributes basesCovered entries
    assert (np.diff(dec.rec["ref_id"].astype(np.int64) << 32 | dec.rec["pos"]) >= 0).all()
    n_start, cov_sum, classes = 0, 0, 0
    for w in ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, skip_zero_coverage=False, window_positions=1 << 22):
        c = w.columns
        n_start += int(c["n_starting_here"].sum()); cov_sum += int(c["coverage"].sum())
        classes += int((c["a"].astype(np.int64) + c["c"] + c["g"] + c["t"] + c["other"] + c["del"]).sum())
    assert n_start == 2_000_000 and cov_sum == classes == info.sum_coverage
    # the oracle on a prefix shard (records are independent, so a prefix of the stream is a valid stream)
    k = 20_000
    st = host_stream(cfg, 0, k)
    stage(ctx, st)
    check_pileup(ctx, st, entries=False)


def _synth_ref_chars(seed, contig, pos):
    """The generator's reference (synth_record.h: ref2): 2-bit base of contig `contig` at positions `pos`."""
    M = np.uint64
    def mix64(z):
        z = z + M(0x9E3779B97F4A7C15)
        z = (z ^ (z >> M(30))) * M(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> M(27))) * M(0x94D049BB133111EB)
        return z ^ (z >> M(31))
    pos = pos.astype(np.uint64)
    with np.errstate(over="ignore"):
        h = mix64(M(seed) ^ M(0x5EEDBA5E) ^ (M(contig) << M(36)) ^ (pos >> M(5)))
    b2 = ((h >> ((pos & M(31)) * M(2))) & M(3)).astype(np.int64)
    return np.frombuffer(b"ACGT", np.uint8)[b2]


@pytest.mark.parametrize("model,n_reads", [("cfg3", 1_000_000), ("cfg2", 1_000_000), ("cfg5", 3_000)])
def test_reference_base_synthetic_prefix_matches_oracle(ctx, model, n_reads):
    """Prefix shards of the bench models against the oracle (cfg3 has N skips: dna(read) leaves the skipped
    bases out, md/reconstruct.d:199, so chunks drift off the reference there -- in BioD and here alike)."""
    cfg = scaled(model, n_reads)
    st = host_stream(cfg, 0, 6000 if model != "cfg5" else 40)
    stage(ctx, st)
    for skip in (True, False):
        _check_refbase(ctx, st, skip_zero=skip, window=100_000)


@pytest.mark.parametrize("model,n_reads", [("cfg2", 2_000_000), ("cfg3", 2_000_000), ("cfg5", 6_000)])
def test_scale_composed_chain_equals_event_loop(ctx, model, n_reads):
    """md.hip's two routes to the same bytes: succ/exit/top/emit (function composition) vs k6_chain (the event
    loop), on shards far beyond the oracle's reach, both skip_zero_coverage modes."""
    cfg = scaled(model, n_reads)
    ctx.reset()
    ctx.synth_device(cfg)
    ctx.flush()
    for skip in (True, False):
        got = {}
        for chain in ("seq", "par"):
            os.environ["BGP_MD_CHAIN"] = chain
            try:
                got[chain] = np.concatenate([w.reference_base for w in ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, use_md_tag=True, skip_zero_coverage=skip, window_positions=1 << 22)])
            finally:
                os.environ.pop("BGP_MD_CHAIN", None)
        bad = np.nonzero(got["seq"] != got["par"])[0]
        assert len(bad) == 0, (len(bad), bad[:10])


@pytest.mark.parametrize("model,n_reads", [("cfg2", 1_000_000), ("cfg5", 3_000)])
def test_scale_reference_base_equals_generator_reference(ctx, model, n_reads):
    """The synthetic reads carry MD tags consistent with a hash-defined reference, so whatever read provides a
    column, reference_base must be that reference -- at a size far beyond the oracle's reach (short reads:
    one chunk fetch per ~read length; long reads with indels: deletion strings and insertion skips)."""
    cfg = scaled(model, n_reads)
    ctx.reset()
    ctx.synth_device(cfg)
    ctx.flush()
    n_cols = 0
    for skip in (True, False):
        for w in ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, use_md_tag=True, skip_zero_coverage=skip, window_positions=1 << 22):
            keep = w.columns["coverage"] > 0
            p = (w.first_position + np.arange(len(w.columns), dtype=np.uint64))[keep]
            exp = _synth_ref_chars(cfg.seed, w.ref_id, p)
            got = w.reference_base[keep]
            bad = np.nonzero(got != exp)[0]
            assert len(bad) == 0, (w.ref_id, [(int(p[k]), chr(got[k]), chr(exp[k])) for k in bad[:8]])
            if not skip:
                assert (w.reference_base[~keep] == ord("N")).all()       # uncovered columns: empty chunk, pileup.d:620-622
            n_cols += int(keep.sum())
    assert n_cols > n_reads


def test_cpp_facade(tmp_path, data_dir):
    """include/biodgpu_pileup.hpp: the reference's own pileup checks (test/unittests.d:320-366,
    examples/make_pileup.d:20-30) through the C++ mirror of BioD's range API."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "facade_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "facade_test.cpp"),
                           "-o", exe, "-L" + os.path.join(root, "biod_amd"), "-lbiodgpu", "-Wl,-rpath," + os.path.join(root, "biod_amd")])
    import shutil
    copy = str(tmp_path / "bins_copy.bam")
    shutil.copy(os.path.join(data_dir, "bins.bam"), copy)
    from test_oracle_kats import p1_stream
    p1 = str(tmp_path / "p1_records.bin")            # the ten literal reads of the reference's pileup unittest (pileup.d:699-749) as record bytes
    open(p1, "wb").write(p1_stream())
    out = subprocess.run([exe, os.path.join(data_dir, "ex1_header.bam"), os.path.join(data_dir, "illu_20_chunk.bam"), os.path.join(data_dir, "bins.bam"), copy, p1,
                          os.path.join(data_dir, "tags.bam")], capture_output=True, text=True)
    assert out.returncode == 0 and "facade ok" in out.stdout and "tags through column.reads" in out.stdout, out.stderr


def test_records_larger_than_framing_segments(ctx):
    """The first record is tiny (so segments are 4 KiB) and later records are hundreds of KB: interior
    segments hold no record start, the guesser finds nothing there, and the verify/re-chase fixpoint has
    to carry the chain across them.  Also a long soft-clipped read and a 30 000-op CIGAR."""
    rng = np.random.default_rng(7)

    def seq(n):
        return "".join("ACGT"[int(x)] for x in rng.integers(0, 4, n))

    recs = [bamio.pack_record("tiny", "ACGT", [(4, "M")], ref_id=0, pos=1)]
    pos = 10
    for k, n in enumerate((70_000, 300_000, 5, 120_000)):
        if k == 1:      # many small ops
            ops, left = [], n
            while left > 0:
                m = min(left, 7); ops.append((m, "M")); left -= m
                if left > 0:
                    ops.append((2, "D")); i = min(left, 3); ops.append((i, "I")); left -= i
            if ops[-1][1] == "I":
                ops.append((0, "P"))
            cigar = ops
        elif k == 3:
            cigar = [(20_000, "S"), (n - 20_000, "M")]
        else:
            cigar = [(n, "M")]
        recs.append(bamio.pack_record(f"big{k}", seq(n), cigar, ref_id=0, pos=pos, qual=bytes(rng.integers(0, 42, n).astype(np.uint8))))
        pos += 1000
    st = b"".join(recs)
    assert len(O.frame(st)) == 5
    stage(ctx, st)
    check_decode(ctx, st)
    info, ora = check_pileup(ctx, st, entries=False)
    assert info.max_coverage >= 3


def test_deep_pile(ctx):
    """12 000 reads stacked on ~150 columns: more reads in one window than its LDS list holds (several
    scan/process rounds), coverage beyond the depth-histogram bins kept in LDS, position groups of hundreds of
    reads, and -- with use_md_tag -- provider windows longer than k6_succ will scan (the event loop takes over)."""
    rng = np.random.default_rng(77)
    recs, pos = [], 1000
    for i in range(12_000):
        pos += int(rng.random() < 0.004)
        ln = int(rng.integers(60, 140))
        seq = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, ln))
        recs.append(bamio.pack_record(f"d{i}", seq, [(ln, "M")], ref_id=0, pos=pos, tags=bamio.tag_z("MD", str(ln))))
    st = b"".join(recs)
    stage(ctx, st)
    check_decode(ctx, st)
    info, ora = check_pileup(ctx, st, entries=False)
    assert info.max_coverage > 8000
    from biod_amd.shard import depth_histogram
    hist = np.zeros(L.BGP_DEPTH_BINS, np.uint64)
    import ctypes as C2
    hip = C2.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(hist.ctypes.data_as(C2.c_void_p), C2.c_void_p(info.device_depth_hist), C2.c_size_t(hist.nbytes), 2) == 0
    assert np.array_equal(hist.astype(np.int64), depth_histogram(O.pileup(st, skip_zero=False).cov))
    for skip in (True, False):
        _check_refbase(ctx, st, skip_zero=skip, window=64)


def test_very_deep_pile_takes_the_32_bit_layout(ctx):
    """70 000 reads on ~60 columns: the window's candidate range exceeds what 16-bit LDS counters can hold, so it is
    left to K4's deep pass (32-bit counters); coverage itself passes 65 535."""
    rng = np.random.default_rng(78)
    recs, pos = [], 500
    for i in range(70_000):
        pos += int(rng.random() < 0.0005)
        ln = int(rng.integers(20, 50))
        seq = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, ln))
        recs.append(bamio.pack_record(f"v{i}", seq, [(ln, "M")], ref_id=0, pos=pos))
    st = b"".join(recs)
    stage(ctx, st)
    info, ora = check_pileup(ctx, st, entries=False)
    assert info.max_coverage > 65_535


def _random_stream(rng, n_reads, quirks):
    """Coordinate-sorted random records: every CIGAR op type, optional zero-length ops, SEQ shorter or
    longer than the CIGAR's query length, unmapped reads, three references."""
    recs, pos, ref = [], 0, 0
    for i in range(n_reads):
        if rng.random() < 0.01 and ref < 2:
            ref += 1; pos = int(rng.integers(0, 50))
        pos += int(rng.integers(0, 6))
        n_ops = int(rng.integers(1, 8))
        ops = []
        for k in range(n_ops):
            op = "MIDNSHP=X"[int(rng.choice(9, p=[0.45, 0.08, 0.08, 0.04, 0.08, 0.03, 0.03, 0.1, 0.11]))]
            ln = int(rng.integers(1, 25))
            if quirks and rng.random() < 0.03:
                ln = 0
            ops.append((ln, op))
        # no leading N (that is the ECIGAR case, tested separately): make the first reference-consuming op an M
        for k, (ln, op) in enumerate(ops):
            if op in "MDN=X":
                if op == "N":
                    ops[k] = (max(ln, 1), "M")
                break
        qlen = sum(ln for ln, op in ops if op in "MIS=X")
        l_seq = qlen
        if quirks and rng.random() < 0.05:
            l_seq = max(0, qlen + int(rng.integers(-10, 11)))
        seq = "".join("ACGTNRYK="[int(x)] for x in rng.choice(9, size=l_seq, p=[0.23, 0.23, 0.23, 0.23, 0.03, 0.01, 0.01, 0.01, 0.02]))
        flag = 4 if rng.random() < 0.03 else (16 if rng.random() < 0.5 else 0)
        qual = bytes(rng.integers(0, 60, l_seq).astype(np.uint8))
        name = f"q{i}" + "x" * int(rng.integers(0, 20))
        recs.append(bamio.pack_record(name, seq, ops, ref_id=ref, pos=pos, flag=flag, qual=qual, mapq=int(rng.integers(0, 61)),
                                      tags=bamio.tag_z("XY", "z" * int(rng.integers(0, 30)))))
    return b"".join(recs)


@pytest.mark.parametrize("seed", range(6))
def test_random_cigars_match_oracle(ctx, seed):
    rng = np.random.default_rng(1000 + seed)
    st = _random_stream(rng, 400, quirks=seed % 2 == 1)
    stage(ctx, st)
    check_decode(ctx, st)
    try:
        ora = O.pileup(st, want_entries=True)
    except O.OracleError as e:                 # zero-length ops can starve the cursor: BioD's assert (pileup.d:224-228)
        assert e.code == O.ORC_ECIGAR
        with pytest.raises(L.BiodGpuError) as g:
            ctx.pileup_run()
        assert g.value.code == L.BGP_ECIGAR
        return
    check_pileup(ctx, st, entries=True, window=1021)


@pytest.mark.parametrize("depth", [3, 20, 40, 90])
def test_entries_staged_bases_edges(ctx, depth):
    """K5's listed reads (<= 3 ops) with what its staging of bases has to get right: insertions of more than eight bases inside a
    block of 64 columns (the entries beyond the staged 72 query bases are written again with loads of their own), soft clips and
    skips in front, reads of more than 255 bases with one op, every number of listed reads (a lane a read with more than 32 of
    them, eight lanes a read with eight or fewer), and a last record that ends the stream (nothing behind its qualities to read)."""
    rng = np.random.default_rng(77 + depth)
    shapes = [lambda: [(70, "M")], lambda: [(30, "M"), (20, "I"), (30, "M")], lambda: [(3, "M"), (40, "I"), (60, "M")], lambda: [(25, "S"), (50, "M")],
              lambda: [(20, "M"), (700, "N"), (40, "M")], lambda: [(40, "M"), (9, "D"), (33, "M")], lambda: [(400, "M")], lambda: [(1, "M"), (1, "I"), (90, "=")],
              lambda: [(10, "H"), (64, "M"), (5, "S")], lambda: [(60, "X"), (15, "I")]]
    recs, pos = [], 100
    for i in range(depth * 12):
        ops = shapes[int(rng.integers(len(shapes)))]()
        l_seq = sum(ln for ln, op in ops if op in "MIS=X")
        seq = "".join("ACGTN"[int(x)] for x in rng.choice(5, size=l_seq, p=[0.24, 0.24, 0.24, 0.24, 0.04]))
        recs.append(bamio.pack_record(f"r{i}", seq, ops, ref_id=0, pos=pos, qual=bytes(rng.integers(0, 60, l_seq).astype(np.uint8))))
        pos += int(rng.integers(0, max(2, 160 // depth)))
    st = b"".join(recs)                                   # the last record's qualities are the stream's last bytes
    stage(ctx, st)
    check_pileup(ctx, st, entries=True, window=997)


# ------------------------------------------------------------------------------------------ MD tags / reference_base
def _check_read_dna(ctx, st, limit=None):
    """The oracle materialises the chars BioD's lazy range yields (also when it then stalls on an exhausted
    query -- the device flags that case BGP_EMD; the flag itself is checked through reference_base parity)."""
    offs = O.frame(st)
    n_ok = n_bad = 0
    for k, off in enumerate(offs[:limit]):
        exp = O.dna_read(st, int(off))
        assert ctx.read_dna(k, cap=64, strict=False) == exp, k       # cap=64 also exercises the grow-and-retry path
        try:
            ctx.read_dna(k)
            n_ok += 1
        except L.BiodGpuError as e:
            assert e.code == L.BGP_EMD
            n_bad += 1
    return n_ok, n_bad


@pytest.mark.parametrize("name", ["illu_20_chunk.bam", "ion_20_chunk.bam", "b7_295_chunk.bam", "mg1655_chunk.bam", "tags.bam"])
def test_read_dna_fixtures(ctx, data_dir, name):
    """dna(read) on the device == the oracle's (md/reconstruct.d:38-214) for every record of the fixtures."""
    _, st = bamio.read_bam(os.path.join(data_dir, name))
    stage(ctx, st)
    n_ok, _ = _check_read_dna(ctx, st, limit=400)
    assert n_ok > 0


def test_read_dna_kats_and_odd_md(ctx):
    """M1 of SURVEY.md 8c (reconstruct.d:216-279) plus MD strings BioD accepts silently: lower case, digits as
    bases, leading/trailing zeros, empty deletions, MD shorter / longer than the CIGAR, H-typed MD, no MD."""
    cases = [("AGGTTTTGTGAGTGGGACAGTTGCAGCAAAACACAACCATAGGTGCCCATCCACCAAGGCAGGCTCTCCATCTTGCTCAGAGTGGCTCTA",
              [(89, "M")], "16", "AGGTTTTGTGAGTGGGACAGTTGCAGCAAAACACAACCATAGGTGCCCATCCACCAAGGCAGGCTCTCCATCTTGCTCAGAGTGGCTCTA"[:16]),
             ("GAGACGGGGTGACAGAGTGAGACCCTGTCTCAAAAACAAAAACAAACAAAAATTAGGCATGGTGGTGCATG", [(53, "M"), (1, "I"), (17, "M")], "53^A17", None),
             ("ACGTACGTAC", [(10, "M")], "4t5", None), ("ACGTACGTAC", [(10, "M")], "0A0C8", None), ("ACGTACGTAC", [(10, "M")], "3^5", None),
             ("ACGTACGTAC", [(4, "M"), (2, "D"), (6, "M")], "4^AC6", None), ("ACGTACGTAC", [(4, "M"), (2, "D"), (6, "M")], "4^ac6", None),
             ("ACGTACGTAC", [(10, "M")], "25", None), ("ACGTACGTAC", [(10, "M")], "2", None), ("ACGTACGTAC", [(10, "M")], "", None),
             ("ACGTACGTAC", [(10, "M")], "10A", None), ("ACGTACGTAC", [(10, "M")], "A", None), ("ACGTACGTAC", [(10, "M")], "^", None),
             ("ACGTACGTAC", [(10, "M")], "5^", None), ("ACGTACGTAC", [(10, "M")], "00010", None), ("ACGTACGTAC", [(10, "M")], "1Z2*3", None),
             ("ACGTACGTAC", [(2, "S"), (3, "M"), (1, "I"), (2, "N"), (4, "M")], "3^GG4", None),
             ("ACGTACGTAC", [(3, "M"), (7, "S")], "1^ACGTACGT2", None), ("", [(5, "M")], "5", None), ("AC", [(5, "M")], "1A3", None)]
    recs = []
    for i, (seq, ops, md, _) in enumerate(cases):
        recs.append(bamio.pack_record(f"k{i}", seq, ops, ref_id=0, pos=10 + i, tags=bamio.tag_z("XA", "q") + bamio.tag_z("MD", md)))
    recs.append(bamio.pack_record("nomd", "ACGT", [(4, "M")], ref_id=0, pos=100, tags=bamio.tag_z("XA", "q")))
    recs.append(bamio.pack_record("hmd", "ACGT", [(4, "M")], ref_id=0, pos=101, tags=b"MDH" + b"4\0"))
    recs.append(bamio.pack_record("imd", "ACGT", [(4, "M")], ref_id=0, pos=102, tags=b"MDi" + (4).to_bytes(4, "little")))
    recs.append(bamio.pack_record("bmd", "ACGT", [(4, "M")], ref_id=0, pos=103, tags=b"XBBs" + (3).to_bytes(4, "little") + b"\1\0\2\0\3\0" + bamio.tag_z("MD", "2A1")))
    st = b"".join(recs)
    stage(ctx, st)
    _check_read_dna(ctx, st)
    assert ctx.read_dna(0) == cases[0][3]


@pytest.mark.parametrize("seed", range(3))
def test_read_dna_random_md(ctx, seed):
    rng = np.random.default_rng(4000 + seed)
    recs = []
    alphabet = "0123456789ACGTN^acgt="
    for i in range(300):
        n = int(rng.integers(0, 24))
        md = "".join(alphabet[int(x)] for x in rng.integers(0, len(alphabet), n))
        ln = int(rng.integers(1, 40))
        ops = [(ln, "M")] if rng.random() < 0.5 else [(ln // 2 + 1, "M"), (2, "D"), (3, "I"), (ln // 2 + 1, "=")]
        qlen = sum(l for l, o in ops if o in "MIS=X")
        seq = "".join("ACGTN"[int(x)] for x in rng.integers(0, 5, qlen))
        recs.append(bamio.pack_record(f"r{i}", seq, ops, ref_id=0, pos=i, tags=bamio.tag_z("MD", md)))
    st = b"".join(recs)
    stage(ctx, st)
    n_ok, n_bad = _check_read_dna(ctx, st)
    assert n_ok > 50


def _check_refbase(ctx, st, mode=L.BGP_PILEUP_COLUMNS, skip_zero=True, start_from=0, end_at=O.U64_MAX, window=777):
    """Both provider-chain implementations (the event loop alone; composed successors with the event loop where
    their premises fail) against the oracle."""
    ora = None
    for chain in ("seq", "par"):
        os.environ["BGP_MD_CHAIN"] = chain
        try:
            ora = _check_refbase_one(ctx, st, mode, skip_zero, start_from, end_at, window)
        finally:
            os.environ.pop("BGP_MD_CHAIN", None)
    return ora


def _check_refbase_one(ctx, st, mode, skip_zero, start_from, end_at, window):
    """reference_base of every column the sweep visits (PileupRangeUsingMdTag, pileup.d:522-654) == oracle's;
    where BioD would throw, the device path must fail with the same status."""
    try:
        ora = O.pileup(st, mode, use_md=True, start_from=start_from, end_at=end_at, skip_zero=skip_zero)
    except O.OracleError as e:
        assert e.code in (O.ORC_EMD, O.ORC_EINVAL)
        with pytest.raises(L.BiodGpuError) as g:
            list(ctx.pileup_windows(mode, True, start_from, end_at, skip_zero, window_positions=window))
        assert g.value.code == (L.BGP_EMD if e.code == O.ORC_EMD else L.BGP_EINVAL)
        return None
    refs, poss, bases = [], [], []
    for w in ctx.pileup_windows(mode, True, start_from, end_at, skip_zero, window_positions=window):
        assert w.reference_base is not None and len(w.reference_base) == len(w.columns)
        keep = w.columns["coverage"] > 0 if skip_zero else np.ones(len(w.columns), bool)
        p = (w.first_position + np.arange(len(w.columns), dtype=np.uint64))[keep]
        refs.append(np.full(len(p), w.ref_id, np.int32)); poss.append(p); bases.append(w.reference_base[keep])
    refs = np.concatenate(refs) if refs else np.zeros(0, np.int32)
    poss = np.concatenate(poss) if poss else np.zeros(0, np.uint64)
    bases = np.concatenate(bases) if bases else np.zeros(0, np.uint8)
    assert np.array_equal(refs, ora.ref) and np.array_equal(poss, ora.pos.astype(np.uint64))
    bad = np.nonzero(bases != ora.refbase)[0]
    assert len(bad) == 0, [(int(refs[k]), int(poss[k]), chr(bases[k]), chr(ora.refbase[k])) for k in bad[:8]]
    return ora


@pytest.mark.parametrize("name", ["illu_20_chunk.bam", "ion_20_chunk.bam", "b7_295_chunk.bam", "mg1655_chunk.bam", "ex1_header.bam", "tags.bam"])
@pytest.mark.parametrize("skip_zero", [True, False])
def test_reference_base_fixtures(ctx, data_dir, name, skip_zero):
    _, st = bamio.read_bam(os.path.join(data_dir, name))
    stage(ctx, st)
    _check_refbase(ctx, st, skip_zero=skip_zero)
    _check_refbase(ctx, st, mode=L.BGP_MAKE_PILEUP, skip_zero=skip_zero)


def test_P1_P2_reference_base_through_api(ctx):
    """The pileup unittest's stream (pileup.d:696-760): reference_base of every column with use_md_tag."""
    s89 = "AGGTTTTGTGAGTGGGACAGTTGCAGCAAAACACAACCATAGGTGCCCATCCACCAAGGCAGGCTCTCCATCTTGCTCAGAGTGGCTCTA"[:89]
    reads = [("r1", s89, [(89, "M")], 60, "89"), ("r2", s89[1:], [(1, "S"), (87, "M")], 61, "87"),
             ("r3", "CCAAGGCAGGCTCTCCATCTTGCTCAGAGTGGCTCTAGGTTTTGTGAGTGGGACAGTTGC", [(20, "M"), (2, "I"), (10, "M"), (3, "D"), (28, "M")], 70, "30^ACG28")]
    st = b"".join(bamio.pack_record(n, s, ops, ref_id=0, pos=p, tags=bamio.tag_z("MD", md)) for n, s, ops, p, md in reads)
    stage(ctx, st)
    for skip in (True, False):
        ora = _check_refbase(ctx, st, skip_zero=skip)
        assert ora is not None and ora.n_columns > 0
        _check_refbase(ctx, st, mode=L.BGP_MAKE_PILEUP, skip_zero=skip, start_from=75, end_at=120)


def _md_stream(rng, n_reads, p_gap=0.02, p_nomd=0.0, p_n=0.0, p_short=0.0, p_long=0.0, nref=2, p_ref=0.01):
    """Sorted reads whose MD is derived from their own CIGAR (with mismatches and deletions), optionally
    damaged: missing, truncated (chunk runs out early) or over-long (BioD's range stalls: EMD)."""
    recs, pos, ref = [], 0, 0
    for i in range(n_reads):
        if rng.random() < p_ref and ref < nref - 1:
            ref += 1; pos = int(rng.integers(0, 30))
        pos += int(rng.integers(0, 8))
        if rng.random() < p_gap:
            pos += int(rng.integers(30, 200))
        kind, ln = rng.random(), int(rng.integers(20, 60))
        if kind < 0.6: ops = [(ln, "M")]
        elif kind < 0.7: ops = [(int(rng.integers(1, 8)), "S"), (ln, "M")]
        elif kind < 0.8: ops = [(ln // 2, "M"), (int(rng.integers(1, 4)), "I"), (ln - ln // 2, "M")]
        elif kind < 0.9: ops = [(ln // 2, "M"), (int(rng.integers(1, 6)), "D"), (ln - ln // 2, "M")]
        else: ops = [(ln // 2, "="), (ln - ln // 2, "X")]
        if rng.random() < p_n:
            ops = [(ln // 2, "M"), (int(rng.integers(5, 40)), "N"), (ln - ln // 2, "M")]
        qlen = sum(l for l, o in ops if o in "MIS=X")
        seq = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, qlen))
        md, run = "", 0
        for l, o in ops:
            if o in "M=X":
                for _ in range(l):
                    if rng.random() < 0.05:
                        md += str(run) + "ACGT"[int(rng.integers(0, 4))]; run = 0
                    else:
                        run += 1
            elif o == "D":
                md += str(run) + "^" + "".join("ACGT"[int(x)] for x in rng.integers(0, 4, l)); run = 0
        md += str(run)
        u = rng.random()
        if u < p_short:
            md = md[:max(1, len(md) // 2)]
        elif u < p_short + p_long:
            md += "A" + str(int(rng.integers(1, 30)))
        tags = bamio.tag_z("XQ", "pad") if rng.random() < p_nomd else bamio.tag_z("MD", md)
        recs.append(bamio.pack_record(f"r{i}", seq, ops, ref_id=ref, pos=pos, tags=tags))
    return b"".join(recs)


_MD_CFGS = [dict(), dict(p_gap=0.1), dict(p_n=0.1), dict(p_nomd=0.03), dict(p_short=0.05), dict(p_long=0.02),
            dict(p_gap=0.05, p_short=0.01, nref=6, p_ref=0.03), dict(p_short=0.05, nref=8, p_ref=0.05),
            dict(p_gap=0.05, p_n=0.05, p_short=0.03, p_long=0.01, nref=6, p_ref=0.03)]


@pytest.mark.parametrize("seed", range(24))
def test_reference_base_random_streams(ctx, seed):
    rng = np.random.default_rng(7000 + seed)
    st = _md_stream(rng, int(rng.integers(5, 400)), **_MD_CFGS[seed % len(_MD_CFGS)])
    stage(ctx, st)
    for skip in (True, False):
        _check_refbase(ctx, st, skip_zero=skip, window=int(rng.integers(50, 5000)))
    info = ctx.pileup_run()
    lo, hi = int(info.runs[0]["start_pos"]), int(info.runs[0]["end_pos"])
    for _ in range(3):
        a = int(rng.integers(lo, hi + 1)); b = int(rng.integers(a, hi + 50))
        _check_refbase(ctx, st, mode=L.BGP_MAKE_PILEUP, skip_zero=bool(rng.integers(0, 2)), start_from=a, end_at=b)


def test_reference_base_stale_chunk_end_across_references(ctx):
    """chunk_end_position is not reset between references (pileup.d:583-595): the first position group of a
    new reference is filtered against the previous reference's LAST provider -- which is not the read with the
    largest end when a chunk is longer than its read's span (here: a 32-base deletion string on a 20M read)."""
    long_md = "10^" + "ACGT" * 8 + "10"
    recs = [bamio.pack_record("a", "A" * 20, [(20, "M")], ref_id=0, pos=0, tags=bamio.tag_z("MD", long_md)),
            bamio.pack_record("b", "C" * 30, [(30, "M")], ref_id=0, pos=5, tags=bamio.tag_z("MD", "30")),
            bamio.pack_record("c", "G" * 10, [(10, "M")], ref_id=1, pos=3, tags=bamio.tag_z("MD", "10")),
            bamio.pack_record("d", "T" * 25, [(25, "M")], ref_id=1, pos=3, tags=bamio.tag_z("MD", "25")),
            bamio.pack_record("e", "A" * 12, [(12, "M")], ref_id=2, pos=7, tags=bamio.tag_z("MD", "12")),
            bamio.pack_record("f", "C" * 40, [(40, "M")], ref_id=2, pos=7, tags=bamio.tag_z("MD", "40"))]
    st = b"".join(recs)
    stage(ctx, st)
    for skip in (True, False):
        ora = _check_refbase(ctx, st, skip_zero=skip)
        assert ora is not None
        col = {(int(r), int(p)): chr(b) for r, p, b in zip(ora.ref, ora.pos, ora.refbase)}
        assert col[(1, 3)] == "T" and col[(2, 7)] == "C"          # d (end 28 > stale 20) and f (end 47 > stale 28) win


# ------------------------------------------------------------------------------------------ isSNP (f4)
def _expected_snp(bases, coverage, refbase, min_cov, min_wit):
    """simple.d:15-41 restated for the test: 0 no, 1 yes, 2 = the answer depends on which of several top-count
    bases druntime's associative array yields first (one of them is reference_base)."""
    if coverage < min_cov % (1 << 64):                      # size_t < int: D converts the int (simple.d:17)
        return 0
    cnt = {}
    for b in bases:
        if b != ord("-"):
            cnt[b] = cnt.get(b, 0) + 1
    if not cnt:
        return 0
    top = max(cnt.values())
    if top < min_wit:
        return 0
    tops = {b for b, v in cnt.items() if v == top}
    if refbase not in tops:
        return 1
    return 0 if len(tops) == 1 else 2


def _check_snp(ctx, st, min_cov, min_wit, with_entries):
    ora = O.pileup(st, use_md=True, want_entries=True)
    exp = {}
    for k in range(ora.n_columns):
        e = ora.entries[int(ora.entry_off[k]):int(ora.entry_off[k + 1])]
        exp[(int(ora.ref[k]), int(ora.pos[k]))] = _expected_snp(bytes(e["base"]), int(ora.cov[k]), int(ora.refbase[k]), min_cov, min_wit)
    seen = {0: 0, 1: 0, 2: 0, 3: 0}
    for w in ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, use_md_tag=True, want_entries=with_entries, window_positions=333, snp=(min_cov, min_wit)):
        assert w.snp is not None and len(w.snp) == len(w.columns)
        for k in np.nonzero(w.columns["coverage"])[0]:
            got, want = int(w.snp[k]), exp[(w.ref_id, int(w.first_position) + int(k))]
            seen[got] += 1
            if got == L.BGP_SNP_NEEDS_ENTRIES:
                assert not with_entries                     # with entries the facade resolves these
                continue
            assert got == want, (w.ref_id, int(w.first_position) + int(k), got, want)
    return seen


@pytest.mark.parametrize("name", ["illu_20_chunk.bam", "ion_20_chunk.bam", "b7_295_chunk.bam", "mg1655_chunk.bam"])
def test_is_snp_fixtures(ctx, data_dir, name):
    _, st = bamio.read_bam(os.path.join(data_dir, name))
    stage(ctx, st)
    for min_cov, min_wit in ((5, 2), (1, 1), (3, 3), (0, 0), (-1, 2)):
        seen = _check_snp(ctx, st, min_cov, min_wit, with_entries=True)
        assert seen[3] == 0


@pytest.mark.parametrize("seed", range(4))
def test_is_snp_random_low_coverage_ties_and_ambiguity_codes(ctx, seed):
    """Shallow random piles: ties between bases at most positions; IUPAC codes and '=' in SEQ so that the merged
    'other' counter can hold the top count (resolved from the entries, or reported as such without them)."""
    rng = np.random.default_rng(9100 + seed)
    recs, pos = [], 10
    for i in range(300):
        pos += int(rng.integers(0, 4))
        ln = int(rng.integers(8, 30))
        seq = "".join("ACGTNRYM="[int(x)] for x in rng.choice(9, size=ln, p=[0.22, 0.22, 0.22, 0.22, 0.04, 0.02, 0.02, 0.02, 0.02]))
        ops = [(ln, "M")] if rng.random() < 0.8 else [(ln // 2, "M"), (2, "D"), (ln - ln // 2, "M")]
        md = str(ln) if len(ops) == 1 else f"{ln // 2}^AC{ln - ln // 2}"
        recs.append(bamio.pack_record(f"s{i}", seq, ops, ref_id=0, pos=pos, tags=bamio.tag_z("MD", md)))
    st = b"".join(recs)
    stage(ctx, st)
    tot = {0: 0, 1: 0, 2: 0, 3: 0}
    for min_cov, min_wit in ((5, 2), (1, 1), (2, 2)):
        a = _check_snp(ctx, st, min_cov, min_wit, with_entries=True)
        b = _check_snp(ctx, st, min_cov, min_wit, with_entries=False)
        assert a[3] == 0
        for k in tot:
            tot[k] += a[k] + b[k]
    assert tot[1] > 0 and tot[2] > 0 and tot[3] > 0 and tot[0] > 0


# ------------------------------------------------------------------------------------------ BAI / region queries (f3)
def _expected_region(st_all_records, ref_id, beg, end):
    import bai_oracle as B
    kept = B.region_filter(B.frame(st_all_records), ref_id, beg, end)
    sizes = {r[0]: 4 + int(np.frombuffer(st_all_records, "<i4", 1, r[0])[0]) for r in kept}
    return kept, b"".join(st_all_records[r[0]:r[0] + sizes[r[0]]] for r in kept)


def _check_subset_decode(dec, exp_stream):
    soa = O.decode_soa(exp_stream)
    assert len(dec) == len(soa.rec_off)
    for f in ("ref_id", "pos", "end_pos", "bin_mq_nl", "flag_nc", "l_seq"):
        assert np.array_equal(dec.rec[f], getattr(soa, f)), f
    assert np.array_equal(dec.seq_off, soa.seq_off) and np.array_equal(dec.cigar_off, soa.cigar_off)
    e = np.zeros(0, np.uint8)
    assert np.array_equal(dec.seq if dec.seq is not None else e, soa.seq) and np.array_equal(dec.qual if dec.qual is not None else e, soa.qual)
    assert np.array_equal(dec.cigar if dec.cigar is not None else np.zeros(0, np.uint32), soa.cigar)


def test_region_queries_match_reference_checks(data_dir):
    """test/unittests.d:145-206 through the device path: bam["large"][beg .. end] == naive filter for the reference's
    40 regions (names, decoded fields, and the pileup of the region's reads), and the first read of every reference
    found through startVirtualOffset + getReadAt."""
    import bai_oracle as B
    from test_oracle_bai import REGIONS
    path = os.path.join(data_dir, "bins.bam")
    hdr, st = bamio.read_bam(path)
    recs = B.frame(st)
    bam = BamReader(path, n_threads=2)
    large = bam["large"]
    for beg, end in REGIONS:
        rr = large[beg:end]
        exp = [r for r in recs if r[1] == large.id and r[2] < end and r[3] > beg]          # unittests.d:152-157
        assert rr.names() == [r[4] for r in exp], (beg, end)
        kept, exp_stream = _expected_region(st, large.id, beg, end)
        assert [r[4] for r in kept] == [r[4] for r in exp]
        _check_subset_decode(rr.decode(), exp_stream)
        if exp:
            info = rr._stage().pileup_run()
            ora = O.pileup(exp_stream)
            assert (info.n_columns, info.sum_coverage, info.max_coverage) == (ora.n_columns, ora.sum_coverage, ora.max_coverage)
    for name in ("tiny", "small", "large"):
        vo = bam[name].startVirtualOffset()
        one = bam.getReadAt(vo)
        rec = one.ctx.decode_fetch(0).rec
        data = one.ctx.staged_bytes()
        assert len(rec) == 1
        off, ln = int(rec["rec_off"][0]) + 36, int(rec["bin_mq_nl"][0]) & 0xFF
        assert data[off:off + ln - 1].tobytes().decode() == f"{name}:r1:0..1:len1:bin4681:hexbin0x1249"
        assert one.ctx.record_voffset(0) == vo
    # virtual offsets of a whole-file load: the first record sits right behind the header
    v0 = bam.ctx.record_voffset(0)
    assert bam.getReadAt(v0).ctx.decode_fetch(0).rec["pos"][0] == bam.ctx.decode_fetch(0).rec["pos"][0]


@pytest.mark.parametrize("name", ["ex1_header.bam", "tags.bam", "bins.bam"])
def test_region_load_and_filter_match_oracle(ctx, data_dir, name):
    """Index-driven load (getChunks + StreamChunksSupplier + BamReadFilter) == the oracle's restatement of it, and
    BamReadFilter over the whole file == the oracle's filter.  The two differ where the index says so: ex1's first
    read (unmapped, placed at 99) lies before the linear index's first offset, so bam["chr1"][0 .. $] starts at the
    second read in BioD (chunk.beg is clamped to it, randomaccessmanager.d:209-219) -- and here."""
    import bai_oracle as B
    from biod_amd.pileup import BaiFile
    path = os.path.join(data_dir, name)
    hdr, st = bamio.read_bam(path)
    bai = BaiFile(path)
    obai = B.parse_bai(path)
    rng = np.random.default_rng(17)
    for rid, (rname, rlen) in enumerate(hdr.refs):
        for beg, end in [(0, max(rlen, 1)), (0, 1), (10, 11)] + [(int(a), int(a) + int(b)) for a, b in zip(rng.integers(0, max(rlen, 2), 6), rng.integers(1, 3000, 6))]:
            cst, ckept = B.region_reads(path, obai, rid, beg, end)
            _, exp_index = _expected_region(cst, rid, beg, end)
            ctx.reset(); ctx.load_bam_region(path, bai, rid, beg, end, 2); ctx.flush()
            _check_subset_decode(ctx.decode_fetch(), exp_index)
            kept, exp_stream = _expected_region(st, rid, beg, end)
            ctx.reset(); ctx.load_bam(path, 2); ctx.flush(); ctx.filter_region(rid, beg, end)
            _check_subset_decode(ctx.decode_fetch(), exp_stream)
            if kept:
                check_pileup_subset = O.pileup(exp_stream)
                info = ctx.pileup_run()
                assert (info.n_columns, info.sum_coverage) == (check_pileup_subset.n_columns, check_pileup_subset.sum_coverage)
    with pytest.raises(L.BiodGpuError):
        ctx.reset(); ctx.load_bam_region(path, bai, 0, 5, 5, 1)          # enforce(start < end), reference.d:76
    with pytest.raises(L.BiodGpuError):
        ctx.reset(); ctx.load_bam_region(path, bai, len(hdr.refs), 0, 5, 1)   # checkRefId, randomaccessmanager.d:205-207
    ctx.reset()


def test_filter_region_stops_like_the_range(ctx):
    """BamReadFilter ends at the first read past the region (randomaccessmanager.d:398-431) -- on unsorted input
    that is not the same as a predicate, and unmapped reads (ref_id -1 = uint.max) end it too."""
    import bai_oracle as B
    rng = np.random.default_rng(23)
    recs = []
    for i in range(600):
        ref = int(rng.choice([0, 1, 1, 1, 2, -1], p=[0.1, 0.3, 0.3, 0.2, 0.05, 0.05])) if i > 50 else int(rng.integers(0, 2))
        pos = int(rng.integers(0, 3000))
        ln = int(rng.integers(0, 80))
        ops = [(ln, "M")] if ln else []
        recs.append(bamio.pack_record(f"u{i}", "A" * ln, ops, ref_id=ref, pos=pos))
    st = b"".join(recs)
    for beg, end in [(0, 3000), (100, 200), (1500, 1501), (2999, 5000), (0, 1)]:
        stage(ctx, st)
        ctx.filter_region(1, beg, end)
        kept, exp_stream = _expected_region(st, 1, beg, end)
        _check_subset_decode(ctx.decode_fetch(), exp_stream)


def test_reference_base_stale_chunk_end_above_the_new_winner(ctx):
    """Found by scratch/fuzz_gpu.py: the stale chunk_end can also be LARGER than the end of the new reference's first
    provider.  add() then turns down a longer read of the first position group (it does not end beyond the stale
    value) and the sweep never looks at it again -- columns past the short provider are 'N' -- while the composed
    successor function would have picked it up as a candidate.  Such a start is flagged and redone by the event loop."""
    recs = [bamio.pack_record("a", "A" * 100, [(100, "M")], ref_id=0, pos=0, tags=bamio.tag_z("MD", "100")),
            bamio.pack_record("c", "G" * 35, [(35, "M")], ref_id=1, pos=6, tags=bamio.tag_z("MD", "35")),
            bamio.pack_record("d", "T" * 56, [(56, "M")], ref_id=1, pos=6, tags=bamio.tag_z("MD", "56")),
            bamio.pack_record("e", "C" * 30, [(30, "M")], ref_id=1, pos=50, tags=bamio.tag_z("MD", "30"))]
    st = b"".join(recs)
    stage(ctx, st)
    for skip in (True, False):
        ora = _check_refbase(ctx, st, skip_zero=skip)
        col = {(int(r), int(p)): chr(b) for r, p, b in zip(ora.ref, ora.pos, ora.refbase)}
        assert col[(1, 40)] == "G" and col[(1, 41)] == "N" and col[(1, 49)] == "N"


def test_hist_reduce_single_rank_rccl(ctx):
    """bgp_hist_reduce through a real RCCL communicator (one rank): resolves librccl at run time,
    sums uint64 in place -- with one rank the histogram must come back unchanged."""
    import torch
    rccl = C.CDLL("librccl.so.1")
    comm = C.c_void_p()
    dev = (C.c_int * 1)(0)
    assert rccl.ncclCommInitAll(C.byref(comm), 1, dev) == 0
    try:
        cfg = make_cfg(n_reads=4000, n_ref=2, ref_len=40_000, read_len=150, profile=1, seed=21)
        st = host_stream(cfg)
        stage(ctx, st)
        info = ctx.pileup_run()
        hist = torch.zeros(L.BGP_DEPTH_BINS, dtype=torch.int64, device="cuda:0")
        hip = C.CDLL("libamdhip64.so")
        assert hip.hipMemcpy(C.c_void_p(hist.data_ptr()), C.c_void_p(info.device_depth_hist), C.c_size_t(L.BGP_DEPTH_BINS * 8), 3) == 0
        before = hist.cpu().numpy().copy()
        L.check(L.lib().bgp_hist_reduce(ctx.handle, comm, C.c_void_p(hist.data_ptr()), L.BGP_DEPTH_BINS), ctx.handle)
        after = hist.cpu().numpy()
        assert np.array_equal(before, after)
        ora = O.pileup(st, skip_zero=False)
        from biod_amd.shard import depth_histogram
        assert np.array_equal(after, depth_histogram(ora.cov))
    finally:
        rccl.ncclCommDestroy(comm)


# ---- f3: createIndex (bai/indexing.d:56-366) -- the .bai of a staged file built from the device record table --------

def _write_bgzf(path, payload, rng, eof=True, small=False):
    """payload cut into BGZF blocks of random sizes: records straddle block boundaries freely."""
    with open(path, "wb") as f:
        at = 0
        while at < len(payload):
            n = int(rng.integers(1, 400)) if small else int(rng.integers(200, 65000))
            f.write(bamio.bgzf_block(payload[at:at + n]))
            at += n
        if eof:
            f.write(bamio.BGZF_EOF)


def _check_created_index(path):
    import bai_oracle as B
    bam = BamReader(path, 2)
    got = bam.ctx.bai_build()
    exp = B.create_index(path)
    assert got == exp
    return bam, got


@pytest.mark.parametrize("name", ["ex1_header.bam", "bins.bam", "tags.bam", "illu_20_chunk.bam", "ion_20_chunk.bam", "mg1655_chunk.bam", "b7_295_chunk.bam"])
def test_create_index_fixtures(data_dir, name):
    _check_created_index(os.path.join(data_dir, name))


def test_created_index_answers_the_reference_region_checks(data_dir, tmp_path):
    """unittests.d:145-185 again, this time through the index this library wrote."""
    import shutil
    src = os.path.join(data_dir, "bins.bam")
    dst = str(tmp_path / "bins.bam")
    shutil.copy(src, dst)
    bam = BamReader(dst, 2)
    assert not bam.has_index
    bam.createIndex()
    assert bam.has_index
    all_rec = bam.reads.decode(0).rec
    for beg, end in [(1400, 1500), (10, 123), (135, 1236), (0, 3000), (50000, 50100), (950000, 950100), (2**20 - 5, 2**20 + 40)]:
        naive = int(np.sum((all_rec["ref_id"] == 2) & (all_rec["pos"] < end) & (all_rec["end_pos"] > beg)))
        assert len(bam["large"][beg:end].names()) == naive, (beg, end)


def _index_stream(rng, n_reads, n_ref, quirks):
    recs, ref, pos = [], int(rng.integers(0, 2)), 0
    if quirks:                                         # reads the builder skips, ahead of the first one it takes
        for k in range(int(rng.integers(0, 3))):
            recs.append(bamio.pack_record(f"pre{k}", "ACGT", [(4, "M")], ref_id=ref, pos=-1, flag=int(rng.choice([0, 4]))))
    for i in range(n_reads):
        if rng.random() < 0.02 and ref < n_ref - 1:
            ref += int(rng.integers(1, 3)); ref = min(ref, n_ref - 1); pos = int(rng.integers(0, 100))
            if quirks and rng.random() < 0.5:          # skipped read of the NEW reference: still counted for the old one
                recs.append(bamio.pack_record(f"neg{i}", "AC", [(2, "M")], ref_id=ref, pos=-1))
        pos += int(rng.choice([0, 1, 7, 300, 5000, 40000], p=[0.2, 0.3, 0.3, 0.1, 0.07, 0.03]))
        ln = int(rng.choice([1, 30, 100, 20000, 70000], p=[0.05, 0.5, 0.35, 0.07, 0.03]))
        ops = [(10, "M"), (ln, "N"), (10, "M")] if ln > 1000 else [(ln, "M")]
        flag = 4 if rng.random() < 0.05 else 0
        if quirks and rng.random() < 0.03:
            ops = []                                   # nothing covered: end window = (pos - 1) / 16384
        seq = "A" * sum(l for l, o in ops if o == "M")
        rec = bytearray(bamio.pack_record(f"r{i}", seq, ops, ref_id=ref, pos=pos, flag=flag))
        if quirks and rng.random() < 0.05:             # the bin field is taken as it is (check_bins off)
            struct_bin = int(rng.choice([0, 1, 9, 73, 585, 4681, 4700, 37449, 40000]))
            rec[14:16] = struct_bin.to_bytes(2, "little")
        recs.append(bytes(rec))
    for k in range(int(rng.integers(0, 4))):
        recs.append(bamio.pack_record(f"u{k}", "ACGT", [], ref_id=-1, pos=-1, flag=4))
    return recs


def test_create_index_synthetic_file(tmp_path):
    """A generator file (cfg3 read model, 60k reads, BamWriter block layout): device index == oracle index, bytes."""
    from biod_amd.synth import CONFIGS
    m = CONFIGS["cfg3"]                                # spliced reads: parent bins break the leaf runs
    cfg = make_cfg(n_reads=60_000, n_ref=3, ref_len=20_000_000, read_len=m["read_len"], profile=m["profile"], seed=5)
    st = host_stream(cfg)
    offs = O.frame(st)
    recs = [bytes(st[int(offs[i]):int(offs[i + 1]) if i + 1 < len(offs) else len(st)]) for i in range(len(offs))]
    path = str(tmp_path / "synth.bam")
    bamio.write_bam(path, [(f"chr{i + 1}", 1 << 28) for i in range(max(1, cfg.n_ref))], recs)
    _check_created_index(path)


@pytest.mark.parametrize("seed", range(8))
def test_create_index_random_files(tmp_path, seed):
    rng = np.random.default_rng(7700 + seed)
    n_ref = int(rng.integers(3, 9))
    refs = [(f"c{i}", 1 << 29) for i in range(n_ref)]
    recs = _index_stream(rng, int(rng.integers(50, 1500)), n_ref, quirks=seed % 2 == 1)
    path = str(tmp_path / f"r{seed}.bam")
    _write_bgzf(path, bamio.bam_header_bytes(refs) + b"".join(recs), rng, eof=seed % 3 != 2, small=seed % 4 == 3)
    _check_created_index(path)


def test_create_index_edges(tmp_path):
    import bai_oracle as B
    rng = np.random.default_rng(5)
    refs = [("a", 1000), ("b", 1000), ("c", 1000)]
    for tag, recs in {
        "none": [],
        "unplaced_only": [bamio.pack_record("u", "AC", [], ref_id=-1, pos=-1, flag=4)] * 3,
        "skipped_only": [bamio.pack_record("s", "AC", [(2, "M")], ref_id=1, pos=-1)] * 2,
        "last_reference": [bamio.pack_record("x", "ACGT", [(4, "M")], ref_id=2, pos=7)],
    }.items():
        path = str(tmp_path / f"{tag}.bam")
        _write_bgzf(path, bamio.bam_header_bytes(refs) + b"".join(recs), rng)
        _check_created_index(path)
    # not coordinate-sorted: the builder's enforce (indexing.d:253-268)
    bad = [bamio.pack_record("x", "ACGT", [(4, "M")], ref_id=1, pos=70), bamio.pack_record("y", "ACGT", [(4, "M")], ref_id=1, pos=69)]
    path = str(tmp_path / "unsorted.bam")
    _write_bgzf(path, bamio.bam_header_bytes(refs) + b"".join(bad), rng)
    with pytest.raises(B.NotSorted):
        B.create_index(path)
    with pytest.raises(L.BiodGpuError) as e:
        BamReader(path, 1).ctx.bai_build()
    assert e.value.code == L.BGP_EUNSORTED
    back = [bamio.pack_record("x", "ACGT", [(4, "M")], ref_id=1, pos=70), bamio.pack_record("y", "ACGT", [(4, "M")], ref_id=0, pos=90)]
    path = str(tmp_path / "refback.bam")
    _write_bgzf(path, bamio.bam_header_bytes(refs) + b"".join(back), rng)
    with pytest.raises(B.NotSorted):
        B.create_index(path)
    with pytest.raises(L.BiodGpuError):
        BamReader(path, 1).ctx.bai_build()
    # anything but one whole file staged
    c = Context()
    c.push_inflated(bad[0]); c.flush()
    with pytest.raises(L.BiodGpuError) as e:
        c.bai_build()
    assert e.value.code == L.BGP_ESTATE


# ---- f2: decompressBgzfBlock on the device (bgzf/block.d:127-216): k_inflate against zlib ---------------------------

def _device_stream(path):
    c = Context(0)
    c.load_bam_device(path)
    c.flush()
    return c, c.staged_bytes().tobytes()


@pytest.mark.parametrize("name", FIXTURES)
def test_device_inflate_fixtures(data_dir, name):
    path = os.path.join(data_dir, name)
    hdr, st = bamio.read_bam(path)
    c, got = _device_stream(path)
    assert got == st
    assert c.references() == [(n, l) for n, l in hdr.refs]
    check_decode(c, st)
    host = Context(0); host.load_bam(path, 2); host.flush()
    n = host.num_records()
    assert c.num_records() == n
    for i in sorted({0, n // 2, n - 1} if n else set()):
        assert c.record_voffset(i) == host.record_voffset(i)
    assert c.bai_build() == host.bai_build()


@pytest.mark.parametrize("name", ["corrupted_zlib_archive.bam", "duplicated_block_size.bam", "no_block_size.bam", "wrong_bc_subfield_length.bam", "wrong_extra_gzip_length.bam"])
def test_device_inflate_negative_fixtures(data_dir, name):
    """test/unittests.d:132-142: every malformed container is refused, on this path too."""
    c = Context(0)
    with pytest.raises(L.BiodGpuError):
        c.load_bam_device(os.path.join(data_dir, name))
        c.flush()


@pytest.mark.parametrize("level,block", [(0, 30000), (1, 65000), (6, 65280), (9, 65536), (6, 700), (9, 40), (1, 5)])
def test_device_inflate_block_types_and_sizes(tmp_path, level, block):
    """Stored, fixed-code and dynamic-code DEFLATE blocks, from 5-byte to 64 KB BGZF blocks, records straddling."""
    import zlib
    rng = np.random.default_rng(level * 1000 + block)
    st = _random_stream(rng, 1500 if block > 1000 else 120, quirks=False)
    payload = bamio.bam_header_bytes([("a", 1 << 20), ("b", 1 << 20), ("c", 1 << 20)]) + st
    path = str(tmp_path / "x.bam")
    with open(path, "wb") as f:
        for at in range(0, len(payload), block):
            chunk = payload[at:at + block]
            co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, zlib.Z_FIXED if (level == 9 and block == 40) else zlib.Z_DEFAULT_STRATEGY)
            cdata = co.compress(chunk) + co.flush()
            f.write(bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0]) + (len(cdata) + 25).to_bytes(2, "little") + cdata
                    + (zlib.crc32(chunk) & 0xFFFFFFFF).to_bytes(4, "little") + len(chunk).to_bytes(4, "little"))
        f.write(bamio.BGZF_EOF)
    c, got = _device_stream(path)
    assert got == st
    check_decode(c, st)


def test_device_inflate_synthetic_and_damage(tmp_path):
    """A 150k-read generator file (long runs of equal quality bytes: overlapping matches) equals the host inflate;
    one flipped bit anywhere in a block's CDATA, CRC or ISIZE is refused."""
    cfg = scaled("cfg2", 150_000)
    st = host_stream(cfg)
    path = str(tmp_path / "s.bam")
    rng = np.random.default_rng(3)
    _write_bgzf(path, bamio.bam_header_bytes([(f"chr{i + 1}", 1 << 28) for i in range(max(1, cfg.n_ref))]) + st, rng)
    c, got = _device_stream(path)
    assert got == st
    info = c.pileup_run()
    host = Context(0); host.load_bam(path, 2); host.flush()
    hinfo = host.pileup_run()
    assert (info.n_columns, info.sum_coverage, info.max_coverage) == (hinfo.n_columns, hinfo.sum_coverage, hinfo.max_coverage)
    raw = bytearray(open(path, "rb").read())
    size = len(raw)
    for trial in range(12):
        bad = bytearray(raw)
        at = int(rng.integers(200, size - 40))
        bad[at] ^= 1 << int(rng.integers(0, 8))
        p2 = str(tmp_path / f"bad{trial}.bam")
        open(p2, "wb").write(bytes(bad))
        host_ok = True
        try:
            h = Context(0); h.load_bam(p2, 2)
        except L.BiodGpuError:
            host_ok = False
        dev_ok = True
        try:
            d = Context(0); d.load_bam_device(p2)
        except L.BiodGpuError:
            dev_ok = False
        assert dev_ok == host_ok, (trial, at)


def test_device_inflate_file_of_several_chunks(tmp_path, monkeypatch):
    """A file of several read chunks (16 MB each): the blocks are inflated in batches while the file is still being uploaded
    (bgzf_load_bam_device's dispatcher: one upload stream, a batch to whichever inflate stream has run dry) -- the
    same bytes as the host inflate and as one launch at the end (BGP_INFLATE_PIPELINE=0)."""
    from concurrent.futures import ThreadPoolExecutor
    cfg = scaled("cfg2", 500_000)
    st = host_stream(cfg)
    path = str(tmp_path / "big.bam")
    b = memoryview(st)
    with open(path, "wb") as f, ThreadPoolExecutor(8) as ex:
        f.write(bamio.bgzf_block(bamio.bam_header_bytes([(f"chr{i + 1}", 1 << 28) for i in range(max(1, cfg.n_ref))])))
        for blk in ex.map(lambda at: bamio.bgzf_block(bytes(b[at:at + 60000])), range(0, len(b), 60000)):
            f.write(blk)
        f.write(bamio.BGZF_EOF)
    assert os.path.getsize(path) > 2 * (32 << 20)
    c, got = _device_stream(path)
    assert got == st
    info = c.pileup_run()
    assert info.n_columns > 0 and c.num_records() == cfg.n_reads
    c2, got2 = _device_stream(path)                       # (again in a fresh ctx, and in the first one: its streams are reused)
    assert got2 == st
    c.reset(); c.load_bam_device(path); c.flush()
    assert c.staged_bytes().tobytes() == st
    monkeypatch.setenv("BGP_INFLATE_BATCH_MIN", "50")          # dozens of batches, both inflate streams busy while the file still arrives
    for _ in range(3):
        c4, got4 = _device_stream(path)
        assert got4 == st
    monkeypatch.delenv("BGP_INFLATE_BATCH_MIN")
    monkeypatch.setenv("BGP_INFLATE_PIPELINE", "0")
    c3, got3 = _device_stream(path)
    assert got3 == st


def test_device_inflate_file_that_outgrows_the_size_estimate(tmp_path):
    """bgzf_load_bam_device sizes the inflated stream from the first read chunk's ratio (and a third more); a file whose later
    blocks inflate far more than that -- here one record repeated a million times behind 17 MB of ordinary reads -- is inflated in
    one launch at the end instead (batches already launched have stayed inside the estimate): the same bytes as the host loader."""
    from concurrent.futures import ThreadPoolExecutor
    cfg = scaled("cfg2", 115_000)
    head = host_stream(cfg)
    at, last = 0, 0
    while at < len(head):                                  # the last record of the ordinary part
        last = at; at += 4 + int.from_bytes(head[at:at + 4], "little")
    st = head + head[last:] * 1_200_000
    path = str(tmp_path / "grow.bam")
    b = memoryview(st)
    with open(path, "wb") as f, ThreadPoolExecutor(8) as ex:
        f.write(bamio.bgzf_block(bamio.bam_header_bytes([(f"chr{i + 1}", 1 << 28) for i in range(max(1, cfg.n_ref))])))
        for blk in ex.map(lambda o: bamio.bgzf_block(bytes(b[o:o + 64000])), range(0, len(b), 64000)):
            f.write(blk)
        f.write(bamio.BGZF_EOF)
    size = os.path.getsize(path)
    assert size > (17 << 20) and len(st) > 3 * size + (64 << 20)       # the first chunk is ordinary reads; the whole outgrows ratio * 1.34 + 64 MB
    c, got = _device_stream(path)
    assert got == st
    assert c.num_records() == cfg.n_reads + 1_200_000
    c.reset(); c.load_bam_device(path); c.flush()                       # (again in the same ctx: the buffers have the real size now)
    assert c.staged_bytes().tobytes() == st


def test_bam_reader_with_device_inflate(data_dir):
    """BamReader(filename, device_inflate=True): the reference's pileup checks (unittests.d:320-366) hold unchanged."""
    bam = BamReader(os.path.join(data_dir, "ex1_header.bam"), device_inflate=True)
    assert [n for n, _ in bam.reference_sequences] == ["chr1", "chr2"]
    expected = {0: 1470, 1: 1567}
    for column in pileupColumns(bam.reads):
        expected[column.ref_id] -= 1
    assert expected == {0: 0, 1: 0}
    assert bam["chr2"][100:200].names() == BamReader(os.path.join(data_dir, "ex1_header.bam"))["chr2"][100:200].names()


def _long_md_read(rng, name, ref, pos, length, md_mode="full"):
    """A long read (indel every ~40 bases) with its MD tag; md_mode: full | short (tag ends early) | over (claims more
    than the CIGAR has) | none."""
    ops, md, run, seq = [], "", 0, []
    left = length
    while left > 0:
        m = int(min(left, rng.integers(10, 80)))
        ops.append((m, "M")); left -= m
        for _ in range(m):
            if rng.random() < 0.04:
                md += str(run) + "ACGT"[int(rng.integers(0, 4))]; run = 0
            else:
                run += 1
        seq.append("".join("ACGT"[int(x)] for x in rng.integers(0, 4, m)))
        if left > 0:
            if rng.random() < 0.5:
                k = int(rng.integers(1, 4)); ops.append((k, "I")); seq.append("".join("ACGT"[int(x)] for x in rng.integers(0, 4, k)))
            else:
                k = int(rng.integers(1, 5)); ops.append((k, "D"))
                md += str(run) + "^" + "".join("ACGT"[int(x)] for x in rng.integers(0, 4, k)); run = 0
    md += str(run)
    if md_mode == "short":
        md = md[:len(md) // 3]
    elif md_mode == "over":
        md += "A999"
    tags = bamio.tag_z("XQ", "pad") if md_mode == "none" else bamio.tag_z("MD", md)
    return bamio.pack_record(name, "".join(seq), ops, ref_id=ref, pos=pos, tags=tags)


@pytest.mark.parametrize("seed", range(4))
def test_reference_base_long_chunks_in_pieces(ctx, seed):
    """Chunks longer than 4096 chars go through checkpoints and one lane per 512 chars (k6_ck_build / k6_emit_long):
    full tags, tags that end early (the pieces behind the end stay 'N'), tags that claim more than the CIGAR, short
    reads in between, two references."""
    rng = np.random.default_rng(880 + seed)
    recs, pos, ref = [], 0, 0
    for i in range(14):
        if i == 8:
            ref, pos = 1, 5
        mode = ["full", "full", "short", "full", "over"][int(rng.integers(0, 5))] if seed else "full"
        length = int(rng.integers(5000, 16000)) if i % 3 != 2 else int(rng.integers(30, 200))
        recs.append(_long_md_read(rng, f"L{i}", ref, pos, length, mode))
        pos += int(rng.integers(200, 7000))
    st = b"".join(recs)
    stage(ctx, st)
    ctx.timing_enable(True); ctx.timing_reset()
    try:
        for skip in (True, False):
            _check_refbase(ctx, st, skip_zero=skip, window=int(rng.integers(500, 9000)))
        assert ctx.timing_get().get("k6_emit_long", {}).get("launches", 0) > 0      # the path under test did run
    finally:
        ctx.timing_enable(False)
    info = ctx.pileup_run()
    lo, hi = int(info.runs[0]["start_pos"]), int(info.runs[0]["end_pos"])
    a = int(rng.integers(lo, hi)); b = int(rng.integers(a, hi + 50))
    _check_refbase(ctx, st, mode=L.BGP_MAKE_PILEUP, skip_zero=False, start_from=a, end_at=b)


def _long_md_read_odd(rng, name, ref, pos, length, kind):
    """A long read whose tag is cut into tokens the unusual way, or damaged: what k6_ck_build's prefix sums must either reproduce or hand to
    the serial walk.  kind: plain | big_match (one M, one number: checkpoints inside a token) | long_dels (deletions of hundreds of bases:
    checkpoints inside a deletion) | del_last (the tag ends with a deletion: the backward parse's token) | mm_last (... with a mismatch) |
    zero_last (... with a 0 behind a mismatch) | lower (lower-case mismatches and one lower-case deleted base) | eq_n (= and N among the
    mismatches) | empty_del | tail (two letters at the end) | digits (a number of eleven digits) | over | short | lead_zero ("0" first)."""
    ops, md, run, seq = [], "", 0, []
    if kind == "big_match":
        seq = ["".join("ACGT"[int(x)] for x in rng.integers(0, 4, length))]
        return bamio.pack_record(name, seq[0], [(length, "M")], ref_id=ref, pos=pos, tags=bamio.tag_z("MD", str(length)))
    left = length
    mm = "acgt" if kind == "lower" else ("ACGT=N" if kind == "eq_n" else "ACGT")
    while left > 0:
        m = int(min(left, rng.integers(10, 80)))
        ops.append((m, "M")); left -= m
        for _ in range(m):
            if rng.random() < 0.04:
                md += str(run) + mm[int(rng.integers(0, len(mm)))]; run = 0
            else:
                run += 1
        seq.append("".join("ACGT"[int(x)] for x in rng.integers(0, 4, m)))
        if left > 0:
            if rng.random() < 0.5:
                k = int(rng.integers(1, 4)); ops.append((k, "I")); seq.append("".join("ACGT"[int(x)] for x in rng.integers(0, 4, k)))
            else:
                k = int(rng.integers(200, 900)) if kind == "long_dels" and rng.random() < 0.04 else int(rng.integers(1, 5))      # (a few: the tag has to fit the wave's 32 KiB of staging)
                ops.append((k, "D"))
                md += str(run) + "^" + "".join("ACGT"[int(x)] for x in rng.integers(0, 4, k)); run = 0
    if kind == "del_last":
        ops.append((3, "D")); md += str(run) + "^ACG"
    elif kind == "mm_last":
        md += str(max(run - 1, 0)) + "T" if run else "5T"
    elif kind == "zero_last":
        md += str(max(run - 1, 0)) + "T0" if run else "T0"
    else:
        md += str(run)
    if kind == "lower" and "^A" in md:
        md = md.replace("^A", "^a", 1)
    elif kind == "empty_del":
        k = md.find("^", len(md) // 2); md = md[:k + 1] + md[k + 1:].lstrip("ACGT") if k >= 0 else md + "^5"
    elif kind == "tail":
        md += "AC"
    elif kind == "digits":
        k = len(md) // 2
        while k < len(md) and not (md[k].isdigit() and not md[k - 1].isdigit()):
            k += 1
        md = md[:k] + "0000000000" + md[k:]           # (the number that starts here, with ten zeros in front: the same value in more than nine digits)
    elif kind == "over":
        md += "A999"
    elif kind == "short":
        md = md[:len(md) // 3]
    elif kind == "lead_zero":
        md = "0" + md
    return bamio.pack_record(name, "".join(seq), ops, ref_id=ref, pos=pos, tags=bamio.tag_z("XA", "q" * int(rng.integers(0, 9))) + bamio.tag_z("MD", md))


@pytest.mark.parametrize("kind", ["plain", "big_match", "long_dels", "del_last", "mm_last", "zero_last", "lower", "eq_n", "empty_del", "tail", "digits", "over", "short", "lead_zero"])
def test_reference_base_long_chunks_token_shapes(ctx, kind, capfd, monkeypatch):
    """k6_ck_build seeds a long chunk's checkpoints from prefix sums over the tag's tokens instead of walking there: every shape of token at
    a checkpoint (inside a number, inside a deletion, the last token -- the backward parse's --, zero matches around it) and every kind of
    tag the wave does not vouch for (those take the serial walk) against the oracle, which parses as md/parse.d does."""
    rng = np.random.default_rng(7100 + sum(map(ord, kind)))
    recs, pos = [], 0
    for i in range(12):
        length = int(rng.integers(5000, 16000)) if i % 4 != 3 else int(rng.integers(30, 200))
        recs.append(_long_md_read_odd(rng, f"T{i}", 0, pos, length, kind if i % 3 != 1 else "plain"))
        pos += int(rng.integers(200, 6000))
    st = b"".join(recs)
    stage(ctx, st)
    monkeypatch.setenv("BGP_VERBOSE", "1")            # (the library then says how many chunks took the serial walk)
    capfd.readouterr()
    ctx.timing_enable(True); ctx.timing_reset()
    try:
        for skip in (True, False):
            _check_refbase(ctx, st, skip_zero=skip, window=int(rng.integers(500, 9000)))
        assert ctx.timing_get().get("k6_emit_long", {}).get("launches", 0) > 0
    finally:
        ctx.timing_enable(False)
    import re
    said = [(int(a), int(b)) for a, b in re.findall(r"MD checkpoints: (\d+) long chunk\(s\), \d+ piece\(s\), (\d+) chunk\(s\) walked serially", capfd.readouterr().err)]
    assert said and all(n > 0 for n, _ in said)
    vouched = kind in ("plain", "big_match", "long_dels", "del_last", "mm_last", "zero_last", "eq_n", "over", "short", "lead_zero")      # ("lower" holds one "^a": an empty deletion, md/parse.d:89-93)
    if vouched:
        assert all(ser == 0 for _, ser in said), said      # the prefix sums took every chunk
    else:
        assert any(ser > 0 for _, ser in said), said       # an empty deletion, trailing letters, a number of more than nine digits: the walk


# ---- the plan / accumulate path of K4 (k4_plan / k4_acc): batches, map overflow, fallbacks; framing repairs; full-size properties ----

@pytest.mark.parametrize("n_reads,ref_len,profile", [(8000, 12_000, 1), (5000, 6_000, 0), (14_000, 12_000, 1)])
def test_accumulate_path_deep_windows(ctx, n_reads, ref_len, profile):
    """Coverage of 100-170: a 1024-position window holds 500-1000 descriptors (a dozen batches a window, every wave
    drawing several; counters well beyond a byte); at 170-fold some windows outgrow their slab and go to the general
    kernel, next to windows that do not.  Columns == oracle everywhere."""
    cfg = make_cfg(n_reads=n_reads, n_ref=1, ref_len=ref_len, read_len=150, profile=profile, seed=77 + n_reads)
    st = host_stream(cfg)
    stage(ctx, st)
    info, ora = check_pileup(ctx, st, entries=False)
    assert info.max_coverage > 100


def test_accumulate_path_short_reads_next_to_long_reads(ctx):
    """Short reads (SEG descriptors) and long indel-heavy reads (ROW descriptors, in pieces of 256 positions) on one
    reference, in the same windows and the same batches of k4_acc."""
    short = make_cfg(n_reads=6000, n_ref=1, ref_len=60_000, read_len=150, profile=1, seed=3)
    recs = O.frame(host_stream(short))
    long_cfg = make_cfg(n_reads=12, n_ref=1, ref_len=60_000, read_len=(2000, 6000), profile=2, seed=4)
    st_long = host_stream(long_cfg)
    st_short = host_stream(short)
    lrec = O.frame(st_long)
    # merge by position (both streams are sorted): (pos, bytes)
    def items(st, fr):
        soa = O.decode_soa(st)
        return [(int(soa.pos[i]), st[int(fr[i][0]):int(fr[i][0]) + int(fr[i][1])]) for i in range(len(fr))]
    def spans(st):
        off = O.decode_soa(st).rec_off
        ends = list(off[1:]) + [len(st)]
        return [(int(o), int(e - o)) for o, e in zip(off, ends)]
    merged = sorted(items(st_short, spans(st_short)) + items(st_long, spans(st_long)), key=lambda x: x[0])
    st = b"".join(b for _, b in merged)
    stage(ctx, st)
    check_decode(ctx, st)
    check_pileup(ctx, st, entries=False)


def test_accumulate_path_long_segments_overflow_the_chunk_map(ctx):
    """250-base reads at 90-fold coverage: 64 descriptors of 32-33 chunks each do not fit a wave's chunk map, the batch
    is worked off in halves / quarters of its lanes; with soft clips so that segments start on odd nibbles."""
    rng = np.random.default_rng(9)
    recs, pos = [], 5
    for i in range(1800):
        n = int(rng.integers(236, 256)); sc = int(rng.integers(0, 4))
        seq = "".join("ACGTN"[int(x)] for x in rng.integers(0, 5, n))
        cig = ([(sc, "S")] if sc else []) + [(n - sc, "M")]
        recs.append(bamio.pack_record("q%d" % i, seq, cig, ref_id=0, pos=pos, qual=bytes(int(x) for x in rng.integers(2, 40, n))))
        pos += int(rng.integers(0, 6))
    st = b"".join(recs)
    stage(ctx, st)
    info, ora = check_pileup(ctx, st, entries=False)
    assert info.max_coverage > 80


def test_framing_with_a_wrong_reference_bound(ctx):
    """bgp_set_n_ref smaller than the ids the records carry: the boundary guesser rejects every true header, no
    segment but the first has an entry, and one chase carries the chain through all of them (instead of one
    verify round per segment).  Results do not depend on the hint."""
    cfg = make_cfg(n_reads=30_000, n_ref=6, ref_len=30_000, read_len=150, profile=1, seed=11)
    st = host_stream(cfg)
    stage(ctx, st, n_ref=1)
    check_decode(ctx, st)
    check_pileup(ctx, st, entries=False)


@pytest.mark.parametrize("decode_at_flush", [False, True])
def test_false_record_headers_inside_a_tag_are_repaired(ctx, decode_at_flush, monkeypatch, capfd):
    """Records whose aux data is a 70 kB B:C array made of complete, plausible BAM records: wherever a framing segment begins
    inside such an array the boundary guesser finds a "record" with a plausible successor and the segment is chased from there.
    k0_verify sees that the segment before ends elsewhere, lists the segment, and the repair round (k0_rechase: the listed
    segments from LDS) chases it again from the true boundary (readrange.d:118-173: the chain of block_size prefixes is the
    only truth).  Both framings: the flush's own segments, and the 64 KiB ones of the decode at flush (k12_lds, with records
    longer than a window)."""
    rng = np.random.default_rng(77)
    decoy = bamio.pack_record("decoy", "ACGTACGTAC", [(10, "M")], ref_id=0, pos=5, qual=bytes([30] * 10))
    payload = decoy * (70_000 // len(decoy))
    big_tag = b"XB" + b"BC" + struct.pack("<I", len(payload)) + payload
    recs, pos = [], 100
    for i in range(2400):
        n = int(rng.integers(80, 160))
        seq = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, n))
        tags = big_tag if i % 400 == 200 else bamio.tag_z("MD", str(n))
        recs.append(bamio.pack_record("r%05d" % i, seq, [(n, "M")], ref_id=0, pos=pos, qual=bytes(int(x) for x in rng.integers(2, 40, n)), tags=tags))
        pos += int(rng.integers(1, 9))
    st = b"".join(recs)
    monkeypatch.setenv("BGP_VERBOSE", "1")
    ctx.reset(); ctx.set_n_ref(1); ctx.push_inflated(st)
    ctx.flush(decode=L.BGP_FETCH_ALL) if decode_at_flush else ctx.flush()
    err = capfd.readouterr().err
    rounds = [int(x) for x in re.findall(r"framing round \d+: (\d+) of", err)]
    assert rounds and rounds[0] > 0 and rounds[-1] == 0, err            # something was repaired, and the last round found nothing left
    check_decode(ctx, st)
    check_pileup(ctx, st, entries=False)


def test_cfg5_read_lengths_columns_and_entries(ctx):
    """BASELINE config 5 at its own read lengths (10-50 kb, an indel every ~15 bases): 48 reads, every column against
    the oracle, per-read entries on sampled windows."""
    cfg = scaled("cfg5", 48)
    assert cfg.len_min == 10_000 and cfg.len_max == 50_000
    st = host_stream(cfg)
    stage(ctx, st)
    check_decode(ctx, st)
    info, _ = check_pileup(ctx, st, entries=False)
    ora = O.pileup(st, want_entries=True)
    k, seen = 0, 0
    for wi, w in enumerate(ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, want_entries=True, window_positions=1500)):
        nz = np.nonzero(w.columns["coverage"])[0]
        if wi % 23 == 0:
            for jj, j in enumerate(nz):
                e = w.entries[int(w.entry_off[j]):int(w.entry_off[j + 1])]
                oe = ora.entries[int(ora.entry_off[k + jj]):int(ora.entry_off[k + jj + 1])]
                assert np.array_equal(e["read_ordinal"], oe["read_ordinal"].astype(np.uint32))
                assert np.array_equal(e["cigar_op_index"], oe["cigar_op_index"].astype(np.uint16)) and np.array_equal(e["cigar_op_offset"], oe["cigar_op_offset"])
                assert np.array_equal(e["query_offset"], oe["query_offset"]) and np.array_equal(e["base"], oe["base"]) and np.array_equal(e["qual"], oe["qual"])
                seen += 1
        k += len(nz)
    assert seen > 5000


@pytest.mark.parametrize("name", ["cfg2", "cfg3", "cfg5"])
def test_full_size_properties(ctx, name):
    """The BASELINE configurations at their FULL sizes (10 M, 100 M, 200 k reads; the stream is generated in HBM):
    identities of the result checked on the device -- sum n_starting_here = reads used, sum coverage = sum
    basesCovered of the record table = the header's total, coverage = sum of classes at every position, the depth
    histogram adds up -- and a prefix shard against the oracle."""
    import torch
    from biod_amd.checks import full_size_invariants
    from biod_amd.synth import CONFIGS
    n = CONFIGS[name]["n_reads"]
    free, _ = torch.cuda.mem_get_info(0)
    if n * (1000 if name != "cfg5" else 600_000) > free * 0.9:
        pytest.skip("not enough free HBM for the full configuration")
    cfg = scaled(name, n)
    ctx.reset()
    ctx.synth_device(cfg)
    ctx.flush()
    info = ctx.pileup_run()
    res = full_size_invariants(torch, torch.device("cuda", 0), ctx, info, n)
    assert res["ok"], res
    assert res["reads_used"] == n
    k = 20_000 if name != "cfg5" else 40
    st = host_stream(cfg, 0, k)
    stage(ctx, st)
    check_pileup(ctx, st, entries=False)


# ---- a14: pileupChunks (splitter.d:35-101, pileup.d:859-1015) ----------------------------------------------------------

def _chunk_columns(chunks, want_refbase=False):
    """Columns of the chunks, in chunk order: (ref, pos, coverage, n_starting_here, six counts[, reference_base])."""
    ref, pos, cols, rb = [], [], [], []
    for ch in chunks:
        for w in ch.windows():
            nz = np.nonzero(w.columns["coverage"])[0]
            ref.append(np.full(len(nz), w.ref_id, np.int64)); pos.append(w.first_position + nz.astype(np.int64)); cols.append(w.columns[nz])
            if want_refbase:
                rb.append(w.reference_base[nz])
            assert ch.start_position <= w.first_position and w.first_position + len(w.columns) <= ch.end_position
    cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
    return cat(ref, np.int64), cat(pos, np.int64), cat(cols, L.COLUMN_DTYPE), (cat(rb, np.uint8) if want_refbase else None)


def _assert_columns_equal_oracle(ref, pos, cols, ora):
    assert len(pos) == ora.n_columns
    assert np.array_equal(ref, ora.ref.astype(np.int64)) and np.array_equal(pos, ora.pos.astype(np.int64))
    assert np.array_equal(cols["coverage"], ora.cov) and np.array_equal(cols["n_starting_here"], ora.nstart)
    for j, f in enumerate(("a", "c", "g", "t", "other", "del")):
        assert np.array_equal(cols[f], ora.counts[:, j]), f


@pytest.mark.parametrize("block", [20_000, 150_000, 16_384_000])
def test_pileup_chunks_plan_and_concatenation(ctx, block):
    """Chunks cut and seamed as BioD does (against the oracle's plan), their columns concatenated == pileupColumns;
    three references, reads without coverage and unmapped reads (ref_id -1) at the end."""
    from biod_amd.pileup import pileupChunks
    cfg = make_cfg(n_reads=9000, n_ref=3, ref_len=40_000, read_len=150, profile=1, seed=31)
    tail = [bamio.pack_record(f"u{k}", "ACGTACGT", [], ref_id=-1, pos=-1, flag=4) for k in range(40)]
    st = host_stream(cfg) + b"".join(tail)
    reads = RecordStream(st, ctx)
    chunks = pileupChunks(reads, False, block, want_entries=False)
    plan = O.chunk_plan(st, block)
    assert [(c.first_record, c.last_record, c.ref_id, c.start_position, c.end_position) for c in chunks] == plan
    if block < 16_384_000:
        assert len(chunks) > 6
    ref, pos, cols, _ = _chunk_columns(chunks)
    _assert_columns_equal_oracle(ref, pos, cols, O.pileup(st))
    # any order, any number of times
    k = len(chunks) // 2
    a = _chunk_columns([chunks[k]]); b = _chunk_columns([chunks[0]]); c2 = _chunk_columns([chunks[k]])
    assert np.array_equal(a[1], c2[1]) and np.array_equal(a[2], c2[2]) and (k == 0 or not np.array_equal(a[1], b[1]))


def test_pileup_chunks_leading_chunks_that_cover_nothing(ctx):
    """pileup.d:894-896: the constructor never hands out leading chunks whose right end is not beyond start_from -- reads that cover
    nothing at position 0 --, and the first chunk kept has no chunk in front of it; later such chunks are handed out (:915-939).
    The lists are the reference's rule worked by hand (tests/test_oracle_kats.py holds the same for the oracle)."""
    from biod_amd.pileup import pileupChunks
    def plan_of(st, block):
        return [(c.first_record, c.last_record, c.ref_id, c.start_position, c.end_position) for c in pileupChunks(RecordStream(st, ctx), False, block, want_entries=False)]
    u = lambda ref: bamio.pack_record("u0", "ACGT", [(4, "S")], ref_id=ref, pos=0)
    m = lambda name, ref, pos: bamio.pack_record(name, "ACGTACGT", [(8, "M")], ref_id=ref, pos=pos)
    st = u(0) + m("r1", 1, 6) + m("r2", 1, 10)
    assert plan_of(st, 16384000) == [(1, 2, 1, 6, 18)] == O.chunk_plan(st, 16384000)
    st = u(1) + m("r1", 1, 6) + m("r2", 1, 10)
    small = len(u(1)) - 1
    assert plan_of(st, small) == [(1, 1, 1, 6, 6), (2, 2, 1, 6, 18)] == O.chunk_plan(st, small)
    st = m("r1", 0, 6) + u(1) + m("r2", 2, 10)
    assert plan_of(st, 16384000) == [(0, 0, 0, 6, 14), (1, 1, 1, 0, 0), (2, 2, 2, 10, 18)] == O.chunk_plan(st, 16384000)
    for st_, block in ((u(0) + m("r1", 1, 6) + m("r2", 1, 10), 16384000), (u(1) + m("r1", 1, 6) + m("r2", 1, 10), small)):
        chunks = pileupChunks(RecordStream(st_, ctx), False, block, want_entries=False)
        ref, pos, cols, _ = _chunk_columns(chunks)
        _assert_columns_equal_oracle(ref, pos, cols, O.pileup(st_))


@pytest.mark.parametrize("name", ["ex1_header.bam", "b7_295_chunk.bam", "tags.bam"])
def test_pileup_chunks_fixtures(ctx, data_dir, name):
    from biod_amd.pileup import pileupChunks
    _, st = bamio.read_bam(os.path.join(data_dir, name))
    for block in (30_000, 16_384_000):
        chunks = pileupChunks(RecordStream(st, ctx), False, block, want_entries=False)
        assert [(c.first_record, c.last_record, c.ref_id, c.start_position, c.end_position) for c in chunks] == O.chunk_plan(st, block)
        ref, pos, cols, _ = _chunk_columns(chunks)
        _assert_columns_equal_oracle(ref, pos, cols, O.pileup(st))


@pytest.mark.parametrize("seed", range(6))
def test_pileup_chunks_of_a_read_or_two(ctx, seed):
    """pileupChunks on random CIGARs with block sizes down to a read or two a chunk (scratch/fuzz_chunks.py's rounds): the plan is the
    oracle's, the chunks' columns concatenated are pileupColumns."""
    from biod_amd.pileup import pileupChunks
    rng = np.random.default_rng(5100 + seed)
    st = _random_stream(rng, int(rng.integers(50, 900)), quirks=False)
    try:
        ora = O.pileup(st)
    except O.OracleError:
        pytest.skip("the oracle rejects this stream (ECIGAR)")
    for block in (int(rng.integers(150, 900)), int(rng.integers(900, 20_000))):
        chunks = pileupChunks(RecordStream(st, ctx), False, block, want_entries=False)
        assert [(c.first_record, c.last_record, c.ref_id, c.start_position, c.end_position) for c in chunks] == O.chunk_plan(st, block)
        ref, pos, cols, _ = _chunk_columns(chunks)
        _assert_columns_equal_oracle(ref, pos, cols, ora)


def test_pileup_chunks_clipping_and_reference_bases(ctx):
    """start_from / end_at: the constructor drops leading chunks, later chunks are clipped (pileup.d:878-913); on one
    reference the concatenation is makePileup(reads, use_md, start_from, end_at).  With use_md_tag every column carries
    the reference base (consistent MD tags: any provider gives the same base)."""
    from biod_amd.pileup import pileupChunks
    cfg = make_cfg(n_reads=6000, n_ref=1, ref_len=60_000, read_len=150, profile=0, seed=8)
    st = host_stream(cfg)
    for s, e in ((0, L_ULONG_MAX), (20_000, 41_234), (59_000, L_ULONG_MAX), (100, 101), (70_000, 80_000)):
        chunks = pileupChunks(RecordStream(st, ctx), True, 40_000, s, e, want_entries=False)
        ref, pos, cols, rb = _chunk_columns(chunks, want_refbase=True)
        ora = O.pileup(st, mode=O.MODE_MAKE_PILEUP, use_md=True, start_from=s, end_at=e)
        _assert_columns_equal_oracle(ref, pos, cols, ora)
        assert np.array_equal(rb, ora.refbase)
        for c in chunks:
            assert c.start_position >= s and c.end_position <= e
    assert pileupChunks(RecordStream(b"", ctx), False, want_entries=False) == []          # (BioD's constructor would spin here, pileup.d:878-900)


def test_more_reference_runs_than_the_first_run_table(ctx):
    """70 000 references with one read each (an assembly of many small contigs): the run table K3 fills starts at
    65 536 entries and is grown."""
    recs = [bamio.pack_record(f"r{k}", "ACGTACGTAC", [(10, "M")], ref_id=k, pos=k % 50) for k in range(70_000)]
    st = b"".join(recs)
    stage(ctx, st)
    info, ora = check_pileup(ctx, st, entries=False)
    assert len(info.runs) == 70_000 and info.n_columns == 700_000


# ---- batched staging with carry-over (bgp_batch_*): a stream in several batches == the stream at once == the oracle ----

def _covered(windows, want_entries=False):
    ref, pos, cols, ents, npos = [], [], [], [], 0
    for w in windows:
        nz = np.nonzero(w.columns["coverage"])[0]
        ref.append(np.full(len(nz), w.ref_id, np.int64)); pos.append(w.first_position + nz.astype(np.int64)); cols.append(w.columns[nz])
        npos += len(w.columns)
        if want_entries:
            for j in nz:
                ents.append(w.entries[int(w.entry_off[j]):int(w.entry_off[j + 1])])
    cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
    return cat(ref, np.int64), cat(pos, np.int64), cat(cols, L.COLUMN_DTYPE), ents, npos


def _pieces(st, size):
    return [st[o:o + size] for o in range(0, len(st), size)] or [b""]


@pytest.mark.parametrize("name,batch", [("ex1_header.bam", 150_000), ("b7_295_chunk.bam", 40_000), ("bins.bam", 30_000)])
def test_batched_staging_fixtures(ctx, data_dir, name, batch):
    _, st = bamio.read_bam(os.path.join(data_dir, name))
    assert len(st) > 3 * batch
    ref, pos, cols, _, npos = _covered(ctx.batched_windows(_pieces(st, 7777), batch))
    ora = O.pileup(st)
    if name != "bins.bam":
        _assert_columns_equal_oracle(ref, pos, cols, ora)
    else:                                               # empty SEQ: coverage only (test_bins_bam_coverage_parity)
        assert np.array_equal(pos, ora.pos.astype(np.int64)) and np.array_equal(cols["coverage"], ora.cov) and np.array_equal(cols["n_starting_here"], ora.nstart)
    stage(ctx, st)
    assert npos == ctx.pileup_run().n_positions        # the zero-coverage positions inside the runs are handed out too, once


@pytest.mark.parametrize("name,batch,piece", [("ex1_header.bam", 150_000, 7777), ("b7_295_chunk.bam", 70_000, 100_000)])
def test_batched_staging_zero_copy(ctx, data_dir, name, batch, piece):
    """bgp_batch_acquire / _commit: the caller writes into the pinned buffer the next copy to HBM starts from (the shape of
    decompressBgzfBlock writing into the block's own buffer, block.d:127-216).  Same columns as the pushes' and the oracle's; and
    between an acquire and its commit the buffer is the caller's -- a push, a second acquire, the end are refused."""
    _, st = bamio.read_bam(os.path.join(data_dir, name))
    assert len(st) > 2 * batch
    ref, pos, cols, _, npos = _covered(ctx.batched_windows_zero_copy(_pieces(st, piece), batch))
    _assert_columns_equal_oracle(ref, pos, cols, O.pileup(st))
    ref2, pos2, cols2, _, npos2 = _covered(ctx.batched_windows(_pieces(st, piece), batch))
    assert npos == npos2 and np.array_equal(pos, pos2) and np.array_equal(cols, cols2)
    # the state rules
    ctx.batch_begin(batch)
    view, rc = ctx.batch_acquire()
    assert rc == L.BGP_OK and len(view) == batch
    for call in (lambda: ctx.batch_push(st[:100]), ctx.batch_acquire, ctx.batch_end):
        with pytest.raises(L.BiodGpuError) as e:
            call()
        assert e.value.code == L.BGP_ESTATE
    ctx.batch_commit(0)
    with pytest.raises(L.BiodGpuError) as e:              # nothing is acquired any more
        ctx.batch_commit(0)
    assert e.value.code == L.BGP_ESTATE
    view, rc = ctx.batch_acquire()
    with pytest.raises(L.BiodGpuError) as e:              # more than was handed out
        ctx.batch_commit(len(view) + 1)
    assert e.value.code == L.BGP_EINVAL
    ctx.batch_commit(0)
    ctx.batch_end()
    assert list(ctx.batch_drain()) == []
    ctx.reset()


def test_batched_staging_200k_reads_in_batches(ctx):
    """200 000 reads of cfg3's model (58 MB) in 16 MiB batches: columns == the stream staged at once, and == the oracle
    on a prefix; cut points of the pushes fall anywhere, also inside records."""
    cfg = scaled("cfg3", 200_000)
    st = host_stream_np(cfg)
    batch = 16 << 20
    assert len(st) > 3 * batch
    ref, pos, cols, _, npos = _covered(ctx.batched_windows(_pieces(st, 5_000_003), batch))
    ctx.reset(); ctx.push_inflated(st.tobytes()); ctx.flush()
    info = ctx.pileup_run()
    dense = device_columns(ctx, info)
    assert npos == info.n_positions == len(dense)
    nz = np.nonzero(dense["coverage"])[0]
    assert len(nz) == len(pos) == info.n_columns
    for f in L.COLUMN_DTYPE.names:
        assert np.array_equal(cols[f], dense[f][nz]), f
    k = 30_000
    pre = host_stream(cfg, 0, k)
    ora = O.pileup(pre)
    last = int(ora.pos[-1]) - 400                       # columns the rest of the stream cannot touch
    m = (ora.ref == 0) & (ora.pos < last)
    assert m.sum() > 1000 and np.array_equal(cols["coverage"][:m.sum()], ora.cov[m]) and np.array_equal(pos[:m.sum()], ora.pos[m].astype(np.int64))


@pytest.mark.parametrize("batch", [4096, 9000, 70_000])
def test_batched_staging_entries_and_small_batches(ctx, batch):
    """Batches of a few dozen records, N-skip reads reaching across several of them, per-read entries with stream-wide
    ordinals; a batch size the stream length is a multiple of (the stream ends exactly on a batch)."""
    cfg = make_cfg(n_reads=1500, n_ref=2, ref_len=9_000, read_len=100, profile=1, seed=13)
    st = host_stream(cfg)
    ora = O.pileup(st, want_entries=True)
    ref, pos, cols, ents, _ = _covered(ctx.batched_windows(_pieces(st, 1013), batch, want_entries=True, window_positions=700), want_entries=True)
    _assert_columns_equal_oracle(ref, pos, cols, ora)
    assert len(ents) == ora.n_columns
    for k in range(0, ora.n_columns, 7):
        oe = ora.entries[int(ora.entry_off[k]):int(ora.entry_off[k + 1])]
        assert np.array_equal(ents[k]["read_ordinal"], oe["read_ordinal"].astype(np.uint32)), k
        assert np.array_equal(ents[k]["query_offset"], oe["query_offset"]) and np.array_equal(ents[k]["base"], oe["base"]) and np.array_equal(ents[k]["qual"], oe["qual"])


def test_batched_staging_edges(ctx):
    """Empty stream; a stream of exactly two batches; records larger than a batch (the carry outgrows its reserve);
    reads without coverage and unmapped reads at the end; a truncated last record is an error, as it is at once."""
    assert list(ctx.batched_windows([b""], 4096)) == []
    cfg = make_cfg(n_reads=400, n_ref=1, ref_len=5_000, read_len=150, profile=0, seed=2)
    st = host_stream(cfg)
    ref, pos, cols, _, _ = _covered(ctx.batched_windows([st], max(4096, -(-len(st) // 2))))      # exactly two batches: the stream ends on a batch
    _assert_columns_equal_oracle(ref, pos, cols, O.pileup(st))
    rng = np.random.default_rng(5)
    big = [bamio.pack_record(f"L{k}", "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 90_000)), [(90_000, "M")], ref_id=0, pos=10 + 500 * k) for k in range(4)]
    tail = [bamio.pack_record("nocov", "ACGT", [(4, "S")], ref_id=0, pos=2600), bamio.pack_record("u", "ACGT", [], ref_id=-1, pos=-1, flag=4)]
    st2 = b"".join(big + tail)
    ref, pos, cols, _, _ = _covered(ctx.batched_windows(_pieces(st2, 30_011), 65_536))
    _assert_columns_equal_oracle(ref, pos, cols, O.pileup(st2))
    with pytest.raises(L.BiodGpuError) as e:
        list(ctx.batched_windows([st[:-7]], 20_000))
    assert e.value.code == L.BGP_ETRUNC
    # a corrupt block_size in the MIDDLE of the stream is refused in the batch it appears in (the one-shot path says ETRUNC /
    # EINVAL at once): it is not "a record whose rest is still to come", carried and re-framed batch after batch
    import struct
    offs = O.frame(st)
    n_ok = len(list(ctx.batched_windows([st], 8192, window_positions=64)))
    for bad in (0x7FFFFFF0, 40):                                         # beyond any record | smaller than the sections its header announces
        dmg = bytearray(st); dmg[int(offs[60]):int(offs[60]) + 4] = struct.pack("<i", bad)
        seen = 0
        with pytest.raises(L.BiodGpuError) as e:
            for w in ctx.batched_windows([bytes(dmg)], 8192, window_positions=64):
                seen += 1
        assert e.value.code == L.BGP_EINVAL and seen < n_ok // 2          # (record 60 of 400 lies in the second of ~15 batches)
    # bgp_set_n_ref before bgp_batch_begin survives the reset inside it (the guesser keeps its bound): same columns
    ctx.reset(); ctx.set_n_ref(1)
    cov = np.concatenate([w.columns["coverage"] for w in ctx.batched_windows([st], 8192)])
    assert np.array_equal(cov[cov > 0], O.pileup(st).cov)


def _batched_refbase(ctx, pieces, batch, skip):
    ref, pos, rb = [], [], []
    for w in ctx.batched_windows(pieces, batch, window_positions=3000, use_md_tag=True, skip_zero_coverage=skip):
        nz = np.nonzero(w.columns["coverage"])[0]
        ref.append(np.full(len(nz), w.ref_id, np.int64)); pos.append(w.first_position + nz.astype(np.int64)); rb.append(w.reference_base[nz])
    cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
    return cat(ref, np.int64), cat(pos, np.int64), cat(rb, np.uint8)


def _long_indel_read(rng, i, pos, n_ops, damage=None):
    """A long read (M / I / D / N ops, a mismatch every ~20 bases) with the MD tag its CIGAR implies, optionally damaged."""
    ops, md, run, seq = [], "", 0, []
    for k in range(n_ops):
        ln = int(rng.integers(8, 40))
        ops.append((ln, "M"))
        for _ in range(ln):
            if rng.random() < 0.05:
                md += str(run) + "ACGT"[int(rng.integers(0, 4))]; run = 0
            else:
                run += 1
        seq.append(ln)
        u = rng.random()
        if k + 1 < n_ops:
            if u < 0.3:
                il = int(rng.integers(1, 6)); ops.append((il, "I")); seq.append(il)
            elif u < 0.6:
                dlen = int(rng.integers(1, 8)); ops.append((dlen, "D"))
                md += str(run) + "^" + "".join("ACGT"[int(x)] for x in rng.integers(0, 4, dlen)); run = 0
            elif u < 0.65:
                ops.append((int(rng.integers(10, 60)), "N"))
    md += str(run)
    if damage == "short":
        md = md[:len(md) // 2]
    elif damage == "long":
        md += "A" + str(int(rng.integers(5, 50)))
    elif damage == "empty_del":
        k = md.find("^"); md = md[:k + 1] + md[k + 1:].lstrip("ACGT") if k >= 0 else md + "^5"
    elif damage == "lower":
        md = md.replace("^A", "^a", 1) if "^A" in md else md.lower()
    elif damage == "tail":
        md += "AC"
    elif damage == "digits":
        md = "00000000001" + md
    elif damage == "none":
        return bamio.pack_record(f"l{i}", "".join("ACGT"[int(x)] for x in rng.integers(0, 4, sum(seq))), ops, ref_id=0, pos=pos, tags=bamio.tag_z("XQ", "nomd"))
    return bamio.pack_record(f"l{i}", "".join("ACGT"[int(x)] for x in rng.integers(0, 4, sum(seq))), ops, ref_id=0, pos=pos,
                             tags=bamio.tag_z("XA", "q" * int(rng.integers(0, 40))) + bamio.tag_z("MD", md))


@pytest.mark.parametrize("damage", [None, "short", "long", "empty_del", "lower", "tail", "digits", "none"])
def test_reference_base_long_reads_checked_by_the_wave(ctx, damage):
    """Providers of more than 2048 bases have their tags checked by k6_verify_long (tokens found in parallel, sums instead of a walk): with
    consistent tags the composed chain stands, and every kind of damage -- a tag cut short, one that asks for more bases than the CIGAR has,
    an empty deletion, lower-case deleted bases, trailing letters, a number of eleven digits, no MD at all -- must come out as the
    reference's own parser has it (the oracle): the check flags what it cannot vouch for and the event loop redoes the run."""
    rng = np.random.default_rng(991)
    recs, pos = [], 100
    for i in range(60):
        recs.append(_long_indel_read(rng, i, pos, int(rng.integers(110, 160)), damage if (damage and i % 7 == 3) else None))
        pos += int(rng.integers(50, 400))
    st = b"".join(recs)
    assert len(st) // 60 >= 2048                      # (the library lists providers for the wave's check on streams of long records)
    stage(ctx, st)
    for skip in (True, False):
        _check_refbase(ctx, st, skip_zero=skip, window=5000)


def test_reference_base_every_read_a_provider(ctx):
    """Short reads at coverage ~1.3: nearly every read of a 2048-record segment is a provider, so k6_emit hands a segment's providers
    out in several stretches (its list in LDS holds 192); mismatches and deletions in the tags so that the bases differ by provider."""
    rng = np.random.default_rng(4242)
    recs, pos = [], 50
    for i in range(5000):
        n = int(rng.integers(8, 14))
        seq = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, n))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            ops, md = [(n, "M")], str(n)
        elif kind == 1:
            k = int(rng.integers(0, n)); ops, md = [(n, "M")], f"{k}{'ACGT'[int(rng.integers(0, 4))]}{n - k - 1}"
        elif kind == 2:
            k = int(rng.integers(1, n)); ops, md = [(k, "M"), (2, "D"), (n - k, "M")], f"{k}^{'ACGT'[int(rng.integers(0, 4))]}{'ACGT'[int(rng.integers(0, 4))]}{n - k}"
        else:
            k = int(rng.integers(1, n - 1)); ops, md = [(k, "M"), (1, "I"), (n - k - 1, "M")], str(n - 1)
        recs.append(bamio.pack_record(f"p{i}", seq, ops, ref_id=0, pos=pos, tags=bamio.tag_z("MD", md)))
        pos += int(rng.integers(5, 11))
    st = b"".join(recs)
    stage(ctx, st)
    for skip in (True, False):
        _check_refbase(ctx, st, skip_zero=skip, window=4000)


@pytest.mark.parametrize("seed", range(8))
def test_batched_reference_base_batches_of_a_few_reads(ctx, seed):
    """use_md_tag across batches of ONE TO TEN reads (bgp_batch_acquire ships whatever a small buffer holds): a batch may hold no event of
    the provider chain in front of its cut -- the chain then stands where it stood and the next batch resumes from the same state
    (until round 4 it was begun afresh there: wrong bases wherever a provider outlived such a batch; found by scratch/fuzz_tiny_batches.py)."""
    rng = np.random.default_rng(7130 + seed)
    st = _md_stream(rng, int(rng.integers(60, 500)), **_MD_CFGS[seed % len(_MD_CFGS)])
    for skip in (True, False):
        try:
            ora = O.pileup(st, use_md=True, skip_zero=skip)
        except O.OracleError:
            continue
        pos, rb = [], []
        for w in ctx.batched_windows_zero_copy(_pieces(st, int(rng.integers(200, 1500))), 4096, window_positions=3000, use_md_tag=True, skip_zero_coverage=skip):
            nz = np.nonzero(w.columns["coverage"])[0]
            pos.append(w.first_position + nz.astype(np.int64)); rb.append(w.reference_base[nz])
        keep = ora.cov > 0
        assert np.array_equal(np.concatenate(pos), ora.pos[keep].astype(np.int64)) and np.array_equal(np.concatenate(rb), ora.refbase[keep]), skip


def _cut_stream(with_winner):
    """r0 covers [0, 51): its chunk is used up at column 50, where the next provider is fetched from the reads that start at or before 50
    and end beyond r0 (pileup.d:583-596, 627-652): D (position 30, an MD tag of five bases: twenty columns in it has nothing left --
    BGP_EMD if it is ever the provider), H (position 50, ends before D) and -- with_winner -- G (position 50, the largest end_position: it wins).
    r0's padding puts the end of a 4096-byte batch between H and G: the batch's cut is at position 50, and it holds H but not G."""
    def rec(name, pos, n, md, pad=0):
        seq = "".join("ACGT"[(7 * i + pos) % 4] for i in range(n))
        return bamio.pack_record(name, seq, [(n, "M")], ref_id=0, pos=pos, tags=bamio.tag_z("MD", md) + (bamio.tag_z("XP", "p" * pad) if pad else b""))
    recs = [rec("r0", 0, 51, "51", pad=3760), rec("D", 30, 30, "5"), rec("H", 50, 5, "5")]
    if with_winner:
        recs.append(rec("G", 50, 60, "60"))
    recs += [rec("t%d" % i, 60 + 20 * i, 40, "40", pad=300) for i in range(12)]
    st = b"".join(recs)
    soa = O.decode_soa(st)
    assert int(soa.rec_off[3]) < 4096 < int(soa.rec_off[3]) + 100      # the first batch ends inside the fourth record
    return st


def test_batched_reference_base_provider_fetched_at_the_cut(ctx):
    """Found by fuzzing (round 5, fuzz_gpu seed 6210148; the bug predates the round): a batch's cut lies at the position of its last
    read, and of the reads that START there the batch may hold only the first few.  A provider fetched AT that position is the best of
    an incomplete group: the batch must not raise on it (here: D's damaged tag, when G in the next batch wins) -- and must still raise,
    one batch later, when the error is real (the same stream without G: the oracle raises too)."""
    st = _cut_stream(True)
    ora = O.pileup(st, use_md=True, skip_zero=True)
    ref, pos, rb = _batched_refbase(ctx, _pieces(st, 1000), 4096, True)
    keep = ora.cov > 0
    assert np.array_equal(pos, ora.pos[keep].astype(np.int64)) and np.array_equal(rb, ora.refbase[keep])
    st = _cut_stream(False)
    with pytest.raises(O.OracleError) as oe:
        O.pileup(st, use_md=True, skip_zero=True)
    assert oe.value.code == O.ORC_EMD
    with pytest.raises(L.BiodGpuError) as ge:
        _batched_refbase(ctx, _pieces(st, 1000), 4096, True)
    assert ge.value.code == L.BGP_EMD
    ctx.reset()


@pytest.mark.parametrize("skip", [True, False])
@pytest.mark.parametrize("name,batch", [("illu_20_chunk.bam", 4096), ("b7_295_chunk.bam", 20_000), ("b7_295_chunk.bam", 70_000), ("ex1_header.bam", 30_000), ("tags.bam", 8192)])
def test_batched_staging_reference_base_fixtures(ctx, data_dir, name, batch, skip):
    """use_md_tag across batches (bgp_batch_set_md): the provider chain's state crosses every seam, reference_base of every
    covered column == PileupRangeUsingMdTag over the whole stream (the oracle), in batches of a few dozen reads."""
    _, st = bamio.read_bam(os.path.join(data_dir, name))
    try:
        ora = O.pileup(st, use_md=True, skip_zero=skip)
    except O.OracleError as e:
        assert e.code in (O.ORC_EMD, O.ORC_EINVAL)
        with pytest.raises(L.BiodGpuError):
            _batched_refbase(ctx, _pieces(st, 5000), batch, skip)
        return
    ref, pos, rb = _batched_refbase(ctx, _pieces(st, 5000), batch, skip)
    keep = ora.cov > 0
    assert np.array_equal(ref, ora.ref[keep].astype(np.int64)) and np.array_equal(pos, ora.pos[keep].astype(np.int64))
    bad = np.nonzero(rb != ora.refbase[keep])[0]
    assert len(bad) == 0, [(int(pos[k]), chr(rb[k]), chr(ora.refbase[keep][k])) for k in bad[:8]]


@pytest.mark.parametrize("profile,batch", [(0, 40_000), (0, 300_000), (1, 60_000), (2, 400_000)])
def test_batched_staging_reference_base_synthetic(ctx, profile, batch):
    """The three read models in many batches, both sweeps: M-only reads with consistent MD tags, the mixed model (N skips: BioD's
    chunk drifts off the reference there -- the same drift must come out), long indel-heavy reads (providers of tens of kb:
    the chain's state reaches far in front of the cut)."""
    cfg = make_cfg(n_reads=6000 if profile < 2 else 60, n_ref=2, ref_len=40_000 if profile < 2 else 300_000, read_len=150 if profile < 2 else (4000, 20_000), profile=profile, seed=70 + profile)
    st = host_stream(cfg)
    for skip in (True, False):
        try:
            ora = O.pileup(st, use_md=True, skip_zero=skip)
        except O.OracleError as e:
            assert e.code == O.ORC_EMD
            with pytest.raises(L.BiodGpuError):
                _batched_refbase(ctx, _pieces(st, 33_333), batch, skip)
            continue
        ref, pos, rb = _batched_refbase(ctx, _pieces(st, 33_333), batch, skip)
        keep = ora.cov > 0
        assert np.array_equal(pos, ora.pos[keep].astype(np.int64))
        bad = np.nonzero(rb != ora.refbase[keep])[0]
        assert len(bad) == 0, (skip, [(int(ref[k]), int(pos[k]), chr(rb[k]), chr(ora.refbase[keep][k])) for k in bad[:8]])


def test_interval_shards_with_exact_halo_on_the_device(ctx):
    """plan_interval_shards: three equal-read-count shards of a two-contig stream, each piled up on the
    device from its byte range (own reads + exact halo) and clipped; concatenation == oracle on the whole stream."""
    from biod_amd.shard import clip_columns, plan_interval_shards
    cfg = make_cfg(n_reads=6000, n_ref=2, ref_len=40_000, read_len=150, profile=1, seed=41)
    st = host_stream(cfg)
    soa = O.decode_soa(st)
    parts = []
    for sh in plan_interval_shards(soa.ref_id, soa.pos, soa.end_pos, soa.rec_off, len(st), 3):
        stage(ctx, st[sh["byte_lo"]:sh["byte_hi"]])
        assert sh["byte_lo"] == int(soa.rec_off[sh["first_record"]])
        ref, pos, cols, _, _ = _covered(ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, skip_zero_coverage=False, window_positions=1 << 20))
        keep = clip_columns(ref, pos, sh["lo"], sh["hi"])
        parts.append((ref[keep], pos[keep], cols[keep]))
    _assert_columns_equal_oracle(np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]), np.concatenate([p[2] for p in parts]), O.pileup(st))


def test_first_record_hints_right_wrong_and_wild(ctx, data_dir):
    """bgp_push_inflated's first_record_uoffset: blocks pushed with the true offset of the first record that starts in
    them, with offsets that are no record starts, and with offsets past the block -- the record table is the same."""
    _, st = bamio.read_bam(os.path.join(data_dir, "mg1655_chunk.bam"))
    st = st * 40                                       # several framing segments
    soa = O.decode_soa(st)
    offs = soa.rec_off.astype(np.int64)
    block = 0xFF00
    rng = np.random.default_rng(3)
    for mode in ("right", "wrong", "wild"):
        ctx.reset()
        for b0 in range(0, len(st), block):
            blk = st[b0:b0 + block]
            i = int(np.searchsorted(offs, b0))
            true_u = int(offs[i] - b0) if i < len(offs) and offs[i] < b0 + len(blk) else L.BGP_UOFFSET_UNKNOWN
            u = true_u if mode == "right" else int(rng.integers(0, len(blk))) if mode == "wrong" else int(rng.integers(len(blk), 1 << 20))
            ctx.push_inflated(blk, b0, u)
        ctx.flush()
        dec = ctx.decode_fetch(0)
        assert np.array_equal(dec.rec["rec_off"], soa.rec_off) and np.array_equal(dec.rec["end_pos"], soa.end_pos), mode


@pytest.mark.parametrize("m_len,n_long,n_short", [(100, 40, 0), (230, 30, 0), (700, 24, 0), (100, 6, 900), (37, 50, 0)])
def test_long_read_rows_chunk_shapes(ctx, m_len, n_long, n_short):
    """The long reads' pre-pass at the shapes its branches turn on: chunks of 64 ops with more than 64 pieces of 14
    columns (m_len 100: several rounds of the piece loop), chunks wider than the 4096-byte LDS row (m_len 230: written
    straight to the arena), ops wider than 512 columns (m_len 700: whole-wave copies), pieces that are not a multiple of
    14 and odd query offsets (37); all reads long (a wave per record) and a few long reads among short ones (a lane per
    record).  Columns against the oracle; entries on small windows."""
    rng = np.random.default_rng(1000 + m_len + n_short)
    ref_len = 120_000
    items = []
    for r in range(n_long):
        cig, seq_len, cols = [], 0, 0
        n_ops = int(rng.integers(150, 400))
        if rng.random() < 0.5:
            k = int(rng.integers(1, 30)); cig.append((k, "S")); seq_len += k
        for j in range(n_ops):
            k = max(1, int(rng.integers(m_len - m_len // 3, m_len + m_len // 3 + 1)))
            cig.append((k, "M=X"[int(rng.integers(0, 3))])); seq_len += k; cols += k
            t = rng.random()
            if j == n_ops - 1:
                break
            if t < 0.45:
                k = int(rng.integers(1, 4)); cig.append((k, "I")); seq_len += k
            elif t < 0.9:
                k = int(rng.integers(1, 4)); cig.append((k, "D")); cols += k
            elif t < 0.95:
                k = int(rng.integers(600, 3000)); cig.append((k, "N")); cols += k
        pos = int(rng.integers(0, 5000))
        seq = "".join(rng.choice(list("ACGTNRY"), size=seq_len, p=[0.24, 0.24, 0.24, 0.24, 0.02, 0.01, 0.01]))
        items.append((pos, bamio.pack_record(f"L{r}", seq, cig, ref_id=0, pos=pos, qual=bytes(rng.integers(0, 60, size=seq_len, dtype=np.uint8)))))
    for r in range(n_short):
        pos = int(rng.integers(0, 60_000))
        seq = "".join(rng.choice(list("ACGT"), size=100))
        items.append((pos, bamio.pack_record(f"s{r}", seq, [(60, "M"), (2, "D"), (40, "M")], ref_id=0, pos=pos)))
    items.sort(key=lambda x: x[0])
    st = b"".join(b for _, b in items)
    stage(ctx, st)
    check_decode(ctx, st)
    check_pileup(ctx, st, entries=(m_len == 37), window=700)


def test_python_windows_as_views(ctx, data_dir):
    """pileup_windows(copy=False): the arrays are views of the ctx's buffers, equal to the copies while their window is
    the current one."""
    _, st = bamio.read_bam(os.path.join(data_dir, "ex1_header.bam"))
    stage(ctx, st)
    ctx.pileup_run()
    copies = [(w.ref_id, int(w.first_position), w.columns.copy(), w.entry_off.copy(), w.entries.copy())
              for w in ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, want_entries=True, window_positions=300)]
    k = 0
    for w in ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, want_entries=True, window_positions=300, copy=False):
        r, p, c, eo, en = copies[k]
        assert (w.ref_id, int(w.first_position)) == (r, p)
        assert np.array_equal(w.columns, c) and np.array_equal(w.entry_off, eo) and np.array_equal(w.entries, en)
        k += 1
    assert k == len(copies) and k > 5


def test_views_outlive_the_windows_that_replace_their_buffers(ctx):
    """include/biodgpu.h: the memory behind a returned pointer stays mapped until bgp_reset / bgp_destroy -- a view kept too long reads
    stale bytes, never freed memory.  Views of small windows are kept while ever larger windows (columns and entries) make the ctx replace
    its pinned buffers several times over; every old view must still be readable and hold what it held."""
    cfg = make_cfg(n_reads=60000, n_ref=1, ref_len=120_000, read_len=100, profile=1, seed=91)
    st = host_stream(cfg)
    stage(ctx, st)
    ctx.pileup_run()
    kept = []
    for wp in (64, 700, 9000, 60_000, 120_000):            # (each window needs a buffer ~10 x the one before: five generations of pinned memory)
        gen = ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, want_entries=True, window_positions=wp, copy=False)
        w = next(gen)
        kept.append((w.columns, w.columns.copy(), w.entries, w.entries.copy()))
        gen.close()
        for view, was, ev, ewas in kept:                    # (the same bytes as when the view was taken, or stale ones of a later window of the same buffer: readable either way)
            assert view.shape == was.shape and ev.shape == ewas.shape
            _ = int(view["coverage"].sum()) + int(ev["read_ordinal"].sum())
    assert np.array_equal(kept[-1][0], kept[-1][1]) and np.array_equal(kept[-1][2], kept[-1][3])


def test_batched_staging_then_a_one_shot_load_on_the_same_ctx(ctx, data_dir):
    """The pinned and device buffers a batched pass parks in the ctx are given back when the ctx stages a stream another way; one-shot and
    batched passes alternate on one ctx with equal results."""
    _, st = bamio.read_bam(os.path.join(data_dir, "ex1_header.bam"))
    ora = O.pileup(st)
    for _round in range(3):
        cov = np.concatenate([w.columns["coverage"][w.columns["coverage"] > 0] for w in ctx.batched_windows(_pieces(st, 50_000), 150_000)])
        assert np.array_equal(cov, ora.cov)
        stage(ctx, st)                                      # bgp_reset + bgp_push_inflated: the parked batch buffers go
        info = ctx.pileup_run()
        assert (info.n_columns, info.sum_coverage) == (ora.n_columns, ora.sum_coverage)


@pytest.mark.parametrize("name,rid,beg,end", [("ex1_header.bam", 0, 100, 900), ("ex1_header.bam", 1, 0, 1500), ("tags.bam", 0, 0, 100000)])
def test_entries_after_a_region_filter(ctx, data_dir, name, rid, beg, end):
    """The record table is compacted by bgp_filter_region (K1's side array of first CIGAR ops is not): columns AND entries of
    the kept reads against the oracle on the kept reads' stream; read ordinals count the kept reads."""
    path = os.path.join(data_dir, name)
    _, st = bamio.read_bam(path)
    kept, exp_stream = _expected_region(st, rid, beg, end)
    if not kept:
        pytest.skip("no read in the region")
    ctx.reset(); ctx.load_bam(path, 2); ctx.flush(); ctx.filter_region(rid, beg, end)
    check_pileup(ctx, exp_stream, entries=True, window=257)
    # pileupChunks over the region's reads: chunksConsumingLessThan adds up the sizes of the reads it is GIVEN (splitter.d:65-91),
    # not the stream bytes between them (the reads the filter dropped still lie there)
    for block in (3000, 20_000):
        n = ctx.pileup_chunks_begin(False, block)
        got = [ctx.pileup_chunk(k) for k in range(n)]
        assert [(c.first_record, c.last_record, c.ref_id, c.start_position, c.end_position) for c in got] == O.chunk_plan(exp_stream, block)


@pytest.mark.parametrize("world,device_inflate", [(2, False), (3, True), (5, False), (8, True)])
def test_file_shards_by_the_index_equal_the_whole_file(ctx, tmp_path, world, device_inflate):
    """shard.plan_file_shards + file_shard_windows: a generator BAM with its index (written by the device index writer),
    cut INSIDE its contigs at linear-index windows; the columns the ranks hand out, put together, are the columns of the
    whole file -- every position once, none missing."""
    from biod_amd import shard
    from biod_amd.pileup import BaiFile
    cfg = make_cfg(n_reads=40_000, n_ref=2, ref_len=1_500_000, read_len=(100, 150), profile=1, seed=23)
    st = host_stream(cfg)
    offs = O.frame(st)
    o = [int(x) for x in offs] + [len(st)]
    recs = [bytes(st[o[i]:o[i + 1]]) for i in range(len(offs))]
    path = str(tmp_path / "shards.bam")
    lens = [1_500_000, 1_500_000]
    bamio.write_bam(path, [(f"chr{i + 1}", lens[i]) for i in range(2)], recs)
    bam = BamReader(path, 2)
    bam.createIndex(overwrite=True)
    # the whole file
    c2 = bam.ctx
    c2.pileup_run()
    whole = {}
    for w in c2.pileup_windows(L.BGP_PILEUP_COLUMNS, window_positions=1 << 16):
        nz = np.nonzero(w.columns["coverage"])[0]
        for j in nz:
            whole[(w.ref_id, int(w.first_position) + int(j))] = w.columns[j].tobytes()
    bai = BaiFile(path)
    plan = shard.plan_file_shards(bai, lens, world)
    assert sum(len(r) for r in plan) >= world and any(b > 0 for rank in plan for _, b, _ in rank)       # cuts inside a contig
    got = {}
    for rank in plan:
        for w in shard.file_shard_windows(ctx, path, bai, rank, 2, device_inflate=device_inflate, window_positions=1 << 16):      # (device_inflate: the chunks' blocks through k_inflate, bgp_load_bam_region_device)
            nz = np.nonzero(w.columns["coverage"])[0]
            for j in nz:
                key = (w.ref_id, int(w.first_position) + int(j))
                assert key not in got
                got[key] = w.columns[j].tobytes()
    assert len(whole) > 100_000 and got == whole
