"""makePileup over a FILTERED range of reads (bgp_set_read_filter): reads.filter!(r => flag / mapping-quality predicate) in front of the
pileup, as BioD's callers write it (pileup.d:683-687 takes any input range of reads; the predicates are the flag properties of
read.d:137-194 and mapping_quality).  The oracle piles up the stream with the failing records REMOVED; the device marks them in place:
same columns, same entries (read ordinals mapped through the kept records), same reference bases -- on both decode routes, through
batched staging and through the file-fed pass."""
import os

import numpy as np
import pytest

import oracle_lib as O
from biod_amd import _lib as L
from biod_amd import bamio
from biod_amd.pileup import Context
from biod_amd.synth import host_stream, make_cfg

from test_gpu_parity import _assert_columns_equal_oracle, _covered, _random_stream

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.set_read_filter(0, 0, 0)
    c.close()


def split(st):
    offs = [int(x) for x in O.frame(st)] + [len(st)]
    return [bytes(st[offs[i]:offs[i + 1]]) for i in range(len(offs) - 1)]


def passes(rec, require, exclude, min_mapq):
    bmn = int.from_bytes(rec[12:16], "little"); fnc = int.from_bytes(rec[16:20], "little")
    flag, mq = fnc >> 16, (bmn >> 8) & 0xFF
    return (flag & require) == require and not (flag & exclude) and mq >= min_mapq


def with_flags(st, rng):
    """the generator's reads with flags and mapping qualities a filter can tell apart"""
    out = []
    for rec in split(st):
        b = bytearray(rec)
        flag = int(rng.choice([0, 16, 0x400, 0x100, 0x200 | 16, 0x2 | 0x1, 0x400 | 0x2 | 0x1]))
        fnc = int.from_bytes(b[16:20], "little")
        b[16:20] = ((fnc & 0xFFFF) | (flag << 16)).to_bytes(4, "little")
        b[13] = int(rng.integers(0, 61))
        out.append(bytes(b))
    return b"".join(out)


FILTERS = [(0, 0x400, 0), (0, 0x700, 20), (0x3, 0, 0), (0, 16, 30), (0, 0, 60), (0xFFFF, 0, 0)]


@pytest.mark.parametrize("flt", FILTERS)
@pytest.mark.parametrize("rows", [False, True])
def test_filtered_pileup_equals_oracle_on_the_filtered_range(ctx, flt, rows):
    """columns, per-read entries (ordinals of the unfiltered stream, mapped) and use_md_tag's reference bases; `rows`: through
    bgp_flush_decode (k12_lds leaves plan rows: the filter clears them) or bgp_flush (the planner reads the marked table)"""
    rng = np.random.default_rng(sum(flt) + rows)
    cfg = make_cfg(n_reads=6000, n_ref=2, ref_len=25_000, read_len=100, profile=1, seed=41 + sum(flt) % 7)
    st = with_flags(host_stream(cfg), rng)
    recs = split(st)
    keep = [i for i, r in enumerate(recs) if passes(r, *flt)]
    assert flt == (0xFFFF, 0, 0) or 0 < len(keep) < len(recs)
    flt_st = b"".join(recs[i] for i in keep)
    ora = O.pileup(flt_st, want_entries=True)
    ctx.set_read_filter(*flt)
    ctx.reset(); ctx.push_inflated(st)
    ctx.flush(decode=L.BGP_FETCH_ALL) if rows else ctx.flush()
    info = ctx.pileup_run()
    assert (info.n_columns, info.sum_coverage, info.max_coverage) == (ora.n_columns, ora.sum_coverage, ora.max_coverage)
    assert info.n_reads_used <= len(keep)
    ref, pos, cols, ents, _ = _covered(ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, want_entries=True, window_positions=4000), want_entries=True)
    if ora.n_columns:
        _assert_columns_equal_oracle(ref, pos, cols, ora)
    kept = np.asarray(keep, np.uint32)
    for k in range(0, ora.n_columns, 13):
        oe = ora.entries[int(ora.entry_off[k]):int(ora.entry_off[k + 1])]
        assert np.array_equal(ents[k]["read_ordinal"], kept[oe["read_ordinal"]]), k
        assert np.array_equal(ents[k]["query_offset"], oe["query_offset"]) and np.array_equal(ents[k]["base"], oe["base"])
    if ora.n_columns:
        try:
            ora_md = O.pileup(flt_st, use_md=True)
        except O.OracleError as e:                       # (BioD's popFront throws where a provider's MD tag runs out -- behind an N skip a chunk drifts; the filtered range has its own providers)
            assert e.code == -7
            with pytest.raises(L.BiodGpuError) as g:
                list(ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, use_md_tag=True, window_positions=4000))
            assert g.value.code == L.BGP_EMD
        else:
            got = np.concatenate([w.reference_base[w.columns["coverage"] > 0] for w in ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, use_md_tag=True, window_positions=4000)])
            assert np.array_equal(got, ora_md.refbase)
    # a dropped read "covers nothing" in the table; createIndex is refused under a filter
    dec = ctx.decode_fetch(0)
    dropped = np.setdiff1d(np.arange(len(recs)), kept)
    assert np.array_equal(dec.rec["end_pos"][dropped], dec.rec["pos"][dropped])
    ctx.set_read_filter(0, 0, 0)
    ctx.reset(); ctx.push_inflated(st); ctx.flush()
    whole = O.pileup(st)
    info = ctx.pileup_run()
    assert (info.n_columns, info.sum_coverage) == (whole.n_columns, whole.sum_coverage)          # the setting is gone with (0, 0, 0)


def test_filter_through_batches_and_a_streamed_file(ctx, tmp_path, data_dir):
    """the reference's ex1 fixture (paired reads, both strands, mapping qualities 0..99): unmapped / reverse-strand / low-quality reads
    dropped -- one-shot, in batches of a few dozen reads and as a file through bgp_stream_bam_begin: the oracle's columns on the filtered range"""
    path = os.path.join(data_dir, "ex1_header.bam")
    _, st = bamio.read_bam(path)
    recs = split(st)
    for flt in [(0, 0x4 | 0x10, 0), (0x1, 0, 40), (0x2, 0x10, 10)]:
        keep = [i for i, r in enumerate(recs) if passes(r, *flt)]
        assert 0 < len(keep) < len(recs)
        ora = O.pileup(b"".join(recs[i] for i in keep))
        ctx.set_read_filter(*flt)
        try:
            ctx.reset(); ctx.push_inflated(st); ctx.flush()
            ref, pos, cols, _, _ = _covered(ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, window_positions=500))
            _assert_columns_equal_oracle(ref, pos, cols, ora)
            ref, pos, cols, _, _ = _covered(ctx.batched_windows([st[o:o + 7001] for o in range(0, len(st), 7001)], 20_000, window_positions=300))
            _assert_columns_equal_oracle(ref, pos, cols, ora)
            ref, pos, cols, _, _ = _covered(ctx.stream_bam_windows(path, 128 << 10, window_positions=700))
            _assert_columns_equal_oracle(ref, pos, cols, ora)
            ctx.reset(); ctx.load_bam(path, 2); ctx.flush()
            with pytest.raises(L.BiodGpuError) as e:
                ctx.bai_build()
            assert e.value.code == L.BGP_ESTATE
        finally:
            ctx.set_read_filter(0, 0, 0)
    with pytest.raises(L.BiodGpuError):
        ctx.set_read_filter(1 << 16, 0, 0)
    ctx.reset()
