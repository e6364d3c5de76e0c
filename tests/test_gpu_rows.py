"""K4 behind the bench's decode route: k12_lds leaves a 16-byte PLAN ROW and a class byte per record (common.h: bgp_plan_row),
k4_acc2<ROWS> takes a window's ordinary reads from those rows and k4_plan_rest plans only what is left (pileup.hip).  Every case
stages through bgp_flush_decode (K0 + k12_lds), asserts that the rows route ran, and compares decode, columns and per-read entries
with the oracle bit for bit (pileup.d:115-134,175-222,345-397)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from biod_amd import _lib as L
from biod_amd import bamio
from biod_amd.pileup import Context
from biod_amd.synth import host_stream, make_cfg

from test_gpu_parity import FIXTURES, _random_stream, check_decode, check_pileup

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def stage_rows(ctx, st, n_ref=0, need_lds=True):
    """The bench's route: framing, then record table + arenas + plan rows in one kernel over LDS windows."""
    ctx.reset()
    if n_ref:
        ctx.set_n_ref(n_ref)
    ctx.push_inflated(st)
    ctx.timing_enable(True); ctx.timing_reset()
    ctx.flush(decode=L.BGP_FETCH_ALL)
    assert not need_lds or ctx.timing_get()["k12_lds"]["launches"] == 1
    return ctx


def ran_rows(ctx):
    tm = ctx.timing_get()
    return "k4_plan_rest" in tm and tm["k4_plan_rest"]["launches"] > 0


def run(ctx, st, entries=True, window=4096, n_ref=0, expect_rows=True):
    stage_rows(ctx, st, n_ref)
    check_decode(ctx, st)
    info, ora = check_pileup(ctx, st, entries=entries, window=window)
    assert ran_rows(ctx) == expect_rows
    ctx.timing_enable(False)
    return info, ora


@pytest.mark.parametrize("name", [f for f in FIXTURES if f != "bins.bam"])
def test_rows_fixtures(ctx, data_dir, name):
    hdr, st = bamio.read_bam(os.path.join(data_dir, name))
    stage_rows(ctx, st, len(hdr.refs), need_lds=False)      # (a file with long CIGARs takes the other decode kernels: no rows then)
    check_decode(ctx, st)
    check_pileup(ctx, st, entries=True)
    ctx.timing_enable(False)


def rec(name, seq_len, cigar, pos, rng, ref_id=0, flag=0):
    seq = "".join("ACGTNRY="[int(x)] for x in rng.choice(8, size=seq_len, p=[0.24, 0.24, 0.24, 0.24, 0.01, 0.01, 0.01, 0.01]))
    return bamio.pack_record(name, seq, cigar, ref_id=ref_id, pos=pos, flag=flag, qual=bytes(int(x) for x in rng.integers(2, 41, seq_len)))


def test_rows_every_mould(ctx):
    """One read of every shape the row takes or refuses, around a window boundary (positions 1000 .. 1300 of a run that starts at 0:
    the windows are cut at 1024): one and two M runs, soft clips and insertions in front, between and behind, D and N between the runs
    within and beyond the row's reach (256 positions), = and X, P and H, a trailing D, three runs, an unmapped read, a read of clips only."""
    rng = np.random.default_rng(5)
    shapes = [
        (150, [(150, "M")]), (150, [(10, "S"), (140, "M")]), (150, [(3, "I"), (147, "M")]), (150, [(70, "M"), (5, "I"), (75, "M")]),
        (150, [(70, "M"), (20, "D"), (80, "M")]), (150, [(70, "M"), (105, "D"), (80, "M")]), (150, [(70, "M"), (106, "D"), (80, "M")]),
        (150, [(70, "M"), (107, "N"), (80, "M")]), (150, [(70, "M"), (3000, "N"), (80, "M")]), (150, [(60, "="), (90, "X")]),
        (150, [(60, "M"), (2, "P"), (90, "M")]), (150, [(5, "H"), (150, "M"), (5, "H")]), (150, [(140, "M"), (10, "S")]),
        (150, [(100, "M"), (5, "D")]), (150, [(50, "M"), (50, "M"), (50, "M")]), (150, [(50, "M"), (1, "D"), (50, "M"), (50, "M")]),
        (255, [(255, "M")]), (255, [(170, "M"), (1, "I"), (84, "M")]), (255, [(127, "M"), (1, "D"), (128, "M")]),
        (100, [(100, "S")]), (1, [(1, "M")]), (8, [(4, "M"), (2, "I"), (1, "D"), (2, "M")]), (150, [(70, "M"), (2, "I"), (3, "D"), (78, "M")]),
        (150, [(70, "M"), (3, "D"), (2, "I"), (78, "M")]), (150, [(1, "M"), (149, "I")]), (200, [(150, "M")]), (100, [(150, "M")]),
    ]
    recs = [(0, rec("first", 150, [(150, "M")], 0, rng))]      # (the run starts at 0: its windows are cut at 1024, 2048, ...)
    pos = 100
    for rnd in range(36):
        for k, (n, cig) in enumerate(shapes):
            recs.append((pos, rec("s%d_%d" % (rnd, k), n, cig, pos, rng)))
            pos += int(rng.integers(0, 4))
        recs.append((pos, rec("u%d" % rnd, 50, [(50, "M")], pos, rng, flag=4)))
    st = b"".join(b for _, b in recs)
    run(ctx, st, entries=True, window=999)


@pytest.mark.parametrize("seed", range(6))
def test_rows_random_cigars(ctx, seed):
    rng = np.random.default_rng(4000 + seed)
    st = _random_stream(rng, 600, quirks=seed % 2 == 1)
    stage_rows(ctx, st)
    check_decode(ctx, st)
    try:
        O.pileup(st, want_entries=False)
    except O.OracleError as e:
        assert e.code == O.ORC_ECIGAR
        with pytest.raises(L.BiodGpuError) as g:
            ctx.pileup_run()
        assert g.value.code == L.BGP_ECIGAR
        ctx.timing_enable(False)
        return
    check_pileup(ctx, st, entries=True, window=1021)
    ctx.timing_enable(False)


@pytest.mark.parametrize("profile,read_len,n_reads,ref_len", [(0, 150, 20000, 120_000), (1, 150, 30000, 150_000), (1, 100, 10000, 50_000),
                                                              (0, 250, 8000, 80_000), (1, (170, 255), 8000, 80_000), (1, (240, 300), 6000, 80_000)])
def test_rows_synthetic(ctx, profile, read_len, n_reads, ref_len):
    cfg = make_cfg(n_reads=n_reads, n_ref=2, ref_len=ref_len, read_len=read_len, profile=profile, seed=11)
    st = host_stream(cfg)
    long_reads = not isinstance(read_len, int) and read_len[1] > 255      # (reads a SEG descriptor cannot hold: the stream is planned read by read, long reads walked from checkpoints)
    info, ora = run(ctx, st, entries=False, expect_rows=not long_reads)
    assert info.n_columns > 0


@pytest.mark.parametrize("n_reads,ref_len,profile,read_len", [(8000, 12_000, 1, 150), (5000, 6_000, 0, 150), (14_000, 12_000, 1, 150), (9000, 12_000, 1, (200, 255)), (30000, 9_000, 1, 40)])
def test_rows_deep_windows(ctx, n_reads, ref_len, profile, read_len):
    """100- to 250-fold coverage: several rounds of rows a window, windows beyond the byte-wide counters handed to the general kernel
    next to windows that are not; short reads at depth: more rows a window than bases a position."""
    cfg = make_cfg(n_reads=n_reads, n_ref=1, ref_len=ref_len, read_len=read_len, profile=profile, seed=77 + n_reads)
    st = host_stream(cfg)
    info, ora = run(ctx, st, entries=False)
    assert info.max_coverage > 90


def test_rows_two_runs_of_pieces(ctx):
    """Reads of 255 bases in two M runs (five pieces a read) at 120-fold coverage next to N-skip reads the planner takes: the rounds
    that make the most pieces."""
    rng = np.random.default_rng(21)
    recs, pos = [], 3
    for i in range(2600):
        r = rng.random()
        if r < 0.6:
            a = int(rng.integers(165, 175)); cig = [(a, "M"), (1, "I"), (254 - a, "M")]; n = 255
        elif r < 0.8:
            a = int(rng.integers(100, 130)); cig = [(a, "M"), (int(rng.integers(300, 2000)), "N"), (250 - a, "M")]; n = 250
        else:
            sc = int(rng.integers(0, 5)); cig = ([(sc, "S")] if sc else []) + [(250 - sc, "M")]; n = 250
        recs.append(rec("p%d" % i, n, cig, pos, rng))
        pos += int(rng.integers(0, 5))
    st = b"".join(recs)
    info, ora = run(ctx, st, entries=False)
    assert info.max_coverage > 100


def test_rows_unsorted_stream_is_refused(ctx):
    """A stream that is not coordinate sorted is an error whichever way K4 would have gone (makePileup's precondition, pileup.d:25-27):
    a window's rows would no longer be a contiguous range of records."""
    rng = np.random.default_rng(8)
    recs = [rec("a%d" % i, 100, [(100, "M")], p, rng) for i, p in enumerate([10, 400, 30, 2000, 1500, 1800, 50])]
    st = b"".join(recs)
    stage_rows(ctx, st)
    with pytest.raises(L.BiodGpuError) as g:
        ctx.pileup_run()
    assert g.value.code == L.BGP_EUNSORTED
    ctx.timing_enable(False)


def test_rows_islands_and_references(ctx):
    """Runs that begin in the middle of nowhere (a window's first position is the run's, not a multiple of 1024), several references,
    reads at position 0, a run shorter than the row's reach."""
    rng = np.random.default_rng(13)
    recs = []
    for ref_id, starts in enumerate([[0, 0, 1, 5, 100, 1000, 1023, 1024, 1025, 1279, 1280, 2047, 2048], [70000, 70001, 70100, 200000, 200255, 200256, 200257, 201279, 201280], [5]]):
        for j, p in enumerate(starts):
            for k in range(3):
                cig = [[(150, "M")], [(70, "M"), (20, "D"), (80, "M")], [(20, "S"), (60, "M"), (4, "I"), (66, "M")]][k]
                recs.append(rec("i%d_%d_%d" % (ref_id, j, k), 150, cig, p, rng, ref_id=ref_id))
    st = b"".join(recs)
    run(ctx, st, entries=True, window=500, n_ref=3)


def test_decode_only_writes_nothing_for_k4(ctx):
    """bgp_flush_decode(BGP_FETCH_ALL | BGP_DECODE_ONLY): the caller says no pileup follows, so k12_lds writes the table and the arenas and neither
    plan rows nor the first-ops side array.  The decode == oracle; a pileup asked for all the same is correct too -- the planner takes every read
    (k4_plan, not k4_plan_rest), and the side array it and K5 read is made from the CIGAR arena first (k_cig4_from_arena)."""
    cfg = make_cfg(n_reads=12000, n_ref=2, ref_len=60_000, read_len=150, profile=1, seed=31)
    st = host_stream(cfg)
    ctx.reset()
    ctx.push_inflated(st)
    ctx.timing_enable(True); ctx.timing_reset()
    ctx.flush(decode=L.BGP_FETCH_ALL | L.BGP_DECODE_ONLY)
    assert ctx.timing_get()["k12_lds"]["launches"] == 1
    check_decode(ctx, st)
    check_pileup(ctx, st, entries=True, window=3000)
    tm = ctx.timing_get()
    assert not ran_rows(ctx) and tm["k4_plan"]["launches"] > 0 and tm["k_cig4_from_arena"]["launches"] == 1
    ctx.timing_enable(False)


def test_rows_entries_and_reference_bases(ctx):
    """Behind plan rows k12_lds leaves the first-ops side array out: K5's entries have it made from the CIGAR arena once (k_cig4_from_arena), K6 takes
    the CIGAR-implied length of dna(read) from a bit beside the class -- end_position - position unless the CIGAR holds an N (md/reconstruct.d:199)
    or was not seen whole.  Reads with N skips, with deletions, clipped ones: entries and reference_base == oracle."""
    from test_gpu_parity import _check_refbase
    rng = np.random.default_rng(61)
    recs, pos = [], 10
    for i in range(900):
        k = int(rng.integers(0, 6))
        if k == 0:
            cig, n, md = [(150, "M")], 150, "150"
        elif k == 1:
            a = int(rng.integers(20, 120)); cig, n, md = [(a, "M"), (int(rng.integers(30, 400)), "N"), (150 - a, "M")], 150, "150"
        elif k == 2:
            a = int(rng.integers(20, 120)); d = int(rng.integers(1, 9)); cig, n = [(a, "M"), (d, "D"), (150 - a, "M")], 150
            md = f"{a}^{'ACGT'[i % 4] * d}{150 - a}"
        elif k == 3:
            cig, n, md = [(5, "S"), (60, "M"), (3, "D"), (85, "M")], 150, "60^ACG85"
        elif k == 4:
            cig, n, md = [(10, "S"), (60, "M"), (70, "N"), (80, "M")], 150, "140"
        else:
            cig, n, md = [(3, "H"), (10, "S"), (140, "M")], 150, "70A69"
        seq = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, n))
        recs.append(bamio.pack_record("e%d" % i, seq, cig, ref_id=0, pos=pos, qual=bytes(int(x) for x in rng.integers(2, 41, n)), tags=bamio.tag_z("MD", md)))
        pos += int(rng.integers(0, 12))
    st = b"".join(recs)
    stage_rows(ctx, st)
    check_decode(ctx, st)
    check_pileup(ctx, st, entries=True, window=2000)
    tm = ctx.timing_get()
    assert ran_rows(ctx) and tm["k_cig4_from_arena"]["launches"] == 1
    for skip in (True, False):
        _check_refbase(ctx, st, skip_zero=skip, window=1500)
    ctx.timing_enable(False)


def test_rows_reads_of_up_to_sixteen_ops(ctx):
    """Behind plan rows the planner (k4_plan_rest) takes the reads a row cannot hold four ops at a time, up to sixteen: doubly-indelled reads, clips
    on both sides, N skips between M runs, at depth -- columns and entries == oracle, the rows route taken."""
    rng = np.random.default_rng(83)
    shapes = [
        [(5, "S"), (40, "M"), (2, "I"), (30, "M"), (3, "D"), (60, "M"), (13, "S")],
        [(30, "M"), (1, "I"), (30, "M"), (1, "D"), (30, "M"), (1, "I"), (30, "M"), (1, "D"), (28, "M")],
        [(3, "H"), (4, "S"), (20, "M"), (300, "N"), (20, "M"), (2, "P"), (20, "M"), (900, "N"), (20, "=", ), (10, "X"), (2, "D"), (30, "M"), (5, "I"), (15, "M"), (6, "S"), (2, "H")],
        [(10, "M")] + [(1, "D"), (9, "M")] * 7 + [(1, "I")],
        [(150, "M")], [(70, "M"), (5, "D"), (80, "M")],
    ]
    recs, pos = [], 40
    for i in range(2400):
        cig = shapes[int(rng.integers(0, len(shapes)))]
        n = sum(ln for ln, op in cig if op in "MIS=X")
        recs.append(rec("s%d" % i, n, cig, pos, rng))
        pos += int(rng.integers(0, 7))
    st = b"".join(recs)
    info, ora = run(ctx, st, entries=True, window=1777)
    assert info.max_coverage > 40


def test_a_read_of_seventeen_ops_keeps_the_stream_off_the_rows_route(ctx):
    """Framing sees every header: one read of more CIGAR ops than the planner takes (K4 walks such reads from checkpoints and plans every read itself)
    and k12_lds writes the first-ops side array instead of plan rows, as it did before there were rows -- nothing is written for nobody, nothing made twice."""
    rng = np.random.default_rng(71)
    recs = [rec("n%d" % i, 150, [(150, "M")], 5 * i, rng) for i in range(400)]
    recs.insert(200, rec("odd", 150, [(10, "M")] + [(1, "D"), (10, "M")] * 8 + [(60, "M")], 5 * 200, rng))
    st = b"".join(recs)
    stage_rows(ctx, st)
    check_decode(ctx, st)
    check_pileup(ctx, st, entries=True, window=700)
    tm = ctx.timing_get()
    assert not ran_rows(ctx) and tm["k4_plan"]["launches"] > 0 and "k_cig4_from_arena" not in tm
    ctx.timing_enable(False)
