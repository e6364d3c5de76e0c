"""f3: the BAI / region-query restatement pinned to the reference's own checks, and the
host side of the C ABI (no GPU needed: index parsing, chunk selection, chunk bytes) against that restatement."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import bai_oracle as B                                     # noqa: E402
import oracle_lib as O                                     # noqa: E402
from biod_amd import bamio                                 # noqa: E402
from biod_amd.pileup import BaiFile, read_chunks          # noqa: E402

# test/unittests.d:161-185
REGIONS = [(1400, 1500), (10, 123), (135, 1236), (1350, 3612), (643, 1732), (267, 1463), (0, 30), (1363, 1612), (361, 1231),
           (322, 612), (912, 938), (0, 3000), (0, 100), (0, 1000), (0, 1900), (1, 279)] + [(i, i + 100) for i in range(50_000, 1_000_000, 50_000)]


@pytest.fixture(scope="module")
def bins(data_dir):
    path = os.path.join(data_dir, "bins.bam")
    hdr, st = bamio.read_bam(path)
    return path, hdr.refs, st, B.frame(st), B.parse_bai(path)


def naive(records, ref_id, beg, end):                      # test/unittests.d:152-157
    return [r for r in records if r[1] == ref_id and r[2] < end and r[3] > beg]


def test_frame_agrees_with_c_oracle(bins):
    path, refs, st, recs, bai = bins
    offs = O.frame(st)
    assert [r[0] for r in recs] == [int(x) for x in offs]
    for r in recs[::97]:
        f = O.fields_at(st, r[0])
        assert (f.ref_id, f.pos, f.end_pos) == (r[1], r[2], r[3])


@pytest.mark.parametrize("beg,end", REGIONS)
def test_region_equals_naive_filter(bins, beg, end):
    """compareWithNaiveApproach, test/unittests.d:148-185: bam["large"][beg .. end] == filter over all reads."""
    path, refs, st, recs, bai = bins
    large = [n for n, _ in refs].index("large")
    _, got = B.region_reads(path, bai, large, beg, end)
    exp = naive(recs, large, beg, end)
    assert [r[4] for r in got] == [r[4] for r in exp]
    assert [(r[1], r[2], r[3]) for r in got] == [(r[1], r[2], r[3]) for r in exp]


def test_first_reads_of_every_reference(bins):
    """test/unittests.d:192-206: startVirtualOffset -> getReadAt names."""
    path, refs, st, recs, bai = bins
    for name in ("tiny", "small", "large"):
        rid = [n for n, _ in refs].index(name)
        length = refs[rid][1]
        stc, got = B.region_reads(path, bai, rid, 0, length)
        assert got[0][4] == f"{name}:r1:0..1:len1:bin4681:hexbin0x1249"


def test_reg2bin_matches_record_bins(bins):
    """Every record's bin field is reg2bin(pos, end) (bai/bin.d:82-92; the generator of bins.bam names reads by bin)."""
    path, refs, st, recs, bai = bins
    for off, ref, pos, end, name in recs[::13]:
        (bmn,) = np.frombuffer(st, "<u4", 1, off + 12)
        if ":bin" in name:
            assert B.reg2bin(pos, end) == int(name.split(":bin")[1].split(":")[0]) == int(bmn) >> 16


# ------------------------------------------------------------------------------------------ C ABI, host side
@pytest.mark.parametrize("fn", ["bins.bam", "ex1_header.bam", "tags.bam"])
def test_abi_bai_parse_and_chunks(data_dir, fn):
    path = os.path.join(data_dir, fn)
    bai = B.parse_bai(path)
    h = BaiFile(path)
    assert h.n_ref == len(bai)
    rng = np.random.default_rng(3)
    for rid, ix in enumerate(bai):
        assert h.shape(rid) == (len(ix["bins"]), sum(len(c) for _, c in ix["bins"]), len(ix["ioffsets"]))
        for pos in (-5, 0, 1, 16383, 16384, 40_000, 1 << 20, (1 << 29) - 1, (1 << 31) - 1):
            assert h.min_offset(rid, pos) == B.min_offset(ix, pos)
        regions = [(0, 1), (0, 100), (0, 1 << 29), (5, 0xFFFFFFFF), (100, 50)] + REGIONS[:8] + \
                  [(int(a), int(a) + int(b)) for a, b in zip(rng.integers(0, 1_100_000, 40), rng.integers(1, 200_000, 40))]
        for beg, end in regions:
            exp = B.get_chunks(bai, rid, beg, end)
            got = h.chunks(rid, beg, end)
            assert [tuple(int(x) for x in r) for r in got] == exp, (rid, beg, end)
    with pytest.raises(Exception):
        h.chunks(len(bai), 0, 10)                          # checkRefId, randomaccessmanager.d:205-207


def test_abi_bin_helpers():
    from biod_amd import _lib as L
    rng = np.random.default_rng(5)
    for _ in range(3000):
        beg = int(rng.integers(0, 1 << 29)); end = beg + int(rng.integers(0, 1 << 18))
        assert L.lib().bgp_reg2bin(beg, end) == B.reg2bin(beg, end)
    ids = [0, 1, 8, 9, 72, 73, 584, 585, 4680, 4681, 4682, 37448, 37449, 37450, 40000] + [int(x) for x in rng.integers(0, 37450, 200)]
    for bid in ids:
        for beg, end in [(0, 1), (0, 16384), (16383, 16385), (100_000, 200_000), (0, 1 << 29), (-7, 5), (5, -1), (1 << 20, (1 << 20) + 1)]:
            assert bool(L.lib().bgp_bin_can_overlap(bid, beg, end)) == B.can_overlap(bid, beg, end), (bid, beg, end)


def test_abi_read_chunks_bytes(bins):
    path, refs, st, recs, bai = bins
    large = [n for n, _ in refs].index("large")
    for beg, end in REGIONS[:10] + [(0, 1 << 29)]:
        ch = B.get_chunks(bai, large, beg, end)
        assert read_chunks(path, np.array(ch, dtype=np.uint64).reshape(-1, 2), n_threads=3) == B.read_chunks(path, ch)
    # hand-made chunk lists: two chunks starting in the same block (folded, inputstream.d:259-273), a chunk ending on a
    # block edge (uoffset 0: that block contributes nothing, :316-322), a chunk inside one block
    all_ch = sorted(c for _, chunks in bai[large]["bins"] for c in chunks)
    a, b = all_ch[0], all_ch[-1]
    hand = [[(a[0], a[0] + 40), (a[0] + 40, a[1])], [(a[0], (b[1] >> 16) << 16)], [(a[0] + 10, a[0] + 300)]]
    for ch in hand:
        assert read_chunks(path, np.array(ch, dtype=np.uint64).reshape(-1, 2)) == B.read_chunks(path, ch)


# ---- the index writer (bai/indexing.d:56-366): restatement checks that need no GPU --------------------------------
# The reference holds no test and no fixture for createIndex (the .bai files in test/data come from another tool:
# their linear index starts at the second read and their last chunk ends behind the EOF block), so the writer's
# oracle is pinned only through what the reference DOES check with an index: unittests.d:145-206 must hold on it.

def test_built_index_answers_the_reference_region_checks(bins, tmp_path):
    path, refs, st, recs, _ = bins
    out = str(tmp_path / "bins.bam.bai")
    open(out, "wb").write(B.create_index(path))
    bai = B.parse_bai(out)
    large = [n for n, _ in refs].index("large")
    for beg, end in REGIONS:
        _, got = B.region_reads(path, bai, large, beg, end)
        assert [r[4] for r in got] == [r[4] for r in naive(recs, large, beg, end)], (beg, end)
    for name in ("tiny", "small", "large"):
        rid = [n for n, _ in refs].index(name)
        _, got = B.region_reads(path, bai, rid, 0, refs[rid][1])
        assert got[0][4] == f"{name}:r1:0..1:len1:bin4681:hexbin0x1249"


def test_built_index_against_the_fixture_index(data_dir):
    """Same bins and chunk starts as the test/data index on ex1 (the two writers differ in two documented spots)."""
    path = os.path.join(data_dir, "ex1_header.bam")
    n, reads = B.records_with_offsets(path)
    mine, no_coord = B.build_index(n, reads)
    fx = B.parse_bai(path)
    assert no_coord == 0 and len(mine) == len(fx) == 2
    for a, b in zip(mine, fx):
        theirs = {k: v for k, v in b["bins"]}
        assert sorted(a["bins"]) + [37450] == sorted(theirs)
        for k, cs in a["bins"].items():
            assert [c[0] for c in cs] == [c[0] for c in theirs[k]]
        assert tuple(theirs[37450][1]) == a["meta"][2:]                       # mapped / unmapped counts
        assert len(a["ioffsets"]) == len(b["ioffsets"])
    assert sum(r["meta"][2] + r["meta"][3] for r in mine) == len(reads)


def test_index_builder_by_hand():
    """Five reads, offsets chosen by hand: one bin run merged across a block edge, one not, a hole in the linear index."""
    K = 1 << 16
    def rd(ref, pos, span, b, s, e, unmapped=False):
        return dict(ref=ref, pos=pos, end_pos=pos + span, bin=b, unmapped=unmapped, start_vo=s, end_vo=e)
    reads = [rd(1, 10, 50, 4681, 5 * K, 5 * K + 100),
             rd(1, 20, 50, 4681, 5 * K + 100, 5 * K + 200),
             rd(1, 16000, 1000, 585, 5 * K + 200, 9 * K + 7),              # straddles two 16 kb windows: parent bin
             rd(1, 16390, 10, 4682, 9 * K + 7, 9 * K + 90),
             rd(1, 70000, 10, 4685, 9 * K + 90, 12 * K),                   # windows 2 and 3 stay untouched
             rd(-1, -1, 0, 4680, 12 * K, 12 * K + 50, True)]
    refs, no_coord = B.build_index(3, reads)
    assert no_coord == 1 and refs[0] is None and refs[2] is None
    r = refs[1]
    assert r["bins"] == {4681: [[5 * K, 5 * K + 200]], 585: [[5 * K + 200, 9 * K + 7]], 4682: [[9 * K + 7, 9 * K + 90]], 4685: [[9 * K + 90, 12 * K]]}
    assert r["meta"] == (5 * K, 12 * K, 5, 0)
    assert r["ioffsets"] == [5 * K, 5 * K + 200, 5 * K + 200, 5 * K + 200, 9 * K + 90]
    raw = B.serialise_index(3, refs, no_coord)
    assert raw[:8] == b"BAI\x01\x03\x00\x00\x00" and raw[-8:] == (1).to_bytes(8, "little")
    # same bin again later in the file: joined only when the new chunk starts in the block the old one ended in
    again = [rd(0, 5, 10, 4681, 2 * K, 2 * K + 40), rd(0, 6, 40000, 0, 2 * K + 40, 2 * K + 80), rd(0, 7, 10, 4681, 2 * K + 80, 3 * K),
             rd(0, 8, 40000, 0, 3 * K, 4 * K + 1), rd(0, 9, 10, 4681, 4 * K + 1, 4 * K + 30)]
    r0 = B.build_index(1, again)[0][0]
    assert r0["bins"][4681] == [[2 * K, 3 * K], [4 * K + 1, 4 * K + 30]] and r0["bins"][0] == [[2 * K + 40, 2 * K + 80], [3 * K, 4 * K + 1]]
    with pytest.raises(B.NotSorted):
        B.build_index(1, [rd(0, 9, 1, 4681, K, K + 1), rd(0, 8, 1, 4681, K + 1, K + 2)])
