"""Pins the parity oracle (oracle/biod_oracle.c) to every known-answer test the reference holds for
the decode -> CIGAR walk -> pileup path.  CPU only.

KAT ids follow SURVEY.md:  P1 pileup.d:699-826, P2 pileup.d:830-856, P3 test/unittests.d:334-366,
P4 test/unittests.d:320-330, P5 examples/make_pileup.d:20-30, R1 read.d:1286-1350,
R2 test/unittests.d:88-105, R3 test/unittests.d:303, C1 cigar.d:116, B1 base.d:100-113,
M1 md/reconstruct.d:226-258, M2 md/reconstruct.d:406-439, M3 md/parse.d:154-181,
Z1 test/unittests.d:132-142.
"""
import os

import numpy as np
import pytest

import oracle_lib as O
from biod_amd import bamio


def _stream(data_dir, name):
    return bamio.read_bam(os.path.join(data_dir, name))


# ------------------------------------------------------------------------------------------ C1 / B1

def test_C1_cigar_type_mask():
    L = O.lib()
    spec = {"M": (1, 1), "I": (1, 0), "D": (0, 1), "N": (0, 1), "S": (1, 0), "H": (0, 0), "P": (0, 0), "=": (1, 1), "X": (1, 1)}
    for code, ch in enumerate("MIDNSHP=X"):
        raw = (77 << 4) | code
        assert L.orc_cigar_length(raw) == 77
        assert L.orc_cigar_type(raw) == ch.encode()
        assert (L.orc_cigar_is_query_consuming(raw), L.orc_cigar_is_reference_consuming(raw)) == spec[ch]
        assert L.orc_cigar_is_match_or_mismatch(raw) == int(spec[ch] == (1, 1))
        assert L.orc_cigar_is_clipping(raw) == int(ch in "SH")
    for code in range(9, 16):
        assert L.orc_cigar_type(code) == b"?"
        assert not L.orc_cigar_is_query_consuming(code) and not L.orc_cigar_is_reference_consuming(code)


def test_B1_base_tables():
    L = O.lib()
    code2char = "=ACMGRSVTWYHKDBN"
    for i, c in enumerate(code2char):
        assert L.orc_base_code2char(i) == c.encode()
        assert L.orc_base_char2code(ord(c)) == i
    # base.d:104-113: every byte maps to N unless it is a digit 0-3 (ACGT) or an IUPAC letter in either case
    for c in range(256):
        c2 = L.orc_base_code2char(L.orc_base_char2code(c)).decode()
        if c2 != "N":
            if ord("0") <= c <= ord("9"):
                assert c2 == "ACGT"[c - ord("0")]
            else:
                assert c2 == chr(c).upper()
    # python mirror used by the record builder agrees
    for c in range(256):
        assert bamio.char2code(chr(c)) == L.orc_base_char2code(c)


# ------------------------------------------------------------------------------------------ R1 / R2 / R3

def test_R1_sequence_pack_unpack():
    rec = bamio.pack_record("readname", "AGCTGACTACGTAATAGCCCTA", [(22, "M")], qual=bytes(range(1, 23)))
    s = O.decode_soa(rec)
    assert s.l_seq[0] == 22 and len(s.cigar) == 1 and s.cigar[0] == (22 << 4)
    assert bytes(s.seq).decode() == "AGCTGACTACGTAATAGCCCTA"
    assert int(s.qual.sum()) == 253
    rec = bamio.pack_record("anothername", "AGCTGGCTACGTAATAGCCCT", [(20, "M"), (2, "X")])
    s = O.decode_soa(rec)
    seq = bytes(s.seq).decode()
    assert s.l_seq[0] == 21 and s.qual[20] == 255 and seq == "AGCTGGCTACGTAATAGCCCT"
    assert seq[::-1][2] == "C" and seq[::-1][0] == "T" and seq[4] == "G" and seq[0] == "A"
    assert seq[0:8] == "AGCTGGCT" and seq[3:5] == "TG" and seq[3:9][1:4] == "GGC"
    f = O.fields_at(rec, 0)
    assert f.l_read_name == 12 and f.n_cigar_op == 2 and f.mapq == 255 and f.ref_id == -1 and f.pos == -1


def test_R2_R3_ex1_first_records(data_dir):
    hdr, st = _stream(data_dir, "ex1_header.bam")
    assert [r[0] for r in hdr.refs] == ["chr1", "chr2"] and [r[1] for r in hdr.refs] == [1575, 1584]
    s = O.decode_soa(st)
    assert len(s.rec_off) == 3270                                         # R3
    seq0 = bytes(s.seq[int(s.seq_off[0]):int(s.seq_off[1])]).decode()
    q0 = bytes(s.qual[int(s.seq_off[0]):int(s.seq_off[1])] + 33).decode()
    assert seq0 == "CTCAAGGTTGTTGCAAGGGGGTCTATGTGAACAAA"
    assert q0 == "<<<7<<<;<<<<<<<<8;;<7;4<;<;;;;;94<;"
    f = O.fields_at(st, int(s.rec_off[0]))
    assert hdr.refs[f.ref_id][0] == "chr1"
    assert st[f.name_off:f.name_off + f.l_read_name - 1] == b"EAS56_57:6:190:289:82"
    assert f.flag == 69 and f.pos == 99 and f.mapq == 0
    # third record: EAS51_64:3:190:727:308 99 chr1 103 99 35M = 263 195 GGTG... <<<<...
    f = O.fields_at(st, int(s.rec_off[2]))
    assert st[f.name_off:f.name_off + f.l_read_name - 1] == b"EAS51_64:3:190:727:308"
    assert (f.flag, f.ref_id, f.pos + 1, f.mapq, f.next_ref_id, f.next_pos + 1, f.tlen) == (99, 0, 103, 99, 0, 263, 195)
    assert s.cigar[int(s.cigar_off[2])] == (35 << 4) and s.cigar_off[3] - s.cigar_off[2] == 1
    assert bytes(s.seq[int(s.seq_off[2]):int(s.seq_off[3])]).decode() == "GGTGCAGAGCCGAGTCACGGGGTTGCCAGCACAGG"
    assert bytes(s.qual[int(s.seq_off[2]):int(s.seq_off[3])] + 33).decode() == "<<<<<<<<<<<<<<<<<<<<<<<<<<<::<<<844"
    import struct
    tags = st[f.tags_off:int(s.rec_off[2]) + 4 + f.block_size]
    assert tags[:3] == b"MFC" and tags[3] == 18                            # MF:i:18 stored as 'C'


def test_framing_edge_cases():
    rec = bamio.pack_record("r", "ACGT", [(4, "M")], ref_id=0, pos=5)
    assert len(O.frame(b"")) == 0
    assert len(O.frame(rec + rec)) == 2
    assert len(O.frame(rec + b"\x01\x02")) == 1                            # partial prefix: silently ends (readrange.d:139-149)
    with pytest.raises(O.OracleError) as e:
        O.frame(rec + rec[:-1])                                            # truncated body: readExact throws (readrange.d:169)
    assert e.value.code == O.ORC_ETRUNC


# ------------------------------------------------------------------------------------------ M1 / M2 / M3

def test_M3_md_operations():
    assert O.md_operations("86") == [("match", 86)]
    assert O.md_operations("0G81") == [("mismatch", "G"), ("match", 81)]
    assert O.md_operations("62^T28") == [("match", 62), ("deletion", "T"), ("match", 28)]
    assert O.md_operations("3C6C0A13^A4C2") == [("match", 3), ("mismatch", "C"), ("match", 6), ("mismatch", "C"),
                                                ("mismatch", "A"), ("match", 13), ("deletion", "A"), ("match", 4),
                                                ("mismatch", "C"), ("match", 2)]
    assert O.md_operations("27^TTT63") == [("match", 27), ("deletion", "TTT"), ("match", 63)]
    assert O.md_operations("") == [] and O.md_operations("0") == [] and O.md_operations("0A0") == [("mismatch", "A")]


M1 = [("CGATACGGGGACATCCGGCCTGCTCCTTCTCACATG", [(36, "M")], "1A0C0C0C1T0C0T27", "CACCCCTCTGACATCCGGCCTGCTCCTTCTCACATG"),
      ("GAGACGGGGTGACATCCGGCCTGCTCCTTCTCACAT", [(6, "M"), (1, "I"), (29, "M")], "0C1C0C1C0T0C27", "CACCCCTCTGACATCCGGCCTGCTCCTTCTCACAT"),
      ("AGTGATGGGGGGGTTCCAGGTGGAGACGAGGACTCC", [(9, "M"), (9, "D"), (27, "M")], "2G0A5^ATGATGTCA27", "AGGAATGGGATGATGTCAGGGGTTCCAGGTGGAGACGAGGACTCC"),
      ("AGTGATGGGAGGATGTCTCGTCTGTGAGTTACAGCA", [(2, "M"), (1, "I"), (7, "M"), (6, "D"), (26, "M")], "3C3T1^GCTCAG26", "AGGCTGGTAGCTCAGGGATGTCTCGTCTGTGAGTTACAGCA")]


@pytest.mark.parametrize("seq,cigar,md,expected", M1)
def test_M1_dna_read(seq, cigar, md, expected):
    rec = bamio.pack_record("r", seq, cigar, ref_id=0, pos=0, tags=bamio.tag_z("MD", md))
    assert O.dna_read(rec, 0) == expected


def test_M2_dna_reads():
    reads = [("AGGTTTTGTGAGTGGGACAGTTGCAGCAAAACACAACCATAGGTGCCCATCCACCAAGGCAGGCTCTCCATCTTGCTCAGAGTGGCTCTA", [(89, "M"), (1, "S")], 60246, "89"),
             ("TGTGAGTGGGACAGTTGCAGCAAAACACAACCATAGGTGCCCATCCACCAAGGCAGGCTCTCCATCTTGCTCAGAGTGGCTCCAGCCCTT", [(83, "M"), (7, "S")], 60252, "82T0"),
             ("CATAGGTGCCCATCCACCAAGGCAGGCTCTCCATCTTGCTCAGAGTGGCTCTAGCCCTTGCTGACTGCTGGGCAGGGAGAGAGCAGAGCT", [(90, "M")], 60283, "90"),
             ("CCCTTGCTGACTGCTGGGCAGGGAGAGAGCAGAGCTAACTTCCTCATGGGACCTGGGTGTGTCTGATCTGTGCACACCACTATCCAACCG", [(90, "M")], 60337, "90"),
             ("GAGGCTCCACCCTGGCCACTCTTGTGTGCACACAGCACAGCCTCTACTGCTACACCTGAGTACTTTGCCAGTGGCCTGGAAGCACTTTGT", [(90, "M")], 60432, "90")]
    # the reference's literal reads keep ref_id == -1 (BamRead ctor default)
    st = b"".join(bamio.pack_record(f"r{i+1}", s, c, pos=p, tags=bamio.tag_z("MD", md)) for i, (s, c, p, md) in enumerate(reads))
    assert O.dna_reads(st) == ("AGGTTTTGTGAGTGGGACAGTTGCAGCAAAACACAACCATAGGTGCCCATCCACCAAGGCAGGCTCTCCATCTTGCTCAGAGTGGCTCTAGCCCTTGCTGACTGCTGGGCAGGGAGAGAGCAGAGCTAACTTCCTCATGGGACCTGGGTGTGTCTGATCTGTGCACACCACTATCCAACCG"
                                "NNNNN" "GAGGCTCCACCCTGGCCACTCTTGTGTGCACACAGCACAGCCTCTACTGCTACACCTGAGTACTTTGCCAGTGGCCTGGAAGCACTTTGT")


# ------------------------------------------------------------------------------------------ P1 / P2

P1_SEQS = ["ATTATGGACATTGTTTCCGTTATCATCATCATCATCATCATCATCATTATCATC",
           "GACATTGTTTCCGTTATCATCATCATCATCATCATCATCATCATCATCATCATC",
           "ATTGTTTCCGTTATCATCATCATCATCATCATCATCATCATCATCATCATCACC",
           "TGTTTCCGTTATCATCATCATCATCATCATCATCATCATCATCATCATCACCAC",
           "TCCGTTATCATCATCATCATCATCATCATCATCATCATCATCATCACCACCACC",
           "GTTATCATCATCATCATCATCATCATCATCATCATCATCATCATCGTCACCCTG",
           "TCATCATCATCATAATCATCATCATCATCATCATCATCGTCACCCTGTGTTGAG",
           "TCATCATCATCGTCACCCTGTGTTGAGGACAGAAGTAATTTCCCTTTCTTGGCT",
           "TCATCATCATCATCACCACCACCACCCTGTGTTGAGGACAGAAGTAATATCCCT",
           "CACCACCACCCTGTGTTGAGGACAGAAGTAATTTCCCTTTCTTGGCTGGTCACC"]
P1_CIGARS = [[(54, "M")], [(54, "M")], [(50, "M"), (3, "I"), (1, "M")], [(54, "M")], [(54, "M")], [(54, "M")],
             [(2, "S"), (52, "M")], [(16, "M"), (15, "D"), (38, "M")], [(13, "M"), (3, "I"), (38, "M")], [(54, "M")]]
P1_POS = [758, 764, 767, 769, 773, 776, 785, 795, 804, 817]
P1_MD = ["47C6", "54", "51", "50T3", "46T7", "45A0C7", "11C24A0C14", "11A3T0^CATCATCATCACCAC38", "15T29T5", "2T45T5"]


def p1_stream():
    return b"".join(bamio.pack_record(f"r{i}", P1_SEQS[i], P1_CIGARS[i], ref_id=0, pos=P1_POS[i], tags=bamio.tag_z("MD", P1_MD[i]))
                    for i in range(10))


def test_P1_pileup_lowlevel():
    st = p1_stream()
    reference = O.dna_reads(st)
    pile = O.pileup(st, O.MODE_MAKE_PILEUP, use_md=True, start_from=796, end_at=849, skip_zero=False, want_entries=True)
    pile2 = O.pileup(st, O.MODE_MAKE_PILEUP, use_md=True, start_from=0, end_at=O.U64_MAX, skip_zero=False, want_entries=True)
    assert pile.pos[0] == 796 and pile.pileup_ref_id == 0
    assert pile2.pos[0] == 758 and pile2.n_columns == 113 and pile2.pos[-1] == 870
    i2 = {int(p): k for k, p in enumerate(pile2.pos)}
    expected_bases = {796: "CCCCCCAC", 805: "TCCCCCCCC", 806: "AAAAAAAGA", 821: "AAGG-AA", 826: "CCCCCC"}
    seen = set()
    for k in range(pile.n_columns):
        p = int(pile.pos[k])
        assert pile.cov[k] == pile2.cov[i2[p]]                                 # pileup.d:779
        assert chr(pile.refbase[k]) == reference[p - 758]                       # pileup.d:783
        if p in expected_bases:
            assert pile.bases(k) == expected_bases[p]
            seen.add(p)
        e = pile.entries[int(pile.entry_off[k]):int(pile.entry_off[k + 1])]
        cov = len(e)
        cig = lambda j: P1_CIGARS[int(e[j]["read_ordinal"])]
        if p == 810:   # cigar_after.front.type == 'D' for read coverage-2 (pileup.d:800)
            assert cig(cov - 2)[int(e[cov - 2]["cigar_op_index"]) + 1][1] == "D"
        if p == 817:   # cigar_before.back.type == 'I' (pileup.d:804)
            assert cig(cov - 2)[int(e[cov - 2]["cigar_op_index"]) - 1][1] == "I"
        if p == 821:   # cigar_operation.type == 'D' (pileup.d:808)
            assert cig(cov - 3)[int(e[cov - 3]["cigar_op_index"])][1] == "D"
    assert seen == set(expected_bases)
    assert pile.pos[-1] == 848                                                  # end_at is exclusive
    assert pile2.bases(i2[849]) == "TAT"                                        # the unreachable `case 849` arm, unclipped
    assert "".join(chr(c) for c in pile2.refbase) == reference                  # every column of the unclipped pileup


def test_P2_zero_coverage_gap():
    reads = [("CCCACATAGAAAGCTTGCTGTTTCTCTGTGGGAAGTTTTAACTTAGGTCAGCTT", 979, "54"),
             ("TAGAAAGCTTGCTGTTTCTCTGTGGGAAGTTTTAACTTAGGTTAGCTTCATCTA", 985, "42C7C3"),
             ("TTTTTCTTTCTTTCTTTGAAGAAGGCAGATTCCTGGTCCTGCCACTCAAATTTT", 1046, "54"),
             ("TTTCTTTCTTTCTTTGAAGAAGGCAGATTCCTGGTCCTGCCACTCAAATTTTCA", 1048, "54")]
    st = b"".join(bamio.pack_record(f"r{i+1}", s, [(54, "M")], ref_id=0, pos=p, tags=bamio.tag_z("MD", md)) for i, (s, p, md) in enumerate(reads))
    pile = O.pileup(st, O.MODE_MAKE_PILEUP, use_md=True, skip_zero=False)
    ref = O.dna_reads(st)
    assert "".join(chr(c) for c in pile.refbase) == ref
    assert ref.count("N") == 7 and int((pile.cov == 0).sum()) == 7
    # default skip_zero_coverage=true drops exactly the zero-coverage columns
    pile_s = O.pileup(st, O.MODE_MAKE_PILEUP, use_md=True, skip_zero=True)
    assert pile_s.n_columns == pile.n_columns - 7 and (pile_s.cov > 0).all()


# ------------------------------------------------------------------------------------------ P3 / P4 / P5

def test_P3_pileup_columns_ex1(data_dir):
    hdr, st = _stream(data_dir, "ex1_header.bam")
    pile = O.pileup(st, O.MODE_PILEUP_COLUMNS, want_entries=True)
    assert int((pile.ref == 0).sum()) == 1470 and int((pile.ref == 1).sum()) == 1567
    assert (np.diff(pile.ref) >= 0).all()
    k0, k1 = int(np.argmax(pile.ref == 0)), int(np.argmax(pile.ref == 1))
    assert pile.pos[k0] == 99 and pile.pos[k1] == 0
    rec_off = O.frame(st)

    def name(ordinal):
        f = O.fields_at(st, int(rec_off[ordinal]))
        return st[f.name_off:f.name_off + f.l_read_name - 1].decode()

    assert name(int(pile.entries[int(pile.entry_off[k0])]["read_ordinal"])) == "EAS56_57:6:190:289:82"
    assert name(int(pile.entries[int(pile.entry_off[k1])]["read_ordinal"])) == "B7_591:8:4:841:340"
    soa = O.decode_soa(st)
    ords = pile.entries["read_ordinal"].astype(np.int64)
    assert ((soa.flag_nc[ords] >> 16) & 4 == 0).all()                           # no unmapped read in any column
    col_of_entry = np.repeat(np.arange(pile.n_columns), np.diff(pile.entry_off).astype(np.int64))
    assert (soa.ref_id[ords] == pile.ref[col_of_entry]).all()
    # figures the surveyor derived independently -- a cross-check, not the authority
    assert pile.n_columns == 3037 and pile.sum_coverage == 113916 and pile.max_coverage == 64
    assert int(pile.nstart.sum()) == 3235
    assert pile.counts.sum(axis=0).tolist()[:4] == [39673, 22894, 19980, 31338] and int(pile.counts[:, 5].sum()) == 2


def test_P4_make_pileup_first_reference_only(data_dir):
    hdr, st = _stream(data_dir, "ex1_header.bam")
    pile = O.pileup(st, O.MODE_MAKE_PILEUP)
    assert pile.pileup_ref_id == 0 and (pile.ref == 0).all() and pile.n_columns == 1470


def test_P5_reads_starting_here(data_dir):
    hdr, st = _stream(data_dir, "illu_20_chunk.bam")
    pile = O.pileup(st, O.MODE_MAKE_PILEUP, use_md=True, want_entries=True)
    n = len(O.frame(st))
    assert n == 29 and int(pile.nstart.sum()) == 29
    starting = []
    for k in range(pile.n_columns):
        e = pile.entries[int(pile.entry_off[k]):int(pile.entry_off[k + 1])]
        starting += [int(x) for x in e["read_ordinal"][len(e) - int(pile.nstart[k]):]]
    assert starting == list(range(29))                                          # joiner(reads_starting_here) == bam.reads


# ------------------------------------------------------------------------------------------ fixtures, cross-checks

SURVEYOR = {"illu_20_chunk.bam": (153, 2266, 29), "ion_20_chunk.bam": (211, 77767, 476), "b7_295_chunk.bam": (802, 147535, 448),
            "mg1655_chunk.bam": (904, 192092, 450), "tags.bam": (28, 1668, 190)}


@pytest.mark.parametrize("name", sorted(SURVEYOR))
def test_fixture_column_totals(data_dir, name):
    _, st = _stream(data_dir, name)
    pile = O.pileup(st)
    assert (pile.n_columns, pile.sum_coverage, pile.max_coverage) == SURVEYOR[name]
    assert (pile.counts.sum(axis=1) == pile.cov).all()


def test_bins_bam_coverage_only(data_dir):
    _, st = _stream(data_dir, "bins.bam")
    pile = O.pileup(st)                     # reads carry CIGARs but empty SEQ -> soft ORC_ESEQ status
    assert pile.status == O.ORC_ESEQ
    assert pile.n_columns == 1114626 and pile.sum_coverage == 368228842 and pile.max_coverage == 580
    assert [int((pile.ref == r).sum()) for r in range(3)] == [514, 65536, 1048576]


@pytest.mark.parametrize("name", ["ex1_header.bam", "ion_20_chunk.bam", "b7_295_chunk.bam", "mg1655_chunk.bam", "illu_20_chunk.bam"])
@pytest.mark.parametrize("chunk_bytes", [2000, 20000, 16384000])
def test_chunked_equals_serial(data_dir, name, chunk_bytes):
    """pileupChunks decomposition (exact halo) concatenates to the serial pileupColumns output."""
    _, st = _stream(data_dir, name)
    a = O.pileup(st, want_entries=True)
    b = O.pileup_chunked(st, chunk_bytes=chunk_bytes, n_threads=3, want_entries=True)
    for f in ("ref", "pos", "cov", "nstart", "counts", "entry_off"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    for f in ("read_ordinal", "cigar_op_index", "cigar_op_offset", "query_offset", "cigar_op", "base", "qual"):
        assert np.array_equal(a.entries[f], b.entries[f]), f


def test_chunk_plan_drops_leading_chunks_that_cover_nothing():
    """PileupChunkRange's constructor (pileup.d:878-900) keeps walking while `_chunk_right_end > start_from` does not hold (:894-896):
    a leading chunk of reads that cover nothing at position 0 -- unmapped-flagged, or a CIGAR of soft clips only -- is never handed
    out, and the first chunk kept begins at ITS first read (no _prev_chunk, :888).  Found by scratch/fuzz_chunks.py (seeds 2930100,
    2931917): the device's plan followed the reference, this helper listed the empty chunk."""
    from biod_amd import bamio
    recs = [bamio.pack_record("u0", "ACGT", [(4, "S")], ref_id=0, pos=0),
            bamio.pack_record("r1", "ACGTACGT", [(8, "M")], ref_id=1, pos=6),
            bamio.pack_record("r2", "ACGTACGT", [(8, "M")], ref_id=1, pos=10)]
    st = b"".join(recs)
    assert O.chunk_plan(st, 16384000) == [(1, 2, 1, 6, 18)]
    # the same reads on ONE reference: the dropped chunk is no _prev_chunk -- the first chunk kept begins at its own first read
    recs = [bamio.pack_record("u0", "ACGT", [(4, "S")], ref_id=1, pos=0),
            bamio.pack_record("r1", "ACGTACGT", [(8, "M")], ref_id=1, pos=6),
            bamio.pack_record("r2", "ACGTACGT", [(8, "M")], ref_id=1, pos=10)]
    st = b"".join(recs)
    small = len(recs[0]) - 1                               # a chunk a read (a chunk closes once it holds more than block_size bytes)
    assert O.chunk_plan(st, small) == [(1, 1, 1, 6, 6), (2, 2, 1, 6, 18)]       # (a chunk ends at its last read's position while its reference goes on, :905-908)
    # a chunk that covers nothing LATER in the stream is handed out (popFront has no such test, :915-939)
    recs = [bamio.pack_record("r1", "ACGTACGT", [(8, "M")], ref_id=0, pos=6),
            bamio.pack_record("u0", "ACGT", [(4, "S")], ref_id=1, pos=0),
            bamio.pack_record("r2", "ACGTACGT", [(8, "M")], ref_id=2, pos=10)]
    st = b"".join(recs)
    assert O.chunk_plan(st, 16384000) == [(0, 0, 0, 6, 14), (1, 1, 1, 0, 0), (2, 2, 2, 10, 18)]


def test_bench_consumer_agrees(data_dir):
    _, st = _stream(data_dir, "ex1_header.bam")
    r1 = O.pileup_bench(st, 1)
    r4 = O.pileup_bench(st, 4, chunk_bytes=30000)
    assert (r1.n_columns, r1.sum_coverage, r1.max_coverage) == (3037, 113916, 64)
    assert (r4.n_columns, r4.sum_coverage, r4.max_coverage) == (3037, 113916, 64)


def test_leading_N_is_cigar_error():
    rec = bamio.pack_record("r", "ACGTACGTAC", [(5, "N"), (10, "M")], ref_id=0, pos=10)
    with pytest.raises(O.OracleError) as e:
        O.pileup(rec)
    assert e.value.code == O.ORC_ECIGAR


def test_unsorted_is_rejected():
    a = bamio.pack_record("a", "ACGT", [(4, "M")], ref_id=0, pos=50)
    b = bamio.pack_record("b", "ACGT", [(4, "M")], ref_id=0, pos=10)
    with pytest.raises(O.OracleError) as e:
        O.pileup(a + b)
    assert e.value.code == O.ORC_EUNSORTED


# ------------------------------------------------------------------------------------------ Z1

@pytest.mark.parametrize("name,exc", [("duplicated_block_size.bam", bamio.BgzfError), ("no_block_size.bam", bamio.BgzfError),
                                      ("wrong_extra_gzip_length.bam", bamio.BgzfError), ("wrong_bc_subfield_length.bam", bamio.BgzfError),
                                      ("corrupted_zlib_archive.bam", bamio.ZlibError)])
def test_Z1_negative_bgzf(data_dir, name, exc):
    with pytest.raises(exc):
        bamio.read_bam(os.path.join(data_dir, name))
