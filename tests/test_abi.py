"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol the header
declares, refuses to run without a GPU (no CPU fallback), and the host-only helpers behave."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as O
from biod_amd import _lib as L
from biod_amd import bamio
from biod_amd.synth import host_stream, make_cfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    n = C.c_int(0)
    try:
        hip = C.CDLL("libamdhip64.so")
        return hip.hipGetDeviceCount(C.byref(n)) == 0 and n.value > 0
    except OSError:
        return False


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "biodgpu.h")).read()
    declared = set(re.findall(r"\b(bgp_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "header parse found nothing"
    lib = L.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/biodgpu.h but not exported by libbiodgpu.so"
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)
    assert lib.bgp_abi_version() == 4          # 2: bgp_columns grew `maq`; 3: bgp_decode_fetch_range, bgp_pileup_begin_ref; 4: bgp_columns grew `columns8` (compact transport), bgp_stream_bam_begin


def test_integration_notes_name_every_entry_point():
    """INTEGRATION.md is where a maintainer looks up what replaces what: every function the header declares has its line there."""
    hdr = open(os.path.join(ROOT, "include", "biodgpu.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [n for n in sorted(set(re.findall(r"\b(bgp_[a-z0-9_]+)\s*\(", hdr))) if n not in doc]
    assert not missing, missing


def test_header_cites_reference_for_every_entry_point():
    hdr = open(os.path.join(ROOT, "include", "biodgpu.h")).read()
    assert len(re.findall(r"[a-z_/]+\.d:\d+", hdr)) >= 40


def test_struct_layouts_match_header():
    assert C.sizeof(L.Records) == 72 and C.sizeof(L.Columns) == 72 and C.sizeof(L.SynthCfg) == 40
    assert L.RECORD_DTYPE.itemsize == 32 and L.COLUMN_DTYPE.itemsize == 32 and L.ENTRY_DTYPE.itemsize == 16 and L.MAQ_CALL_DTYPE.itemsize == 8


def test_strerror_covers_all_codes():
    lib = L.lib()
    for code in range(-9, 2):
        assert lib.bgp_strerror(code)
    assert b"unknown" in lib.bgp_strerror(-99)


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a box without a GPU")
def test_no_cpu_fallback_without_gpu():
    h = C.c_void_p()
    rc = L.lib().bgp_init(0, C.byref(h))
    assert rc == L.BGP_EDEVICE and not h.value
    with pytest.raises(L.BiodGpuError):
        from biod_amd.pileup import Context
        Context(0)


def test_null_arguments_are_rejected():
    lib = L.lib()
    assert lib.bgp_init(0, None) == L.BGP_EINVAL
    assert lib.bgp_flush(None) == L.BGP_EINVAL
    assert lib.bgp_pileup_run(None, None) == L.BGP_EINVAL
    lib.bgp_destroy(None)


@pytest.mark.parametrize("profile,read_len", [(0, 150), (1, 150), (1, 100), (2, (400, 2500))])
def test_synthetic_generator_host(profile, read_len):
    """The generator's records are well formed, sorted, and its MD tags agree with its CIGARs
    (checked by rebuilding the reference from MD with the oracle)."""
    cfg = make_cfg(n_reads=3000, n_ref=3, ref_len=150_000, read_len=read_len, profile=profile, seed=11)
    st = host_stream(cfg)
    assert host_stream(cfg) == st                                   # deterministic
    s = O.decode_soa(st)
    assert len(s.rec_off) == 3000
    key = s.ref_id.astype(np.int64) << 32 | s.pos
    assert (np.diff(key) >= 0).all() and set(s.ref_id) == {0, 1, 2}
    assert (s.pos >= 0).all() and (s.end_pos <= 150_000).all()
    # any slice of the record index space reproduces the same bytes (records are pure functions of i)
    part = host_stream(cfg, 1000, 500)
    assert part == st[int(s.rec_off[1000]):int(s.rec_off[1500])]
    # MD consistency: reference rebuilt from MD is identical no matter which overlapping read provides it
    p = O.pileup(st, use_md=True, want_entries=True)
    assert p.status == 0
    if profile != 1:
        # (profile 1 has N skips: BioD's dna(read) leaves skipped bases out, md/reconstruct.d:199, so its
        # reference_base drifts on spliced reads -- reference behaviour, not a generator property)
        assert all(chr(c) in "ACGT" for c in p.refbase)
        ents, off = p.entries, p.entry_off
        col = np.repeat(np.arange(p.n_columns), np.diff(off).astype(np.int64))
        is_m = np.isin(ents["cigar_op"] & 15, [0, 7, 8])
        match = ents["base"][is_m] == p.refbase[col[is_m]]
        assert 0.97 < match.mean() < 0.999                           # ~1% substitutions + 0.1% N
    if profile == 1:
        kinds = {tuple(int(x) & 15 for x in s.cigar[int(a):int(b)]) for a, b in zip(s.cigar_off[:-1], s.cigar_off[1:])}
        assert kinds == {(0,), (4, 0), (0, 1, 0), (0, 2, 0), (0, 3, 0)}


def test_bamio_roundtrip(tmp_path):
    cfg = make_cfg(n_reads=500, n_ref=2, ref_len=30_000, read_len=100, profile=1, seed=3)
    st = host_stream(cfg)
    offs = O.frame(st)
    recs = [st[int(a):int(b)] for a, b in zip(offs, list(offs[1:]) + [len(st)])]
    path = str(tmp_path / "t.bam")
    bamio.write_bam(path, [("chr1", 30_000), ("chr2", 30_000)], recs, text="@HD\tVN:1.6\tSO:coordinate\n")
    hdr, back = bamio.read_bam(path)
    assert back == st and hdr.refs == [("chr1", 30_000), ("chr2", 30_000)]


def test_bai_open_rejects_reference_count_the_file_cannot_hold(tmp_path):
    """A .bai whose n_ref promises more references than its bytes can describe is refused with a status code
    (every reference takes >= 8 bytes); no C++ exception may cross the C ABI (baifile.d:126-169 throws a
    catchable exception here)."""
    import struct
    from biod_amd import _lib as L
    p = tmp_path / "x.bai"
    p.write_bytes(b"BAI\1" + struct.pack("<i", 0x7FFFFFFF))
    h = C.c_void_p()
    assert L.lib().bgp_bai_open(str(p).encode(), C.byref(h)) == L.BGP_EINVAL
    p.write_bytes(b"BAI\1" + struct.pack("<i", 2) + struct.pack("<ii", 0, 0))          # second reference missing
    assert L.lib().bgp_bai_open(str(p).encode(), C.byref(h)) < 0


def test_chunk_offsets_past_their_block_are_refused(data_dir):
    """StreamChunksSupplier (inputstream.d:254-330) with a chunk whose uoffset lies beyond the block's ISIZE: an
    index that does not belong to the file.  Status code, not stale bytes."""
    import os
    from biod_amd import _lib as L
    bam = os.path.join(data_dir, "ex1_header.bam").encode()
    n = C.c_uint64()
    for beg, end in ((0, 65000), (65000, (1 << 16) | 5)):
        chunk = (C.c_uint64 * 2)(beg, end)
        assert L.lib().bgp_bgzf_read_chunks(bam, chunk, 1, 1, None, 0, C.byref(n)) == L.BGP_EINVAL
    chunk = (C.c_uint64 * 2)(0, 100)                        # inside the first block (103 bytes): fine
    assert L.lib().bgp_bgzf_read_chunks(bam, chunk, 1, 1, None, 0, C.byref(n)) == 0 and n.value == 100


def test_file_shard_plan_covers_every_position_once():
    """shard.plan_file_shards over the reference's indexed fixtures: whatever the number of ranks, every position of every
    indexed reference lies in exactly one region, regions follow file order, and rank shares are contiguous."""
    import os
    from biod_amd import bamio, shard
    from biod_amd.pileup import BaiFile
    data = os.path.join(os.path.dirname(__file__), "golden", "data")
    for name in ("ex1_header.bam", "tags.bam", "bins.bam"):
        path = os.path.join(data, name)
        hdr, _ = bamio.read_bam(path)
        lens = [l for _, l in hdr.refs]
        bai = BaiFile(path)
        indexed = [r for r in range(min(bai.n_ref, len(lens))) if bai.shape(r)[2] > 0 and lens[r] > 0]
        for world in (1, 2, 3, 5, 8):
            plan = shard.plan_file_shards(bai, lens, world)
            assert len(plan) == world
            flat = [reg for rank in plan for reg in rank]
            assert flat == sorted(flat)                                   # file order, rank after rank
            per_ref = {}
            for r, b, e in flat:
                assert 0 <= b < e <= lens[r]
                per_ref.setdefault(r, []).append((b, e))
            assert sorted(per_ref) == indexed
            for r, iv in per_ref.items():
                assert iv[0][0] == 0 and iv[-1][1] == lens[r]
                assert all(iv[k][1] == iv[k + 1][0] for k in range(len(iv) - 1))
