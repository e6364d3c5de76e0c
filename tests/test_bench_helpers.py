"""Host-side helpers of bench.py that run without a GPU: the BGZF file the file-E2E leg writes is a BAM BioD's own reader layout
accepts (bamio.read_bam restates bgzf/inputstream.d:54-199 and block.d:127-216 on the host), and the librccl probe answers with a list."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_file_e2e_writes_a_bam_the_reader_takes_back(tmp_path):
    import bench_e2e
    from biod_amd import bamio
    from biod_amd.synth import host_stream_np, make_cfg
    cfg = make_cfg(n_reads=3000, n_ref=3, ref_len=20000, read_len=100, profile=1, seed=4)
    st = host_stream_np(cfg)
    path = str(tmp_path / "w.bam")
    size = bench_e2e.write_bam(path, st, 3, threads=4, block=7000)
    assert size == os.path.getsize(path) and size < len(st)
    hdr, back = bamio.read_bam(path)
    assert [n for n, _ in hdr.refs] == ["chr1", "chr2", "chr3"]
    assert np.array_equal(np.frombuffer(back, np.uint8), st)


def test_rccl_probe_lists_what_is_mapped():
    import bench
    libs = bench.rccl_libraries_mapped()
    assert isinstance(libs, list) and all("librccl" in os.path.basename(p) for p in libs)


def test_generator_reference_property_holds_for_the_oracle():
    """What bench.py's whole-stream check of `use_md_tag` rests on (interior_parity.md_generator_reference_check): the generator writes its MD tags against
    one hash-defined reference, so on a stream without N skips the ORACLE's reference_base (PileupRangeUsingMdTag restated, pileup.d:522-654) is that
    reference at every covered column -- and on a stream with N skips it is not (dna(read) leaves the skipped bases out, md/reconstruct.d:199)."""
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import interior_parity as IP
    import oracle_lib as O
    from biod_amd.synth import host_stream, make_cfg
    for profile, expect_equal in ((0, True), (1, False)):
        cfg = make_cfg(n_reads=3000, n_ref=2, ref_len=30_000, read_len=150, profile=profile, seed=9)
        ora = O.pileup(host_stream(cfg), use_md=True, skip_zero=True)
        same = True
        for r in (0, 1):
            sel = ora.ref == r
            pos = ora.pos[sel].astype(np.int64)
            if not len(pos):
                continue
            ref = IP.synth_reference_chars(int(cfg.seed), r, int(pos.min()), int(pos.max() - pos.min()) + 1)
            same = same and bool((ora.refbase[sel] == ref[pos - pos.min()]).all())
        assert same == expect_equal, profile
