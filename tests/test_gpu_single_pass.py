"""bgp_flush_decode, and the ONE-PASS decode kernel behind BGP_FRAME_MODE=1 (k_stream_decode: framing by look-back over
LDS windows, record table and arenas in the same pass).  The product path is K0 + k12_lds (K0 / k1_rows / k2_unpack for
records too long for its windows); the one-pass kernel is an experiment that was measured and lost (LABNOTES.md):
it is compiled only by `make -C biod_amd/csrc EXPERIMENTS=1` into biod_amd/libbiodgpu_exp.so, and its half of these tests
runs when that library is the one loaded (`BIODGPU_SO=biod_amd/libbiodgpu_exp.so pytest tests/test_gpu_single_pass.py -m gpu`).
Both must give the oracle's bytes."""
import os

import numpy as np
import pytest

import oracle_lib as O
from biod_amd import _lib as L
from biod_amd import bamio
from biod_amd.synth import host_stream, scaled
from test_gpu_parity import FIXTURES, check_decode, check_pileup

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from biod_amd.pileup import Context
    c = Context(0)
    yield c
    c.close()


@pytest.fixture(params=["0", "1"], ids=["three_kernels", "one_pass"])
def mode(request):
    if request.param == "1" and not hasattr(L.lib(), "bgp_experiments_built"):
        pytest.skip("the one-pass kernel is not part of the product build (make EXPERIMENTS=1)")
    old = os.environ.get("BGP_FRAME_MODE")
    os.environ["BGP_FRAME_MODE"] = request.param
    yield request.param
    if old is None:
        del os.environ["BGP_FRAME_MODE"]
    else:
        os.environ["BGP_FRAME_MODE"] = old


def stage_decode(ctx, st, n_ref=0, what=L.BGP_FETCH_ALL):
    ctx.reset()
    if n_ref:
        ctx.set_n_ref(n_ref)
    ctx.push_inflated(st)
    ctx.flush(decode=what)


@pytest.mark.parametrize("name", FIXTURES)
def test_flush_decode_fixtures(ctx, data_dir, mode, name):
    hdr, st = bamio.read_bam(os.path.join(data_dir, name))
    stage_decode(ctx, st, len(hdr.refs))
    check_decode(ctx, st)
    if name != "bins.bam":
        check_pileup(ctx, st, entries=False)


@pytest.mark.parametrize("cfg_name,n", [("cfg3", 60_000), ("cfg2", 40_000), ("cfg5", 40)])
def test_flush_decode_many_chunks(ctx, mode, cfg_name, n):
    """Streams of hundreds of LDS windows (cfg5: records far longer than a window): every product against the oracle,
    and the kernel that ran is the one the mode names."""
    st = host_stream(scaled(cfg_name, n))
    ctx.timing_enable(True); ctx.timing_reset()
    stage_decode(ctx, st)
    names = set(k for k, v in ctx.timing_get().items() if v["launches"])
    ctx.timing_enable(False)
    assert ("k012_stream" in names) == (mode == "1"), names
    assert ("k0_chase" in names) == (mode == "0"), names
    check_decode(ctx, st)
    check_pileup(ctx, st, entries=False)


def test_flush_decode_partial_sections(ctx, mode):
    st = host_stream(scaled("cfg3", 5_000))
    soa = O.decode_soa(st)
    for what in (L.BGP_FETCH_SEQ, L.BGP_FETCH_QUAL | L.BGP_FETCH_CIGAR, 0):
        stage_decode(ctx, st, what=what)
        dec = ctx.decode_fetch(what)
        assert np.array_equal(dec.rec["end_pos"], soa.end_pos) and np.array_equal(dec.seq_off, soa.seq_off)
        if what & L.BGP_FETCH_SEQ:
            assert np.array_equal(dec.seq, soa.seq)
        if what & L.BGP_FETCH_QUAL:
            assert np.array_equal(dec.qual, soa.qual)
        if what & L.BGP_FETCH_CIGAR:
            assert np.array_equal(dec.cigar, soa.cigar)
    check_decode(ctx, st)                      # the sections not asked for at the flush are made on demand


def test_one_pass_ragged_and_malformed(ctx, mode):
    rec = bamio.pack_record("r", "ACGT", [(4, "M")], ref_id=0, pos=5)
    for st, n in [(b"", 0), (b"\x01\x02", 0), (rec, 1), (rec + b"\x07", 1), (rec + rec + b"\x00\x00\x00", 2)]:
        stage_decode(ctx, st)
        assert ctx.num_records() == n
    for st, code in [(rec + rec[:-1], L.BGP_ETRUNC)]:
        with pytest.raises(L.BiodGpuError) as e:
            stage_decode(ctx, st)
        assert e.value.code == code
    bad = bytearray(rec); bad[0:4] = (0xFFFFFFF0).to_bytes(4, "little")
    with pytest.raises(L.BiodGpuError) as e:
        stage_decode(ctx, bytes(bad))
    assert e.value.code == L.BGP_EINVAL
    bad = bytearray(rec); bad[20:24] = (5000).to_bytes(4, "little")       # l_seq larger than the record
    with pytest.raises(L.BiodGpuError) as e:
        stage_decode(ctx, bytes(bad) + rec)
    assert e.value.code == L.BGP_EINVAL
    # damage deep inside a stream of many windows: the chain stops there in both modes, with the same status
    st = bytearray(host_stream(scaled("cfg3", 30_000)))
    soa = O.decode_soa(bytes(st))
    off = int(soa.rec_off[20_000])
    st[off + 20:off + 24] = (1 << 29).to_bytes(4, "little")
    with pytest.raises(L.BiodGpuError) as e:
        stage_decode(ctx, bytes(st))
    assert e.value.code == L.BGP_EINVAL
    st2 = bytes(st[:off + 40])                                              # ... and a stream cut inside a record
    with pytest.raises(L.BiodGpuError) as e:
        stage_decode(ctx, st2)
    assert e.value.code == L.BGP_ETRUNC


def test_one_pass_names_that_are_not_printable(ctx, mode):
    """The boundary guesser wants printable read names (what tells a header from itself read four bytes early); a stream
    whose names are not is framed all the same -- by the chain of records alone."""
    recs = [bamio.pack_record("\x01\x02n%d" % i, "ACGTACGTAC", [(10, "M")], ref_id=0, pos=10 + i) for i in range(3000)]
    st = b"".join(recs)
    stage_decode(ctx, st)
    check_decode(ctx, st)
    check_pileup(ctx, st, entries=False)


def test_one_pass_records_smaller_and_larger_than_everything(ctx, mode):
    rng = np.random.default_rng(11)
    recs = []
    for i in range(4000):                                                    # 40-byte records next to 20 kb ones
        if i % 97 == 5:
            n = int(rng.integers(15_000, 30_000))
            recs.append(bamio.pack_record("big%d" % i, "".join("ACGT"[int(x)] for x in rng.integers(0, 4, n)), [(n, "M")], ref_id=0, pos=100 + i))
        else:
            recs.append(bamio.pack_record("s", "A", [(1, "M")], ref_id=0, pos=100 + i))
    st = b"".join(recs)
    stage_decode(ctx, st)
    check_decode(ctx, st)
    check_pileup(ctx, st, entries=False)


def test_batches_hints_and_chunks_through_the_one_pass_kernel(ctx, data_dir):
    """Batched staging (a stream that ends inside a record: the chain stops in front of it), caller hints and the
    region / chunk paths all sit on flush_core: the same tests, framing by the one-pass kernel."""
    import test_gpu_parity as T
    if not hasattr(L.lib(), "bgp_experiments_built"):
        pytest.skip("the one-pass kernel is not part of the product build (make EXPERIMENTS=1)")
    old = os.environ.get("BGP_FRAME_MODE")
    os.environ["BGP_FRAME_MODE"] = "1"
    try:
        T.test_batched_staging_edges(ctx)
        T.test_batched_staging_200k_reads_in_batches(ctx)
        T.test_first_record_hints_right_wrong_and_wild(ctx, data_dir)
        T.test_empty_and_ragged_streams(ctx)
        T.test_odd_cigars(ctx)
    finally:
        if old is None:
            del os.environ["BGP_FRAME_MODE"]
        else:
            os.environ["BGP_FRAME_MODE"] = old
