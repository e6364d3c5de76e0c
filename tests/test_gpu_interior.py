"""Interior-shard parity at FULL size: with a whole BASELINE configuration resident on the device, record ranges anywhere
in the stream -- past 4 / 8 / 16 GiB of stream, arena and column offsets, across contig boundaries, at the deepest column,
at the end -- are regenerated on the host, run through the oracle and compared bit for bit: record rows, the records'
own bytes, SEQ / QUAL / CIGAR arena slices, every column the range determines, per-read entries on two ranges
(tests/interior_parity.py).  Sums cannot see a base counted at the wrong position or an offset that wrapped; this does."""
import numpy as np
import pytest

import interior_parity as IP
from biod_amd import _lib as L
from biod_amd.pileup import Context
from biod_amd.synth import CONFIGS, host_stream, make_cfg, scaled

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def test_range_fetch_and_one_reference_pileup_small(ctx):
    """bgp_decode_fetch_range / bgp_pileup_begin_ref on a stream small enough to compare whole: every slice of the
    table is the corresponding slice of bgp_decode_fetch, and makePileup over one reference is that reference's part of
    pileupColumns."""
    cfg = make_cfg(n_reads=6000, n_ref=3, ref_len=40_000, read_len=100, profile=1, seed=21)
    st = host_stream(cfg)
    ctx.reset(); ctx.push_inflated(st); ctx.flush()
    full = ctx.decode_fetch()
    for first, count in ((0, 1), (0, 6000), (1999, 2), (4000, 1500), (5999, 1), (6000, 0)):
        d = ctx.decode_fetch_range(first, count, L.BGP_FETCH_ALL, stream=True)
        assert np.array_equal(d.rec, full.rec[first:first + count])
        assert np.array_equal(d.seq_off, full.seq_off[first:first + count + 1]) and np.array_equal(d.cigar_off, full.cigar_off[first:first + count + 1])
        a, b = int(full.seq_off[first]), int(full.seq_off[first + count])
        assert np.array_equal(d.seq, full.seq[a:b]) and np.array_equal(d.qual, full.qual[a:b])
        assert np.array_equal(d.cigar, full.cigar[int(full.cigar_off[first]):int(full.cigar_off[first + count])])
        if count:
            s0 = int(full.rec["rec_off"][first]); s1 = int(full.rec["rec_off"][first + count]) if first + count < 6000 else len(st)
            assert d.data.tobytes() == st[s0:s1]
    with pytest.raises(L.BiodGpuError):
        ctx.decode_fetch_range(5999, 2)
    whole = {}
    for w in ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, skip_zero_coverage=False, window_positions=3000):
        whole.setdefault(w.ref_id, []).append(w.columns)
    for r in (0, 1, 2):
        got = np.concatenate([w.columns for w in ctx.pileup_windows(L.BGP_MAKE_PILEUP, ref_id=r, skip_zero_coverage=False, window_positions=7000)])
        assert np.array_equal(got, np.concatenate(whole[r]))
    assert list(ctx.pileup_windows(L.BGP_MAKE_PILEUP, ref_id=7)) == []
    # ranges of the same stream through the interior checker, entries included (halo, contig boundary, the end)
    info = ctx.pileup_run()
    for first in (0, 700, 1900, 3990, 5500):
        IP.check_range(ctx, cfg, first, 500, entries=True, entry_positions=300)


@pytest.mark.parametrize("name", ["cfg2", "cfg3", "cfg5"])
def test_full_size_interior_parity(ctx, name):
    import torch
    n = CONFIGS[name]["n_reads"]
    free, _ = torch.cuda.mem_get_info(0)
    if n * (1000 if name != "cfg5" else 600_000) > free * 0.9:
        pytest.skip("not enough free HBM for the full configuration")
    cfg = scaled(name, n)
    ctx.reset()
    ctx.synth_device(cfg)
    ctx.flush(decode=L.BGP_FETCH_ALL)                 # the bench's decode path (K0 + k12_lds; cfg5: k1_rows + k2_unpack)
    info = ctx.pileup_run()
    res = IP.run(ctx, cfg, info, 20_000 if name != "cfg5" else 200, torch, torch.device("cuda", 0), want=16)
    assert res["all_equal"], res["failed"]
    assert res["ranges"] >= 16 and res["entries"] > 0, res
    # use_md_tag against the oracle at the start of every contig (reference_base far into the stream, a provider chain per reference)
    per = IP.md_contig_prefix_check(ctx, cfg, 20_000 if name != "cfg5" else 200)
    assert len(per) == int(cfg.n_ref) and all(e["columns"] > 0 for e in per), per
    if name != "cfg3":      # no N skips: every covered column of the whole stream must show the generator's own reference, whichever read provides it
        whole = IP.md_generator_reference_check(ctx, cfg)
        assert whole["equal"] and whole["covered_columns"] == info.n_columns, (whole, info.n_columns)
