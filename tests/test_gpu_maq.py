"""K8, the MAQ consensus model per column (bgp_pileup_maq; bio/std/hts/snpcallers/maq.d:388-486), against
oracle/maq_oracle.py on the reference's fixtures and on random piles.  Genotype, flags and base count are compared
exactly; the call quality to one float ulp of the two scores it is the difference of (the reference subtracts in 80-bit
`real` where the device has doubles: include/biodgpu.h)."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as O
from biod_amd import _lib as L
from biod_amd import bamio
from biod_amd.pileup import MaqSnpCaller, RecordStream

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import maq_oracle as M  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from biod_amd.pileup import Context
    c = Context(0)
    yield c
    c.close()


def check_maq(ctx, st, depcorr=0.17, eta=0.03, min_bq=13, use_md=True, window=777):
    ctx.reset(); ctx.push_inflated(st); ctx.flush()
    ctx.pileup_run()
    soa = O.decode_soa(st)
    ora = O.pileup(st, use_md=use_md, want_entries=True)
    tab = M.tables(depcorr, eta)
    exp = {}
    for k in range(ora.n_columns):
        e = ora.entries[int(ora.entry_off[k]):int(ora.entry_off[k + 1])]
        ordn = e["read_ordinal"].astype(np.int64)
        rb = M.column_read_bases(e["base"], e["qual"], (soa.bin_mq_nl[ordn] >> 8) & 0xFF, (soa.flag_nc[ordn] >> 20) & 1, min_bq)
        exp[(int(ora.ref[k]), int(ora.pos[k]))] = M.make_call(rb, int(ora.refbase[k]) if use_md else ord("N"), tab)
    seen = dict(calls=0, none=0, variant=0, het=0, exact=0, sub=0)
    for w in ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, use_md_tag=use_md, window_positions=window, maq=(depcorr, eta, min_bq)):
        assert w.maq is not None and len(w.maq) == len(w.columns)
        for k in np.nonzero(w.columns["coverage"])[0]:
            got, want = w.maq[k], exp[(w.ref_id, int(w.first_position) + int(k))]
            if want is None:
                assert int(got["flags"]) == 0 and int(got["genotype"]) == 0xFF
                seen["none"] += 1
                continue
            g, qual, r5, var, n, sub = want
            flags = L.BGP_MAQ_CALLED | (L.BGP_MAQ_VARIANT if var else 0) | (L.BGP_MAQ_SUBSAMPLED if sub else 0)
            assert (int(got["n_bases"]), int(got["reference_base5"])) == (n, r5), (w.ref_id, int(w.first_position) + int(k))
            tol = 4 * np.spacing(np.float32(max(abs(float(qual)), 1.0)) * 64)        # the scores are sums up to ~n * 63; an ulp of either moves their difference
            assert abs(float(got["quality"]) - float(qual)) <= tol, (float(got["quality"]), float(qual))
            if float(qual) > tol:                                                     # (a tie within an ulp may fall either way)
                assert (int(got["genotype"]), int(got["flags"])) == (g, flags), (w.ref_id, int(w.first_position) + int(k), int(got["genotype"]), g)
            seen["calls"] += 1; seen["variant"] += bool(var); seen["het"] += (g // 5 != g % 5); seen["sub"] += bool(sub)
            seen["exact"] += float(got["quality"]) == float(qual)
    return seen


@pytest.mark.parametrize("name", ["illu_20_chunk.bam", "ion_20_chunk.bam", "b7_295_chunk.bam", "mg1655_chunk.bam"])
def test_maq_fixtures(ctx, data_dir, name):
    _, st = bamio.read_bam(os.path.join(data_dir, name))
    seen = check_maq(ctx, st)
    assert seen["calls"] > 100 and seen["exact"] >= 0.999 * seen["calls"]
    seen = check_maq(ctx, st, depcorr=0.3, eta=0.01, min_bq=0, use_md=False, window=4096)
    assert seen["variant"] == seen["calls"]                                            # reference 'N': every call is a variant


def _random_pile(rng, n_reads, span, deep=False):
    recs, pos = [], 100
    for i in range(n_reads):
        pos += int(rng.integers(0, 2 if deep else 5))
        ln = int(rng.integers(20, 60))
        truth = "ACGT"[(pos // 7) % 4]
        seq = "".join(truth if rng.random() < 0.85 else "ACGTNRY="[int(rng.integers(0, 8))] for _ in range(ln))
        qual = bytes(int(x) for x in rng.choice([2, 5, 12, 13, 20, 30, 30, 40, 41, 63, 70], size=ln))
        ops = [(ln, "M")] if rng.random() < 0.7 else [(ln // 2, "M"), (3, "D"), (ln - ln // 2, "M")]
        md = str(ln) if len(ops) == 1 else f"{ln // 2}^ACG{ln - ln // 2}"
        recs.append(bamio.pack_record(f"q{i}", seq, ops, ref_id=0, pos=min(pos, 100 + span), mapq=int(rng.choice([0, 3, 20, 37, 60, 255])),
                                      flag=16 if rng.random() < 0.5 else 0, qual=qual, tags=bamio.tag_z("MD", md)))
    return b"".join(recs)


@pytest.mark.parametrize("seed", range(3))
def test_maq_random_piles(ctx, seed):
    """Both strands, qualities on both sides of every threshold (4, 13, 63, the mapping quality), ambiguity codes,
    deletions in the columns."""
    rng = np.random.default_rng(5100 + seed)
    st = _random_pile(rng, 400, 2000)
    seen = check_maq(ctx, st)
    assert seen["calls"] > 300 and seen["variant"] > 0 and seen["none"] >= 0
    check_maq(ctx, st, min_bq=30, window=64)


def test_maq_columns_deeper_than_255(ctx):
    rng = np.random.default_rng(77)
    st = _random_pile(rng, 1500, 300, deep=True)
    seen = check_maq(ctx, st, min_bq=5)
    assert seen["sub"] > 20
    # half of the bases below the bar: columns of 65 .. 128 entries with at most 64 that count (the two-pass pick of k8_maq64's
    # fast path) next to columns that stay with the wave-per-column routine
    seen = check_maq(ctx, _random_pile(np.random.default_rng(78), 900, 700, deep=True), min_bq=30, window=500)
    assert seen["calls"] > 300


def test_maq_snp_caller_facade(ctx):
    """MaqSnpCaller.findSNPs (maq.d:488-541) through the facade == the oracle's calls filtered the same way."""
    st = _random_pile(np.random.default_rng(31), 500, 1500)
    caller = MaqSnpCaller()
    assert (caller.depcorr, caller.eta, caller.minimum_call_quality, caller.minimum_base_quality) == (0.17, 0.03, 6.0, 13)     # maq.d:329-333
    got = [(c.position, c.genotype, c.reference_base) for c in caller.findSNPs(RecordStream(st, ctx))]
    soa = O.decode_soa(st)
    ora = O.pileup(st, use_md=True, want_entries=True)
    tab = M.tables(0.17, 0.03)
    want = []
    for k in range(ora.n_columns):
        e = ora.entries[int(ora.entry_off[k]):int(ora.entry_off[k + 1])]
        ordn = e["read_ordinal"].astype(np.int64)
        call = M.make_call(M.column_read_bases(e["base"], e["qual"], (soa.bin_mq_nl[ordn] >> 8) & 0xFF, (soa.flag_nc[ordn] >> 20) & 1, 13), int(ora.refbase[k]), tab)
        if call is not None and call[3] and float(call[1]) > 6.0:
            want.append((int(ora.pos[k]), "ACGTN"[call[0] // 5] + "|" + "ACGTN"[call[0] % 5], "ACGTN"[call[2]]))
    assert got == want and len(got) > 0


@pytest.mark.parametrize("name,batch", [("ex1_header.bam", 150_000), ("b7_295_chunk.bam", 40_000)])
def test_maq_calls_across_batches(ctx, data_dir, name, batch):
    """bgp_pileup_maq after bgp_batch_begin: the calls on a stream handed over in batches (reads carried over the seams)
    are the calls of the one-shot pileup without MD tags, position by position."""
    _, st = bamio.read_bam(os.path.join(data_dir, name))
    ctx.reset(); ctx.push_inflated(st); ctx.flush(); ctx.pileup_run()
    want = {}
    for w in ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, window_positions=500, maq=(0.17, 0.03, 13)):
        for k in np.nonzero(w.columns["coverage"])[0]:
            want[(w.ref_id, int(w.first_position) + int(k))] = w.maq[k].tobytes()
    got = {}
    pieces = [st[o:o + 7001] for o in range(0, len(st), 7001)]
    for w in ctx.batched_windows(pieces, batch_bytes=batch, window_positions=333, maq=(0.17, 0.03, 13)):
        assert w.maq is not None and len(w.maq) == len(w.columns)
        for k in np.nonzero(w.columns["coverage"])[0]:
            key = (w.ref_id, int(w.first_position) + int(k))
            assert key not in got
            got[key] = w.maq[k].tobytes()
    assert len(want) > 500 and got == want
