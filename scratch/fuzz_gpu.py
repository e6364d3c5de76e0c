"""One-off differential fuzz on the GPU box: random record streams through decode / pileup (entries) / reference_base,
device vs oracle.  usage: python scratch/fuzz_gpu.py [n_rounds] [seed0]"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
import pytest
import oracle_lib as O
import test_gpu_parity as T
from biod_amd import _lib as L
from biod_amd.pileup import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000
k0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # (replay: round k0 of an earlier run)
ctx = Context(0)
t0 = time.time(); bad = 0
for k in range(k0, k0 + n):
    rng = np.random.default_rng(seed0 + k)
    try:
        st = T._random_stream(rng, int(rng.integers(50, 3000)), quirks=bool(k % 2))
        T.stage(ctx, st)
        T.check_decode(ctx, st)
        ctx.reset(); ctx.push_inflated(st); ctx.flush(decode=L.BGP_FETCH_ALL)      # the decode at flush: 64 KiB segments, 16-bit side offsets, k12_lds
        T.check_decode(ctx, st)
        try:
            O.pileup(st)
            T.check_pileup(ctx, st, entries=(k % 3 == 0), window=int(rng.integers(100, 9000)))
        except O.OracleError as e:
            assert e.code == O.ORC_ECIGAR
        st = T._md_stream(rng, int(rng.integers(50, 1500)), **T._MD_CFGS[k % len(T._MD_CFGS)])
        T.stage(ctx, st)
        T.check_pileup(ctx, st, entries=False)
        for skip in (True, False):
            T._check_refbase(ctx, st, skip_zero=skip, window=int(rng.integers(50, 5000)))
        for skip in (True, False):                 # use_md_tag across batches of random size == the whole-stream chain
            try:
                ora = O.pileup(st, use_md=True, skip_zero=skip)
            except O.OracleError:
                continue
            ref, pos, rb = T._batched_refbase(ctx, T._pieces(st, int(rng.integers(1000, 50_000))), int(rng.integers(4096, 60_000)), skip)
            keep = ora.cov > 0
            assert np.array_equal(pos, ora.pos[keep].astype(np.int64)) and np.array_equal(rb, ora.refbase[keep]), ("batched md", skip)
        T.stage(ctx, st)
        info = ctx.pileup_run()
        if len(info.runs):
            lo, hi = int(info.runs[0]["start_pos"]), int(info.runs[0]["end_pos"])
            a0 = int(rng.integers(lo, hi + 1)); b0 = int(rng.integers(a0, hi + 60))
            T._check_refbase(ctx, st, mode=L.BGP_MAKE_PILEUP, skip_zero=bool(rng.integers(0, 2)), start_from=a0, end_at=b0)
        try:
            O.pileup(st, use_md=True)
            T._check_snp(ctx, st, int(rng.integers(0, 6)), int(rng.integers(0, 4)), with_entries=bool(k % 2))
        except O.OracleError:
            pass
        # region filter on a stacked stream: many reads per window (several scan rounds), random region
        deep = T._md_stream(rng, int(rng.integers(1500, 4000)), p_gap=0.0, nref=2, p_ref=0.0005)
        T.stage(ctx, deep)
        T.check_pileup(ctx, deep, entries=False)
        rid = int(rng.integers(0, 2)); rb = int(rng.integers(0, 9000)); re_ = rb + int(rng.integers(1, 4000))
        T.stage(ctx, deep)
        ctx.filter_region(rid, rb, re_)
        kept, exp_stream = T._expected_region(deep, rid, rb, re_)
        T._check_subset_decode(ctx.decode_fetch(), exp_stream)
        if kept:
            inf2 = ctx.pileup_run(); o2 = O.pileup(exp_stream)
            assert (inf2.n_columns, inf2.sum_coverage, inf2.max_coverage) == (o2.n_columns, o2.sum_coverage, o2.max_coverage)
        # createIndex: a random file with block edges anywhere, device bytes == oracle bytes
        import os, tempfile
        from biod_amd import bamio
        n_ref = int(rng.integers(2, 9))
        recs = T._index_stream(rng, int(rng.integers(20, 2500)), n_ref, quirks=bool(k % 2))
        path = os.path.join(tempfile.gettempdir(), "fuzz_index.bam")
        T._write_bgzf(path, bamio.bam_header_bytes([(f"c{i}", 1 << 29) for i in range(n_ref)]) + b"".join(recs), rng, eof=bool(k % 3), small=(k % 5 == 0))
        T._check_created_index(path)
    except (Exception, pytest.fail.Exception) as e:          # noqa
        bad += 1
        print('MISMATCH at seed', seed0 + k, type(e).__name__, str(e)[:300], flush=True)
        if n == 1:
            import traceback; traceback.print_exc()
print(f'{n} rounds, {bad} failures, {time.time() - t0:.1f} s')
