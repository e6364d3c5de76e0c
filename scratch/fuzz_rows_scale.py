"""K4's rows route at medium scale on the GPU box: random generator configurations (200 k - 2 M reads, 36 - 250 bases, depth 5 - 250, one to four
contigs, M-only or mixed CIGARs) decoded by k12_lds and piled up through k4_acc2<ROWS>; every column vs the oracle.
usage: python scratch/fuzz_rows_scale.py [rounds] [seed0]"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import oracle_lib as O
import test_gpu_parity as T
import test_gpu_rows as R
from biod_amd import _lib as L
from biod_amd.pileup import Context
from biod_amd.synth import host_stream_np, make_cfg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
ctx = Context(0)
t0 = time.time(); bad = 0
for k in range(n):
    rng = np.random.default_rng(seed0 + k)
    n_reads = int(rng.integers(200_000, 2_000_000)); rl = int(rng.choice([36, 50, 76, 100, 150, 200, 250])); depth = float(rng.choice([5, 12, 30, 60, 120, 250]))
    n_ref = int(rng.integers(1, 5)); profile = int(rng.integers(0, 2))
    ref_len = max(2000, int(n_reads * rl / depth / n_ref))
    try:
        cfg = make_cfg(n_reads=n_reads, n_ref=n_ref, ref_len=ref_len, read_len=rl, profile=profile, seed=seed0 + k)
        st = host_stream_np(cfg).tobytes()
        R.stage_rows(ctx, st, n_ref)
        info = ctx.pileup_run()
        assert R.ran_rows(ctx)
        ctx.timing_enable(False)
        ora = O.pileup(st)
        assert (info.n_columns, info.sum_coverage, info.max_coverage) == (ora.n_columns, ora.sum_coverage, ora.max_coverage), "totals"
        dense = T.device_columns(ctx, info)
        exp = T.dense_from_oracle(ora, info.runs)
        for f in L.COLUMN_DTYPE.names:
            assert np.array_equal(dense[f], exp[f]), f
        print(f"round {k}: {n_reads} reads x {rl}, depth {depth:g}, {n_ref} contigs, profile {profile}: {info.n_columns} columns, max {info.max_coverage} ok", flush=True)
    except Exception as e:          # noqa
        bad += 1
        print('MISMATCH at seed', seed0 + k, type(e).__name__, str(e)[:300], flush=True)
print(f'{n} rounds, {bad} failures, {time.time() - t0:.1f} s')
