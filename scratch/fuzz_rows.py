"""Differential fuzz of K4's rows route on the GPU box (k12_lds plan rows -> k4_acc2<ROWS> + k4_plan_rest): random sorted streams of reads with
at most sixteen CIGAR ops (most: four) of every type and up to 255 bases (so that the stream holds no long read and the route is taken), every depth from a
few to several hundred, runs that start anywhere, several references; decode, columns and (every third round) per-read entries vs the oracle.
usage: python scratch/fuzz_rows.py [n_rounds] [seed0]"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import pytest
import oracle_lib as O
import test_gpu_parity as T
import test_gpu_rows as R
from biod_amd import _lib as L
from biod_amd import bamio
from biod_amd.pileup import Context


def stream(rng, n_reads, quirks):
    recs, pos, ref = [], int(rng.integers(0, 5000)), 0
    step_hi = int(rng.choice([1, 2, 3, 6, 20, 200]))
    lmax = int(rng.choice([30, 100, 150, 255]))
    for i in range(n_reads):
        if rng.random() < 0.004 and ref < 3:
            ref += 1; pos = int(rng.integers(0, 3000))
        if rng.random() < 0.002:
            pos += int(rng.integers(1000, 200_000))          # an island (a new run behind 65 536 uncovered positions, or not)
        pos += int(rng.integers(0, step_hi + 1))
        n_ops = int(rng.integers(1, 5)) if rng.random() < 0.8 else int(rng.integers(5, 17))      # (one read in five: up to 16 ops, the planner's, four at a time)
        budget = int(rng.integers(1, lmax + 1))
        ops = []
        for k in range(n_ops):
            op = "MIDNSHP=X"[int(rng.choice(9, p=[0.42, 0.09, 0.09, 0.06, 0.09, 0.03, 0.03, 0.09, 0.10]))]
            if op in "MIS=X":
                ln = int(rng.integers(1, max(2, budget // (n_ops - k) + 1))); budget = max(1, budget - ln)
            elif op == "N":
                ln = int(rng.choice([1, 10, 200, 255, 256, 257, 1000, 5000, 70000]))
            else:
                ln = int(rng.integers(1, 40))
            if quirks and rng.random() < 0.02:
                ln = 0
            ops.append((ln, op))
        for k, (ln, op) in enumerate(ops):              # no leading D / N: the first reference-consuming op is an M (a leading N is BGP_ECIGAR, tested elsewhere)
            if op in "MDN=X":
                if op in "DN":
                    ops[k] = (max(min(ln, 60), 1), "M")
                break
        qlen = sum(ln for ln, op in ops if op in "MIS=X")
        if qlen > 255:
            ops = [(min(ln, 40), op) if op in "MIS=X" else (ln, op) for ln, op in ops]
            qlen = sum(ln for ln, op in ops if op in "MIS=X")
        l_seq = qlen
        if quirks and rng.random() < 0.04:
            l_seq = max(0, min(255, qlen + int(rng.integers(-8, 9))))
        seq = "".join("ACGTNRYK="[int(x)] for x in rng.choice(9, size=l_seq, p=[0.23, 0.23, 0.23, 0.23, 0.03, 0.01, 0.01, 0.01, 0.02]))
        flag = 4 if rng.random() < 0.02 else (16 if rng.random() < 0.5 else 0)
        recs.append(bamio.pack_record(f"q{i}" + "x" * int(rng.integers(0, 12)), seq, ops, ref_id=ref, pos=pos, flag=flag, qual=bytes(rng.integers(0, 60, l_seq).astype(np.uint8)),
                                      mapq=int(rng.integers(0, 61)), tags=bamio.tag_z("XY", "z" * int(rng.integers(0, 20)))))
    return b"".join(recs)


n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 90_000
ctx = Context(0)
t0 = time.time(); bad = 0; took = 0
for k in range(n):
    rng = np.random.default_rng(seed0 + k)
    try:
        st = stream(rng, int(rng.integers(20, 6000)), quirks=bool(k % 2))
        R.stage_rows(ctx, st)
        T.check_decode(ctx, st)
        try:
            O.pileup(st)
        except O.OracleError as e:
            assert e.code == O.ORC_ECIGAR
            with pytest.raises(L.BiodGpuError):
                ctx.pileup_run()
            continue
        T.check_pileup(ctx, st, entries=(k % 3 == 0), window=int(rng.integers(100, 9000)))
        took += int(R.ran_rows(ctx))
        ctx.timing_enable(False)
    except (Exception, pytest.fail.Exception) as e:          # noqa
        bad += 1
        print('MISMATCH at seed', seed0 + k, type(e).__name__, str(e)[:300], flush=True)
        if n == 1:
            import traceback; traceback.print_exc()
print(f'{n} rounds, {bad} failures, rows route taken in {took}, {time.time() - t0:.1f} s')
