"""The rows route with reads of five to sixteen CIGAR ops at medium scale (10^5 reads a round, depth 20 - 400): k4_plan_rest's rounds of four ops, slabs that fill
up, windows handed to the general kernel -- decode and every column vs the oracle, entries on every other round.
usage: python scratch/fuzz_rows_odd_scale.py [rounds] [seed0]"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import oracle_lib as O
import test_gpu_parity as T
import test_gpu_rows as R
from biod_amd import bamio
from biod_amd.pileup import Context


def shape(rng, lmax):
    n_ops = int(rng.integers(1, 5)) if rng.random() < 0.6 else int(rng.integers(5, 17))
    ops, left = [], lmax
    for k in range(n_ops):
        op = "MIDNS=X"[int(rng.choice(7, p=[0.5, 0.1, 0.12, 0.04, 0.06, 0.09, 0.09]))]
        if k == 0 and op in "DN":
            op = "M"
        if op in "MIS=X":
            ln = int(rng.integers(1, max(2, left // (n_ops - k) + 1))); left = max(1, left - ln)
        elif op == "N":
            ln = int(rng.choice([5, 60, 300, 1500]))
        else:
            ln = int(rng.integers(1, 12))
        ops.append((ln, op))
    if not any(op in "M=X" for _, op in ops):
        ops[0] = (max(1, ops[0][0]), "M")
    for k, (ln, op) in enumerate(ops):      # the first reference-consuming op is an M
        if op in "MDN=X":
            if op in "DN":
                ops[k] = (ln if ln < 60 else 30, "M")
            break
    q = sum(ln for ln, op in ops if op in "MIS=X")
    if q > 255:
        ops = [(min(ln, 15), op) if op in "MIS=X" else (ln, op) for ln, op in ops]
    return ops


n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 500
ctx = Context(0)
bad = 0; took = 0; t0 = time.time()
for k in range(n):
    rng = np.random.default_rng(seed0 + k)
    n_reads = int(rng.integers(60_000, 140_000)); step = int(rng.choice([1, 2, 4, 8])); lmax = int(rng.choice([100, 150, 250]))
    shapes = [shape(rng, lmax) for _ in range(300)]
    packed = []
    for ops in shapes:
        q = sum(ln for ln, op in ops if op in "MIS=X")
        seq = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, q))
        packed.append((ops, seq, bytes(rng.integers(2, 41, q).astype(np.uint8))))
    recs, pos = [], 100
    pick = rng.integers(0, len(packed), n_reads); steps = rng.integers(0, step + 1, n_reads)
    for i in range(n_reads):
        ops, seq, ql = packed[int(pick[i])]
        recs.append(bamio.pack_record("r%d" % i, seq, ops, ref_id=0, pos=pos, qual=ql))
        pos += int(steps[i])
    st = b"".join(recs)
    try:
        R.stage_rows(ctx, st)
        T.check_decode(ctx, st)
        info, ora = T.check_pileup(ctx, st, entries=(k % 2 == 0), window=int(rng.integers(2000, 60000)))
        took += int(R.ran_rows(ctx)); ctx.timing_enable(False)
        print("round", k, "reads", n_reads, "max coverage", info.max_coverage, "rows route", R.ran_rows(ctx), flush=True)
    except Exception as e:
        bad += 1; print("ODDSCALE seed", seed0 + k, type(e).__name__, str(e)[:300], flush=True); ctx.reset()
print(n, "rounds,", bad, "failures, rows route taken in", took, ",", round(time.time() - t0, 1), "s")
