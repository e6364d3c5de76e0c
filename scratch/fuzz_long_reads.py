"""Long reads (hundreds to thousands of CIGAR ops, indels every few dozen bases, N skips) at random depth, a few short reads in between: columns, every
entry, and reference_base (consistent tags, sometimes a damaged one) == the oracle.  The routes only long reads take: k4_ck_build, k4_acc2<LONG>, K5's
wave-settled ops, k6_verify_long, k6_ck_build / k6_emit_long.  usage: python scratch/fuzz_long_reads.py [rounds] [seed0]"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
import oracle_lib as O
import test_gpu_parity as T
from biod_amd import _lib as L, bamio
from biod_amd.pileup import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 900
ctx = Context(0)
bad = 0; t0 = time.time()
for k in range(n):
    rng = np.random.default_rng(seed0 + k)
    recs, pos = [], int(rng.integers(0, 3000))
    n_reads = int(rng.integers(8, 80)); step = int(rng.integers(20, 1500))
    dmg_every = int(rng.choice([0, 0, 5, 9]))
    for i in range(n_reads):
        if rng.random() < 0.15:      # a short read in between
            ln = int(rng.integers(20, 120)); seq = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, ln))
            recs.append(bamio.pack_record(f"s{i}", seq, [(ln, "M")], ref_id=0, pos=pos, tags=bamio.tag_z("MD", str(ln))))
        else:
            dmg = None if not dmg_every or i % dmg_every else str(rng.choice(["short", "long", "empty_del", "lower", "tail", "none"]))
            recs.append(T._long_indel_read(rng, i, pos, int(rng.integers(6, 600)), dmg))
        pos += int(rng.integers(0, step))
    st = b"".join(recs)
    try:
        T.stage(ctx, st)
        T.check_pileup(ctx, st, entries=True, window=int(rng.integers(500, 20000)))
        for skip in (True, False):
            T._check_refbase(ctx, st, skip_zero=skip, window=int(rng.integers(500, 20000)))
    except Exception as e:
        bad += 1; print("LONG seed", seed0 + k, "reads", n_reads, "len", len(st), type(e).__name__, str(e)[:200])
        ctx.reset()
print(n, "rounds,", bad, "failures,", round(time.time() - t0, 1), "s")
