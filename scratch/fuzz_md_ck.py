"""k6_ck_build's seeded checkpoints against the oracle: long reads (4 k - 40 k bases) whose MD tags are cut into tokens every way tests/test_gpu_parity.py's
_long_md_read_odd knows, a random kind a read, overlapping so that chunks begin anywhere inside a provider; reference_base of every column, both chain modes.
usage: python scratch/fuzz_md_ck.py [rounds] [seed0]"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
import test_gpu_parity as T
from biod_amd.pileup import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
KINDS = ["plain", "plain", "big_match", "long_dels", "del_last", "mm_last", "zero_last", "lower", "eq_n", "empty_del", "tail", "digits", "over", "short", "lead_zero"]
ctx = Context(0)
bad = 0; t0 = time.time()
for k in range(n):
    rng = np.random.default_rng(seed0 + k)
    recs, pos = [], int(rng.integers(0, 2000))
    n_reads = int(rng.integers(3, 14)); step = int(rng.choice([300, 2000, 9000, 30000]))
    clean = rng.random() < 0.4
    for i in range(n_reads):
        length = int(rng.integers(4100, 40000)) if rng.random() < 0.85 else int(rng.integers(30, 3000))
        kind = str(rng.choice(KINDS[:9] if clean else KINDS))
        recs.append(T._long_md_read_odd(rng, f"F{i}", 0, pos, length, kind))
        pos += int(rng.integers(1, step))
    st = b"".join(recs)
    try:
        T.stage(ctx, st)
        for skip in (True, False):
            T._check_refbase(ctx, st, skip_zero=skip, window=int(rng.integers(500, 30000)))
    except Exception as e:
        bad += 1; print("MDCK seed", seed0 + k, "reads", n_reads, "len", len(st), type(e).__name__, str(e)[:300], flush=True)
        ctx.reset()
print(n, "rounds,", bad, "failures,", round(time.time() - t0, 1), "s")
