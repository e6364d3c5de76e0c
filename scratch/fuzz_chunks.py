"""pileupChunks (splitter.d:65-91, pileup.d:859-1015) on random streams with random block sizes down to a read or two a chunk:
the plan == the oracle's, the chunks' columns concatenated == pileupColumns.  usage: python scratch/fuzz_chunks.py [rounds] [seed0]"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
import oracle_lib as O
import test_gpu_parity as T
from biod_amd import _lib as L
from biod_amd.pileup import Context, RecordStream, pileupChunks
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
ctx = Context(0)
bad = 0; t0 = time.time()
for k in range(n):
    rng = np.random.default_rng(seed0 + k)
    st = T._random_stream(rng, int(rng.integers(20, 1200)), quirks=False)
    try:
        ora = O.pileup(st)
    except O.OracleError:
        continue
    block = int(rng.choice([int(rng.integers(150, 1500)), int(rng.integers(1500, 30_000)), 16_384_000]))
    s_from = 0; e_at = T.L_ULONG_MAX
    try:
        chunks = pileupChunks(RecordStream(st, ctx), False, block, want_entries=False)
        plan = O.chunk_plan(st, block)
        assert [(c.first_record, c.last_record, c.ref_id, c.start_position, c.end_position) for c in chunks] == plan, "plan"
        ref, pos, cols, _ = T._chunk_columns(chunks)
        T._assert_columns_equal_oracle(ref, pos, cols, ora)
    except Exception as e:
        bad += 1; print("CHUNKS seed", seed0 + k, "block", block, "len", len(st), type(e).__name__, str(e)[:160])
print(n, "rounds,", bad, "failures,", round(time.time() - t0, 1), "s")
