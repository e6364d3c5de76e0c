"""use_md_tag across TINY batches (a few records each): bgp_batch_acquire ships whatever the buffer holds when less than an inflated
BGZF block of room is left -- with batch_bytes = 4096 every commit is a batch.  usage: python scratch/fuzz_tiny_batches.py [rounds] [seed0]"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
import oracle_lib as O
import test_gpu_parity as T
from biod_amd import _lib as L
from biod_amd.pileup import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
ctx = Context(0)
bad = 0; t0 = time.time()
for k in range(n):
    rng = np.random.default_rng(seed0 + k)
    st = T._md_stream(rng, int(rng.integers(30, 600)), **T._MD_CFGS[k % len(T._MD_CFGS)])
    for skip in (True, False):
        try:
            ora = O.pileup(st, use_md=True, skip_zero=skip)
        except O.OracleError:
            continue
        piece = int(rng.integers(200, 3000))
        try:
            ref, pos, rb = [], [], []
            for w in ctx.batched_windows_zero_copy(T._pieces(st, piece), 4096, window_positions=3000, use_md_tag=True, skip_zero_coverage=skip):
                nz = np.nonzero(w.columns["coverage"])[0]
                pos.append(w.first_position + nz.astype(np.int64)); rb.append(w.reference_base[nz])
            pos = np.concatenate(pos) if pos else np.zeros(0, np.int64); rb = np.concatenate(rb) if rb else np.zeros(0, np.uint8)
            keep = ora.cov > 0
            ok = np.array_equal(pos, ora.pos[keep].astype(np.int64)) and np.array_equal(rb, ora.refbase[keep])
            if not ok:
                bad += 1
                d = np.nonzero(rb != ora.refbase[keep])[0] if len(rb) == int(keep.sum()) else []
                print("MISMATCH seed", seed0 + k, "skip", skip, "piece", piece, "len", len(st), "positions equal", np.array_equal(pos, ora.pos[keep].astype(np.int64)), "first diffs", [(int(pos[i]), chr(rb[i]), chr(ora.refbase[keep][i])) for i in d[:5]], "n diff", len(d))
        except Exception as e:
            bad += 1; print("EXC seed", seed0 + k, "skip", skip, "piece", piece, type(e).__name__, e)
    # columns and per-read entries (stream-wide ordinals) across the same kind of batches: random CIGARs, every field of every entry
    st = T._random_stream(rng, int(rng.integers(30, 500)), quirks=False)
    try:
        ora = O.pileup(st, want_entries=True)
    except O.OracleError:
        continue
    piece = int(rng.integers(150, 2500))
    try:
        ref, pos, cols, ents, _ = T._covered(ctx.batched_windows_zero_copy(T._pieces(st, piece), 4096, want_entries=True, window_positions=int(rng.integers(50, 2000))), want_entries=True)
        T._assert_columns_equal_oracle(ref, pos, cols, ora)
        assert len(ents) == ora.n_columns
        for kk in range(ora.n_columns):
            oe = ora.entries[int(ora.entry_off[kk]):int(ora.entry_off[kk + 1])]
            for f in ("read_ordinal", "cigar_op_index", "cigar_op_offset", "query_offset", "base", "qual"):
                assert np.array_equal(ents[kk][f].astype(np.int64), oe[f].astype(np.int64)), (kk, f)
    except Exception as e:
        bad += 1; print("ENTRIES seed", seed0 + k, "piece", piece, type(e).__name__, str(e)[:200])
print(n, "rounds,", bad, "failures,", round(time.time() - t0, 1), "s")
