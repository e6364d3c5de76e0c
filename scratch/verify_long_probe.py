"""Does k6_verify_long flag the damaged long tags (the run is redone by the event loop) and leave consistent ones alone?  BGP_VERBOSE prints it."""
import os, sys
os.environ["BGP_VERBOSE"] = "1"
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
import test_gpu_parity as T
from biod_amd import _lib as L
from biod_amd.pileup import Context
ctx = Context(0)
for damage in [None, "short", "long", "empty_del", "lower", "tail", "digits", "none"]:
    rng = np.random.default_rng(991)
    recs, pos = [], 100
    for i in range(60):
        recs.append(T._long_indel_read(rng, i, pos, int(rng.integers(110, 160)), damage if (damage and i % 7 == 3) else None))
        pos += int(rng.integers(50, 400))
    st = b"".join(recs)
    T.stage(ctx, st)
    sys.stderr.write(f"--- damage={damage}\n"); sys.stderr.flush()
    try:
        n = sum(len(w.columns) for w in ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, True, 0, (1 << 64) - 1, True, window_positions=5000))
        sys.stderr.write(f"    ok, {n} positions\n")
    except L.BiodGpuError as e:
        sys.stderr.write(f"    error {e.code}\n")
