"""Replays the batched use_md_tag failure of fuzz_gpu seed 6210148 (scratch/repro_md.bin + .json made on the CPU side)."""
import sys, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import oracle_lib as O
import test_gpu_parity as T
from biod_amd import _lib as L
from biod_amd.pileup import Context
st = open('/root/repo/scratch/repro_md.bin', 'rb').read()
params = json.load(open('/root/repo/scratch/repro_md.json'))
ctx = Context(0)
for skip, piece, batch in params:
    ora = O.pileup(st, use_md=True, skip_zero=skip)
    for b in ([batch] + [int(x) for x in sys.argv[1:]]):
        try:
            ref, pos, rb = T._batched_refbase(ctx, T._pieces(st, piece), b, skip)
            keep = ora.cov > 0
            print("skip", skip, "piece", piece, "batch", b, "ok, equal:", np.array_equal(pos, ora.pos[keep].astype(np.int64)) and np.array_equal(rb, ora.refbase[keep]))
        except L.BiodGpuError as e:
            print("skip", skip, "piece", piece, "batch", b, "ERROR", e.code)
            ctx.reset()
