"""One round of scratch/fuzz_chunks.py, the two plans side by side where they differ.  usage: python scratch/chunks_repro.py seed"""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
import oracle_lib as O
import test_gpu_parity as T
from biod_amd.pileup import Context, RecordStream, pileupChunks
seed = int(sys.argv[1])
ctx = Context(0)
rng = np.random.default_rng(seed)
st = T._random_stream(rng, int(rng.integers(20, 1200)), quirks=False)
ora = O.pileup(st)
block = int(rng.choice([int(rng.integers(150, 1500)), int(rng.integers(1500, 30_000)), 16_384_000]))
chunks = pileupChunks(RecordStream(st, ctx), False, block, want_entries=False)
got = [(c.first_record, c.last_record, c.ref_id, c.start_position, c.end_position) for c in chunks]
plan = O.chunk_plan(st, block)
print("seed", seed, "block", block, "records", len(O.decode_soa(st).rec_off), "chunks device", len(got), "oracle", len(plan))
for i in range(max(len(got), len(plan))):
    g = got[i] if i < len(got) else None; p = plan[i] if i < len(plan) else None
    if g != p:
        print("  chunk", i, "device", g, "oracle", p)
        lo = max(0, i - 2)
        for j in range(lo, min(len(plan), i + 3)):
            print("     ctx", j, "device", got[j] if j < len(got) else None, "oracle", plan[j])
        break
soa = O.decode_soa(st)
i0 = max(0, (got[i][0] if i < len(got) else plan[i][0]) - 3)
for r in range(i0, min(len(soa.rec_off), i0 + 12)):
    size = int(soa.rec_off[r + 1] - soa.rec_off[r]) if r + 1 < len(soa.rec_off) else len(st) - int(soa.rec_off[r])
    print("     rec", r, "ref", int(soa.ref_id[r]), "pos", int(soa.pos[r]), "end", int(soa.end_pos[r]), "flag_nc", hex(int(soa.flag_nc[r])), "bytes", size)
