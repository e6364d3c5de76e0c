"""Whole-shard passes of the window consumers (views, no facade copies): columns, reference bases, isSNP, MAQ with reference bases, pileupChunks."""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from biod_amd import _lib as L
from biod_amd.pileup import Context
from biod_amd.synth import scaled
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ctx = Context(0)
ctx.reset(); ctx.synth_device(scaled('cfg3', n)); ctx.flush()
info = ctx.pileup_run()
print('records', ctx.num_records(), 'positions', info.n_positions, flush=True)
def run(tag, **kw):
    ctx.timing_enable(True); ctx.timing_reset()
    t0 = time.time(); npos = 0; nw = 0
    for w in ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, window_positions=1 << 20, copy=False, **kw):
        npos += len(w.columns); nw += 1
    dt = time.time() - t0
    tm = ctx.timing_get()
    print(f"{tag}: {nw} windows, {npos} positions in {dt:.3f} s = {npos / dt / 1e6:.0f} Mpositions/s", {k: round(v['total_ms'], 1) for k, v in tm.items() if v['total_ms'] > 0.5}, flush=True)
run('columns')
run('reference_base', use_md_tag=True)
run('isSNP', use_md_tag=True, snp=(5, 2))
run('MAQ + reference_base', use_md_tag=True, maq=(0.17, 0.03, 13))
t0 = time.time(); nch = ctx.pileup_chunks_begin(block_size=16_384_000, window_positions=1 << 20); npos = 0
for k in range(nch):
    for w in ctx.chunk_windows(k, copy=False):
        npos += len(w.columns)
print(f"pileupChunks: {nch} chunks, {npos} positions in {time.time() - t0:.3f} s", flush=True)
