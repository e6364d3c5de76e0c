"""createIndex at scale on the GPU box: a 2M-read generator file (cfg3 read model), device index vs wall time.
Usage: python scratch/bai_scale.py [n_reads]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import oracle_lib as O
from biod_amd import bamio
from biod_amd.pileup import BamReader
from biod_amd.synth import CONFIGS, host_stream_np, make_cfg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
m = CONFIGS["cfg3"]
cfg = make_cfg(n_reads=n, n_ref=8, ref_len=60_000_000, read_len=m["read_len"], profile=m["profile"], seed=11)
st = host_stream_np(cfg)
path = "/tmp/bai_scale.bam"
t = time.time()
with open(path, "wb") as f:                      # 64 000-byte blocks cut anywhere: records straddle
    hb = bamio.bam_header_bytes([(f"chr{i + 1}", 1 << 28) for i in range(8)])
    f.write(bamio.bgzf_block(hb))
    b = st.tobytes()
    for at in range(0, len(b), 64000):
        f.write(bamio.bgzf_block(b[at:at + 64000]))
    f.write(bamio.BGZF_EOF)
print(f"wrote {os.path.getsize(path) / 1e6:.0f} MB in {time.time() - t:.1f} s")
t = time.time(); bam = BamReader(path, 0); print(f"load+flush {time.time() - t:.2f} s")
bam.ctx.timing_enable(True)
for k in range(3):
    bam.ctx.timing_reset()
    t = time.time(); raw = bam.ctx.bai_build(); dt = time.time() - t
    print(f"bai_build: {len(raw)} bytes in {dt * 1e3:.1f} ms ({n / dt / 1e6:.0f} Mreads/s)", {k: round(v["total_ms"], 2) for k, v in bam.ctx.timing_get().items()})
