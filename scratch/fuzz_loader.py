"""bgp_load_bam_device on files of several read chunks (the pipelined dispatcher: batches of blocks inflated while the file uploads):
equal to the host loader on the intact file; a flipped bit or a truncation is accepted or refused exactly as the host loader does.
usage: python scratch/fuzz_loader.py [rounds] [seed0]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from concurrent.futures import ThreadPoolExecutor
import numpy as np
from biod_amd import bamio, _lib as L
from biod_amd.pileup import Context
from biod_amd.synth import host_stream_np, scaled
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t0 = time.time(); fails = 0
dev = Context(0); host = Context(0)
def load(ctx, fn, path):
    ctx.reset()
    try:
        fn(path); ctx.flush()
        return True, ctx.staged_bytes().tobytes()
    except L.BiodGpuError as e:
        return False, e.code
for k in range(n):
    rng = np.random.default_rng(seed0 + k)
    os.environ["BGP_INFLATE_BATCH_MIN"] = str(int(rng.choice([16, 100, 500, 2048])))      # (read at every load: many small batches on both inflate streams, or few)
    cfg = scaled(("cfg2", "cfg3")[k & 1], int(rng.integers(250_000, 500_000)))
    st = host_stream_np(cfg); b = memoryview(st)
    cuts = [0]
    while cuts[-1] < len(b):
        cuts.append(min(len(b), cuts[-1] + int(rng.integers(20000, 65000))))
    path = "/tmp/fuzz_loader.bam"
    with open(path, "wb") as f, ThreadPoolExecutor(16) as ex:
        f.write(bamio.bgzf_block(bamio.bam_header_bytes([(f"chr{i + 1}", 1 << 28) for i in range(max(1, cfg.n_ref))])))
        for blk in ex.map(lambda i: bamio.bgzf_block(bytes(b[cuts[i]:cuts[i + 1]])), range(len(cuts) - 1)):
            f.write(blk)
        f.write(bamio.BGZF_EOF)
    raw = open(path, "rb").read()
    ok, got = load(dev, dev.load_bam_device, path)
    if not ok or got != st.tobytes():
        fails += 1; print("MISMATCH intact, seed", seed0 + k, ok, flush=True); continue
    for trial in range(4):
        bad = bytearray(raw)
        if trial < 3:
            at = int(rng.integers(100, len(raw) - 30)); bad[at] ^= 1 << int(rng.integers(0, 8)); what = f"bit flipped at {at}"
        else:
            cut = int(rng.integers(len(raw) // 4, len(raw) - 1)); bad = bad[:cut]; what = f"truncated at {cut}"
        p2 = "/tmp/fuzz_loader_bad.bam"
        open(p2, "wb").write(bytes(bad))
        h_ok, h_got = load(host, lambda p: host.load_bam(p, 8), p2)
        d_ok, d_got = load(dev, dev.load_bam_device, p2)
        if h_ok != d_ok or (h_ok and h_got != d_got):
            fails += 1; print("MISMATCH", what, "seed", seed0 + k, "host", h_ok if not h_ok else "ok", h_got if not h_ok else "", "device", d_ok, d_got if not d_ok else "", flush=True)
print(f"{n} rounds ({len(raw) >> 20} MB files, 4 damaged variants each), {fails} failures, {time.time() - t0:.1f} s")
