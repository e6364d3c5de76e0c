"""Two host threads, a ctx each, streaming files through the device at the same time (one ctx per (thread, GPU): include/biodgpu.h) while a third
keeps loading one-shot: columns == the one-shot result of each file.  usage: python scratch/stress_stream_threads.py [rounds]"""
import ctypes as C, os, sys, threading, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, R + "/tests"); sys.path.insert(0, R + "/oracle")
import numpy as np
from biod_amd import _lib as L, bamio
from biod_amd.pileup import Context
from biod_amd.synth import host_stream, make_cfg
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
files = []
for k in range(3):
    cfg = make_cfg(n_reads=120_000 + 40_000 * k, n_ref=2, ref_len=400_000, read_len=150, profile=1, seed=90 + k)
    st = host_stream(cfg)
    path = f"/tmp/stress_{k}.bam"
    bamio.write_bam(path, [("chr1", 1 << 28), ("chr2", 1 << 28)], [st], level=1) if False else None
    from concurrent.futures import ThreadPoolExecutor
    b = memoryview(st)
    with open(path, "wb") as f, ThreadPoolExecutor(8) as ex:
        f.write(bamio.bgzf_block(bamio.bam_header_bytes([("chr1", 1 << 28), ("chr2", 1 << 28)])))
        for blk in ex.map(lambda at: bamio.bgzf_block(bytes(b[at:at + 60000])), range(0, len(b), 60000)):
            f.write(blk)
        f.write(bamio.BGZF_EOF)
    c = Context(0); c.load_bam_device(path); c.flush(); info = c.pileup_run()
    files.append((path, int(info.n_positions), int(info.n_columns), int(info.sum_coverage))); c.close()
errors = []
def streamer(tid):
    try:
        c = Context(0)
        for r in range(rounds):
            path, npos, ncol, scov = files[(tid + r) % 3]
            c.set_compact_columns(bool((tid + r) & 1))
            got_pos = got_col = got_cov = 0
            for w in c.stream_bam_windows(path, [256 << 10, 4 << 20, 0][(tid + r) % 3], window_positions=65536):
                cov = w.columns["coverage"]; got_pos += len(cov); got_col += int(np.count_nonzero(cov)); got_cov += int(cov.sum(dtype=np.uint64))
            if (got_pos, got_col, got_cov) != (npos, ncol, scov): errors.append((tid, r, path, (got_pos, got_col, got_cov), (npos, ncol, scov)))
        c.close()
    except Exception as e:
        errors.append((tid, "exception", repr(e)))
def loader():
    try:
        c = Context(0)
        for r in range(rounds):
            path, npos, ncol, scov = files[r % 3]
            c.reset(); c.load_bam_device(path); c.flush(); info = c.pileup_run()
            if (int(info.n_positions), int(info.n_columns), int(info.sum_coverage)) != (npos, ncol, scov): errors.append(("loader", r))
        c.close()
    except Exception as e:
        errors.append(("loader", "exception", repr(e)))
t0 = time.time()
th = [threading.Thread(target=streamer, args=(k,)) for k in range(2)] + [threading.Thread(target=loader)]
for t in th: t.start()
for t in th: t.join()
print(rounds, "rounds x 2 streaming threads + 1 loading thread,", len(errors), "failures,", round(time.time() - t0, 1), "s", errors[:3])
