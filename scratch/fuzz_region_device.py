"""bgp_load_bam_region_device (the .bai's chunks, their blocks inflated on the device, the cut blocks gathered) against bgp_load_bam_region (the same chunks,
zlib on host threads) on random files -- BGZF blocks of random sizes and compression levels, records straddling them, tiny blocks, empty blocks -- indexed on
the device (createIndex), and random regions: the staged bytes, the record count, virtual offsets of a few records and the pileup's totals must agree.
usage: python scratch/fuzz_region_device.py [rounds] [seed0]"""
import os, sys, time, zlib
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, R + "/tests"); sys.path.insert(0, R + "/oracle")
import numpy as np
import test_gpu_parity as T
from biod_amd import _lib as L, bamio
from biod_amd.pileup import Context, BaiFile
from biod_amd.synth import host_stream
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
dev, host = Context(0), Context(0)
bad = ran = nreg = 0; t0 = time.time()
path = "/tmp/fuzz_region.bam"
for k in range(n):
    rng = np.random.default_rng(seed0 + k)
    cfg = T.make_cfg(n_reads=int(rng.integers(300, 40000)), n_ref=int(rng.integers(1, 4)), ref_len=int(rng.integers(20_000, 300_000)), read_len=int(rng.integers(50, 151)), profile=int(rng.integers(0, 2)), seed=int(rng.integers(1, 1 << 30)))
    try:
        st = host_stream(cfg)
    except L.BiodGpuError:
        continue
    payload = bamio.bam_header_bytes([(f"chr{i + 1}", 1 << 28) for i in range(cfg.n_ref)], text="@HD\tVN:1.0\n" * int(rng.integers(0, 2000))) + st
    mode = int(rng.integers(0, 3))
    with open(path, "wb") as f:
        at = 0
        while at < len(payload):
            m = int(rng.integers(1, 400)) if (mode == 0 and rng.random() < 0.1) else int(rng.integers(200, 65000))
            chunk = payload[at:at + m]; at += m
            if rng.random() < 0.03: f.write(bamio.BGZF_EOF)
            co = zlib.compressobj(int(rng.choice([0, 1, 6, 9])), zlib.DEFLATED, -15)
            cdata = co.compress(chunk) + co.flush()
            f.write(bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0]) + (len(cdata) + 25).to_bytes(2, "little") + cdata + (zlib.crc32(chunk) & 0xFFFFFFFF).to_bytes(4, "little") + len(chunk).to_bytes(4, "little"))
        f.write(bamio.BGZF_EOF)
    try:
        dev.reset(); dev.load_bam(path, 2); dev.flush()
        with open(path + ".bai", "wb") as f: f.write(dev.bai_build())
        bai = BaiFile(path)
        for _ in range(int(rng.integers(2, 7))):
            rid = int(rng.integers(0, cfg.n_ref))
            beg = int(rng.integers(0, cfg.ref_len)); end = beg + int(rng.choice([1, int(rng.integers(1, 300)), int(rng.integers(300, 30000)), cfg.ref_len]))
            dev.reset(); dev.load_bam_region_device(path, bai, rid, beg, end); dev.flush()
            host.reset(); host.load_bam_region(path, bai, rid, beg, end, int(rng.integers(1, 5))); host.flush()
            nr = dev.num_records()
            assert nr == host.num_records(), ("records", nr, host.num_records())
            assert dev.staged_bytes().tobytes() == host.staged_bytes().tobytes(), "staged bytes"
            for i in sorted({0, nr // 2, nr - 1} if nr else set()):
                assert dev.record_voffset(i) == host.record_voffset(i), ("voffset", i)
            if nr:
                a, b = dev.pileup_run(), host.pileup_run()
                assert (a.n_positions, a.n_columns, a.sum_coverage, a.max_coverage) == (b.n_positions, b.n_columns, b.sum_coverage, b.max_coverage), "pileup totals"
            nreg += 1
        ran += 1
    except Exception as e:
        bad += 1; print("REGION seed", seed0 + k, "len", len(st), type(e).__name__, str(e)[:300])
        dev.reset(); host.reset()
print(n, "rounds,", ran, "ran,", nreg, "regions,", bad, "failures,", round(time.time() - t0, 1), "s")
