"""Differential fuzz of k_inflate on the GPU box: payloads of every compressibility, zlib at random level / strategy /
window / memLevel, BGZF blocks of random size -> device bytes == the payload; and garbage CDATA must be refused, not
hang.  usage: python scratch/fuzz_inflate.py [n_rounds] [seed0]"""
import os, sys, time, zlib, tempfile
sys.path.insert(0, '/root/repo')
import numpy as np
from biod_amd import _lib as L, bamio
from biod_amd.pileup import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
HDR = bamio.bam_header_bytes([("a", 1 << 20)])


def payload(rng, size):
    kind = int(rng.integers(0, 6))
    if kind == 0:
        return rng.integers(0, 256, size, dtype=np.uint8).tobytes()                       # incompressible
    if kind == 1:
        return bytes([int(rng.integers(0, 256))]) * size                                   # one long run
    if kind == 2:
        pat = rng.integers(0, 256, int(rng.integers(2, 300)), dtype=np.uint8).tobytes()
        return (pat * (size // len(pat) + 1))[:size]                                       # periodic
    if kind == 3:
        return rng.choice(np.frombuffer(b"ACGTN", np.uint8), size, p=[.3, .2, .2, .29, .01]).tobytes()   # few symbols: short codes
    if kind == 4:
        a = rng.integers(0, 256, size, dtype=np.uint8)
        a[rng.random(size) < 0.7] = 65                                                     # skewed: very long + very short codes
        return a.tobytes()
    words = [rng.integers(97, 123, int(rng.integers(2, 12)), dtype=np.uint8).tobytes() for _ in range(int(rng.integers(5, 400)))]
    out = bytearray()
    while len(out) < size:
        out += words[int(rng.integers(0, len(words)))] + b" "
    return bytes(out[:size])                                                               # text-like: far and near matches


def bgzf(chunk, rng):
    level = int(rng.integers(0, 10)); strat = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
    co = zlib.compressobj(level, zlib.DEFLATED, -int(rng.integers(9, 16)), int(rng.integers(1, 10)), strat)
    parts = []
    at = 0
    while at < len(chunk):                                                                  # several DEFLATE blocks in one BGZF block
        step = int(rng.integers(1, len(chunk) + 1))
        parts.append(co.compress(chunk[at:at + step]))
        if rng.random() < 0.3:
            parts.append(co.flush(zlib.Z_FULL_FLUSH if rng.random() < 0.5 else zlib.Z_SYNC_FLUSH))   # empty stored blocks in between
        at += step
    cdata = b"".join(parts) + co.flush()
    if len(cdata) + 26 > 65536:
        return None
    return (bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0]) + (len(cdata) + 25).to_bytes(2, "little") + cdata
            + (zlib.crc32(chunk) & 0xFFFFFFFF).to_bytes(4, "little") + len(chunk).to_bytes(4, "little"))


path = os.path.join(tempfile.gettempdir(), "fuzz_inflate.bam")
t0 = time.time(); bad = 0
for k in range(n):
    rng = np.random.default_rng(seed0 + k)
    try:
        # the payloads travel as the tag bytes of otherwise empty records (framing never looks inside tags)
        recs = [bamio.pack_record("p", "", [], ref_id=-1, pos=-1, flag=4,
                                  tags=payload(rng, int(rng.choice([int(rng.integers(1, 200)), int(rng.integers(200, 20000)), int(rng.integers(20000, 200000))]))))
                for _ in range(int(rng.integers(1, 30)))]
        st = b"".join(recs)
        blocks, at = [bgzf(HDR, rng)], 0
        while at < len(st):
            size = int(rng.choice([int(rng.integers(1, 200)), int(rng.integers(200, 20000)), int(rng.integers(20000, 65537))]))
            b = bgzf(st[at:at + size], rng)
            if b is None:                                          # did not fit a BGZF block at this level: halve
                size = max(1, size // 2); b = bgzf(st[at:at + size], rng)
                if b is None:
                    size = min(size, 30000); b = bgzf(st[at:at + size], np.random.default_rng(0))
            blocks.append(b); at += size
        open(path, "wb").write(b"".join(blocks) + bamio.BGZF_EOF)
        c = Context(0)
        c.load_bam_device(path); c.flush()
        assert c.staged_bytes().tobytes() == st
        assert c.num_records() == len(recs)
        c.close()
        # damage: garbage CDATA of the same length in one block -> refused (or, one time in 2^32, a CRC collision)
        raw = bytearray(b"".join(blocks) + bamio.BGZF_EOF)
        off = len(blocks[0])
        if len(blocks) > 1:
            j = int(rng.integers(1, len(blocks)))
            off = sum(len(x) for x in blocks[:j])
            clen = len(blocks[j]) - 26
            raw[off + 18:off + 18 + clen] = rng.integers(0, 256, clen, dtype=np.uint8).tobytes()
            open(path, "wb").write(bytes(raw))
            c = Context(0)
            try:
                c.load_bam_device(path)
                raise AssertionError("garbage CDATA accepted")
            except L.BiodGpuError:
                pass
            c.close()
    except Exception as e:      # noqa
        bad += 1
        print("MISMATCH at seed", seed0 + k, type(e).__name__, str(e)[:200], flush=True)
print(f"{n} rounds, {bad} failures, {time.time() - t0:.1f} s")
