"""Host-side BAM/BGZF container helpers (pure Python + zlib).

These mirror the *container* layer of BioD that sits below the accelerated hot path
(``bio/core/bgzf/inputstream.d:54-199`` block header parse, ``bio/core/bgzf/block.d:127-216``
inflate + crc32 check, ``bio/std/hts/bam/reader.d:111-116,579-597`` magic/header/reference table,
``bio/core/bgzf/compress.d:43-104`` + ``constants.d:26-61`` for writing).  They are used by the
tests, the golden-vector generator and small tools; the production host pipeline is the C++
inflater inside ``libbiodgpu.so`` (``biod_amd/csrc/bgzf_host.cpp``).
"""
from __future__ import annotations

import struct
import zlib
from dataclasses import dataclass
from typing import Iterable, List, Sequence, Tuple

BGZF_MAGIC = b"\x1f\x8b\x08\x04"
BGZF_BLOCK_SIZE = 0xFF00          # constants.d:61
BGZF_MAX_BLOCK_SIZE = 65536       # constants.d:60
BGZF_EOF = bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])


class BgzfError(Exception):
    """Mirrors BioD's BgzfException (inputstream.d:41-43)."""


class ZlibError(Exception):
    """Mirrors BioD's ZlibException (bio/core/utils/zlib.d:247-273)."""


def iter_bgzf_blocks(buf: bytes) -> Iterable[Tuple[int, bytes]]:
    """Yield (compressed_offset, inflated_bytes) per BGZF block; raises on malformed blocks the
    way ``fillBgzfBufferFromStream`` (inputstream.d:54-199) and ``decompressBgzfBlock``
    (block.d:127-216) do."""
    off = 0
    n = len(buf)
    while off < n:
        start = off
        if n - off < 4:
            return
        if buf[off:off + 4] != BGZF_MAGIC:
            raise BgzfError(f"block at {start}: wrong BGZF magic")
        if n - off < 12:
            raise BgzfError(f"block at {start}: stream error")
        xlen = struct.unpack_from("<H", buf, off + 10)[0]
        p = off + 12
        end_extra = p + xlen
        bsize = None
        ln = 0
        while ln < xlen:
            if p + 4 > n:
                raise BgzfError(f"block at {start}: stream error")
            si1, si2, slen = struct.unpack_from("<BBH", buf, p)
            p += 4
            if si1 == 66 and si2 == 67:
                if slen != 2:
                    raise BgzfError(f"block at {start}: wrong BC subfield length: {slen}; expected 2")
                if bsize is not None:
                    raise BgzfError(f"block at {start}: duplicate field with block size")
                bsize = struct.unpack_from("<H", buf, p)[0]
            p += slen
            ln += 4 + slen
        if ln != xlen:
            raise BgzfError(f"block at {start}: total length of subfields ({ln}) != gzip_extra_length ({xlen})")
        if bsize is None:
            raise BgzfError(f"block at {start}: block size was not found in any subfield")
        cdata_size = bsize - xlen - 19
        if cdata_size > BGZF_MAX_BLOCK_SIZE or cdata_size < 0:
            raise BgzfError(f"block at {start}: bad compressed data size {cdata_size}")
        cdata = buf[end_extra:end_extra + cdata_size]
        if len(cdata) != cdata_size or end_extra + cdata_size + 8 > n:
            raise BgzfError(f"block at {start}: stream error: not enough data")
        crc, isize = struct.unpack_from("<II", buf, end_extra + cdata_size)
        off = end_extra + cdata_size + 8
        if isize == 0:
            yield start, b""
            continue
        if isize > BGZF_MAX_BLOCK_SIZE:
            raise BgzfError(f"block at {start}: uncompressed block size must be within {BGZF_MAX_BLOCK_SIZE}")
        try:
            d = zlib.decompressobj(-15)
            data = d.decompress(cdata)
            if not d.eof:
                raise ZlibError("inflate did not reach Z_STREAM_END")
        except zlib.error as e:
            raise ZlibError(str(e)) from e
        if len(data) != isize:
            raise ZlibError(f"block at {start}: inflated {len(data)} bytes, header says {isize}")
        if zlib.crc32(data) & 0xFFFFFFFF != crc:
            raise ZlibError(f"block at {start}: crc32 mismatch")
        yield start, data


def bgzf_inflate(buf: bytes) -> bytes:
    return b"".join(d for _, d in iter_bgzf_blocks(buf))


@dataclass
class BamHeader:
    text: str
    refs: List[Tuple[str, int]]
    records_offset: int  # offset of the first record in the inflated stream


def parse_bam_header(inflated: bytes) -> BamHeader:
    """reader.d:111-116 (magic), :579-584 (text), :587-597 + referenceinfo.d (reference table)."""
    if inflated[:4] != b"BAM\x01":
        raise ValueError("wrong BAM magic")
    (l_text,) = struct.unpack_from("<i", inflated, 4)
    text = inflated[8:8 + l_text].decode("latin-1")
    p = 8 + l_text
    (n_ref,) = struct.unpack_from("<i", inflated, p)
    p += 4
    refs = []
    for _ in range(n_ref):
        (l_name,) = struct.unpack_from("<i", inflated, p)
        p += 4
        name = inflated[p:p + l_name - 1].decode("latin-1")
        p += l_name
        (l_ref,) = struct.unpack_from("<i", inflated, p)
        p += 4
        refs.append((name, l_ref))
    return BamHeader(text, refs, p)


def read_bam(path: str) -> Tuple[BamHeader, bytes]:
    """Returns (header, record_stream) where record_stream is the inflated payload starting at the
    first record's 4-byte length prefix -- exactly what ``BamReadRange`` (readrange.d) consumes."""
    with open(path, "rb") as f:
        raw = f.read()
    inflated = bgzf_inflate(raw)
    hdr = parse_bam_header(inflated)
    return hdr, inflated[hdr.records_offset:]


# ---------------------------------------------------------------------------------- record building

_CIGAR_OPS = "MIDNSHP=X"
_NT16 = "=ACMGRSVTWYHKDBN"


def char2code(c: str) -> int:
    """bio/core/base.d:42-62."""
    if c in "0123":
        return 1 << int(c)
    c = c.upper()
    if c == "=":
        return 0
    k = _NT16.find(c)
    return k if k > 0 else 15


def reg2bin(beg: int, end: int) -> int:
    """bai/bin.d:82-92."""
    if end == beg:
        end = beg + 1
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def pack_record(name: str, seq: str, cigar: Sequence[Tuple[int, str]], *, ref_id: int = -1, pos: int = -1,
                mapq: int = 255, flag: int = 0, qual: bytes | None = None, tags: bytes = b"",
                next_ref_id: int = -1, next_pos: int = -1, tlen: int = 0) -> bytes:
    """One BAM record including its 4-byte block_size prefix.  Defaults follow the literal-read
    constructor ``BamRead(name, seq, cigar)`` (read.d:543-582): refID=-1, pos=-1, mapq=255,
    quals 0xFF; ``bin`` is recomputed from position like ``_recalculate_bin``."""
    nm = name.encode("latin-1") + b"\x00"
    cig = b"".join(struct.pack("<I", (ln << 4) | _CIGAR_OPS.index(op)) for ln, op in cigar)
    l_seq = len(seq)
    packed = bytearray((l_seq + 1) // 2)
    for i, ch in enumerate(seq):
        code = char2code(ch)
        packed[i >> 1] |= code << 4 if (i & 1) == 0 else code
    q = qual if qual is not None else b"\xff" * l_seq
    assert len(q) == l_seq
    covered = 0 if (flag & 4) else sum(ln for ln, op in cigar if op in "MDN=X")
    b = reg2bin(pos, pos + covered) if pos >= 0 else 4680
    bin_mq_nl = (b << 16) | (mapq << 8) | len(nm)
    flag_nc = (flag << 16) | len(cigar)
    body = struct.pack("<iiIIiiii", ref_id, pos, bin_mq_nl, flag_nc, l_seq, next_ref_id, next_pos, tlen)
    body += nm + cig + bytes(packed) + q + tags
    return struct.pack("<i", len(body)) + body


def tag_z(key: str, value: str) -> bytes:
    return key.encode() + b"Z" + value.encode() + b"\x00"


def bam_header_bytes(refs: Sequence[Tuple[str, int]], text: str = "") -> bytes:
    t = text.encode("latin-1")
    out = b"BAM\x01" + struct.pack("<i", len(t)) + t + struct.pack("<i", len(refs))
    for name, ln in refs:
        nm = name.encode("latin-1") + b"\x00"
        out += struct.pack("<i", len(nm)) + nm + struct.pack("<i", ln)
    return out


def bgzf_block(chunk: bytes, level: int = 1) -> bytes:
    """compress.d:43-104."""
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, zlib.Z_DEFAULT_STRATEGY)
    cdata = c.compress(chunk) + c.flush()
    total = 18 + len(cdata) + 8
    hdr = bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0]) + struct.pack("<H", total - 1)
    return hdr + cdata + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))


def write_bam(path: str, refs: Sequence[Tuple[str, int]], records: Iterable[bytes], text: str = "", level: int = 1) -> None:
    """Writes a BAM the way BioD's BamWriter lays it out: header in its own block(s), a record starts
    a new BGZF block when it does not fit in the current one (writer.d:257-267), EOF block last."""
    with open(path, "wb") as f:
        hb = bam_header_bytes(refs, text)
        for i in range(0, len(hb), BGZF_BLOCK_SIZE):
            f.write(bgzf_block(hb[i:i + BGZF_BLOCK_SIZE], level))
        cur = bytearray()
        for rec in records:
            if len(cur) + len(rec) > BGZF_BLOCK_SIZE and cur:
                f.write(bgzf_block(bytes(cur), level))
                cur = bytearray()
            cur += rec
            while len(cur) > BGZF_BLOCK_SIZE:       # a record larger than one block spans blocks
                f.write(bgzf_block(bytes(cur[:BGZF_BLOCK_SIZE]), level))
                del cur[:BGZF_BLOCK_SIZE]
        if cur:
            f.write(bgzf_block(bytes(cur), level))
        f.write(BGZF_EOF)
