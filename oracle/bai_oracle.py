"""CPU restatement of BioD's BAI / region-query layer -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(biod_amd/, libbiodgpu.so) never does.  Pure Python + the stdlib zlib, small inputs only.

Follows, function by function:
  parse_bai            bio/std/hts/bam/baifile.d:126-169      (section 4.2 of the SAM/BAM spec)
  bai_path             bio/std/hts/bam/baifile.d:92-121
  min_offset           bio/std/hts/bam/baifile.d:77-82        (Index.getMinimumOffset)
  can_overlap          bio/std/hts/bam/bai/bin.d:55-77        (Bin.canOverlapWith, uint arithmetic kept)
  reg2bin              bio/std/hts/bam/bai/bin.d:82-92
  get_chunks           bio/std/hts/bam/randomaccessmanager.d:209-244 (appendChunks, sort, nonOverlappingChunks)
  non_overlapping      bio/core/utils/algo.d:95-147
  read_chunks          bio/core/bgzf/inputstream.d:254-330    (StreamChunksSupplier: which bytes a chunk list yields)
  region_filter        bio/std/hts/bam/randomaccessmanager.d:366-462 (BamReadFilter, one region)
  records_with_offsets readrange.d:51-68 + bgzf/inputstream.d:425-452,497-530 (reads!withOffsets)
  build_index          bio/std/hts/bam/bai/indexing.d:56-350  (IndexBuilder.put / finish, check_bins off)
  serialise_index      indexing.d:164-218,276-277,347         (bins ascending; the reference's order is hash order)
The reader half is pinned by tests/test_oracle_bai.py against the reference's own checks: test/unittests.d:145-185
(bam["large"][beg .. end] == naive filter for its 40 regions on bins.bam) and :192-206 (names of the first
read of every reference found through startVirtualOffset + getReadAt).
The WRITER half (build_index) is PARITY UNPINNED: the reference has no test, golden vector or fixture for
createIndex (test/data's .bai files were written by another tool and differ in two spots).  It is checked by hand
cases and through the reference's index consumers: the checks above must hold on the index it writes.
"""
import os
import struct
import zlib

MAX_BIN_ID = 37449          # bio/std/hts/bam/constants.d:35
WINDOW = 16384              # constants.d:37
U32 = 0xFFFFFFFF


def bai_path(filename):
    if filename.endswith(".bai"):
        return filename
    first = filename + ".bai"
    dot = filename.rfind(".")
    second = (filename[:dot + 1] if dot >= 0 else "") + "bai"
    if os.path.exists(first):
        return first
    if os.path.exists(second):
        return second
    raise FileNotFoundError(f"searched for {first} or {second}, found neither")


def parse_bai(filename):
    """-> list over references of dict(bins=[(id, [(beg, end), ...]), ...] in file order, ioffsets=[...])"""
    b = open(bai_path(filename), "rb").read()
    assert b[:4] == b"BAI\1", "Invalid file format: expected BAI\\1"
    off = 4
    (n_ref,) = struct.unpack_from("<i", b, off); off += 4
    out = []
    for _ in range(n_ref):
        (n_bin,) = struct.unpack_from("<i", b, off); off += 4
        bins = []
        for _ in range(n_bin):
            bid, n_chunk = struct.unpack_from("<Ii", b, off); off += 8
            ch = []
            for _ in range(n_chunk):
                ch.append(struct.unpack_from("<QQ", b, off)); off += 16
            bins.append((bid, ch))
        (n_intv,) = struct.unpack_from("<i", b, off); off += 4
        io = list(struct.unpack_from(f"<{n_intv}Q", b, off)); off += 8 * n_intv
        out.append(dict(bins=bins, ioffsets=io))
    return out


def _i32(x):
    x &= U32
    return x - (1 << 32) if x >> 31 else x


def min_offset(index, position):
    pos = max(0, _i32(position))
    i = min(pos // WINDOW, len(index["ioffsets"]) - 1)
    return 0 if i == -1 else index["ioffsets"][i]


def can_overlap(bin_id, begin, end):
    if bin_id == 0:
        return True
    if bin_id > MAX_BIN_ID:
        return False
    begin, end = _i32(begin), _i32(end)
    if begin < 0:
        begin = 0
    magic = 4681
    b, e = begin >> 14, end >> 14
    while True:
        delta = (bin_id - magic) & U32                   # uint - int -> uint in D
        if (b & U32) <= delta <= (e & U32):              # int operands are converted to uint
            return True
        magic >>= 3
        if magic == 0:
            return False
        b >>= 3
        e >>= 3


def reg2bin(beg, end):
    if end == beg:
        end = beg + 1
    end -= 1
    if beg >> 14 == end >> 14: return (((1 << 15) - 1) // 7 + (beg >> 14)) & 0xFFFF
    if beg >> 17 == end >> 17: return (((1 << 12) - 1) // 7 + (beg >> 17)) & 0xFFFF
    if beg >> 20 == end >> 20: return (((1 << 9) - 1) // 7 + (beg >> 20)) & 0xFFFF
    if beg >> 23 == end >> 23: return (((1 << 6) - 1) // 7 + (beg >> 23)) & 0xFFFF
    if beg >> 26 == end >> 26: return (((1 << 3) - 1) // 7 + (beg >> 26)) & 0xFFFF
    return 0


def non_overlapping(chunks):
    out, i = [], 0
    while i < len(chunks):
        fb, fe = chunks[i]; i += 1
        while i < len(chunks) and fe >= chunks[i][0]:
            fe = max(fe, chunks[i][1]); i += 1
        out.append((fb, fe))
    return out


def get_chunks(bai, ref_id, beg, end):
    assert 0 <= ref_id < len(bai), "Invalid reference sequence index"
    index = bai[ref_id]
    mo = min_offset(index, beg)
    ch = []
    for bid, chunks in index["bins"]:
        if not can_overlap(bid, beg, end):
            continue
        for cb, ce in chunks:
            if ce > mo:
                ch.append((max(cb, mo), ce))
    ch.sort()
    return non_overlapping(ch)


def _block_at(f, off):
    """-> (isize, payload bytes, next offset) of the BGZF block at file offset off"""
    assert f[off:off + 4] == b"\x1f\x8b\x08\x04"
    (xlen,) = struct.unpack_from("<H", f, off + 10)
    p, xend, bsize = off + 12, off + 12 + xlen, None
    while p < xend:
        si1, si2, slen = struct.unpack_from("<BBH", f, p)
        if si1 == 66 and si2 == 67:
            (bsize,) = struct.unpack_from("<H", f, p + 4)
        p += 4 + slen
    cdata = bsize - xlen - 19
    data = zlib.decompress(f[xend:xend + cdata], -15)
    crc, isize = struct.unpack_from("<II", f, xend + cdata)
    assert len(data) == isize and (zlib.crc32(data) & U32) == crc
    return isize, data, xend + cdata + 8


def read_chunks(bam_path, chunks):
    f = open(bam_path, "rb").read()
    q = [list(c) for c in chunks]
    out = bytearray()
    at = 0

    def move_to_next(at):                                  # inputstream.d:259-273
        if at >= len(q):
            return at
        beg = q[at][0]
        i = at + 1
        while i < len(q) and (q[i][0] >> 16) <= (q[at][0] >> 16):
            i += 1
        at = i - 1
        q[at][0] = beg
        return at

    at = move_to_next(at)
    while at < len(q):
        cb, ce = q[at]
        off = cb >> 16
        while off < len(f):
            isize, data, nxt = _block_at(f, off)
            skip_start = (cb & 0xFFFF) if off == (cb >> 16) else 0
            skip_end = isize - (ce & 0xFFFF) if off == (ce >> 16) else 0
            if isize - skip_start - skip_end > 0:
                out += data[skip_start:isize - skip_end]
            if off >= (ce >> 16):
                break
            off = nxt
        at = move_to_next(at + 1)
    return bytes(out)


def frame(stream):
    """[(offset, ref_id, pos, end_pos, name)] of the records in an inflated stream (readrange.d:118-173 + read.d)."""
    out, off = [], 0
    while off + 4 <= len(stream):
        (bs,) = struct.unpack_from("<i", stream, off)
        if off + 4 + bs > len(stream):
            break
        ref, pos, bmn, fnc, lseq = struct.unpack_from("<iiIIi", stream, off + 4)
        l_name, n_cig = bmn & 0xFF, fnc & 0xFFFF
        span = 0
        for k in range(n_cig):
            (raw,) = struct.unpack_from("<I", stream, off + 36 + l_name + 4 * k)
            if (0x3C1A7 >> ((raw & 15) * 2 + 1)) & 1:
                span += raw >> 4
        name = stream[off + 36:off + 36 + l_name - 1].decode("latin-1")
        out.append((off, ref, pos, pos + span, name))
        off += 4 + bs
    return out


def region_filter(records, ref_id, beg, end):
    """BamReadFilter with one region over records = [(off, ref, pos, end_pos, name)] in stream order."""
    out = []
    for r in records:
        _, ref, pos, end_pos, _ = r
        if (ref & U32) > ref_id:
            break
        if (ref & U32) < ref_id:
            continue
        if (pos & U32) >= end:
            break
        if (pos & U32) > beg:
            out.append(r); continue
        if (end_pos & U32) <= beg:
            continue
        out.append(r)
    return out


def region_reads(bam_path, bai, ref_id, beg, end):
    st = read_chunks(bam_path, get_chunks(bai, ref_id, beg, end))
    return st, region_filter(frame(st), ref_id, beg, end)


# ---- BAI writer: IndexBuilder / createIndex (bio/std/hts/bam/bai/indexing.d:56-366) ---------------------------

def records_with_offsets(bam_path):
    """readsWithProgress!withOffsets (readrange.d:51-68, bgzf/inputstream.d:425-452,497-530) over a whole file:
    -> (n_refs, [dict(ref, pos, end_pos, bin, unmapped, start_vo, end_vo)]).  The virtual offset of a stream
    position on a block boundary is (start of the NEXT block, 0) -- readBlock moves on as soon as a block is spent."""
    f = open(bam_path, "rb").read()
    blocks, off, data = [], 0, bytearray()
    while off + 4 <= len(f):
        isize, d, nxt = _block_at(f, off)
        blocks.append((len(data), isize, off, nxt))
        data += d
        off = nxt
    assert data[:4] == b"BAM\1"
    (l_text,) = struct.unpack_from("<i", data, 4)
    q = 8 + l_text
    (n_ref,) = struct.unpack_from("<i", data, q); q += 4
    for _ in range(n_ref):
        (l_name,) = struct.unpack_from("<i", data, q); q += 4 + l_name + 4
    filled = [b for b in blocks if b[1] > 0]
    end_coff = filled[-1][3] if filled else 0

    def vo(u):
        for ustart, isize, coff, _ in filled:
            if ustart <= u < ustart + isize:
                return (coff << 16) | (u - ustart)
        return end_coff << 16

    out = []
    for off, ref, pos, end_pos, _ in frame(bytes(data[q:])):
        bmn, fnc = struct.unpack_from("<II", data, q + off + 12)
        (bs,) = struct.unpack_from("<i", data, q + off)
        unmapped = bool((fnc >> 16) & 4)
        out.append(dict(ref=ref, pos=pos, end_pos=pos if unmapped else end_pos, bin=bmn >> 16, unmapped=unmapped,
                        start_vo=vo(q + off), end_vo=vo(q + off + 4 + bs)))
    return n_ref, out


class NotSorted(Exception):
    pass


def build_index(n_refs, reads):
    """IndexBuilder.put / finish (indexing.d:281-350) with check_bins off, statement by statement.
    -> (refs, no_coord): refs[i] = None for writeEmptyReference, else dict(bins={id: [[beg, end], ...]},
    meta=(beg_vo, end_vo, mapped, unmapped), ioffsets=[...]).  The reference writes bins in druntime's
    associative-array order (:193), which nothing pins -- callers compare the mapping, or serialise ascending."""
    size = 37449 - 4680 + 1                                           # :273
    st = dict(first=True, prev=None, chunks={}, beg=None, lin=[0] * size, wl=0,
              no_coord=0, beg_vo=(1 << 64) - 1, end_vo=0, unmapped=0, mapped=0)
    refs = []

    def tolin(p):                                                     # :51-53
        return 0 if p < 0 else p // WINDOW

    def update_linear():                                              # :134-162
        p = st["prev"]
        b = tolin(p["pos"])
        e = b if p["unmapped"] else tolin(p["pos"] + (p["end_pos"] - p["pos"]) - 1)
        for i in range(b, e + 1):
            if st["lin"][i] == 0:                                     # a D RangeError past the array: IndexError here
                st["lin"][i] = p["start_vo"]
        st["wl"] = max(st["wl"], e + 1)

    def update_chunks():                                              # :221-246
        p = st["prev"]
        end = p["end_vo"]
        cs = st["chunks"].setdefault(p["bin"], [])
        if not cs or (cs[-1][1] >> 16) != (st["beg"] >> 16):
            cs.append([st["beg"], end])
        else:
            cs[-1][1] = end
        st["beg"] = end

    def dump():                                                       # :164-218
        io, last = [], 0
        for v in st["lin"][:st["wl"]]:
            if v == 0:
                v = last
            else:
                last = v
            io.append(v)
        refs.append(dict(bins=st["chunks"], meta=(st["beg_vo"], st["end_vo"], st["mapped"], st["unmapped"]), ioffsets=io))
        st["lin"] = [0] * size; st["wl"] = 0; st["chunks"] = {}
        st["beg"] = st["prev"]["end_vo"]
        st["beg_vo"] = st["end_vo"] = st["beg"]
        st["unmapped"] = st["mapped"] = 0

    def meta(r):                                                      # :118-132, scope(exit) of put
        if r["ref"] == -1:
            st["no_coord"] += 1
        else:
            if r["unmapped"]:
                st["unmapped"] += 1
            else:
                st["mapped"] += 1
            if st["beg_vo"] == (1 << 64) - 1:
                st["beg_vo"] = r["start_vo"]
            st["end_vo"] = r["end_vo"]

    def put(r):
        p = st["prev"]
        if not st["first"] and r["ref"] != -1 and not (p["ref"] < r["ref"]):          # :253-268
            if not (r["ref"] == p["ref"] and r["pos"] >= p["pos"]):
                raise NotSorted()
        try:
            if r["ref"] < 0 or r["pos"] < 0:
                return
            if st["first"]:
                st["prev"] = r; st["first"] = False; st["beg"] = r["start_vo"]
                refs.extend([None] * r["ref"])
                return
            if r["ref"] > p["ref"]:
                update_linear(); update_chunks(); dump()
                refs.extend([None] * (r["ref"] - p["ref"] - 1))
            if r["ref"] == p["ref"]:
                update_linear()
                if r["bin"] != p["bin"]:
                    update_chunks()
            st["prev"] = r
        finally:
            meta(r)

    for r in reads:
        put(r)
    if not st["first"]:                                               # finish, :336-350
        update_linear(); update_chunks(); dump()
    last = st["prev"]["ref"] if st["prev"] else -1
    refs.extend([None] * (n_refs - last - 1))
    return refs, st["no_coord"]


def serialise_index(n_refs, refs, no_coord):
    """The bytes IndexBuilder streams out (:170-218,276-277,347), bins in ascending id order."""
    out = bytearray(b"BAI\1") + struct.pack("<i", n_refs)
    for r in refs:
        if r is None:
            out += struct.pack("<ii", 0, 0)
            continue
        out += struct.pack("<i", len(r["bins"]) + 1)
        for bid in sorted(r["bins"]):
            cs = r["bins"][bid]
            out += struct.pack("<Ii", bid, len(cs))
            for b, e in cs:
                out += struct.pack("<QQ", b, e)
        out += struct.pack("<IiQQQQ", 37450, 2, *r["meta"])
        out += struct.pack("<i", len(r["ioffsets"])) + struct.pack(f"<{len(r['ioffsets'])}Q", *r["ioffsets"])
    out += struct.pack("<Q", no_coord)
    return bytes(out)


def create_index(bam_path):
    n_refs, reads = records_with_offsets(bam_path)
    refs, no_coord = build_index(n_refs, reads)
    return serialise_index(n_refs, refs, no_coord)
