"""The D module a BioD maintainer adds (d/bio/std/hts/bam/gpu/pileup.d) against include/biodgpu.h, without a D compiler
(none in this image: SURVEY.md 8c): every function's return type and parameter list, every struct's field order, types
and size, and every constant of the header must be mirrored exactly by the module's `extern (C)` block; the facade types
must define the properties BioD's consumers use (pileup.d:86-290, examples/read_bam_file.d:33-41, snpcallers/simple.d:24-28,
snpcallers/maq.d:396-412); and a few cheap lints stand in for the compiler (balanced delimiters, no address of a lambda
parameter -- the dangling pointer the first draft of the binding had)."""
import ctypes as C
import os
import re

from biod_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "biodgpu.h")
DMOD = os.path.join(ROOT, "d", "bio", "std", "hts", "bam", "gpu", "pileup.d")

C2D = {"int": "int", "int32_t": "int", "uint32_t": "uint", "uint64_t": "ulong", "int64_t": "long", "uint8_t": "ubyte", "uint16_t": "ushort",
       "size_t": "size_t", "char": "char", "float": "float", "double": "double", "void": "void"}
SIZES = {"int": 4, "uint": 4, "ulong": 8, "long": 8, "ubyte": 1, "ushort": 2, "size_t": 8, "char": 1, "float": 4, "double": 8}


def _strip_c(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _split_params(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _c_type(decl, has_name=True):
    """'const uint8_t* data' -> (const, base, pointer depth) in D's vocabulary."""
    d = decl.replace("struct ", " ").strip()
    ptr = d.count("*")
    toks = d.replace("*", " ").split()
    const = "const" in toks
    toks = [t for t in toks if t != "const"]
    if has_name and len(toks) > 1:
        toks = toks[:-1]
    assert len(toks) == 1, decl
    return (const, C2D.get(toks[0], toks[0]), ptr)


def _d_type(decl, has_name=True):
    """'const(ubyte)* data' -> (const, base, pointer depth)."""
    d = decl.strip()
    m = re.match(r"const\((\w+)\)\s*(\**)\s*(\w*)$", d)
    if m:
        return (True, m.group(1), len(m.group(2)))
    m = re.match(r"(\w+)\s*(\**)\s*(\w*)$", d)
    assert m, decl
    return (False, m.group(1), len(m.group(2)))


def parse_c():
    text = _strip_c(open(HDR).read())
    consts = {m.group(1): m.group(2) for m in re.finditer(r"#define\s+(BGP_\w+)\s+(\S+)", text)}
    for m in re.finditer(r"enum\s*\{(.*?)\}", text, flags=re.S):
        for e in re.finditer(r"(BGP_\w+)\s*=\s*(-?\w+)", m.group(1)):
            consts[e.group(1)] = e.group(2)
    structs, opaque = {}, set()
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for stmt in m.group(2).split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            names = [x.strip() for x in stmt.split(",")]
            first = names[0]
            base = _c_type(first)
            fname = re.search(r"(\w+)\s*$", first).group(1)
            fields.append((fname, base))
            for extra in names[1:]:
                fields.append((extra.replace("*", "").strip(), (base[0], base[1], extra.count("*"))))
        structs[m.group(3)] = fields
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", text):
        opaque.add(m.group(2))
    body = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    body = re.sub(r"enum\s*\{.*?\}\s*;", " ", body, flags=re.S)
    body = re.sub(r"^\s*#.*$", " ", body, flags=re.M)
    body = body.replace('extern "C" {', " ")
    funcs = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(bgp_\w+)\s*\(([^)]*)\)\s*;", body):
        ret = _c_type(m.group(1), has_name=False)
        params = [] if m.group(3).strip() in ("", "void") else [_c_type(p) for p in _split_params(m.group(3))]
        funcs[m.group(2)] = (ret, params)
    return consts, structs, opaque, funcs


def d_extern_block():
    text = open(DMOD).read()
    a = text.index("extern (C) nothrow @nogc {")
    b = text.index("\n}\n", a)
    return re.sub(r"//[^\n]*", " ", text[a + len("extern (C) nothrow @nogc {"):b])


def parse_d():
    blk = d_extern_block()
    consts = {m.group(1): m.group(2) for m in re.finditer(r"enum\s+(BGP_\w+)\s*=\s*([^;]+);", blk)}
    structs, opaque = {}, set()
    for m in re.finditer(r"struct\s+(\w+)\s*\{(.*?)\}", blk, flags=re.S):
        fields = []
        for stmt in m.group(2).split(";"):
            stmt = stmt.strip()
            if stmt:
                fields.append((re.search(r"(\w+)\s*$", stmt).group(1), _d_type(stmt)))
        structs[m.group(1)] = fields
    for m in re.finditer(r"struct\s+(\w+)\s*;", blk):
        opaque.add(m.group(1))
    body = re.sub(r"struct\s+\w+\s*\{.*?\}", " ", blk, flags=re.S)
    body = re.sub(r"struct\s+\w+\s*;", " ", body)
    body = re.sub(r"enum\s+BGP_\w+\s*=[^;]+;", " ", body)
    funcs = {}
    for m in re.finditer(r"([\w\(\)\*\s]+?)\b(bgp_\w+)\s*\((.*?)\)\s*;", body, flags=re.S):
        ret = _d_type(m.group(1).strip() + " x")
        params = [_d_type(p) for p in _split_params(m.group(3))]
        funcs[m.group(2)] = (ret, params)
    return consts, structs, opaque, funcs


def _num(v):
    v = v.strip().rstrip("uUlL")
    return int(v, 0)


def _sizeof(fields, structs):
    off, align = 0, 1
    for _, (_, base, ptr) in fields:
        sz = 8 if ptr else SIZES[base] if base in SIZES else _sizeof(structs[base], structs)[0]
        al = min(sz, 8)
        off = (off + al - 1) // al * al + sz
        align = max(align, al)
    return (off + align - 1) // align * align, align


def test_every_function_of_the_header_is_bound_with_the_same_signature():
    _, _, _, cf = parse_c()
    _, _, _, df = parse_d()
    assert set(cf) == set(L.SYMBOLS), set(cf) ^ set(L.SYMBOLS)          # the parser sees what the library exports
    assert set(df) == set(cf), sorted(set(df) ^ set(cf))
    for name, (ret, params) in cf.items():
        dret, dparams = df[name]
        assert dret == ret, (name, "return", ret, dret)
        assert len(dparams) == len(params), (name, "arity", len(params), len(dparams))
        for i, (a, b) in enumerate(zip(params, dparams)):
            assert a == b, (name, f"parameter {i}", a, b)


def test_every_struct_has_the_same_fields_in_the_same_order_and_the_same_size():
    _, cs, copaque, _ = parse_c()
    _, ds, dopaque, _ = parse_d()
    assert copaque == dopaque == {"bgp_ctx", "bgp_bai"}
    assert set(cs) == set(ds), set(cs) ^ set(ds)
    for name, fields in cs.items():
        assert [f for f, _ in fields] == [f for f, _ in ds[name]], (name, "field names / order")
        for (f, a), (_, b) in zip(fields, ds[name]):
            assert a == b, (name, f, a, b)
        assert _sizeof(fields, cs) == _sizeof(ds[name], ds), name
    # ... and the sizes are the ones the compiled library uses (ctypes mirrors of the same header, tests/test_abi.py)
    for name, ct in (("bgp_records", L.Records), ("bgp_columns", L.Columns), ("bgp_pileup_info", L.PileupInfo), ("bgp_chunk_info", L.ChunkInfo),
                     ("bgp_kernel_time", L.KernelTime), ("bgp_synth_cfg_c", L.SynthCfg)):
        assert _sizeof(ds[name], ds)[0] == C.sizeof(ct), name
    for name, dt in (("bgp_record", L.RECORD_DTYPE), ("bgp_column", L.COLUMN_DTYPE), ("bgp_entry", L.ENTRY_DTYPE), ("bgp_run", L.RUN_DTYPE), ("bgp_maq_call", L.MAQ_CALL_DTYPE)):
        assert _sizeof(ds[name], ds)[0] == dt.itemsize, name
        assert [f for f, _ in ds[name]] == list(dt.names), name


def test_every_constant_of_the_header_has_the_same_value():
    cc, _, _, _ = parse_c()
    dc, _, _, _ = parse_d()
    cc = {k: v for k, v in cc.items() if k != "BGP_ABI_VERSION" or True}
    assert set(cc) == set(dc), set(cc) ^ set(dc)
    for k, v in cc.items():
        assert _num(v) == _num(dc[k]), (k, v, dc[k])
    assert _num(dc["BGP_ABI_VERSION"]) == L.lib().bgp_abi_version()


def _body(text, header):
    a = text.index(header)
    i = text.index("{", a)
    depth = 0
    for j in range(i, len(text)):
        depth += text[j] == "{"
        depth -= text[j] == "}"
        if depth == 0:
            return text[i:j + 1]
    raise AssertionError(header)


def test_facade_types_define_what_the_reference_consumers_use():
    text = open(DMOD).read()
    read = _body(text, "struct GpuPileupRead {")
    for prop in ("current_base", "current_base_quality", "cigar_operation", "cigar_operation_offset", "query_offset", "cigar_before", "cigar_after"):   # pileup.d:86-149
        assert re.search(r"\b%s\(\)" % prop, read), prop
    # ... and IS BioD's own read, as PileupRead is (pileup.d:86-89): position, ref_id, mapping_quality, is_reverse_strand, name, sequence,
    # base_qualities, cigar, cigarString, read["RG"] (read.d:1070-1087) -- SURVEY 8(b)'s list and everything else -- are BamRead's, over the
    # host copy of the record (read.d:482), end_position the wrapper's field (read.d:1380-1383)
    assert re.search(r"EagerBamRead!BamRead\s+read;", read) and re.search(r"alias\s+read\s+this;", read)
    assert re.search(r"read\.read\s*=\s*BamRead\(cast\(ubyte\[\]\)\s*p\[4 \.\. 4 \+ block_size\]\);", read) and "read.end_position = r.end_pos;" in read
    assert "import bio.std.hts.bam.read : BamRead, EagerBamRead;" in text
    ref_read = open("/root/reference/bio/std/hts/bam/read.d").read() if os.path.exists("/root/reference/bio/std/hts/bam/read.d") else None
    if ref_read is not None:        # (here, not on the GPU box: the names the facade relies on exist in the reference as it is used)
        assert "this(ubyte[] chunk, bool fix_byte_order=true)" in ref_read and "struct EagerBamRead(R=BamRead)" in ref_read and "int end_position;" in ref_read
        for prop in ("name", "sequence", "base_qualities", "cigar", "cigarString", "mapping_quality", "is_reverse_strand", "position", "ref_id", "opIndex(string key)"):
                assert prop in ref_read, prop
    col = _body(text, "struct GpuPileupColumn {")
    for prop in ("coverage", "ref_id", "position", "reads", "reads_starting_here", "bases", "base_qualities", "read_qualities", "reference_base"):      # pileup.d:236-290
        assert re.search(r"\b%s\(\)" % prop, col), prop
    pile = _body(text, "final class GpuPileup {")
    for prop in ("empty", "front", "popFront", "start_position", "end_position", "ref_id"):                                                              # pileup.d:335-397,439-457
        assert re.search(r"\b%s\(\)" % prop, pile), prop
    reads = _body(text, "struct GpuColumnReads {")
    for prop in ("empty", "front", "back", "popFront", "popBack", "save", "length", "opIndex", "opSlice"):                                              # column.reads[column.coverage - 2], pileup.d:803
        assert re.search(r"\b%s\(" % prop, reads), prop
    # same names, parameter lists and defaults as pileup.d:683-687, :509, :1011-1012
    assert re.search(r"makePileup\(R\)\(GpuReads!R reads, bool use_md_tag = false, ulong start_from = 0, ulong end_at = ulong\.max,\s*bool skip_zero_coverage = true\)", text)
    assert re.search(r"pileupColumns\(R\)\(GpuReads!R reads, bool use_md_tag = false, bool skip_zero_coverage = true\)", text)
    assert re.search(r"pileupChunks\(R\)\(GpuReads!R reads, bool use_md_tag = false, size_t block_size = 16_384_000,\s*ulong start_from = 0, ulong end_at = ulong\.max\)", text)


def test_cheap_lints_in_place_of_a_compiler():
    text = open(DMOD).read()
    code = re.sub(r"//[^\n]*", "", re.sub(r"/\*\*.*?\*/", "", text, flags=re.S))
    code = re.sub(r'"(\\.|[^"\\])*"', '""', code)
    code = re.sub(r"'(\\.|[^'\\])'", "' '", code)
    for a, b in ("{}", "()", "[]"):
        assert code.count(a) == code.count(b), (a, code.count(a), code.count(b))
    assert not re.search(r"=>\s*\w+\(\s*&", code), "address of a lambda parameter"
    assert "module bio.std.hts.bam.gpu.pileup;" in text
    # every bgp_ function the facade calls is declared in the extern block
    _, _, _, df = parse_d()
    called = set(re.findall(r"\b(bgp_[a-z0-9_]+)\s*\(", code))
    assert called <= set(df) | {"bgp_ctx"}, called - set(df)
    # `this` is not touched before super(...) in the exception's constructor (a D rule)
    exc = _body(text, "class BiodGpuException : Exception {")
    ctor = exc[exc.index("this(int code"):]
    assert ctor.index("super(") < ctor.index("this.code")
