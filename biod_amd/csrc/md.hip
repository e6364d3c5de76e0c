// md.hip -- reference_base from MD tags on the device.
//
// Reference semantics: PileupRangeUsingMdTag (bio/std/hts/bam/pileup.d:522-654) keeps a "chunk" of
// reconstructed reference bases, dna(provider) (bio/std/hts/bam/md/reconstruct.d:38-214, MD parsed by
// mdOperations, bio/std/hts/bam/md/parse.d:13-142), and replaces it -- when it runs empty -- by the
// chunk of the best candidate seen since (add(), pileup.d:563-596).  The candidate chain is sequential
// per reference run, but only at EVENT granularity: between two chunk fetches nothing depends on read
// order except "first arg-max of end_position among the reads that pass add()'s filter", and the chunk
// cursor moves one base per visited column, i.e. in closed form.
//
//   k6_prep     one lane per record : a 16-byte row -- position, end, length of dna(read) + flags, and the
//               max end_position of the earlier position groups (is the previous column uncovered?)
//   k6_chain    one wave per reference run : the event loop (fetch to fetch), emits provider intervals
//   k6_fill     one lane per interval : dna(provider)[c0 ...] written to the dense reference_base array
//
// The event-level restatement was validated against the column-by-column oracle on random streams
// (gaps, N ops, missing / truncated / over-long MD, several references, both skip_zero_coverage modes)
// before it was written in HIP; tests/test_gpu_parity.py repeats that comparison through the ABI.
#include "common.h"

#include <algorithm>
#include <cstdio>

namespace {

// ---------------------------------------------------------------------------------------------- MD parsing
// mdOperations(md): a two-ended cache over the string (front parsed forwards, back parsed backwards)
// behind filterBidirectional!(not zero-match)  (md/parse.d:13-142).  Only forward iteration is used by
// dna(read), but the backward pre-parse decides where the forward parse stops, so it is kept literally.

struct MdOp { uint32_t type; /* 0 match, 1 mismatch, 2 deletion */ uint32_t match; uint32_t ch; uint32_t del_off, del_len; };
struct MdRes { const uint8_t* md; uint32_t b, e; MdOp front, back; int rem;
               uint64_t wlo, whi; int32_t wbase; bool wok, slide; };     // the tag's last 16 chars in registers: chars [wbase, wbase + 16) (wok: loaded); slide: a long tag, the window follows the parse

// char i of the tag: from the register window where it covers i (a tag of up to 16 chars -- "75A74" -- is parsed without
// another load: every operation used to cost a dependent load of its own), else from memory
// (the window slides: a char outside it brings the sixteen from that char on -- or, near the range's end, the sixteen in front of the end:
// a tag's value starts at least 39 bytes into its record, so md + e - 16 is inside the stream whatever e is.  A byte a load made a
// 30 kb read's tag -- ten thousand chars -- ten thousand dependent round trips: cfg5's provider check, 15 ms)
__device__ __forceinline__ uint8_t md_ch(MdRes& r, uint32_t i) {
    int32_t d = (int32_t)i - r.wbase;
    if (!(r.wok && d >= 0 && d < 16)) {
        if (!r.slide) return r.md[i];                  // (a short tag: the few chars in front of the window one by one)
        const int32_t nb = min((int32_t)i, (int32_t)r.e - 16);
        uint64_t w2[2]; __builtin_memcpy(w2, r.md + (int64_t)nb, 16);
        r.wlo = w2[0]; r.whi = w2[1]; r.wbase = nb; r.wok = true;
        d = (int32_t)i - nb;
    }
    return (uint8_t)((d < 8 ? r.wlo : r.whi) >> (8 * (d & 7)));
}

__device__ __forceinline__ bool is_digit_(uint8_t c) { return c >= '0' && c <= '9'; }
__device__ __forceinline__ bool is_upper_(uint8_t c) { return c >= 'A' && c <= 'Z'; }

// Base16(c).asCharacter: char -> nt16 code -> char (bio/core/base.d:42-62,85; md/operation.d:92-97)
__device__ __forceinline__ uint8_t base16_norm(uint8_t c) {
    if (c >= '0' && c <= '3') return (uint8_t)"ACGT"[c - '0'];
    if (c == '=') return '=';
    if (c >= 'a' && c <= 'z') c = (uint8_t)(c - 'a' + 'A');
    switch (c) {
        case 'A': case 'C': case 'M': case 'G': case 'R': case 'S': case 'V': case 'T':
        case 'W': case 'Y': case 'H': case 'K': case 'D': case 'B': case 'N': return c;
        default: return 'N';
    }
}


__device__ bool md_cache_front(MdRes& r) {                      // md/parse.d:85-109
    if (r.b >= r.e) return false;
    const uint32_t avail = r.e - r.b;
    auto at = [&](uint32_t i) -> uint8_t { return md_ch(r, r.b + i); };
    const uint8_t c0 = at(0);
    MdOp o{0, 0, 0, 0, 0};
    if (c0 == '^') {
        uint32_t len = 0;
        while (1 + len < avail && is_upper_(at(1 + len))) ++len;
        o.type = 2; o.del_off = r.b + 1; o.del_len = len; r.b += 1 + len;
    } else if (is_digit_(c0)) {
        uint32_t len = 0; uint64_t v = 0;
        while (len < avail && is_digit_(at(len))) { v = v * 10 + (uint64_t)(at(len) - '0'); ++len; }
        o.type = 0; o.match = (uint32_t)v; r.b += len;
    } else { o.type = 1; o.ch = base16_norm(c0); r.b++; }
    r.front = o;
    return true;
}

__device__ bool md_cache_back(MdRes& r) {                       // md/parse.d:111-131
    if (r.b >= r.e) return false;
    MdOp o{0, 0, 0, 0, 0};
    // bk(k) is the char k places before the end (from the register window mdf_init loaded, where it reaches)
    auto bk = [&](uint32_t k) -> uint8_t { return md_ch(r, r.e - 1 - k); };
    if (is_digit_(bk(0))) {
        uint32_t len = 0; uint64_t v = 0, p10 = 1;
        while (r.e - len > r.b && is_digit_(bk(len))) { v += p10 * (uint64_t)(bk(len) - '0'); p10 *= 10; ++len; }
        o.type = 0; o.match = (uint32_t)v; r.e -= len;
    } else if (r.e - r.b == 1 || is_digit_(bk(1))) {
        o.type = 1; o.ch = base16_norm(bk(0)); r.e -= 1;
    } else {
        uint32_t len = 0;
        while (r.e - len > r.b && bk(len) != '^') ++len;
        o.type = 2; o.del_off = r.e - len; o.del_len = len;
        r.e = (r.e - len > r.b) ? r.e - len - 1 : r.b;
    }
    r.back = o;
    return true;
}

__device__ __forceinline__ bool md_empty(const MdRes& r) { return r.rem == 0; }
__device__ void md_pop_front(MdRes& r) {                        // md/parse.d:57-69
    if (r.b >= r.e) { if (r.rem == 255) { r.front = r.back; r.rem = 1; } else r.rem = 0; }
    else if (!md_cache_front(r)) r.rem = 0;
}
__device__ void md_pop_back(MdRes& r) {                         // md/parse.d:71-83
    if (r.b >= r.e) { if (r.rem == 255) { r.back = r.front; r.rem = 1; } else r.rem = 0; }
    else if (!md_cache_back(r)) r.rem = 0;
}
__device__ __forceinline__ bool md_zero(const MdOp& o) { return o.type == 0 && o.match == 0; }
__device__ void mdf_init(MdRes& r, const uint8_t* md, uint32_t n, bool slide = false) {   // md/parse.d:23-33 + filterBidirectional ctor (slide: a long read's tag, see md_ch -- a constant at every call, so that the short reads' kernels do not carry the window's code)
    r.md = md; r.b = 0; r.e = n; r.rem = 255;
    // the sixteen bytes in front of the tag's end in one load (a tag's value starts at least 39 bytes into its record, so
    // md + n - 16 is inside the stream whatever n is)
    r.wbase = (int32_t)n - 16; r.wok = n > 0; r.slide = slide;
    if (r.wok) { uint64_t w2[2]; __builtin_memcpy(w2, md + (int64_t)r.wbase, 16); r.wlo = w2[0]; r.whi = w2[1]; } else { r.wlo = 0; r.whi = 0; }
    if (!md_cache_front(r)) r.rem = 0;
    else if (!md_cache_back(r)) { r.back = r.front; r.rem = 1; }
    while (!md_empty(r) && md_zero(r.front)) md_pop_front(r);
    while (!md_empty(r) && md_zero(r.back)) md_pop_back(r);
}
__device__ void mdf_pop_front(MdRes& r) { do { md_pop_front(r); } while (!md_empty(r) && md_zero(r.front)); }

// length of the NUL-terminated string at t[0 .. lim) (lim if there is no NUL): eight bytes a load while eight are there --
// a byte a load is a dependent round trip per character of every MD tag
__device__ __forceinline__ uint64_t strnlen8(const uint8_t* t, uint64_t lim) {
    uint64_t n = 0;
    while (n + 8 <= lim) {
        uint64_t w; __builtin_memcpy(&w, t + n, 8);
        const uint64_t z = (w - 0x0101010101010101ull) & ~w & 0x8080808080808080ull;      // the lowest set bit marks the first zero byte
        if (z) return n + (uint64_t)(__ffsll((long long)z) - 1) / 8;
        n += 8;
    }
    while (n < lim && t[n] != 0) ++n;
    return n;
}

// ---------------------------------------------------------------------------------------------- tag lookup
// BamRead.opIndex("MD") (read.d:1070-1087) with skipValue (read.d:1219-1230, tagvalue.d:78-119).
// Returns 1 found (a Z/H string: *val, *len), 0 absent or not a string, -1 malformed aux data.
__device__ int find_md(const uint8_t* s, const bgp_record& r, const uint8_t** val, uint32_t* len, uint64_t rec_end = 0) {     // rec_end: where the record ends if the caller knows (the next record's offset in an uncompacted table), else 0: block_size is read
    const uint64_t end = rec_end ? rec_end : r.rec_off + 4 + ld_u32(s + r.rec_off);
    const uint32_t l_name = r.bin_mq_nl & 0xFFu, n_cig = r.flag_nc & 0xFFFFu;
    const uint64_t tags = r.rec_off + 36 + l_name + 4ull * n_cig + (uint64_t)((r.l_seq + 1) / 2) + (uint64_t)r.l_seq;
    if (tags > end) return -1;
    const uint8_t* t = s + tags;
    const uint64_t tl = end - tags;
    if (tl < 4) return 0;
    uint64_t off = 0;
    while (off + 1 < tl) {
        const bool match = t[off] == 'M' && t[off + 1] == 'D';
        off += 2;
        if (off >= tl) return -1;
        const uint8_t ty = t[off++];
        if (match) {
            if (ty != 'Z' && ty != 'H') return 0;
            const uint32_t n = (uint32_t)strnlen8(t + off, tl - off);
            *val = t + off; *len = n;
            return 1;
        }
        if (ty == 'Z' || ty == 'H') { off += strnlen8(t + off, tl - off); if (off >= tl) return -1; ++off; }
        else if (ty == 'B') {
            if (off + 5 > tl) return -1;
            const uint8_t et = t[off];
            const int es = (et == 'c' || et == 'C') ? 1 : (et == 's' || et == 'S') ? 2 : (et == 'i' || et == 'I' || et == 'f') ? 4 : -1;
            if (es < 0) return -1;
            off += 5 + (uint64_t)es * ld_u32(t + off + 1);
        } else {
            const int es = (ty == 'A' || ty == 'c' || ty == 'C') ? 1 : (ty == 's' || ty == 'S') ? 2 : (ty == 'i' || ty == 'I' || ty == 'f') ? 4 : -1;
            if (es < 0) return -1;
            off += (uint64_t)es;
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------- dna(read)
// The query bases dna(read) draws from: SEQ under the M/=/X ops only (md/reconstruct.d:58-96,199).
struct QIter { const uint8_t* cg; const uint8_t* sq; uint32_t n_cig, op, in_op, q, raw; bool empty; };   // raw: CIGAR op `op`, loaded once per op (not once per base)
__device__ __forceinline__ void q_load(QIter& it) { it.raw = it.op < it.n_cig ? ld_u32(it.cg + 4ull * it.op) : 0u; }
__device__ void q_settle(QIter& it) {
    while (it.op < it.n_cig) {
        const uint32_t raw = it.raw;
        if (cig_qc(raw)) {
            if (cig_rc(raw) && it.in_op < (raw >> 4)) return;
            if (!cig_rc(raw)) it.q += raw >> 4;
        }
        it.op++; it.in_op = 0;
        q_load(it);
    }
    it.empty = true;
}
__device__ __forceinline__ uint8_t q_front(const QIter& it) {
    const uint32_t byte = it.sq[it.q >> 1];
    const uint32_t code = (it.q & 1u) ? (byte & 0xFu) : (byte >> 4);
    const uint64_t lo = 0x565352474D43413Dull, hi = 0x4E42444B48595754ull;        // "=ACMGRSVTWYHKDBN", base.d:85
    return (uint8_t)(((code & 8u) ? hi : lo) >> ((code & 7u) * 8));
}
__device__ __forceinline__ void q_pop(QIter& it) { it.q++; it.in_op++; q_settle(it); }

// dna(read) (md/reconstruct.d:98-182) as a resumable walk.  The state at the top of the loop -- MD parser, current
// operation, CIGAR cursor, chars yielded so far -- is everything the range carries, so a snapshot of it (DnaCk) lets
// another lane continue from that char: long reads are emitted in pieces (k6_ck_build / k6_emit_long below).
struct DnaState { MdRes m; MdOp cur; uint32_t del_i; QIter q; uint32_t n; bool live; };
struct DnaCk { uint64_t md_off; uint32_t b, e; int32_t rem; MdOp front, back, cur; uint32_t del_i, q_op, q_in, q_q, q_empty, n, node, pad; };
constexpr uint32_t DNA_CK = 512, DNA_LONG = 4096;     // checkpoint spacing; chunks longer than DNA_LONG are emitted in pieces

__device__ void dna_begin(DnaState& st, const uint8_t* s, const bgp_record& r, bool* err, bool slide = false) {
    *err = false; st.live = false; st.n = 0; st.del_i = 0;
    const uint8_t* md = nullptr; uint32_t mdn = 0;
    const int f = find_md(s, r, &md, &mdn);
    if (f < 0) { *err = true; return; }
    if (f == 0) mdn = 0;
    const uint32_t l_name = r.bin_mq_nl & 0xFFu, n_cig = r.flag_nc & 0xFFFFu;
    QIter& q = st.q; q.cg = s + r.rec_off + 36 + l_name; q.sq = q.cg + 4ull * n_cig; q.n_cig = n_cig; q.op = 0; q.in_op = 0; q.q = 0; q.empty = false;
    q_load(q);
    q_settle(q);
    mdf_init(st.m, md, mdn, slide);
    if (md_empty(st.m)) return;
    st.cur = st.m.front; mdf_pop_front(st.m);
    st.live = true;
}
__device__ void dna_save(const DnaState& st, const uint8_t* s, uint32_t node, DnaCk& ck) {
    ck.md_off = (uint64_t)(st.m.md - s); ck.b = st.m.b; ck.e = st.m.e; ck.rem = st.m.rem; ck.front = st.m.front; ck.back = st.m.back; ck.cur = st.cur;
    ck.del_i = st.del_i; ck.q_op = st.q.op; ck.q_in = st.q.in_op; ck.q_q = st.q.q; ck.q_empty = st.q.empty; ck.n = st.n; ck.node = node; ck.pad = st.live;
}
__device__ void dna_restore(DnaState& st, const uint8_t* s, const bgp_record& r, const DnaCk& ck) {
    st.m.md = s + ck.md_off; st.m.b = ck.b; st.m.e = ck.e; st.m.rem = ck.rem; st.m.front = ck.front; st.m.back = ck.back; st.cur = ck.cur;
    st.m.wok = false; st.m.wlo = 0; st.m.whi = 0; st.m.wbase = 0; st.m.slide = true;      // (a restored parser -- a long read's -- reads its chars through the sliding window)
    st.del_i = ck.del_i; st.n = ck.n; st.live = ck.pad != 0;
    const uint32_t l_name = r.bin_mq_nl & 0xFFu, n_cig = r.flag_nc & 0xFFFFu;
    QIter& q = st.q; q.cg = s + r.rec_off + 36 + l_name; q.sq = q.cg + 4ull * n_cig; q.n_cig = n_cig; q.op = ck.q_op; q.in_op = ck.q_in; q.q = ck.q_q; q.empty = ck.q_empty != 0;
    q_load(q);
}
// eight bases of the packed SEQ from query offset q on, as chars ("=ACMGRSVTWYHKDBN", base.d:85): one 8-byte load, the nibbles brought
// to an even start, four codes a v_perm pair -- where a base a step was a byte load, a table shift and a byte store
__device__ __forceinline__ uint64_t seq_chars8(const uint8_t* sq, uint32_t q) {
    unsigned long long w; __builtin_memcpy(&w, sq + (q >> 1), 8);
    if (q & 1u) w = ((w << 4) & 0xF0F0F0F0F0F0F0F0ull) | ((w >> 12) & 0x0F0F0F0F0F0F0F0Full);      // byte i: low nibble of byte i, high nibble of byte i + 1
    const uint32_t w4 = (uint32_t)w, hi = (w4 >> 4) & 0x0F0F0F0Fu, lo = w4 & 0x0F0F0F0Fu;            // the even and the odd bases of four bytes
    const uint32_t ca = __builtin_amdgcn_perm(hi, lo, 0x01050004u), cb = __builtin_amdgcn_perm(hi, lo, 0x03070206u);   // codes of bases 0..3 and 4..7, a byte each
    auto chars4 = [](uint32_t c) {
        const uint32_t sel = c & 0x07070707u, m8 = ((c >> 3) & 0x01010101u) * 0xFFu;
        const uint32_t l = __builtin_amdgcn_perm(0x56535247u, 0x4D43413Du, sel), h = __builtin_amdgcn_perm(0x4E42444Bu, 0x48595754u, sel);
        return (l & ~m8) | (h & m8);
    };
    return ((unsigned long long)chars4(cb) << 32) | chars4(ca);
}

// Runs until `stop` chars have been yielded or the range ends; chars with index >= from go to out[index - from]
// (out may be null: then SEQ is never read).
__device__ void dna_run(DnaState& st, const uint8_t* s, uint32_t from, uint32_t stop, uint8_t* out, bool* bad) {
    *bad = false;
    if (!st.live) return;
    const uint8_t* md = st.m.md;
    for (;;) {
        if (st.n >= stop) return;
        const bool keep = out && st.n >= from;
        if (st.cur.type == 1) {
            if (keep) out[st.n - from] = (uint8_t)st.cur.ch;
            ++st.n;
            if (st.q.empty) { *bad = true; st.live = false; return; }
            q_pop(st.q);
        } else if (st.cur.type == 0) {
            if (st.q.empty) { *bad = true; st.live = false; return; }
            // a match run inside one M / = / X op: eight bases at a time (the pops of eight bases that stay inside the op are
            // q += 8, in_op += 8; SEQ's bytes behind the eighth base are the read's own -- SEQ, then QUAL)
            if (keep && st.cur.match >= 8u && (st.q.raw >> 4) - st.q.in_op >= 8u && stop - st.n >= 8u) {
                const unsigned long long c8 = seq_chars8(st.q.sq, st.q.q);
                __builtin_memcpy(out + (st.n - from), &c8, 8);
                st.n += 8; st.cur.match -= 8; st.q.q += 8; st.q.in_op += 8;
                q_settle(st.q);
                if (st.cur.match != 0) continue;
                if (md_empty(st.m)) { st.live = false; return; }
                st.cur = st.m.front; mdf_pop_front(st.m);
                continue;
            }
            if (keep) out[st.n - from] = q_front(st.q);
            ++st.n;
            --st.cur.match;
            q_pop(st.q);
            if (st.cur.match != 0) continue;
        } else {
            if (st.cur.del_len == 0) { *bad = true; st.live = false; return; }
            if (keep) out[st.n - from] = base16_norm(md[st.cur.del_off + st.del_i]);
            ++st.n;
            if (++st.del_i < st.cur.del_len) continue;
            st.del_i = 0;
        }
        if (md_empty(st.m)) { st.live = false; return; }
        st.cur = st.m.front; mdf_pop_front(st.m);
    }
}

// The same walk one MD OPERATION at a time, nothing stored: the state dna_run would have at char index `target`
// (or where the range ends / stalls before it).  A char-at-a-time walk makes a wave pay one operation's parse
// latency at almost every step -- some lane is always at an operation boundary -- so the checkpoint pass uses this.
__device__ bool q_advance(QIter& q, uint32_t k, uint32_t* done) {      // k times q_pop, as far as the CIGAR allows
    *done = 0;
    while (k) {
        if (q.empty) return false;
        const uint32_t left = (q.raw >> 4) - q.in_op, t = left < k ? left : k;
        q.q += t; q.in_op += t; k -= t; *done += t;
        q_settle(q);
    }
    return true;
}
__device__ void dna_skip(DnaState& st, uint32_t target, bool* stall = nullptr) {      // *stall: the range stalls on an empty query or an empty deletion before `target` (dna_run's *bad)
    while (st.live && st.n < target) {
        if (st.cur.type == 1) {
            ++st.n;
            if (st.q.empty) { st.live = false; if (stall) *stall = true; return; }
            q_pop(st.q);
        } else if (st.cur.type == 0) {
            if (st.q.empty) { st.live = false; if (stall) *stall = true; return; }
            const uint32_t want = st.cur.match < target - st.n ? st.cur.match : target - st.n;
            uint32_t got;
            const bool all = q_advance(st.q, want, &got);
            st.n += got; st.cur.match -= got;
            if (!all) { st.live = false; if (stall) *stall = true; return; }
            if (st.cur.match != 0) continue;
        } else {
            if (st.cur.del_len == 0) { st.live = false; if (stall) *stall = true; return; }
            const uint32_t left = st.cur.del_len - st.del_i, t = left < target - st.n ? left : target - st.n;
            st.n += t; st.del_i += t;
            if (st.del_i < st.cur.del_len) continue;
            st.del_i = 0;
        }
        if (md_empty(st.m)) { st.live = false; return; }
        st.cur = st.m.front; mdf_pop_front(st.m);
    }
}

// Chars with index in [from, from+count) are stored to out (out may be null: the whole range is walked).  Returns
// the number of chars the lazy range yields before it ends (or from+count when out stops it there); *bad is set
// when the range would stall on an empty query (the oracle's `bad`), *err on malformed aux data.
__device__ uint32_t dna_walk(const uint8_t* s, const bgp_record& r, uint32_t from, uint32_t count, uint8_t* out, bool* bad, bool* err) {
    *bad = false;
    DnaState st;
    dna_begin(st, s, r, err);
    if (*err) return 0;
    if (out && from) dna_skip(st, from);       // an MD operation at a time up to the first char wanted (a range that stalls in front of it yields none: *bad is only told for from == 0)
    dna_run(st, s, from, out ? from + count : 0xFFFFFFFFu, out, bad);
    return st.n;
}

// Length of dna(read) and its `bad` flag without touching SEQ: the walk above only asks the query iterator
// "is there another M/=/X base", so the answer is a function of the MD operations and of the number of such
// bases in the CIGAR (a match run takes min(run, bases left) and stalls if it wanted more; a mismatch is
// yielded and then stalls on an empty query; an empty deletion stalls).
__device__ uint32_t dna_length(const uint8_t* s, const bgp_record& r, bool* bad, bool* err, const uint4* c4 = nullptr, uint64_t rec_end = 0) {      // c4: the record's first four CIGAR ops (K1's side array) or null
    *bad = false; *err = false;
    const uint8_t* md = nullptr; uint32_t mdn = 0;
    const int f = find_md(s, r, &md, &mdn, rec_end);
    if (f < 0) { *err = true; return 0; }
    if (f == 0) mdn = 0;
    const uint32_t l_name = r.bin_mq_nl & 0xFFu, n_cig = r.flag_nc & 0xFFFFu;
    const uint8_t* cg = s + r.rec_off + 36 + l_name;
    uint64_t qleft = 0;
    if (c4 && n_cig <= 4) {               // one coalesced 16-byte load instead of a dependent 4-byte read of the stream per op
        const uint4 v = *c4;
        const uint32_t ops[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) if (k < n_cig && cig_qc(ops[k]) && cig_rc(ops[k])) qleft += ops[k] >> 4;
    } else
    {
        auto takeq = [&](uint32_t raw) { if (cig_qc(raw) && cig_rc(raw)) qleft += raw >> 4; };
        uint32_t k = 0;
        for (; k + 8 <= n_cig; k += 8) {
            uint4 v0, v1; __builtin_memcpy(&v0, cg + 4ull * k, 16); __builtin_memcpy(&v1, cg + 4ull * k + 16, 16);
            takeq(v0.x); takeq(v0.y); takeq(v0.z); takeq(v0.w); takeq(v1.x); takeq(v1.y); takeq(v1.z); takeq(v1.w);
        }
        for (; k < n_cig; ++k) takeq(ld_u32(cg + 4ull * k));
    }
    MdRes m;
    mdf_init(m, md, mdn);
    uint64_t n = 0;
    while (!md_empty(m)) {
        const MdOp cur = m.front; mdf_pop_front(m);
        if (cur.type == 1) { ++n; if (qleft == 0) { *bad = true; break; } --qleft; }
        else if (cur.type == 0) {
            const uint64_t take = cur.match < qleft ? cur.match : qleft;
            n += take; qleft -= take;
            if (take < cur.match) { *bad = true; break; }
        } else { if (cur.del_len == 0) { *bad = true; break; } n += cur.del_len; }
    }
    return n > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)n;
}

// ---------------------------------------------------------------------------------------------- k6_prep
// One 16-byte row per record: what the controller needs, and nothing it has to chase pointers for.
//   lf : dna(read) length | MD_COV (basesCovered() > 0) | MD_GST (first read of a position group)
//        | MD_ERR (malformed aux data: EINVAL if ever fetched) | MD_BAD (range stalls on an empty query)
//   mb : max end_position over the reads of EARLIER position groups of the same reference (0 if none), so
//        "_prev_coverage == 0" at the read's column (pileup.d:565,625) is  mb < pos.
struct alignas(16) MdRow { uint32_t pos, end, lf, mb; };      // one dwordx4 / ds_read_b128 per row
constexpr uint32_t MD_BAD = 1u << 31, MD_ERR = 1u << 30, MD_COV = 1u << 29, MD_GST = 1u << 28, MD_LMASK = (1u << 28) - 1;

// what dna(read) is long when the MD tag says what the CIGAR says: the M / = / X bases and the deleted ones (md/reconstruct.d:38-214: a match
// or a mismatch of the tag takes a query base under M / = / X, a deletion yields its own chars; N, S, I, H, P appear in neither).
// k6_prep<false> puts this in a row WITHOUT reading the tag: only a provider's dna is ever looked at, by the reference
// (pileup.d:560-562 "minimise the number of reads for which MD tag will be decoded") and here (the successor of P depends on P's length
// and on the winner's flags alone), and k6_emit checks every provider's real tag against it -- a run with a provider whose tag disagrees,
// stalls or is malformed is redone by the event loop over rows made by k6_prep<true>.
__device__ __forceinline__ uint32_t dna_length_by_cigar(const uint8_t* s, const bgp_record& r, const uint4* c4) {
    const uint32_t l_name = r.bin_mq_nl & 0xFFu, n_cig = r.flag_nc & 0xFFFFu;
    uint64_t n = 0;
    auto take = [&](uint32_t raw) { const uint32_t op = raw & 15u; if ((cig_qc(raw) && cig_rc(raw)) || op == 2u) n += raw >> 4; };
    if (c4 && n_cig <= 4) {
        const uint4 v = *c4;
        const uint32_t ops[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) if (k < n_cig) take(ops[k]);
    } else {
        const uint8_t* cg = s + r.rec_off + 36 + l_name;
        uint32_t k = 0;
        for (; k + 8 <= n_cig; k += 8) {      // (eight ops, two loads in flight, a step: a long read's thousands of ops were as many round trips one after the other)
            uint4 v0, v1; __builtin_memcpy(&v0, cg + 4ull * k, 16); __builtin_memcpy(&v1, cg + 4ull * k + 16, 16);
            take(v0.x); take(v0.y); take(v0.z); take(v0.w); take(v1.x); take(v1.y); take(v1.z); take(v1.w);
        }
        for (; k < n_cig; ++k) take(ld_u32(cg + 4ull * k));
    }
    return n > (uint64_t)MD_LMASK ? MD_LMASK : (uint32_t)n;
}

// (a CIGAR of more than PREP_LONG_OPS ops is not summed by its record's lane -- a 30 kb read's 8 000 ops were a thousand dependent steps of 64 scattered loads a
// wave, 1.7 ms of cfg5's use_md_tag leg for 200 000 records: the record goes on a list and k6_prep_long gives a WAVE to it)
constexpr uint32_t PREP_LONG_OPS = 64;
template <bool FULL> __global__ __launch_bounds__(256) void k6_prep(const uint8_t* s, const bgp_record* rec, const uint4* cig4, const uint8_t* rcls, const uint64_t* pmax_end, const uint64_t* pmax_pos, uint64_t n, MdRow* rows, uint32_t* bmax, uint32_t* long_list, uint32_t long_cap) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bgp_record r{};
    if (i < n) r = rec[i];
    MdRow o{(uint32_t)r.pos, (uint32_t)r.end_pos, 0u, 0u};
    if (i < n && r.end_pos != r.pos) {
        bool bad = false, err = false;
        // (requested before the MD tag is walked: the common case -- this read is alone at its position -- needs nothing else)
        const uint64_t pm_prev = i ? pmax_pos[i - 1] : 0, pe_prev = i ? pmax_end[i - 1] : 0;
        uint32_t L;
        if (FULL) L = dna_length(s, r, &bad, &err, cig4 ? cig4 + i : nullptr, cig4 && i + 1 < n ? (uint64_t)rec[i + 1].rec_off : 0);   // (cig4: the table is K1's own, record i ends where i + 1 begins -- its header line is not touched at all)
        else if (rcls && (rcls[i] & BGP_CLS_NO_SKIP)) L = (uint32_t)r.end_pos - (uint32_t)r.pos;      // (k12_lds saw the whole CIGAR and no N in it: M / = / X and D bases are all the read covers)
        else {
            bool listed = false;
            if (long_list && (r.flag_nc & 0xFFFFu) > PREP_LONG_OPS) {      // (its length: k6_prep_long -- while the list has room; a lane's own walk behind that)
                const uint32_t slot = atomicAdd(long_list, 1u);
                if (slot < long_cap) { long_list[1 + slot] = (uint32_t)i; listed = true; }
            }
            L = listed ? 0u : dna_length_by_cigar(s, r, cig4 ? cig4 + i : nullptr);
        }
        if (L > MD_LMASK) L = MD_LMASK;
        // first record of this read's position group: the prefix maximum of the position key reaches the
        // read's own key exactly there (records are sorted; K3 has already rejected unsorted input).
        const uint64_t key = (ref_key(r.ref_id) << 32) | (uint32_t)r.pos;
        uint64_t hi = i, lo = 0, step = 1;
        if (i && pm_prev < key) lo = i;                                      // the record before belongs to an earlier group: hi == i
        else
        while (hi > 0) {
            const uint64_t t = hi > step ? hi - step : 0;
            if (pmax_pos[t] >= key) { hi = t; step <<= 1; } else { lo = t + 1; break; }
        }
        while (lo < hi) { const uint64_t m = (lo + hi) >> 1; if (pmax_pos[m] >= key) hi = m; else lo = m + 1; }
        uint32_t mb = 0;
        if (hi > 0) { const uint64_t e = hi == i ? pe_prev : pmax_end[hi - 1]; if ((e >> 32) == (key >> 32)) mb = (uint32_t)e; }
        o.lf = L | MD_COV | (hi == i ? MD_GST : 0u) | (bad ? MD_BAD : 0u) | (err ? MD_ERR : 0u);
        o.mb = mb;
    }
    {   // largest end_position of every 64 consecutive rows, and in its top bit whether one of them starts a position group behind
        // uncovered ground: lets k6_succ step over stretches in which nothing ends beyond a chunk and no GAP can be taken
        uint32_t m = (o.lf & MD_COV) ? o.end : 0u;
        for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
        const bool gap = (o.lf & MD_COV) && (o.lf & MD_GST) && o.mb < o.pos;
        if (__ballot(gap)) m |= 0x80000000u;
        if ((threadIdx.x & 63) == 0 && i < n) bmax[i >> 6] = m;
    }
    if (i < n) rows[i] = o;
}

// dna_length_by_cigar of the listed records, a wave each: 64 ops a step, coalesced (the same sum: M / = / X and D lengths; the row's other bits stay)
__global__ __launch_bounds__(64) void k6_prep_long(const uint8_t* s, const bgp_record* rec, const uint32_t* long_list, uint32_t long_cap, MdRow* rows) {
    const uint32_t cnt = min(long_list[0], long_cap);
    const int lane = threadIdx.x;
    for (uint32_t e = blockIdx.x; e < cnt; e += gridDim.x) {
        const uint32_t i = long_list[1 + e];
        const bgp_record r = rec[i];
        const uint32_t l_name = r.bin_mq_nl & 0xFFu, n_cig = r.flag_nc & 0xFFFFu;
        const uint8_t* cg = s + r.rec_off + 36 + l_name;
        unsigned long long sum = 0;
        auto take = [&](uint32_t raw) { const uint32_t op = raw & 15u; if ((cig_qc(raw) && cig_rc(raw)) || op == 2u) sum += raw >> 4; };
        uint32_t k = (uint32_t)lane;
        for (; k + 192 < n_cig; k += 256) {      // (four loads in flight a lane: the sum does not wait for one round trip an iteration)
            const uint32_t r0 = ld_u32(cg + 4ull * k), r1 = ld_u32(cg + 4ull * (k + 64)), r2 = ld_u32(cg + 4ull * (k + 128)), r3 = ld_u32(cg + 4ull * (k + 192));
            take(r0); take(r1); take(r2); take(r3);
        }
        for (; k < n_cig; k += 64) take(ld_u32(cg + 4ull * k));
        for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
        if (lane == 0) { const uint32_t L = sum > (unsigned long long)MD_LMASK ? MD_LMASK : (uint32_t)sum; rows[i].lf = (rows[i].lf & ~MD_LMASK) | L; }
    }
}

// ---------------------------------------------------------------------------------------------- k6_chain
// The provider chain of one reference run, one wave per run.  Lanes hold a batch of 64 consecutive rows;
// every decision is a ballot / wave arg-max over that batch, the state itself is wave-uniform.
//
// Restatement of PileupRangeUsingMdTag at event granularity.  With the chunk of provider P fetched (cursor
// ci, chunk_end_position CE) and the sweep at column y, nothing observable happens until the first of
//   FETCH at F = max(first column whose check sees an empty chunk, position of the first read that add()
//           would accept under CE)                                         (pileup.d:627-635, 563-596)
//   GAP   at E = where coverage ends before the next position group (skip_zero_coverage only): the sweep
//           jumps and re-initialises there, chunk or not                    (pileup.d:382-393, 598-611)
//   END   of the run at E = max end_position.
// add() keeps "the first read with the largest end_position among those it accepts", which does not
// depend on the order the reads are looked at -- so a batch is reduced with one arg-max.
// Batched staging (batch.cpp): the chain of a reference that began in an earlier batch is RESUMED -- between two events its whole
// state is (provider, cursor, chunk_end_position, column, next record to look at), nothing of add()'s bookkeeping is pending --
// and the last such state in front of `stop_at` (the position from which the next batch hands out columns) is saved for it.
struct ChainRun { uint64_t first, endrec, iv_base; uint32_t run_end, ce_in, active, resume; uint64_t rs_P, rs_i; uint32_t rs_ci, rs_y, stop_at, pad; };
struct ChainOut { uint32_t ce_out, n_iv; int32_t err; uint32_t sv_have; int64_t err_from; uint64_t sv_P, sv_i; uint32_t sv_ci, sv_CE, sv_y, err_dep; };      // err_dep: the error was decided from the reads that start at or before this position (batched staging: only final if the batch holds all of those)
struct MdIv { uint32_t col, c0; uint64_t prov; };
constexpr uint64_t MD_NONE = ~0ull, MD_INF = 1ull << 62;

// the largest value of the wave, in every lane: DPP inside the rows of 16, the rows' maxima carried over (six instructions and a
// readlane; the butterfly of __shfl_xor was six ds_bpermute round trips through the LDS for 32 bits, twelve for 64)
#define MD_DPP(v, ctrl, rmask) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rmask, 0xF, false)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = max(v, MD_DPP(v, 0xB1, 0xF)); v = max(v, MD_DPP(v, 0x4E, 0xF)); v = max(v, MD_DPP(v, 0x141, 0xF)); v = max(v, MD_DPP(v, 0x140, 0xF));
    v = max(v, MD_DPP(v, 0x142, 0xA)); v = max(v, MD_DPP(v, 0x143, 0xC));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {      // (the high words' maximum, then the low words' among the lanes that hold it)
    const uint32_t hi = (uint32_t)(v >> 32), mh = wave_max_u32(hi);
    const uint32_t ml = wave_max_u32(hi == mh ? (uint32_t)v : 0u);
    return ((uint64_t)mh << 32) | ml;
}

__global__ __launch_bounds__(64) void k6_chain(const MdRow* rows, const ChainRun* cr, ChainOut* co, MdIv* iv, int skip_zero) {
    const ChainRun R = cr[blockIdx.x];
    if (!R.active) return;
    const int lane = threadIdx.x;
    // chunk state (pileup.d:523-533)
    uint64_t P = MD_NONE; uint32_t L = 0, ci = 0, CE = R.ce_in; bool bad = false;
    bool has_next = false; uint64_t NX = MD_NONE; uint32_t NXend = 0, NXpos = 0, NXlf = 0;
    uint64_t y = 0, i = R.first;
    uint32_t n_iv = 0; int err = 0; int64_t err_from = -1; uint32_t err_dep = 0, init_x = 0;
    const uint64_t iv_cap = R.endrec - R.first + 2;
    uint64_t b = MD_NONE; MdRow row{0, 0, 0, 0};

    uint64_t svP = MD_NONE, svI = 0; uint32_t svci = 0, svCE = 0, svy = 0, svhave = 0;
    auto save = [&]() { if (y < (uint64_t)R.stop_at) { svP = P; svI = i; svci = ci; svCE = CE; svy = (uint32_t)y; svhave = 1; } };      // an event boundary: has_next is false here
    auto load = [&](uint64_t base) { b = base; const uint64_t idx = base + lane; row = idx < R.endrec ? rows[idx] : MdRow{0, 0, 0, 0}; };
    auto emit = [&](uint64_t col, uint32_t c0, uint64_t prov) {
        if (n_iv < iv_cap) { if (lane == 0) iv[R.iv_base + n_iv] = MdIv{(uint32_t)col, c0, prov}; ++n_iv; }
        else if (!err) { err = BGP_EINVAL; }
    };
    // add() over the lanes with `pass` set: first arg-max of end_position; an earlier candidate wins ties
    auto take = [&](bool pass) {
        if (!__ballot(pass)) return;
        const uint64_t key = wave_max_u64(pass ? ((uint64_t)row.end << 32) | (uint32_t)(63 - lane) : 0ull);
        const int wl = 63 - (int)(key & 63u);
        const uint32_t e = (uint32_t)(key >> 32);
        const uint32_t wp = __shfl(row.pos, wl, 64), wf = __shfl(row.lf, wl, 64);
        if (!has_next || e > NXend) { has_next = true; NX = b + wl; NXend = e; NXpos = wp; NXlf = wf; }
    };
    // initNewReference (pileup.d:399-424 + 598-611) at the position group that starts at record j
    auto init_at = [&](uint64_t j, bool new_ref) -> bool {
        if (j < b || j >= b + 64) load(j);
        const uint32_t x = __shfl(row.pos, (int)(j - b), 64);
        init_x = x;
        uint64_t k = j, ge;
        for (;;) {
            if (k >= b + 64) load(k);
            const uint64_t idx = b + lane;
            const bool cov = idx >= k && idx < R.endrec && (row.lf & MD_COV);
            const uint64_t stop = __ballot(cov && row.pos != x);
            const int fs = stop ? __ffsll((unsigned long long)stop) - 1 : 64;
            const bool ing = cov && lane < fs;
            if (new_ref && j >= b && j < b + 64 && k == j) {
                // the first read of a new reference is taken unconditionally (pileup.d:568-575)
                const int fl = (int)(j - b);
                has_next = true; NX = j; NXend = __shfl(row.end, fl, 64); NXpos = x; NXlf = __shfl(row.lf, fl, 64);
                take(ing && idx != j && row.end > CE);
            } else take(ing && row.end > CE);
            if (fs < 64) { ge = b + fs; break; }
            if (b + 64 >= R.endrec) { ge = R.endrec; break; }
            k = b + 64;
        }
        if (has_next) {
            P = NX; L = NXlf & MD_LMASK; bad = (NXlf & MD_BAD) != 0; ci = 0; CE = NXend; has_next = false;
            if (NXlf & MD_ERR) { err = BGP_EINVAL; return false; }
            if (L == 0) { err = BGP_EMD; return false; }          // chunk.front of an empty chunk
            emit(x, 0, P); ci = 1;
        } else {
            emit(x, 0, MD_NONE);                                   // 'N' here ...
            if (P != MD_NONE && ci < L) emit((uint64_t)x + 1, ci, P);   // ... and the old chunk simply goes on
        }
        y = x; i = ge;
        save();
        return true;
    };

    bool started;
    if (R.resume) {                                               // the state an earlier batch saved (record indices already in this batch's numbering)
        P = R.rs_P; ci = R.rs_ci; y = R.rs_y; i = R.rs_i;
        if (P != MD_NONE) { load(P); const uint32_t lf = __shfl(row.lf, 0, 64); L = lf & MD_LMASK; bad = (lf & MD_BAD) != 0; if (ci < L) emit(y + 1, ci, P); }
        started = true;
        save();      // a batch of a few reads may hold no event in front of its cut: the chain then stands where it stood, and the next batch resumes from this very state (it is not "begun afresh")
    } else { started = init_at(R.first, true); if (!started) err_dep = init_x; }
    if (started) {
        for (;;) {
            const uint64_t empty_at = bad ? MD_INF : y + (L > ci + 1 ? L - ci : 1u);
            bool found = false; uint64_t F = MD_INF;
            int outcome = 0; uint64_t jn = R.endrec; uint32_t egap = 0;            // 1 FETCH, 2 GAP, 3 END
            uint64_t j = i;
            for (;;) {
                if (j >= R.endrec) { const uint64_t E = R.run_end > y + 1 ? R.run_end : y + 1; outcome = (found && F < E) ? 1 : 3; jn = R.endrec; break; }
                if (j < b || j >= b + 64) load(j);
                const uint64_t idx = b + lane;
                const bool valid = idx >= j && idx < R.endrec;
                const bool cov = valid && (row.lf & MD_COV);
                const bool hz = row.mb < row.pos;
                const bool cand = cov && (row.pos <= CE || hz) && row.end > CE;      // add()'s filter, pileup.d:583-588
                uint64_t mg = __ballot(skip_zero && cov && (row.lf & MD_GST) && hz);
                int from = 0;
                if (!found) {
                    const uint64_t mc = __ballot(cand);
                    const int fc = mc ? __ffsll((unsigned long long)mc) - 1 : 64, fg = mg ? __ffsll((unsigned long long)mg) - 1 : 64;
                    if (fg < 64 && fg <= fc) { outcome = 2; jn = b + fg; egap = __shfl(row.mb, fg, 64); break; }
                    if (fc == 64) { j = b + 64; continue; }
                    found = true; from = fc;
                    const uint64_t pc = __shfl(row.pos, fc, 64);
                    F = empty_at > pc ? empty_at : pc;
                }
                const bool in_b = valid && lane >= from;
                const uint64_t mbey = __ballot(in_b && cov && (uint64_t)row.pos > F);
                mg &= ~0ull << from;
                const int fb = mbey ? __ffsll((unsigned long long)mbey) - 1 : 64, fg = mg ? __ffsll((unsigned long long)mg) - 1 : 64;
                if (fg < 64 && fg <= fb) {
                    const uint32_t mbv = __shfl(row.mb, fg, 64);
                    const uint64_t E = mbv > y + 1 ? mbv : y + 1;
                    if (fg < fb || E <= F) { take(in_b && cand && lane < fg); outcome = 2; jn = b + fg; egap = mbv; break; }
                    take(in_b && cand && lane < fb); outcome = 1; jn = b + fb; break;
                }
                if (fb < 64) { take(in_b && cand && lane < fb); outcome = 1; jn = b + fb; break; }
                take(in_b && cand);
                j = b + 64;
            }
            if (outcome == 1) {                                       // pileup.d:627-652
                P = NX; L = NXlf & MD_LMASK; bad = (NXlf & MD_BAD) != 0; CE = NXend; has_next = false;
                if (NXlf & MD_ERR) { err = BGP_EINVAL; err_from = (int64_t)F - 1; err_dep = (uint32_t)F; break; }      // (the provider is the best of the reads that start at or before F)
                const uint64_t skip = F - NXpos;                      // popFrontN(position - next_provider.position)
                ci = skip < L ? (uint32_t)skip : L;
                if (ci >= L) { err = BGP_EMD; err_from = (int64_t)F - 1; err_dep = (uint32_t)F; break; }
                emit(F, ci, P); ci += 1; y = F; i = jn;
                save();
                continue;
            }
            const uint64_t E = outcome == 2 ? (egap > y + 1 ? egap : y + 1) : (R.run_end > y + 1 ? R.run_end : y + 1);
            const uint64_t adv = E - y;                               // every popFront consumed a char first (pileup.d:615-622)
            if (bad && ci + adv > L) { err = BGP_EMD; err_from = (int64_t)(y + (L > ci ? L - ci : 0)); err_dep = (uint32_t)err_from; break; }
            ci = ci + adv < L ? (uint32_t)(ci + adv) : L;
            if (outcome == 3) break;
            if (!init_at(jn, false)) { err_from = (int64_t)E - 1; err_dep = init_x; break; }      // (the new chunk's provider is the best of the position group at init_x)
        }
    }
    if (lane == 0) co[blockIdx.x] = ChainOut{CE, n_iv, err, svhave, err_from, svP, svI, svci, svCE, svy, err_dep};
}

// ---------------------------------------------------------------------------------------------- k6_fill
// One lane per provider interval: dna(provider)[c0 ...] into the dense reference_base array, up to the next
// interval of the run (or the run's end).  The array is pre-filled with 'N' (empty chunk, pileup.d:620-622).
struct FillRun { uint64_t iv_base, iv_prefix, column_base; uint32_t start_pos, end_pos, n_iv, pad; };

__global__ __launch_bounds__(256) void k6_fill(const uint8_t* s, const bgp_record* rec, const MdIv* iv, const FillRun* fr, uint32_t n_runs, uint64_t total, uint8_t* refbase) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    uint32_t lo = 0, hi = n_runs;                                      // last run with iv_prefix <= t
    while (hi - lo > 1) { const uint32_t m = (lo + hi) >> 1; if (fr[m].iv_prefix <= t) lo = m; else hi = m; }
    const FillRun R = fr[lo];
    const uint64_t k = t - R.iv_prefix;
    const MdIv v = iv[R.iv_base + k];
    if (v.prov == MD_NONE) return;
    uint64_t endc = k + 1 < R.n_iv ? iv[R.iv_base + k + 1].col : R.end_pos;
    if (endc > R.end_pos) endc = R.end_pos;
    if (v.col >= endc) return;
    bool bad, err;
    dna_walk(s, rec[v.prov], v.c0, (uint32_t)(endc - v.col), refbase + R.column_base + (v.col - R.start_pos), &bad, &err);
}

// ---------------------------------------------------------------------------------------------- the chain, in parallel
// k6_chain walks one run fetch by fetch; 2*10^5 dependent steps per run is 10^2 ms.  But the step is a FUNCTION:
// in the regular regime the next provider depends on the current provider alone.  Proof sketch (state after a
// fetch = provider P fetched at column F, cursor F - pos_P + 1, chunk_end CE = end_P; "regular" = F < pos_P + L_P - 1
// and F < end_P):
//   * the chunk runs empty at A_P = pos_P + L_P - 1 whatever F was;
//   * add() only ever looks at reads behind the ones it has seen.  Reads between P and the fetch column were seen
//     under the old chunk_end: those it accepted lost to P (end <= end_P, so they fail the new filter too); those
//     it ignored (position beyond the old chunk_end, covered predecessor column) are the one way F can matter --
//     so a step is only trusted if NO read of its window, accepted or not, ends beyond the winner;
//   * a skip_zero_coverage jump re-initialises on the first arg-max of the gap's position group, and every read
//     of that group passes add() because coverage was zero in front of it: independent of the state altogether.
// So: k6_succ computes succ(P) = (next provider, its fetch column) for EVERY read at once and flags the steps
// whose premises fail; k6_exit composes succ within 2048-record segments by pointer doubling in LDS (exit(P) =
// first provider outside P's segment); k6_top walks a run segment by segment (10^3 dependent loads instead of
// 10^5) and notes where the real chain enters each segment; k6_emit re-walks each segment from its entry and
// expands the chunks.  A run whose chain touches a flagged step (inconsistent MD, bad chunk, the corner where a
// chunk is fetched on its last base, ...) is redone by k6_chain -- the exact event loop stays the authority,
// and every BGP_EMD comes from it.
constexpr uint32_t SX_END = 0xFFFFFFFFu, SX_NONE = 0xFFFFFFFEu, SX_FLAG = 1u << 31;
#ifndef BGP_SUCC_LV
#define BGP_SUCC_LV 6
#endif
#ifndef BGP_MD_SEG
#define BGP_MD_SEG 2048                       // (a test build with 32 makes a thousand reads many segments: the fuzz loop then runs k6_top's parts and meetings)
#endif
constexpr int SEG = BGP_MD_SEG, SUCC_CAP = 4096;
#ifndef BGP_SUCC_LANE_CAP
#define BGP_SUCC_LANE_CAP 128                 // rows a lane of the general walk looks at before it hands its read to the wave
#endif
#ifndef BGP_SUCC_SCAN
#define BGP_SUCC_SCAN 0                       // 1: every successor by the general row-by-row walk (rounds 2 and 3), for A/B runs
#endif
struct Succ { uint32_t nxt, fcol; };              // fcol: fetch column of nxt | SX_FLAG

struct ParRun { uint64_t first, endrec; uint32_t run_end, ce_in, active, pad; };
struct ParOut { uint32_t ce_out, flagged, n_nodes, pad; };

__device__ __forceinline__ uint32_t par_run_of(const ParRun* pr, uint32_t n_runs, uint64_t i) {   // last run with first <= i
    uint32_t lo = 0, hi = n_runs;
    while (hi - lo > 1) { const uint32_t m = (lo + hi) >> 1; if (pr[m].first <= i) lo = m; else hi = m; }
    return lo;
}

// arrival checks for provider row r fetched at column F (the premises of the next step)
__device__ __forceinline__ bool arrival_irregular(const MdRow& r, uint32_t F) {
    const uint32_t L = r.lf & MD_LMASK;
    if ((r.lf & (MD_BAD | MD_ERR)) || L < 2) return true;
    return !((uint64_t)F + 1 < (uint64_t)r.pos + L && F < r.end);
}

// The same successor, computed by a whole wave (64 rows per step, ballots + one arg-max): for the few providers whose
// window is long -- a read with a 5 kb N skip has chunk_end 5 kb away and must look at a thousand rows before one
// ends beyond it.  Left to its own lane such a read would hold its 63 neighbours for a thousand iterations.
// (rows the workgroup has staged in LDS -- the 512 from `sbase` on -- are taken from there: a deferred read's candidates mostly lie among them,
// and with every step's rows fetched from memory the two or three deferred reads of a wave were two or three chains of round trips)
__device__ Succ succ_coop(const MdRow* rows, const MdRow* srows, uint64_t sbase, uint32_t sn, const uint32_t* bmax, const ParRun& R, uint64_t P, const MdRow& rp, int skip_zero, int lane) {
    auto row_at = [&](uint64_t idx) -> MdRow { const uint64_t d = idx - sbase; return d < (uint64_t)sn ? srows[d] : rows[idx]; };
    const uint32_t CE = rp.end, A = rp.pos + (rp.lf & MD_LMASK) - 1u;
    bool found = false, flag = false;
    uint32_t F = 0, best = SX_END, best_end = 0, allmax = 0;
    MdRow bestrow{0, 0, 0, 0};
    auto wmax32 = [](uint32_t v) { return wave_max_u32(v); };
    auto bcast_row = [](const MdRow& r, int l) { return MdRow{(uint32_t)__builtin_amdgcn_readlane((int)r.pos, l), (uint32_t)__builtin_amdgcn_readlane((int)r.end, l), (uint32_t)__builtin_amdgcn_readlane((int)r.lf, l), (uint32_t)__builtin_amdgcn_readlane((int)r.mb, l)}; };      // (l is the same in every lane)
    uint64_t jb = P + 1;
    int budget = SUCC_CAP;
    int width = 64;
    for (; jb < R.endrec; jb += width) {
        // Before the first candidate the steps end on multiples of 64 rows, and whole blocks of 64 in which nothing ends beyond the chunk
        // and no position group stands behind uncovered ground (k6_prep's per-block maximum, its top bit) are stepped over 64 blocks at a
        // time: nothing in them is a candidate or a GAP, and what they add to "the largest end in front of the window" is their maximum.
        // (A read with a 5 kb N skip looked at a thousand rows, 16 steps, before one ended beyond its chunk: these reads, 4 % of cfg3's,
        // were most of k6_succ's 5.1 ms.)
        width = found ? 64 : 64 - (int)(jb & 63);
        if (!found && (jb & 63) == 0) {
            bool moved = false;
            for (;;) {
                const uint64_t blk = (jb >> 6) + (uint64_t)lane;
                const bool whole = (blk << 6) + 64 <= R.endrec;
                const uint32_t bm = whole ? bmax[blk] : 0xFFFFFFFFu;
                const uint64_t stop = __ballot(bm > CE);
                const int nskip = stop ? __ffsll((unsigned long long)stop) - 1 : 64;
                if (nskip) { allmax = max(allmax, wmax32(lane < nskip ? bm : 0u)); jb += 64ull * (uint64_t)nskip; moved = true; }
                if (nskip < 64) break;
            }
            if (moved) { width = 0; continue; }                              // (the loop's condition looks at the new jb)
        }
        if ((budget -= 64) < 0) { flag = true; break; }
        const uint64_t idx = jb + lane;
        const bool inb = lane < width && idx < R.endrec;
        const MdRow r = inb ? row_at(idx) : MdRow{0, 0, 0, 0};
        const bool cov = inb && (r.lf & MD_COV);
        const bool hz = r.mb < r.pos;
        const bool isg = cov && skip_zero && (r.lf & MD_GST) && hz;
        const bool cand = cov && (r.pos <= CE || hz) && r.end > CE;
        int from = 0;
        int gap_at = 64;
        if (!found) {
            const uint64_t mc = __ballot(cand), mg = __ballot(isg);
            const int fc = mc ? __ffsll((unsigned long long)mc) - 1 : 64, fg = mg ? __ffsll((unsigned long long)mg) - 1 : 64;
            if (fg < 64 && fg <= fc) gap_at = fg;
            else if (fc == 64) { allmax = max(allmax, wmax32(cov ? r.end : 0u)); continue; }
            else {
                found = true; from = fc;
                const uint32_t pc = (uint32_t)__shfl((int)r.pos, fc, 64);
                F = max(A, pc);
                allmax = max(allmax, wmax32(cov && lane < fc ? r.end : 0u));
            }
        }
        int fb = 64;
        if (gap_at == 64) {
            const uint64_t mb_ = __ballot(cov && lane >= from && r.pos > F);
            fb = mb_ ? __ffsll((unsigned long long)mb_) - 1 : 64;
            const uint64_t mg2 = __ballot(isg && lane >= from && lane <= fb && (r.pos <= F || r.mb <= F));
            if (mg2) gap_at = __ffsll((unsigned long long)mg2) - 1;
        }
        const int upto = min(gap_at, fb);                                   // lanes [from, upto) are inside the window
        if (found) {
            const bool in_w = cov && lane >= from && lane < upto;
            allmax = max(allmax, wmax32(in_w ? r.end : 0u));
            const bool c2 = in_w && cand;
            const uint64_t key = wave_max_u64(c2 ? ((uint64_t)r.end << 32) | (uint32_t)(63 - lane) : 0ull);
            if (key && (uint32_t)(key >> 32) > best_end) { const int wl = 63 - (int)(key & 63u); best = (uint32_t)(jb + wl); best_end = (uint32_t)(key >> 32); bestrow = bcast_row(r, wl); }
        }
        if (gap_at < 64) {
            // initNewReference on the gap's position group: first arg-max of end_position
            const MdRow g0 = bcast_row(r, gap_at);
            uint32_t g = (uint32_t)(jb + gap_at), ge = g0.end; MdRow gr = g0;
            for (uint64_t kb = jb + gap_at + 1; kb < R.endrec; kb += 64) {
                if ((budget -= 64) < 0) { flag = true; break; }
                const uint64_t ik = kb + lane;
                const MdRow q = ik < R.endrec ? row_at(ik) : MdRow{0, 0, 0, 0};
                const bool qc = ik < R.endrec && (q.lf & MD_COV);
                const uint64_t stop = __ballot(qc && q.pos != g0.pos);
                const int fs = stop ? __ffsll((unsigned long long)stop) - 1 : 64;
                const bool ing = qc && lane < fs;
                const uint64_t key = wave_max_u64(ing ? ((uint64_t)q.end << 32) | (uint32_t)(63 - lane) : 0ull);
                if (key && (uint32_t)(key >> 32) > ge) { const int wl = 63 - (int)(key & 63u); g = (uint32_t)(kb + wl); ge = (uint32_t)(key >> 32); gr = bcast_row(q, wl); }
                if (fs < 64) break;
            }
            flag |= arrival_irregular(gr, g0.pos) || gr.pos != g0.pos;
            return Succ{g, g0.pos | (flag ? SX_FLAG : 0u)};
        }
        if (fb < 64) {                                                      // FETCH at F
            flag |= best == SX_END || arrival_irregular(bestrow, F) || allmax > best_end;
            return Succ{best, F | (flag ? SX_FLAG : 0u)};
        }
    }
    if (found && F < R.run_end && !(flag && best == SX_END)) {              // the run's reads are exhausted: FETCH only if it comes before the end of coverage
        flag |= best == SX_END || arrival_irregular(bestrow, F) || allmax > best_end;
        return Succ{best, F | (flag ? SX_FLAG : 0u)};
    }
    return Succ{SX_END, flag ? SX_FLAG : 0u};
}

__global__ __launch_bounds__(256) void k6_succ(const MdRow* rows, const uint32_t* bmax, uint64_t n, const ParRun* pr, uint32_t n_runs, int skip_zero, Succ* succ) {
    // the window a lane scans is the ~50 rows behind its own: the block's 256 rows + 256 more sit in LDS (adjacent
    // lanes read adjacent rows, conflict free); a lane that needs more than LANE_CAP rows hands its read to the wave
    constexpr int STAGE = 512;
#if BGP_SUCC_SCAN
    constexpr int LANE_CAP = BGP_SUCC_LANE_CAP;
#endif
    __shared__ MdRow s_rows[STAGE];
    const uint64_t base = (uint64_t)blockIdx.x * blockDim.x;
    for (int t = threadIdx.x; t < STAGE; t += 256) s_rows[t] = base + t < n ? rows[base + t] : MdRow{0, 0, 0, 0};
    __syncthreads();
#if !BGP_SUCC_SCAN
    // "First read with the largest end among rows (P, hi]" for 256 overlapping windows of the same 512 staged rows is a range-maximum
    // query: a sparse table over the rows' ends (key = end relative to the first staged position, then 511 - row, so that the
    // maximum of two keys is the larger end and, of equal ends, the earlier row), built once by the workgroup -- nine levels of two
    // entries a thread -- and asked twice a lane; where the window ends is a binary search over the staged positions, whether a
    // position group in it stands behind uncovered ground a difference of two prefix counts.  (A lane looking at its ~30-50 rows
    // one by one, all lanes as long as the slowest: 810 of the kernel's 1270 VALU instructions a wave.)  Ends that do not fit the
    // key's 22 bits (an N skip of megabases) send the workgroup through the row loop below instead.
    constexpr int LV = BGP_SUCC_LV;                                           // windows of up to 2^LV - 1 rows are asked of the table, longer ones (rare) walk their rows
    __shared__ uint32_t s_key[LV][STAGE];
    __shared__ uint32_t s_sp[STAGE + 1];                                      // rows in front of row j that stand behind uncovered ground
    __shared__ uint32_t s_wsum[8], s_fallback;
    const uint32_t pos_base = s_rows[0].pos;
    if (threadIdx.x == 0) s_fallback = 0u;
    __syncthreads();
    {
        uint32_t spc[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = 2 * (int)threadIdx.x + h;
            const MdRow r = s_rows[j];
            const bool cov = (r.lf & MD_COV) != 0;
            const uint32_t rel = r.end - pos_base;
            if (cov && (r.end < pos_base || rel >= (1u << 22))) s_fallback = 1u;
            s_key[0][j] = cov ? ((rel + 1u) << 9) | (uint32_t)(STAGE - 1 - j) : 0u;      // (never 0 for a covering row)
            spc[h] = cov && r.mb < r.pos ? 1u : 0u;
        }
        // exclusive prefix of the "behind uncovered ground" marks: a thread's two rows, the wave, the four waves
        const uint32_t mine = spc[0] + spc[1];
        uint32_t inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64); if ((int)(threadIdx.x & 63) >= d) inc += o; }
        if ((threadIdx.x & 63) == 63) s_wsum[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t before = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += s_wsum[w];
        const uint32_t ex = before + inc - mine;
        s_sp[2 * threadIdx.x] = ex; s_sp[2 * threadIdx.x + 1] = ex + spc[0];
        if (threadIdx.x == 255) s_sp[STAGE] = ex + mine;
        // two levels a barrier: level k and k + 1 both from level k - 1 (four reads)
#pragma unroll 1
        for (int k = 1; k < LV; k += 2) {
            __syncthreads();
            const int half = 1 << (k - 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int j = (int)threadIdx.x + 256 * h;
                const uint32_t a0 = s_key[k - 1][j], a1 = j + half < STAGE ? s_key[k - 1][j + half] : 0u;
                const uint32_t m01 = max(a0, a1);
                s_key[k][j] = m01;
                if (k + 1 < LV) {
                    const uint32_t a2 = j + 2 * half < STAGE ? s_key[k - 1][j + 2 * half] : 0u, a3 = j + 3 * half < STAGE ? s_key[k - 1][j + 3 * half] : 0u;
                    s_key[k + 1][j] = max(m01, max(a2, a3));
                }
            }
        }
        __syncthreads();
    }
    const bool table_ok = s_fallback == 0u;
#endif
#if BGP_SUCC_SCAN
    auto row_at = [&](uint64_t j) -> MdRow { const uint64_t d = j - base; return d < (uint64_t)STAGE ? s_rows[d] : rows[j]; };
#endif
    const int lane = threadIdx.x & 63;
    const uint64_t P = base + threadIdx.x;
    const MdRow rp = s_rows[threadIdx.x];
    ParRun R{0, 0, 0, 0, 0, 0};
    bool node = P < n && (rp.lf & MD_COV);
    if (node) { R = pr[par_run_of(pr, n_runs, P)]; node = P >= R.first && P < R.endrec; }
    const uint32_t CE = rp.end, A = rp.pos + (rp.lf & MD_LMASK) - 1u;
#if !BGP_SUCC_SCAN
    // The regular step in closed form.  With A = the column the chunk empties at (pos + len(dna) - 1) in front of chunk_end (CE = the
    // provider's end) and no position group behind uncovered ground among the reads that start up to A or right behind them, the sweep's
    // bookkeeping collapses: some read starting at or before A ends beyond CE, so add() has a candidate before the chunk is empty, the
    // fetch column is A (pileup.d:627-635), and the new provider is the FIRST read with the LARGEST end among those starting in
    // (P, A] (pileup.d:563-596: every such read ending beyond CE passes the filter, the others cannot win).  That is one arg-max
    // over ~30 staged rows -- four instructions a row where the general walk below spends ~90 on its cases -- and the same for all
    // lanes of a wave (a read with a 5 kb N skip has a long CE, not a long dna).  Whatever does not fit (no candidate by A: N-skip
    // providers and coverage edges; uncovered ground; rows beyond the staged ones) is handed to the wave, succ_coop, as the walk's
    // long windows always were.
    bool flag = false, deferred = false;
    uint32_t out_n = node ? SX_END : SX_NONE, out_f = 0;
    {
        const uint32_t L = rp.lf & MD_LMASK;
        bool ok = node && L >= 1u && A >= rp.pos && A < CE;
        uint32_t best = SX_END, best_end = 0;
        // (a step without a branch: written with one per case the loop spent more scalar instructions on exec masks -- 35 a row -- than
        // vector ones on the rows, 1373 SALU + 894 VALU a wave)
        uint32_t jr = (uint32_t)(P + 1 - base);                              // the row looked at, as an index into the staged rows
        const uint32_t limr = (uint32_t)(min(R.endrec, base + (uint64_t)STAGE) - base);
        bool special = false, ended = false;                                 // ended: a covering read beyond A closed the window
        bool run = ok && !table_ok;
        if (table_ok && ok) {
            // rows [jr, je): the rows behind P that start at or before A (the staged rows of a run are sorted by position)
            uint32_t lo = jr, hi = max(jr, limr);
            while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (s_rows[m].pos <= A) lo = m + 1; else hi = m; }
            const uint32_t je = lo;
            if (je > jr && je - jr >= (1u << LV)) run = true;                 // (a window longer than the table's levels reach)
            else {
            if (je < limr) { const MdRow rc = s_rows[je]; const bool cc = (rc.lf & MD_COV) != 0; ended = cc; special = !cc || rc.mb < rc.pos; }      // the row that closes the window (its group's GAP included)
            special |= s_sp[je] != s_sp[min(jr, je)];
            if (je > jr) {
                const uint32_t len = je - jr, k = 31u - (uint32_t)__builtin_clz(len);
                const uint32_t m = max(s_key[k][jr], s_key[k][je - (1u << k)]);
                if (m) { best_end = (m >> 9) - 1u + pos_base; best = (uint32_t)base + (uint32_t)(STAGE - 1) - (m & 511u); }
            }
            jr = max(jr, je);
            }
        }
        while (__ballot(run)) {
            const MdRow r = s_rows[min(jr, (uint32_t)STAGE - 1u)];
            const bool inr = run && jr < limr;
            const bool cov = (r.lf & MD_COV) != 0, beyond = r.pos > A;
            const bool take = inr && !beyond;
            special |= inr && ((cov && r.mb < r.pos) || (beyond && !cov));    // (the closing read's group included: a GAP would be taken there, pileup.d:382-393)
            ended |= inr && beyond && cov;
            const uint32_t e = take && cov ? r.end : 0u;
            best = e > best_end ? (uint32_t)base + jr : best;
            best_end = max(best_end, e);
            jr += take ? 1u : 0u;
            run = take;
        }
        const uint64_t j = base + jr;
        const bool ok0 = ok;
        ok = ok && !special && best_end > CE && (ended || j >= R.endrec);      // (stopped by the staged rows' end: not decided here)
        if (ok) {
            if (ended || A < R.run_end) {                                    // FETCH at A; at the run's end only if it comes before the end of coverage
                out_n = best; out_f = A;
                flag = arrival_irregular(s_rows[best - base], A);
            }
        } else {
            // No read that starts by A ends beyond the chunk (the provider has an N skip: its chunk_end lies kilobases behind its last char): the
            // chunk runs empty, the sweep goes on to the FIRST read that ends beyond chunk_end -- add() takes nothing before it, pileup.d:583-588 --,
            // fetches at F = max(A, its position) and keeps the first read with the largest end among those that start by F.  With the table that
            // is a search by doubling for the first row whose end exceeds CE, a binary search for the window's end and one more range maximum;
            // the wave's walk (succ_coop) did it with three or four steps of 64 rows for each such read, one read after the other -- half of this
            // kernel's time for cfg3's 4 % of reads with a skip.  Whatever is not plainly this case is still handed to it.
            bool done2 = false;
            if (table_ok && ok0 && !special && ended && best_end <= CE && CE >= pos_base && CE - pos_base < (1u << 22)) {
                const uint32_t ce_key = ((CE - pos_base + 1u) << 9) | 511u;      // a key is larger exactly when its row ends beyond CE
                uint32_t x = jr;                                                  // (the row that closed the window by A)
                while (x + (1u << (LV - 1)) <= limr && s_key[LV - 1][x] <= ce_key) x += 1u << (LV - 1);
#pragma unroll
                for (int k = LV - 2; k >= 0; --k) if (x + (1u << k) <= limr && s_key[k][x] <= ce_key) x += 1u << k;
                if (x < limr && s_key[0][x] > ce_key) {
                    const MdRow rj = s_rows[x];
                    if (rj.pos <= CE) {
                        const uint32_t F = max(A, rj.pos);
                        uint32_t lo2 = x + 1u, hi2 = limr;
                        while (lo2 < hi2) { const uint32_t m = (lo2 + hi2) >> 1; if (s_rows[m].pos <= F) lo2 = m + 1; else hi2 = m; }
                        const uint32_t y = lo2;                                   // rows [x, y) start by F; row y closes the window
                        if (y < limr && y - x < (1u << LV) && (s_rows[y].lf & MD_COV) && s_sp[y + 1] == s_sp[jr]) {
                            const uint32_t len = y - x, k = 31u - (uint32_t)__builtin_clz(len);
                            const uint32_t m = max(s_key[k][x], s_key[k][y - (1u << k)]);
                            const uint32_t b2 = (uint32_t)base + (uint32_t)(STAGE - 1) - (m & 511u);
                            out_n = b2; out_f = F;
                            flag = arrival_irregular(s_rows[b2 - base], F);
                            done2 = true;
                        }
                    }
                }
            }
            if (!done2) deferred = node;
        }
    }
#else
    bool found = false, flag = false, deferred = false, done = !node;
    uint32_t F = 0, best = SX_END, best_end = 0, allmax = 0;
    MdRow bestrow{0, 0, 0, 0};
    uint32_t out_n = node ? SX_END : SX_NONE, out_f = 0;
    uint64_t j = P + 1;
    int budget = LANE_CAP;
    // One loop for the wave, the cases of a row as predicates: written as a per-lane loop with a branch per case the kernel
    // issued more scalar instructions (exec-mask bookkeeping at every `continue` / `break`: ~96 a row) than vector ones,
    // and the scalar pipe -- one issue per SIMD every four cycles -- was what it waited for (SQ counters: 9.6 k SALU against
    // 4.3 k VALU instructions a wave, 52 % of the wave cycles waiting for an issue slot).
    bool run = !done;                                                        // this lane is still walking rows
    while (__ballot(run && j < R.endrec)) {
        bool act = run && j < R.endrec;
        budget -= act ? 1 : 0;
        if (act && budget < 0) { deferred = true; done = true; run = false; act = false; }
        MdRow r{0, 0, 0, 0};
        if (__ballot(act && ((!found && (j & 63) == 0) || j - base >= (uint64_t)STAGE))) {      // (one test for the two rare cases of a step)
            // nothing a row must be looked at for -- a candidate, a position group behind uncovered ground, an end beyond
            // the winner's -- exists in a block of 64 rows none of which ends beyond this read (a read with a 5 kb N skip
            // would otherwise walk a thousand rows, and hold its 63 neighbours for as long)
            if (act && !found && (j & 63) == 0) {
                while (j + 64 <= R.endrec && bmax[j >> 6] <= CE) j += 64;
                if (j >= R.endrec) { run = false; act = false; }
            }
            if (act && j - base >= (uint64_t)STAGE) r = rows[j];                                      // a row behind the staged ones
        }
        if (act && j - base < (uint64_t)STAGE) r = s_rows[j - base];
        const bool cov = act && (r.lf & MD_COV);
        const bool hz = r.mb < r.pos;
        const bool isg = skip_zero && (r.lf & MD_GST) && hz;
        const bool cand = (r.pos <= CE || hz) && r.end > CE;
        const bool gap_here = cov && isg && (!found || r.pos <= F || r.mb <= F);   // GAP: before any candidate, inside the window, or the window's successor group uncovered by F
        if (__ballot(gap_here)) {
            if (gap_here) {
                // initNewReference on this position group: first arg-max of end_position (every read of it passes add())
                uint32_t g = (uint32_t)j, ge = r.end; MdRow gr = r;
                for (uint64_t k = j + 1; k < R.endrec; ++k) {
                    if (--budget < 0) { deferred = true; break; }
                    const MdRow q = row_at(k);
                    if (!(q.lf & MD_COV)) continue;
                    if (q.pos != r.pos) break;
                    if (q.end > ge) { g = (uint32_t)k; ge = q.end; gr = q; }
                }
                out_n = g; out_f = r.pos;
                flag |= arrival_irregular(gr, r.pos) || gr.pos != r.pos;
                done = true; run = false;
            }
        }
        const bool row = cov && !gap_here;                                    // a covering row that is not a gap's
        const bool first = row && !found && cand;                              // the first candidate fixes the fetch column
        if (row && !found && !cand) allmax = max(allmax, r.end);
        if (first) { found = true; F = max(A, r.pos); }
        const bool inwin = row && found;
        const bool beyond = inwin && r.pos > F;                                 // first read beyond the fetch column: the walk ends ON it (j stays)
        if (beyond) run = false;
        if (inwin && !beyond) {
            allmax = max(allmax, r.end);
            if (cand && r.end > best_end) { best = (uint32_t)j; best_end = r.end; bestrow = r; }
        }
        if (act && run) ++j;
    }
    if (!done && found && (j < R.endrec || F < R.run_end)) {                // FETCH at F (pileup.d:627-652); at the run's end only if it comes before the end of coverage
        out_n = best; out_f = F;
        flag |= best == SX_END || arrival_irregular(bestrow, F) || allmax > best_end;
    }
#endif
    Succ res{out_n, out_f | (flag ? SX_FLAG : 0u)};
    // long windows: one read at a time, the whole wave on it
    for (uint64_t dm = __ballot(deferred); dm; dm &= dm - 1) {
        const int src = __ffsll((unsigned long long)dm) - 1;
        const uint64_t Ps = base + (threadIdx.x & ~63u) + src;
        const MdRow rs = s_rows[(threadIdx.x & ~63u) + src];
        ParRun Rs;
        Rs.first = __shfl(R.first, src, 64); Rs.endrec = __shfl(R.endrec, src, 64); Rs.run_end = (uint32_t)__shfl((int)R.run_end, src, 64); Rs.ce_in = 0; Rs.active = 0; Rs.pad = 0;
        const Succ c = succ_coop(rows, s_rows, base, (uint32_t)STAGE, bmax, Rs, Ps, rs, skip_zero, lane);
        if (lane == src) res = c;
    }
    if (P < n) succ[P] = res;
}

// exit(P): the first provider behind P's segment on the chain from P (or SX_END), its fetch column, and the OR of
// the flags on the way -- pointer doubling over the segment's 2048 entries in LDS (a hop advances by at least
// one record, so 11 rounds always suffice; typical chains need 7).
__global__ __launch_bounds__(256) void k6_exit(const Succ* succ, uint64_t n, Succ* ex) {
    // Pointer jumping IN PLACE: an entry is written by its own thread only and read by others as one 64-bit word, and whatever a reader
    // finds there -- the entry as it was or already jumped -- is a stretch of the chain from that record on with the flags met on it, so
    // composing with it is right either way (and a round gets further than with a copy of the round before: 16 KB of LDS a block, not 32).
    __shared__ unsigned long long buf[SEG];
    __shared__ int changed;
    static_assert(sizeof(Succ) == 8, "an entry is one 64-bit word");
    const uint64_t lo = (uint64_t)blockIdx.x * SEG;
    auto pack = [](const Succ& e) { return ((unsigned long long)e.fcol << 32) | e.nxt; };
    auto unpack = [](unsigned long long w) { return Succ{(uint32_t)w, (uint32_t)(w >> 32)}; };
    for (int t = threadIdx.x; t < SEG; t += 256) buf[t] = pack(lo + t < n ? succ[lo + t] : Succ{SX_NONE, 0u});
    __syncthreads();
    for (int round = 0; round < 12; ++round) {
        if (threadIdx.x == 0) changed = 0;
        __syncthreads();
        bool any = false;
        for (int t = threadIdx.x; t < SEG; t += 256) {
            Succ e = unpack(buf[t]);
            if (e.nxt < SX_NONE && e.nxt >= lo && e.nxt < lo + SEG) {
                const Succ f = unpack(buf[e.nxt - lo]);
                e = Succ{f.nxt, f.fcol | (e.fcol & SX_FLAG) | (f.nxt == SX_NONE ? SX_FLAG : 0u)};
                buf[t] = pack(e);
                any = true;
            }
        }
        if (any) changed = 1;
        __syncthreads();
        if (!changed) break;
    }
    for (int t = threadIdx.x; t < SEG; t += 256) if (lo + t < n) ex[lo + t] = unpack(buf[t]);
}

// One wave per run: the first provider (initNewReference on the run's first position group, filtered against the
// chunk_end carried over from the previous reference), then segment to segment through the exit table.
//
// The walk is a chain of dependent loads, one per segment the chain enters (cfg3: 6 100 hops a reference, 2.9 ms with one lane at
// it).  But the chain forgets where it came from: exit(P) depends on P alone, so two walks that ever stand on the same provider are
// one walk from there on -- and they meet within a segment or two, because whatever the start, the next provider is "the read
// reaching furthest" of nearly the same reads.  So the run's segments are dealt out to the 64 lanes in equal parts; lane 0 starts
// on the real first provider, every other lane on the first covering read of its part (a guess), and all walk their parts side
// by side, leaving their trails in seg_entry.  Then lane p walks ON into part p + 1 until it stands on a provider of that part's
// trail: the true chain enters part p + 1 the way lane p does and is lane p + 1's trail from the meeting point on (by induction
// from lane 0, whose trail is true).  The slots in front of the meeting point are rewritten by the lane that walked in.  A part
// whose trail is never met (or a chain that ends on the way) sends the whole run through the one-lane walk: the guess only ever
// costs time.  Flags are gathered over everything any lane walked -- a superset of the chain's.
__device__ __forceinline__ Succ ld_entry(const Succ* p) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return Succ{(uint32_t)v, (uint32_t)(v >> 32)};
}
__global__ __launch_bounds__(64) void k6_top(const MdRow* rows, const Succ* ex, const ParRun* pr, ParOut* po, Succ* seg_entry) {
    const ParRun R = pr[blockIdx.x];
    if (!R.active) return;
    const int lane = threadIdx.x;
    bool flag = false;
    uint32_t node0 = 0, F0 = 0;
    if (lane == 0) {
        // add() of the first group: the first read unconditionally, a later one if it ends beyond the stale chunk_end and beyond the best so far
        const MdRow r0 = rows[R.first];
        uint32_t node = (uint32_t)R.first, best_end = r0.end, later_max = 0; MdRow nr = r0;
        int budget = SUCC_CAP;
        for (uint64_t k = R.first + 1; k < R.endrec; ++k) {
            if (--budget < 0) { flag = true; break; }
            const MdRow q = rows[k];
            if (!(q.lf & MD_COV)) continue;
            if (q.pos != r0.pos) break;
            if (q.end > R.ce_in && q.end > best_end) { node = (uint32_t)k; best_end = q.end; nr = q; later_max = 0; }
            else later_max = max(later_max, q.end);
        }
        // The one place where chunk_end can DROP is here: the stale value of the previous reference may exceed the winner's
        // end.  A read of this group that add() turned down for not ending beyond the stale value, but that ends beyond the
        // winner, would look like a candidate to succ() although the sweep never sees it again: not a regular start.
        flag |= later_max > best_end;
        flag |= arrival_irregular(nr, r0.pos);
        node0 = node; F0 = r0.pos;
    }
    const uint64_t seg_f = R.first / SEG, seg_l = (R.endrec - 1) / SEG;
    const uint64_t per = (seg_l - seg_f + 64) / 64;                           // segments a part
    const uint64_t p_lo = seg_f + (uint64_t)lane * per, p_hi = min(p_lo + per, seg_l + 1);
    const bool has = p_lo <= seg_l;
    const uint32_t n_parts = (uint32_t)((seg_l - seg_f + per) / per);
    auto slot = [&](uint64_t seg) { return seg + blockIdx.x; };                // slot = segment + run ordinal: two runs may share a segment
    const bool flag0 = __builtin_amdgcn_readfirstlane((int)flag) != 0;         // the first provider's own premises
    // ---- every lane its own part
    uint32_t node = node0, F = F0, hops = 0;
    bool live = lane == 0;                                                     // this lane stands on a provider
    if (lane != 0 && has && !flag0) {
        const uint64_t from = max(p_lo * SEG, R.first), to = min(min(p_hi * SEG, R.endrec), from + 256);
        for (uint64_t k = from; k < to; ++k) { const MdRow q = rows[k]; if (q.lf & MD_COV) { node = (uint32_t)k; F = q.pos; live = true; break; } }
    }
    // a walk from (node, F) while the provider lies in a segment < hi.  write: leave the trail; meet: stop on a provider of the trail that is
    // there.  Returns the last segment a flagged hop (or the chain's irregular end) left from, -1 if none.
    auto walk = [&](uint64_t hi, bool write, bool meet, uint64_t* met_seg) -> int64_t {
        int64_t lfs = -1;
        while (live) {
            const uint64_t seg = node / SEG;
            if (seg >= hi) break;
            if (meet) { const Succ t = ld_entry(seg_entry + slot(seg)); if (t.nxt == node) { *met_seg = seg; break; } }
            if (write) seg_entry[slot(seg)] = Succ{node, F};
            const Succ e = ex[node];
            if ((e.fcol & SX_FLAG) || e.nxt == SX_NONE) lfs = (int64_t)seg;
            ++hops;
            if (e.nxt >= SX_NONE) { live = false; break; }                    // the chain ends here
            node = e.nxt; F = e.fcol & ~SX_FLAG;
        }
        return lfs;
    };
    uint64_t met = ~0ull;
    const int64_t lfs_own = walk(p_hi, true, false, &met);
    __threadfence();
    __builtin_amdgcn_wave_barrier();
    // ---- on into the next part until the trails meet
    const bool has_next = has && (uint32_t)lane + 1 < n_parts;
    const uint32_t x_node = node, x_F = F;                                     // where this lane left its part
    bool fine = true;                                                          // (lane 0 .. n_parts - 2: the next part's trail was met)
    int64_t lfs_ext = -1;
    if (has_next) {
        if (!live) fine = false;                                               // no provider to walk on with (the chain ended in this part, or no guess)
        else { lfs_ext = walk(min(p_hi + per, seg_l + 1), false, true, &met); fine = met != ~0ull; }
    }
    const bool all_fine = __ballot(has_next && !fine) == 0 && !flag0;
    if (all_fine) {
        // which flags lie on the chain: lane 0's all; of another lane's own walk those from the segment on where the lane before it met its trail;
        // of a walk into the next part all (it stops where it meets)
        const uint64_t met_prev = __shfl_up(met, 1, 64);
        flag = has && ((lane == 0 ? lfs_own >= 0 : lfs_own >= (int64_t)met_prev) || lfs_ext >= 0);
        if (has_next) {                                                        // the true entries of the next part in front of the meeting point: this lane's
            for (uint64_t sg = p_hi; sg <= met; ++sg) seg_entry[slot(sg)] = Succ{SX_NONE, 0u};
            node = x_node; F = x_F;
            // (the same hops again, this time written; it stops in front of the meeting segment, whose entry is the provider met -- with THIS walk's fetch column)
            for (;;) {
                const uint64_t seg = node / SEG;
                if (seg >= met) break;
                seg_entry[slot(seg)] = Succ{node, F};
                const Succ e = ex[node];
                node = e.nxt; F = e.fcol & ~SX_FLAG;
            }
            seg_entry[slot(met)] = Succ{node, F};
        }
    } else {
        // the one-lane walk (the round 2 kernel): every slot of the run first
        for (uint64_t sg = seg_f + lane; sg <= seg_l; sg += 64) seg_entry[slot(sg)] = Succ{SX_NONE, 0u};
        __threadfence();
        __builtin_amdgcn_wave_barrier();
        flag = false; hops = 0;
        if (lane == 0) {
            flag = flag0; node = node0; F = F0;
            for (;;) {
                seg_entry[slot(node / SEG)] = Succ{node, F};
                const Succ e = ex[node];
                flag |= (e.fcol & SX_FLAG) != 0;
                ++hops;
                if (e.nxt >= SX_NONE || flag) { flag |= e.nxt == SX_NONE; break; }
                node = e.nxt; F = e.fcol & ~SX_FLAG;
            }
        }
    }
    const bool any_flag = __ballot(flag) != 0;
    uint32_t h = hops;
    for (int d = 32; d >= 1; d >>= 1) h += (uint32_t)__shfl_xor((int)h, d, 64);
    if (lane == 0) po[blockIdx.x] = ParOut{0u, any_flag ? 1u : 0u, h, 0u};
}

// One wave per segment the chain enters: lane 0 re-walks the segment from its entry (nodes to LDS), then the
// lanes expand dna(provider)[F - pos ...] up to the next provider's fetch column.
struct EmitRun { uint64_t first, endrec, column_base; uint32_t start_pos, end_pos; };
// A chunk longer than DNA_LONG chars (one long read providing tens of kilobases) is not walked here: it goes to
// `longs`, gets one checkpoint per DNA_CK chars from k6_ck_build (a walk that never touches SEQ), and k6_emit_long
// emits the pieces one lane each.  ctr[0] = long nodes, ctr[1] = pieces.
struct LongNode { uint32_t rec, from, count, ck_base; uint64_t out_off; };
struct VEntry { uint32_t node, run; };
constexpr uint32_t VERIFY_LONG = 2048, VMD_CAP = 32768, FAR_CAP = 1u << 16;      // reads longer than this are checked by k6_verify_long; the longest tag it stages; far chunks a pass lists

// inclusive scans over the wave's lanes (a sum, a running maximum)
__device__ __forceinline__ uint32_t wave_incl_u32(uint32_t v, int lane) { for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)v, d, 64); if (lane >= d) v += o; } return v; }
__device__ __forceinline__ int32_t wave_incl_max_i32(int32_t v, int lane) { for (int d = 1; d < 64; d <<= 1) { const int32_t o = __shfl_up(v, d, 64); if (lane >= d) v = max(v, o); } return v; }

// length of the NUL-terminated string at t[0 .. lim) (lim if there is no NUL), by the wave: 512 bytes a step
__device__ uint64_t wave_strnlen(const uint8_t* t, uint64_t lim, int lane) {
    for (uint64_t b0 = 0; b0 < lim; b0 += 512) {
        const uint64_t o = b0 + 8ull * (uint64_t)lane;
        unsigned long long w = ~0ull;
        if (o + 8 <= lim) __builtin_memcpy(&w, t + o, 8);
        else if (o < lim) { for (uint64_t k = 0; k < lim - o; ++k) w = (w & ~(0xFFull << (8 * k))) | ((unsigned long long)t[o + k] << (8 * k)); }
        const unsigned long long z = (w - 0x0101010101010101ull) & ~w & 0x8080808080808080ull;      // the lowest set bit marks the first zero byte
        const unsigned long long m = __ballot(z != 0);
        if (m) {
            const int L = __ffsll((long long)m) - 1;
            const uint32_t idx = (uint32_t)(__ffsll((long long)z) - 1) / 8u;
            return b0 + 8ull * (uint64_t)L + (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)idx, L);
        }
    }
    return lim;
}

// BamRead.opIndex("MD") (find_md above) by a whole wave, every lane the same steps; a string's length by the wave (a long read's tag is
// ten thousand chars: a lane's strnlen8 is 1 250 dependent loads).  1: found (*val, *len); 0: no MD tag / not a string; -1: malformed aux data.
__device__ int find_md_wave(const uint8_t* s, const bgp_record& r, int lane, const uint8_t** val, uint64_t* len) {
    const uint32_t l_name = r.bin_mq_nl & 0xFFu, n_cig = r.flag_nc & 0xFFFFu;
    const uint64_t end = r.rec_off + 4 + ld_u32(s + r.rec_off);
    const uint64_t tags = r.rec_off + 36 + l_name + 4ull * n_cig + (uint64_t)((r.l_seq + 1) / 2) + (uint64_t)r.l_seq;
    if (tags > end) return -1;
    const uint8_t* t = s + tags;
    const uint64_t tl = end - tags;
    uint64_t off = 0;
    while (tl >= 4 && off + 1 < tl) {
        const bool match = t[off] == 'M' && t[off + 1] == 'D';
        off += 2;
        if (off >= tl) return -1;
        const uint8_t ty = t[off++];
        if (match) {
            if (ty != 'Z' && ty != 'H') return 0;
            *val = t + off; *len = wave_strnlen(t + off, tl - off, lane);
            return 1;
        }
        if (ty == 'Z' || ty == 'H') { off += wave_strnlen(t + off, tl - off, lane); if (off >= tl) return -1; ++off; }
        else if (ty == 'B') {
            if (off + 5 > tl) return -1;
            const uint8_t et = t[off];
            const int es = (et == 'c' || et == 'C') ? 1 : (et == 's' || et == 'S') ? 2 : (et == 'i' || et == 'I' || et == 'f') ? 4 : -1;
            if (es < 0) return -1;
            off += 5 + (uint64_t)es * ld_u32(t + off + 1);
        } else {
            const int es = (ty == 'A' || ty == 'c' || ty == 'C') ? 1 : (ty == 's' || ty == 'S') ? 2 : (ty == 'i' || ty == 'I' || ty == 'f') ? 4 : -1;
            if (es < 0) return -1;
            off += (uint64_t)es;
        }
    }
    return 0;
}

// k6_emit's check of a provider -- dna_length(tag) against dna_length_by_cigar -- for a LONG read, by a whole wave.  The tag's tokens are
// found without walking it: a char starts a number when it is a digit behind a non-digit, a deletion when it is '^' (its bases: the
// upper-case run behind it, md/parse.d:85-109), and is a mismatch when it is anything else outside such a run; what the lazy range yields
// is then a sum (md/reconstruct.d:98-182: every match takes its bases, every mismatch one, every deleted base itself) as long as the CIGAR
// has the M / = / X bases the tag asks for.  The check only ever errs to the safe side: whatever it cannot vouch for -- a tag it cannot
// stage, a number of more than nine digits, an empty deletion, more bases asked than there are, a last operation the reference's backward
// parse (parse.d:111-131) would cut differently from the forward one -- FLAGS the run, and the event loop, which parses every tag the
// reference's way, redoes it.
#ifndef BGP_VERIFY_LDS
#define BGP_VERIFY_LDS 0      // 1: the tag staged in LDS first (round 4's kernel: 32 KB a wave, five waves a CU -- the kernel waited for its own loads at that occupancy)
#endif
__global__ __launch_bounds__(64) void k6_verify_long(const uint8_t* s, uint64_t stream_len, const bgp_record* rec, const VEntry* vl, const unsigned int* ctr, uint32_t vcap, ParOut* po) {
#if BGP_VERIFY_LDS
    __shared__ uint8_t smd_lds[VMD_CAP + 16];
#endif
    const uint32_t cnt = min(ctr[2], vcap);
    if (blockIdx.x >= cnt) return;
    const int lane = threadIdx.x;
    const VEntry v = vl[blockIdx.x];
    const bgp_record r = rec[v.node];
    const uint32_t l_name = r.bin_mq_nl & 0xFFu, n_cig = r.flag_nc & 0xFFFFu;
    auto wsum = [](unsigned long long x) { for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64); return x; };
    // the CIGAR's M / = / X bases (what the tag may ask for) and deleted bases
    unsigned long long q = 0, dl = 0;
    {
        const uint8_t* cg = s + r.rec_off + 36 + l_name;
        auto take = [&](uint32_t raw) { if (cig_qc(raw) && cig_rc(raw)) q += raw >> 4; else if ((raw & 15u) == 2u) dl += raw >> 4; };
        uint32_t k = (uint32_t)lane;
        for (; k + 192 < n_cig; k += 256) {      // (four loads in flight a lane)
            const uint32_t r0 = ld_u32(cg + 4ull * k), r1 = ld_u32(cg + 4ull * (k + 64)), r2 = ld_u32(cg + 4ull * (k + 128)), r3 = ld_u32(cg + 4ull * (k + 192));
            take(r0); take(r1); take(r2); take(r3);
        }
        for (; k < n_cig; k += 64) take(ld_u32(cg + 4ull * k));
        q = wsum(q); dl = wsum(dl);
    }
    const unsigned long long by_cigar = q + dl > (unsigned long long)MD_LMASK ? (unsigned long long)MD_LMASK : q + dl;
    // BamRead.opIndex("MD") by the wave (find_md_wave above)
    bool flag = false, found = false;
    const uint8_t* md = nullptr; uint64_t mdn = 0;
    { const int f = find_md_wave(s, r, lane, &md, &mdn); flag = f < 0; found = f > 0; }
    unsigned long long lt = 0;
    if (!flag && found) {
        if (mdn > VMD_CAP) flag = true;
        else {
            const uint32_t n = (uint32_t)mdn;
#if BGP_VERIFY_LDS
            uint8_t* smd = smd_lds;
            const uint64_t md_off = (uint64_t)(md - s);
            for (uint32_t o = 16u * (uint32_t)lane; o < n; o += 1024u) {
                if (md_off + o + 16 <= stream_len) { uint4 w; __builtin_memcpy(&w, md + o, 16); *reinterpret_cast<uint4*>(smd + o) = w; }
                else for (uint32_t k = 0; k < 16u && o + k < n; ++k) smd[o + k] = md[o + k];
            }
            __syncthreads();
#else
            // (the tag where it lies: every char is looked at by its own lane and by a neighbour or two -- consecutive lanes, consecutive bytes, the
            // lines stay in the CU's cache -- and without 32 KB of LDS a wave the CU holds its full set of waves instead of five)
            const uint8_t* smd = md; (void)stream_len;
#endif
            unsigned long long num = 0, mm = 0, del = 0;
#ifndef BGP_VERIFY_ABL
#define BGP_VERIFY_ABL 0      // timing switches (wrong verdicts): 1 without the scan of the tag
#endif
#if !BGP_VERIFY_LDS
            // The tag's units as k6_ck_build finds them, 64 chars a step: whether an upper-case letter is a deleted base is asked of the last
            // char that is NOT one at or in front of it (a running maximum over the wave), not of a walk back to the '^'; a deletion's length
            // is the count of its letters.  (A lane looking left and right of its own char by loops of its own -- the digits of a number, the
            // bases of a deletion, the way back -- was 1.05 of this kernel's 1.17 ms: the wave runs every lane's loop.)
            {
                int32_t lnu = -1;                                              // the last char that is not an upper-case letter, in front of this step
                for (uint32_t b0 = 0; b0 < (BGP_VERIFY_ABL == 1 ? 0u : n); b0 += 64) {
                    const uint32_t c_i = b0 + (uint32_t)lane;
                    const bool in = c_i < n;
                    const uint8_t c = in ? smd[c_i] : (uint8_t)0;
                    const bool up = is_upper_(c);
                    const int32_t nu = wave_incl_max_i32(max(lnu, in && !up ? (int32_t)c_i : -1), lane);
                    if (in) {
                        if (c == '^') { if (c_i + 1 >= n || !is_upper_(smd[c_i + 1])) flag = true; }      // an empty deletion: the range stalls
                        else if (is_digit_(c)) {
                            if (c_i == 0 || !is_digit_(smd[c_i - 1])) {
                                unsigned long long val = 0; uint32_t k = 0;
                                while (c_i + k < n && is_digit_(smd[c_i + k])) { val = val * 10 + (unsigned long long)(smd[c_i + k] - '0'); if (++k > 9) { flag = true; break; } }
                                num += val;
                            }
                        } else if (up && nu >= 0 && smd[nu] == '^') ++del;
                        else ++mm;
                    }
                    lnu = __shfl(nu, 63, 64);
                }
            }
#else
            for (uint32_t i = (uint32_t)lane; i < (BGP_VERIFY_ABL == 1 ? 0u : n); i += 64) {
                const uint8_t c = smd[i];
                if (c == '^') {
                    uint32_t len = 0;
                    while (i + 1 + len < n && is_upper_(smd[i + 1 + len])) ++len;
                    if (len == 0) flag = true;                                 // an empty deletion: the range stalls
                    del += len;
                } else if (is_digit_(c)) {
                    if (i == 0 || !is_digit_(smd[i - 1])) {
                        unsigned long long val = 0; uint32_t k = 0;
                        while (i + k < n && is_digit_(smd[i + k])) { val = val * 10 + (unsigned long long)(smd[i + k] - '0'); if (++k > 9) { flag = true; break; } }
                        num += val;
                    }
                } else {
                    bool in_del = false;
                    if (is_upper_(c)) {
                        uint32_t j = i, steps = 0;
                        while (j > 0 && is_upper_(smd[j - 1])) { --j; if (++steps > 4096u) { flag = true; break; } }
                        in_del = j > 0 && smd[j - 1] == '^';
                    }
                    if (!in_del) ++mm;
                }
            }
#endif
            if (lane == 0 && n) {      // the last operation as the backward parse cuts it (parse.d:111-131): a number, a single mismatch, or '^' with upper-case bases to the end
                const uint8_t last = smd[n - 1];
                if (!is_digit_(last) && !(n == 1 || is_digit_(smd[n - 2]))) {
                    uint32_t j = n - 1; bool upper = true;
                    while (smd[j] != '^') { upper = upper && is_upper_(smd[j]); if (j == 0) { upper = false; break; } --j; }
                    if (!upper || j == n - 1) flag = true;
                }
            }
            num = wsum(num); mm = wsum(mm); del = wsum(del);
            if (num + mm > q) flag = true;                                     // the tag asks for more bases than the CIGAR has: the range stalls
            lt = num + mm + del;
        }
    }
    if (__ballot(flag) || lt != by_cigar) { if (lane == 0) po[v.run].flagged = 1u; }
}

template <bool LONGS> __global__ __launch_bounds__(64) void k6_emit(const uint8_t* s, const bgp_record* rec, const MdRow* rows, const Succ* succ, const Succ* seg_entry, uint64_t slot_lo, uint64_t slot_n,
                                              const EmitRun* er, uint32_t n_runs, ParOut* po, uint8_t* refbase, LongNode* longs, uint32_t longs_cap, unsigned int* ctr, int verify,
                                              VEntry* vlist, uint32_t vcap) {
    // (the segment's providers in stretches of EM_CH: with room for a whole segment's 2048 -- 16 KB a block of one wave -- ten blocks shared
    // a CU and its 40 wave slots; a segment of cfg3 has ~32 providers)
    constexpr uint32_t EM_CH = 192;
    __shared__ uint32_t nd[EM_CH + 1], fc[EM_CH + 1];
    __shared__ uint32_t s_cnt, s_run, s_more;
    if (blockIdx.x >= slot_n) return;
    const Succ ent = seg_entry[slot_lo + blockIdx.x];
    if (ent.nxt >= SX_NONE) return;
    const uint64_t seg = ent.nxt / SEG;
    uint32_t node = ent.nxt, F0 = ent.fcol, hops = 0;                 // (thread 0's: where the chain stands between two stretches)
    if (threadIdx.x == 0) {
        uint32_t lo_r = 0, hi_r = n_runs;
        while (hi_r - lo_r > 1) { const uint32_t m = (lo_r + hi_r) >> 1; if (er[m].first <= ent.nxt) lo_r = m; else hi_r = m; }
        s_run = lo_r;
    }
    __syncthreads();
    const EmitRun R = er[s_run];
    for (;;) {
        if (threadIdx.x == 0) {
            const uint64_t hi = seg * SEG + SEG;
            uint32_t k = 0, more = 0;
            for (;;) {
                nd[k] = node; fc[k] = F0; ++k; ++hops;
                const Succ e = succ[node];
                if (e.nxt >= SX_NONE) { fc[k] = R.end_pos; po[s_run].ce_out = rows[node].end; break; }     // the chain ends here: this run's chunk_end carried to the next reference
                node = e.nxt; F0 = e.fcol & ~SX_FLAG;
                if (node >= hi || hops >= (uint32_t)SEG) { fc[k] = F0; break; }
                if (k >= EM_CH) { fc[k] = F0; more = 1; break; }
            }
            s_cnt = k; s_more = more;
        }
        __syncthreads();
        const uint32_t cnt = s_cnt, more = s_more;
        for (uint32_t k = threadIdx.x; k < cnt; k += 64) {
            const uint32_t F = fc[k];
            uint32_t endc = fc[k + 1];
            if (endc > R.end_pos) endc = R.end_pos;
            const MdRow rw = rows[nd[k]];
            bool check = verify != 0;                             // this provider's real tag against what the rows assumed of it (k6_prep<false>: the length its CIGAR implies)
            if (check && LONGS && vlist && (uint32_t)rec[nd[k]].l_seq > VERIFY_LONG) {      // a long read's tag: a wave of its own (k6_verify_long), not a lane's ten thousand steps
                const uint32_t at = atomicAdd(&ctr[2], 1u);
                if (at < vcap) { vlist[at] = VEntry{nd[k], s_run}; check = false; }
            }
            // a long chunk is emitted in pieces from checkpoints; so is a chunk far into a long read (getting there is the walk of thousands of
            // operations: a wave of its own, k6_ck_build) while the list has room for it (long chunks are bounded by the columns, far ones by FAR_CAP)
            bool walk = F < endc;
            if (walk && (endc - F > DNA_LONG || (LONGS && F - rw.pos > DNA_LONG && atomicAdd(&ctr[3], 1u) < FAR_CAP))) {
                const uint32_t at = atomicAdd(&ctr[0], 1u), pieces = (endc - F + DNA_CK - 1) / DNA_CK;
                const uint32_t base = atomicAdd(&ctr[1], pieces);
                if (at < longs_cap + FAR_CAP) longs[at] = LongNode{nd[k], F - rw.pos, endc - F, base, R.column_base + (F - R.start_pos)};
                walk = false;
            }
            if (!walk && !check) continue;
            // ONE walk of the tag for both: the chunk's chars, then on to the range's end an operation at a time for its length (the tag was
            // found and parsed twice, once by dna_length and once by dna_walk)
            DnaState st; bool err, bad = false, stall = false;
            dna_begin(st, s, rec[nd[k]], &err);
            if (err) { if (check) po[s_run].flagged = 1u; continue; }
            if (walk) {
                const uint32_t from = F - rw.pos;
                if (from) dna_skip(st, from, &stall);
                dna_run(st, s, from, from + (endc - F), refbase + R.column_base + (F - R.start_pos), &bad);
            }
            if (check) {
                dna_skip(st, 0xFFFFFFFFu, &stall);
                if (bad || stall || st.n != (rw.lf & MD_LMASK)) po[s_run].flagged = 1u;
            }
        }
        if (!more) break;
        __syncthreads();                                          // (the stretch's entries are read: thread 0 may write the next)
    }
}

// The checkpoints of ONE long chunk by a lane's walk from the read's first char (the first version of k6_ck_build, and still what a tag
// the wave cannot vouch for gets): ~300 instructions an MD operation, ~8 000 operations a 30 kb read -- 9 ms of cfg5's use_md_tag.
__device__ void ck_build_serial(const uint8_t* s, const bgp_record* rec, const LongNode& L, uint32_t i, uint32_t pieces, DnaCk* ck) {
    DnaState st; bool err;
    dna_begin(st, s, rec[L.rec], &err, true);
    if (err) return;                                              // the pool is zero-filled: every piece of this node stays dead
    for (uint32_t j = 0; j < pieces; ++j) {
        dna_skip(st, L.from + j * DNA_CK);
        if (!st.live) return;                                     // the range ends (or stalls) before this piece
        dna_save(st, s, i, ck[L.ck_base + j]);
    }
}


// A WAVE per long chunk.  The state of dna(read)'s lazy range at char index t (md/reconstruct.d:98-182) is a function of two numbers: the
// MD token that yields char t, and the M / = / X bases the tokens in front of it have taken from the query.  So instead of walking there:
//   * the tag is staged in LDS and cut into UNITS by the wave, 64 chars a step -- a number (its value: that many chars yielded, that many
//     bases asked of the query), a mismatch (1, 1), a deleted base (1, 0); md/parse.d:85-109's tokens are a number, a char outside a
//     deletion, or '^' with the upper-case run behind it -- and prefix sums of (yield, asked) over the units give, for every checkpoint
//     index from + j * 512, the token that holds it;
//   * prefix sums of the CIGAR's M / = / X and query-consuming lengths turn "bases asked so far" into the query cursor (op, in_op, q) that
//     q_settle would have reached (md/reconstruct.d:58-96);
//   * a LANE per checkpoint then builds the parser at its token's first char -- e / back / rem from the ordinary mdf_init (parse.d:23-33: the
//     backward pre-parse touches the last token only), front by md_cache_front at the token, `cur = front; mdf_pop_front` as the walk does
//     between two operations -- and finishes inside the token with the walk's own dna_skip.
// What the wave cannot vouch for goes to the serial walk: a tag longer than the staging buffer, a number of more than nine digits, an
// empty deletion (the range stalls there), a last operation the backward parse cuts differently, a token behind the backward parse's cut.
// Pieces behind a stall stay dead as the walk leaves them: a token in front of which the tag has asked for more bases than the CIGAR has.
constexpr uint32_t CKP_MAX = 256, CK_NONE = 0xFFFFFFFFu;
__device__ unsigned int g_ck_serial;             // long chunks that took the serial walk (read and reset by the host under BGP_VERBOSE: the tests look at it)
__global__ __launch_bounds__(64) void k6_ck_build(const uint8_t* s, uint64_t stream_len, const bgp_record* rec, const LongNode* longs, uint32_t n_long, DnaCk* ck, uint32_t* piece_node, int serial_only) {
#if BGP_VERIFY_LDS
    __shared__ __attribute__((aligned(16))) uint8_t smd_lds[VMD_CAP + 32];
#endif
    __shared__ uint32_t s_tok[CKP_MAX], s_y[CKP_MAX], s_d[CKP_MAX], s_op[CKP_MAX], s_in[CKP_MAX], s_q[CKP_MAX];
    const uint32_t i = blockIdx.x;
    const int lane = threadIdx.x;
    if (i >= n_long) return;
    const LongNode L = longs[i];
    const uint32_t pieces = (L.count + DNA_CK - 1) / DNA_CK;
    for (uint32_t j = (uint32_t)lane; j < pieces; j += 64) piece_node[L.ck_base + j] = i;
    const bgp_record r = rec[L.rec];
    const uint8_t* md = nullptr; uint64_t mdn = 0;
    const int f = find_md_wave(s, r, lane, &md, &mdn);
    if (f <= 0 || mdn == 0) return;                               // malformed aux data, no tag, an empty tag: the range yields nothing, every piece stays dead
    bool flag = serial_only != 0 || mdn > VMD_CAP || pieces > CKP_MAX;
    const uint32_t n = (uint32_t)mdn;
    const uint32_t l_name = r.bin_mq_nl & 0xFFu, n_cig = r.flag_nc & 0xFFFFu;
    const uint8_t* cg = s + r.rec_off + 36 + l_name;
    uint64_t m_total = 0, q_total = 0;
    if (!flag) {
#if BGP_VERIFY_LDS
        uint8_t* smd = smd_lds;
        const uint64_t md_off = (uint64_t)(md - s);
        for (uint32_t o = 16u * (uint32_t)lane; o < n + 16u; o += 1024u) {
            uint4 w = make_uint4(0u, 0u, 0u, 0u);
            if (o < n) {
                if (md_off + o + 16 <= stream_len) __builtin_memcpy(&w, md + o, 16);
                else { uint8_t b[16]; for (uint32_t k = 0; k < 16u; ++k) b[k] = o + k < n ? md[o + k] : (uint8_t)0; __builtin_memcpy(&w, b, 16); }
            }
            *reinterpret_cast<uint4*>(smd + o) = w;
        }
#else
        const uint8_t* smd = md; (void)stream_len;     // (the tag where it lies, as in k6_verify_long: 6 KB of LDS a wave instead of 38)
#endif
        for (uint32_t j = (uint32_t)lane; j < CKP_MAX; j += 64) { s_tok[j] = CK_NONE; s_y[j] = 0u; s_d[j] = CK_NONE; s_op[j] = CK_NONE; s_in[j] = 0u; s_q[j] = 0u; }
        __syncthreads();
        // ---- the tag's units and the token that holds each checkpoint index
        uint32_t Y = 0, D = 0; int32_t lnu = -1;                  // chars yielded, bases asked, the last char that is not an upper-case letter: in front of this step
        for (uint32_t b0 = 0; b0 < n; b0 += 64) {
            const uint32_t c_i = b0 + (uint32_t)lane;
            const bool in = c_i < n;
            const uint8_t c = in ? smd[c_i] : (uint8_t)0;         // (behind the tag: a NUL, not upper-case)
            const bool up = is_upper_(c);
            const int32_t nu = wave_incl_max_i32(max(lnu, in && !up ? (int32_t)c_i : -1), lane);      // the last non-upper char at or in front of this one
            const bool in_del = up && nu >= 0 && smd[nu] == '^';
            uint32_t y = 0, d = 0;
            if (in) {
                if (c == '^') { if (c_i + 1 >= n || !is_upper_(smd[c_i + 1])) flag = true; }       // an empty deletion: the range stalls
                else if (is_digit_(c)) {
                    if (c_i == 0 || !is_digit_(smd[c_i - 1])) {
                        uint32_t val = 0, k = 0;
                        while (c_i + k < n && is_digit_(smd[c_i + k])) { if (++k > 9u) { flag = true; break; } val = val * 10u + (uint32_t)(smd[c_i + k - 1] - '0'); }
                        if (val > (1u << 24)) flag = true;        // (no read has that many bases: the walk decides what such a tag means)
                        else y = d = val;
                    }
                } else if (in_del) y = 1;
                else y = d = 1;
            }
            const uint32_t iy = wave_incl_u32(y, lane), id = wave_incl_u32(d, lane);
            const uint32_t yex = Y + iy - y, dex = D + id - d;
            if (y && yex + y > L.from) {
                const uint32_t lo = yex <= L.from ? 0u : (yex - L.from + DNA_CK - 1) / DNA_CK;
                const uint32_t hi = min(pieces, (yex + y - L.from + DNA_CK - 1) / DNA_CK);      // from + j * 512 < yex + y
                for (uint32_t j = lo; j < hi; ++j) {
                    s_tok[j] = in_del ? (uint32_t)nu : c_i;
                    s_y[j] = in_del ? yex - (c_i - (uint32_t)nu - 1u) : yex;
                    s_d[j] = dex;
                }
            }
            Y += (uint32_t)__shfl((int)iy, 63, 64); D += (uint32_t)__shfl((int)id, 63, 64);
            lnu = __shfl(nu, 63, 64);
            if (Y > (1u << 30) || D > (1u << 30)) flag = true;
            if (__ballot(flag)) break;
        }
        if (lane == 0 && n) {      // the last operation as the backward parse cuts it (parse.d:111-131): a number, a single mismatch, or '^' with upper-case bases to the end
            const uint8_t last = smd[n - 1];
            if (!is_digit_(last) && !(n == 1 || is_digit_(smd[n - 2]))) {
                uint32_t j = n - 1; bool upper = true;
                while (smd[j] != '^') { upper = upper && is_upper_(smd[j]); if (j == 0) { upper = false; break; } --j; }
                if (!upper || j == n - 1) flag = true;
            }
        }
        flag = __ballot(flag) != 0;
        __syncthreads();
        // ---- the CIGAR: where the query cursor stands once `asked` M / = / X bases have been taken
        if (!flag) {
            for (uint32_t k0 = 0; k0 < n_cig; k0 += 64) {
                const uint32_t k = k0 + (uint32_t)lane;
                const uint32_t raw = k < n_cig ? ld_u32(cg + 4ull * k) : 0u;
                const uint32_t len = raw >> 4;
                const uint32_t ml = cig_qc(raw) && cig_rc(raw) ? len : 0u, ql = cig_qc(raw) ? len : 0u;
                // (an op is below 2^28: 64 of them fit a 32-bit scan)
                const uint32_t im = wave_incl_u32(ml, lane), iq = wave_incl_u32(ql, lane);
                const uint64_t mex = m_total + im - ml, qex = q_total + iq - ql;
                if (ml) {
                    uint32_t lo = 0, hi = pieces;                 // the first checkpoint whose `asked` is not below mex (s_d is non-decreasing; the dead ones, at the end, hold the maximum)
                    while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if ((uint64_t)s_d[m] < mex) lo = m + 1; else hi = m; }
                    for (uint32_t j = lo; j < pieces && (uint64_t)s_d[j] < mex + ml; ++j) { s_op[j] = k; s_in[j] = (uint32_t)((uint64_t)s_d[j] - mex); s_q[j] = (uint32_t)(qex + ((uint64_t)s_d[j] - mex)); }
                }
                m_total += (uint32_t)__shfl((int)im, 63, 64); q_total += (uint32_t)__shfl((int)iq, 63, 64);
            }
            __syncthreads();
        }
    }
    // ---- a lane a checkpoint
    MdRes m0;
    if (!flag) {
        mdf_init(m0, md, n, true);
        if (md_empty(m0)) return;                                 // (only zero matches: nothing is yielded)
        bool behind = false;
        for (uint32_t j = (uint32_t)lane; j < pieces; j += 64) behind = behind || (s_tok[j] != CK_NONE && s_tok[j] > m0.e);
        flag = __ballot(behind) != 0;                             // a token behind the backward parse's cut: not a tag this code reasons about
    }
    if (flag) { if (lane == 0) { atomicAdd(&g_ck_serial, 1u); ck_build_serial(s, rec, L, i, pieces, ck); } return; }
    for (uint32_t j = (uint32_t)lane; j < pieces; j += 64) {
        const uint32_t T = s_tok[j];
        if (T == CK_NONE) continue;                               // the range ends before this piece
        const uint64_t asked = s_d[j];
        if (asked > m_total) continue;                            // the range has stalled in front of this token (a match or a mismatch met an empty query)
        DnaState st;
        st.m = m0; st.del_i = 0; st.n = s_y[j]; st.live = true;
        QIter& q = st.q; q.cg = cg; q.sq = cg + 4ull * n_cig; q.n_cig = n_cig;
        if (asked == m_total) { q.op = n_cig; q.in_op = 0; q.q = (uint32_t)q_total; q.empty = true; }
        else { q.op = s_op[j]; q.in_op = s_in[j]; q.q = s_q[j]; q.empty = false; }
        q_load(q);
        if (T == m0.e) {                                          // the tag's LAST token: the one the backward pre-parse took (parse.d:57-69: the front hands it over, then the range is empty)
            st.m.b = m0.e; st.m.front = m0.back; st.m.rem = 0; st.cur = m0.back;
        } else {
            st.m.b = T;
            if (!md_cache_front(st.m)) continue;
            st.cur = st.m.front; mdf_pop_front(st.m);
        }
        dna_skip(st, L.from + j * DNA_CK);
        if (!st.live) continue;
        dna_save(st, s, i, ck[L.ck_base + j]);
    }
}

__global__ __launch_bounds__(64) void k6_emit_long(const uint8_t* s, const bgp_record* rec, const LongNode* longs, const DnaCk* ck, const uint32_t* piece_node, uint32_t n_pieces, uint8_t* refbase) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pieces) return;
    const LongNode L = longs[piece_node[p]];
    const uint32_t j = p - L.ck_base;
    const DnaCk c = ck[p];
    if (!c.pad) return;                                           // the range ended before this piece
    DnaState st; bool bad;
    dna_restore(st, s, rec[L.rec], c);
    const uint32_t lo = L.from + j * DNA_CK, hi = min(L.from + L.count, lo + DNA_CK);
    dna_run(st, s, lo, hi, refbase + L.out_off + (uint64_t)j * DNA_CK, &bad);
}

// ---------------------------------------------------------------------------------------------- isSNP
// simple.isSNP (bio/std/hts/snpcallers/simple.d:15-41) from the column summary + reference_base; see biodgpu.h for
// the two cases that are reported instead of guessed.
__global__ __launch_bounds__(256) void k7_snp(const bgp_column* cols, const uint8_t* refbase, uint64_t np, long long min_cov, int min_wit, uint8_t* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= np) return;
    const bgp_column c = cols[i];
    const uint8_t rb = refbase[i];
    uint8_t res = BGP_SNP_NO;
    const unsigned long long cov = c.coverage;
    const bool enough = !(cov < (unsigned long long)min_cov);          // size_t < int: the int is converted (simple.d:17)
    if (enough && (c.a | c.c | c.g | c.t | c.other)) {
        const uint32_t m4 = max(max(c.a, c.c), max(c.g, c.t));
        const long long wit = min_wit;
        if (c.other != 0 && c.other >= m4) {
            // some ambiguity code may hold the largest count: undecidable here unless every count is below the witness bar
            res = ((long long)c.other < wit && (long long)m4 < wit) ? BGP_SNP_NO : BGP_SNP_NEEDS_ENTRIES;
        } else if (!((long long)m4 < wit)) {
            const int n_top = (c.a == m4) + (c.c == m4) + (c.g == m4) + (c.t == m4);
            const bool ref_top = (rb == 'A' && c.a == m4) || (rb == 'C' && c.c == m4) || (rb == 'G' && c.g == m4) || (rb == 'T' && c.t == m4);
            res = !ref_top ? BGP_SNP_YES : (n_top == 1 ? BGP_SNP_NO : BGP_SNP_TIE);
        }
    }
    out[i] = res;
}

// debug/test hook: dna(read) of one record into a host-visible buffer
__global__ void k6_dna_one(const uint8_t* s, const bgp_record* rec, uint64_t i, uint8_t* out, uint32_t cap, uint32_t* len_bad) {
    bool bad, err, b2, e2;
    const uint32_t L = dna_walk(s, rec[i], 0, 0, nullptr, &bad, &err);        // full length (no early stop)
    if (cap) dna_walk(s, rec[i], 0, cap, out, &b2, &e2);
    bool b3, e3;                                                               // self-check: the SEQ-free length pass agrees with the walk
    const uint32_t L3 = dna_length(s, rec[i], &b3, &e3);
    len_bad[0] = L; len_bad[1] = (bad ? 1u : 0u) | (err ? 2u : 0u) | ((L3 != L || b3 != bad || e3 != err) ? 4u : 0u);
}

}  // namespace

int k7_snp_simple(bgp_ctx* c, uint64_t col_index, uint64_t ref_index, uint64_t np, int min_cov, int min_wit) {
    BGP_TRY(c->d_snp.reserve(np + 16));
    if (!np) return BGP_OK;
    hipLaunchKernelGGL(k7_snp, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, c->stream, c->d_columns.as<bgp_column>() + col_index, c->d_refbase.as<uint8_t>() + ref_index, np,
                       (long long)min_cov, min_wit, c->d_snp.as<uint8_t>());
    BGP_HIP(hipGetLastError());
    return BGP_OK;
}

// rows for every record of the staged stream (needs K3's prefix maxima: call after k34_pileup)
static int k6_rows(bgp_ctx* c, bool full) {
    if (c->md_rows_ready && (c->md_rows_full || !full)) return BGP_OK;
    const uint64_t n = c->n_records;
    BGP_TRY(c->d_md_row.reserve((n + 1) * sizeof(MdRow)));
    BGP_TRY(c->d_md_bmax.reserve((n / 64 + 2) * 4));
    // (records with long CIGARs are listed for k6_prep_long -- only where framing has seen a read beyond the short-read shape at all, and
    // while a 32-bit index names every record; a stream of short reads takes neither the list nor the second launch)
    const bool list_long = !full && c->odd_reads && n < 0xFFFFFFFFull;
    uint32_t* long_list = nullptr;
    const uint32_t long_cap = (uint32_t)std::min<uint64_t>(n, 1ull << 22);
    if (list_long) { BGP_TRY(c->d_md_plong.reserve(((size_t)long_cap + 2) * 4)); long_list = c->d_md_plong.as<uint32_t>(); BGP_HIP(hipMemsetAsync(long_list, 0, 4, c->stream)); }
    if (n) {
        TimedRegion t(c, "k6_prep", n * (32 + sizeof(MdRow)));
        if (full) BGP_TRY(ensure_cig4(c));      // (rows from the tags themselves walk every read's ops)
        const uint8_t* rcls = c->rows_ok ? c->d_rcls.as<uint8_t>() : nullptr;
        if (full) hipLaunchKernelGGL(k6_prep<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->d_stream, c->d_rec.as<bgp_record>(), c->cig4_ok ? c->d_cig4.as<uint4>() : nullptr, rcls,
                           c->d_pmax_end.as<uint64_t>(), c->d_pmax_pos.as<uint64_t>(), n, c->d_md_row.as<MdRow>(), c->d_md_bmax.as<uint32_t>(), (uint32_t*)nullptr, 0u);
        else hipLaunchKernelGGL(k6_prep<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->d_stream, c->d_rec.as<bgp_record>(), c->cig4_ok ? c->d_cig4.as<uint4>() : nullptr, rcls,
                           c->d_pmax_end.as<uint64_t>(), c->d_pmax_pos.as<uint64_t>(), n, c->d_md_row.as<MdRow>(), c->d_md_bmax.as<uint32_t>(), long_list, long_cap);
        if (list_long) hipLaunchKernelGGL(k6_prep_long, dim3(8192), dim3(64), 0, c->stream, c->d_stream, c->d_rec.as<bgp_record>(), long_list, long_cap, c->d_md_row.as<MdRow>());
        BGP_HIP(hipGetLastError());
    }
    c->md_rows_ready = true; c->md_rows_full = full;
    return BGP_OK;
}

int k6_dna_of_record(bgp_ctx* c, uint64_t i, uint8_t* host_out, uint32_t cap, uint32_t* len, uint32_t* flags) {
    if (i >= c->n_records) return BGP_EINVAL;
    BGP_TRY(c->d_entries.reserve((size_t)cap + 64));
    uint8_t* d = c->d_entries.as<uint8_t>();
    uint32_t* lb = (uint32_t*)(d + ((cap + 15) & ~15u));
    BGP_TRY(c->d_entries.reserve((size_t)((cap + 15) & ~15u) + 16));
    d = c->d_entries.as<uint8_t>(); lb = (uint32_t*)(d + ((cap + 15) & ~15u));
    hipLaunchKernelGGL(k6_dna_one, dim3(1), dim3(1), 0, c->stream, c->d_stream, c->d_rec.as<bgp_record>(), i, d, cap, lb);
    uint32_t h[2] = {0, 0};
    BGP_HIP(hipMemcpyAsync(h, lb, 8, hipMemcpyDeviceToHost, c->stream));
    BGP_HIP(hipStreamSynchronize(c->stream));
    const uint32_t ncopy = std::min(h[0], cap);
    if (ncopy) BGP_HIP(hipMemcpy(host_out, d, ncopy, hipMemcpyDeviceToHost));
    *len = h[0]; *flags = h[1];
    return BGP_OK;
}

// reference_base for the runs [run_first, run_end) into the dense array d_refbase (indexed like d_columns).
// `first_record` != ~0 makes the sweep of run_first start at that record (makePileup's first survivor,
// pileup.d:482-489).  An error BioD would raise while sweeping is returned in *md_err together with the
// column whose popFront raises it (*md_err_from, -1 = at construction); the caller decides whether the
// sweep it mirrors gets that far.
//
// Runs are independent except for one scalar: chunk_end_position is NOT reset between references, and
// add() compares the reads of a new reference's first position group against the stale value
// (pileup.d:583-595).  Each run is chained with a guess (the previous run's max end_position, which is what
// the previous chain ends on unless an MD tag is inconsistent), the guesses are checked against the
// chains' real final values in order, and wrong ones are re-chained.
int k6_reference_bases(bgp_ctx* c, int skip_zero, uint32_t run_first, uint32_t run_end, uint64_t first_record, int* md_err, int64_t* md_err_from, K6Batch* bt) {
    *md_err = 0; *md_err_from = -1;
    if (bt) { bt->out_have = 0; bt->out_ce = bt->in_ce; }
    if (c->n_records >= 0xFFFFFFF0ull) { c->last_error = "too many records for the MD path"; return BGP_EINVAL; }
    const char* mode = getenv("BGP_MD_CHAIN");                 // "seq": the event loop only; default: composed successors, event loop where premises fail
    const bool use_par = !(mode && !strcmp(mode, "seq")) && !bt;      // (a batch's chain resumes and saves state: the event loop)
    static const bool lazy_md = !(getenv("BGP_MD_LAZY") && atoi(getenv("BGP_MD_LAZY")) == 0);
    BGP_TRY(k6_rows(c, !(use_par && lazy_md)));                 // the event loop reads real lengths; the composed chain assumes them and k6_emit checks its providers
    // The chain runs over whole REFERENCES: PileupRangeUsingMdTag keeps its chunk across a reference's uncovered stretches (it
    // walks them, or jumps and lets "the old chunk simply go on"), so the runs K3 cut at such stretches are put together again here
    // -- `mr` -- and d_refbase is dense over every reference's [first position, largest end) (a byte a position: 2 GB at worst);
    // c->md_run_base says where a run's first position lies in it.
    struct MRun { int32_t ref_id; uint64_t first_record, end_record, start_pos, end_pos, column_base; };
    std::vector<MRun> mr;
    c->md_run_base.assign(c->runs.size(), 0);
    uint64_t n_pos = 0;
    for (uint32_t j = run_first; j < run_end; ++j) {
        const bgp_run& r = c->runs[j];
        if (mr.empty() || mr.back().ref_id != r.ref_id) { mr.push_back(MRun{r.ref_id, r.first_record, r.end_record, r.start_pos, r.end_pos, n_pos}); }
        else { n_pos -= mr.back().end_pos - mr.back().start_pos; mr.back().end_record = r.end_record; mr.back().end_pos = r.end_pos; }
        n_pos += mr.back().end_pos - mr.back().start_pos;
    }
    {
        size_t k = 0;
        for (uint32_t j = run_first; j < run_end; ++j) {
            while (mr[k].ref_id != c->runs[j].ref_id || c->runs[j].first_record >= mr[k].end_record) ++k;
            c->md_run_base[j] = mr[k].column_base + (c->runs[j].start_pos - mr[k].start_pos);
        }
    }
    if (n_pos > (16ull << 30)) { c->last_error = "use_md_tag: the references' spans add up to more than 16 G positions"; return BGP_ENOMEM; }
    BGP_TRY(c->d_refbase.reserve(n_pos + 16));
    if (n_pos) BGP_HIP(hipMemsetAsync(c->d_refbase.p, 'N', n_pos, c->stream));
    const uint32_t nr = (uint32_t)mr.size();
    if (nr == 0) return BGP_OK;
    std::vector<ChainRun> cr(nr);
    std::vector<ChainOut> co(nr);
    std::vector<ParRun> pr(nr);
    std::vector<EmitRun> er(nr);
    for (uint32_t k = 0; k < nr; ++k) {
        const MRun& r = mr[k];
        ChainRun& x = cr[k];
        x.first = (k == 0 && first_record != ~0ull) ? first_record : r.first_record;
        x.endrec = r.end_record; x.iv_base = x.first + 2ull * k; x.run_end = (uint32_t)r.end_pos;
        x.ce_in = k == 0 ? (bt ? bt->in_ce : 0u) : (uint32_t)mr[k - 1].end_pos;
        x.active = 0; x.pad = 0; x.resume = 0; x.rs_P = 0; x.rs_i = 0; x.rs_ci = 0; x.rs_y = 0; x.stop_at = 0;
        if (bt && k == 0 && bt->in_have) { x.resume = 1; x.rs_P = bt->in_P; x.rs_i = bt->in_i; x.rs_ci = bt->in_ci; x.rs_y = bt->in_y; }
        if (bt && k + 1 == nr) x.stop_at = bt->stop_at;
        pr[k] = ParRun{x.first, x.endrec, x.run_end, x.ce_in, 0u, 0u};
        er[k] = EmitRun{x.first, x.endrec, r.column_base, (uint32_t)r.start_pos, (uint32_t)r.end_pos};
    }
    auto al16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t off_out = al16(sizeof(ChainRun) * nr), off_fill = off_out + al16(sizeof(ChainOut) * nr), off_pr = off_fill + al16(sizeof(FillRun) * nr),
                 off_po = off_pr + al16(sizeof(ParRun) * nr), off_er = off_po + al16(sizeof(ParOut) * nr);
    BGP_TRY(c->d_md_state.reserve(off_er + sizeof(EmitRun) * nr));
    BGP_TRY(c->d_md_iv.reserve((c->n_records + 2ull * nr + 2) * sizeof(MdIv)));
    uint8_t* st = c->d_md_state.as<uint8_t>();
    const uint64_t n = c->n_records;
    const uint64_t n_slots = (n + SEG - 1) / SEG + nr + 1;
    const uint32_t longs_cap = (uint32_t)std::min<uint64_t>(n_pos / DNA_LONG + nr + 64, 0x7FFFFFFFull);    // a long chunk covers > DNA_LONG columns of its run
    if (use_par) {
        BGP_TRY(c->d_md_long.reserve(16 + ((size_t)longs_cap + FAR_CAP) * sizeof(LongNode)));
        BGP_TRY(c->d_md_succ.reserve((n + 1) * sizeof(Succ))); BGP_TRY(c->d_md_exit.reserve((n + 1) * sizeof(Succ))); BGP_TRY(c->d_md_entry.reserve(n_slots * sizeof(Succ)));
        BGP_HIP(hipMemcpyAsync(st + off_pr, pr.data(), sizeof(ParRun) * nr, hipMemcpyHostToDevice, c->stream));
        BGP_HIP(hipMemcpyAsync(st + off_er, er.data(), sizeof(EmitRun) * nr, hipMemcpyHostToDevice, c->stream));
        {
            TimedRegion t(c, "k6_succ", n * (sizeof(MdRow) + sizeof(Succ)));
            hipLaunchKernelGGL(k6_succ, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->d_md_row.as<MdRow>(), c->d_md_bmax.as<uint32_t>(), n, (const ParRun*)(st + off_pr), nr, skip_zero, c->d_md_succ.as<Succ>());
        }
        {
            TimedRegion t(c, "k6_exit", n * 2 * sizeof(Succ));
            hipLaunchKernelGGL(k6_exit, dim3((unsigned)((n + SEG - 1) / SEG)), dim3(256), 0, c->stream, c->d_md_succ.as<Succ>(), n, c->d_md_exit.as<Succ>());
        }
        BGP_HIP(hipGetLastError());
    }
    // (re)solve the runs in `ks` under their current carried-in chunk_end
    auto solve = [&](const std::vector<uint32_t>& ks, bool clear) -> int {
        std::vector<uint32_t> seq;
        if (clear) for (uint32_t k : ks) { const MRun& r = mr[k]; if (r.end_pos > r.start_pos) BGP_HIP(hipMemsetAsync(c->d_refbase.as<uint8_t>() + r.column_base, 'N', r.end_pos - r.start_pos, c->stream)); }
        if (use_par) {
            for (uint32_t k = 0; k < nr; ++k) pr[k].active = 0;
            for (uint32_t k : ks) { pr[k].active = 1; pr[k].ce_in = cr[k].ce_in; }
            BGP_HIP(hipMemcpyAsync(st + off_pr, pr.data(), sizeof(ParRun) * nr, hipMemcpyHostToDevice, c->stream));
            for (size_t g0 = 0; g0 < ks.size();) {      // (the slots of runs next to each other in one piece: k6_emit takes them in one launch, and a slot between two runs' segments must say "nothing entered" too)
                size_t g1 = g0 + 1;
                while (g1 < ks.size() && ks[g1] == ks[g1 - 1] + 1) ++g1;
                const uint32_t kf = ks[g0], kl = ks[g1 - 1];
                const uint64_t lo = cr[kf].first / SEG + kf, hi = (cr[kl].endrec ? (cr[kl].endrec - 1) / SEG : 0) + kl + 1;
                BGP_HIP(hipMemsetAsync(c->d_md_entry.as<Succ>() + lo, 0xFF, (hi - lo) * sizeof(Succ), c->stream));
                g0 = g1;
            }
            BGP_HIP(hipMemsetAsync(c->d_md_long.p, 0, 16, c->stream));
            // (a stream of long records: the providers' tags are long too -- k6_emit lists them for k6_verify_long)
            const uint32_t vcap = 1u << 14;
            VEntry* vlist = nullptr;
            if (c->n_records && c->stream_len / c->n_records >= 2048) { BGP_TRY(c->d_md_vlist.reserve((size_t)vcap * sizeof(VEntry))); vlist = c->d_md_vlist.as<VEntry>(); }
            {
                TimedRegion t(c, "k6_top", 0);
                hipLaunchKernelGGL(k6_top, dim3(nr), dim3(64), 0, c->stream, c->d_md_row.as<MdRow>(), c->d_md_exit.as<Succ>(), (const ParRun*)(st + off_pr), (ParOut*)(st + off_po), c->d_md_entry.as<Succ>());
            }
            // (runs next to each other are emitted by ONE launch: their segments' slots are next to each other too, and a block finds its run from
            // its entry -- eight launches of 6 100 one-wave blocks, a launch a contig of cfg3, were eight tails: 2.1 ms)
            for (size_t g0 = 0; g0 < ks.size();) {
                size_t g1 = g0 + 1;
                while (g1 < ks.size() && ks[g1] == ks[g1 - 1] + 1) ++g1;
                const uint32_t kf = ks[g0], kl = ks[g1 - 1];
                const uint64_t lo = cr[kf].first / SEG + kf, hi = (cr[kl].endrec ? (cr[kl].endrec - 1) / SEG : 0) + kl + 1;
                uint64_t span = 0;
                for (size_t g = g0; g < g1; ++g) span += mr[ks[g]].end_pos - mr[ks[g]].start_pos;
                g0 = g1;
                TimedRegion t(c, "k6_emit", span);
                if (vlist)      // (a stream of long records: its instantiation lists long providers for k6_verify_long and sends far chunks the checkpointed way)
                hipLaunchKernelGGL(k6_emit<true>, dim3((unsigned)(hi - lo)), dim3(64), 0, c->stream, c->d_stream, c->d_rec.as<bgp_record>(), c->d_md_row.as<MdRow>(), c->d_md_succ.as<Succ>(),
                                   c->d_md_entry.as<Succ>(), lo, hi - lo, (const EmitRun*)(st + off_er), nr, (ParOut*)(st + off_po), c->d_refbase.as<uint8_t>(),
                                   (LongNode*)(c->d_md_long.as<uint8_t>() + 16), longs_cap, c->d_md_long.as<unsigned int>(), 1, vlist, vcap);
                else
                hipLaunchKernelGGL(k6_emit<false>, dim3((unsigned)(hi - lo)), dim3(64), 0, c->stream, c->d_stream, c->d_rec.as<bgp_record>(), c->d_md_row.as<MdRow>(), c->d_md_succ.as<Succ>(),
                                   c->d_md_entry.as<Succ>(), lo, hi - lo, (const EmitRun*)(st + off_er), nr, (ParOut*)(st + off_po), c->d_refbase.as<uint8_t>(),
                                   (LongNode*)(c->d_md_long.as<uint8_t>() + 16), longs_cap, c->d_md_long.as<unsigned int>(), 1, vlist, vcap);
            }
            if (vlist) {      // the long reads among the providers: their tags checked a wave each (blocks beyond the list's count leave at once)
                TimedRegion t(c, "k6_verify_long", 0);
                hipLaunchKernelGGL(k6_verify_long, dim3(vcap), dim3(64), 0, c->stream, c->d_stream, c->stream_len, c->d_rec.as<bgp_record>(), vlist, c->d_md_long.as<unsigned int>(), vcap, (ParOut*)(st + off_po));
            }
            BGP_HIP(hipGetLastError());
            std::vector<ParOut> po(nr);
            unsigned int ctr[2] = {0, 0};
            BGP_HIP(hipMemcpyAsync(po.data(), st + off_po, sizeof(ParOut) * nr, hipMemcpyDeviceToHost, c->stream));
            BGP_HIP(hipMemcpyAsync(ctr, c->d_md_long.p, sizeof ctr, hipMemcpyDeviceToHost, c->stream));
            BGP_HIP(hipStreamSynchronize(c->stream));
            if (ctr[0]) {                                          // long chunks: checkpoints, then one lane per DNA_CK chars
                if (ctr[0] > longs_cap + FAR_CAP) { c->last_error = "MD: more long chunks than columns / DNA_LONG"; return BGP_EDEVICE; }
                BGP_TRY(c->d_md_ck.reserve((size_t)ctr[1] * (sizeof(DnaCk) + 4) + 64));
                DnaCk* ck = c->d_md_ck.as<DnaCk>(); uint32_t* piece_node = (uint32_t*)(ck + ctr[1]);
                BGP_HIP(hipMemsetAsync(ck, 0, (size_t)ctr[1] * sizeof(DnaCk), c->stream));
                const LongNode* longs = (const LongNode*)(c->d_md_long.as<uint8_t>() + 16);
                const bool ck_verbose = getenv("BGP_VERBOSE") != nullptr;
                if (ck_verbose) { unsigned int zero = 0; BGP_HIP(hipStreamSynchronize(c->stream)); BGP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_ck_serial), &zero, sizeof zero)); }      // (the count is this launch's)
                static const int ck_serial = getenv("BGP_MD_CK_SERIAL") && atoi(getenv("BGP_MD_CK_SERIAL")) != 0;      // (tests: the walk a lane did before the wave's prefix sums)
                { TimedRegion t(c, "k6_ck_build", 0);
                  hipLaunchKernelGGL(k6_ck_build, dim3(ctr[0]), dim3(64), 0, c->stream, c->d_stream, c->stream_len, c->d_rec.as<bgp_record>(), longs, ctr[0], ck, piece_node, ck_serial); }
                { TimedRegion t(c, "k6_emit_long", (uint64_t)ctr[1] * DNA_CK);
                  hipLaunchKernelGGL(k6_emit_long, dim3((ctr[1] + 63) / 64), dim3(64), 0, c->stream, c->d_stream, c->d_rec.as<bgp_record>(), longs, ck, piece_node, ctr[1], c->d_refbase.as<uint8_t>()); }
                BGP_HIP(hipGetLastError());
                if (ck_verbose) {
                    unsigned int ser = 0;
                    BGP_HIP(hipStreamSynchronize(c->stream));
                    BGP_HIP(hipMemcpyFromSymbol(&ser, HIP_SYMBOL(g_ck_serial), sizeof ser));
                    fprintf(stderr, "[biodgpu] MD checkpoints: %u long chunk(s), %u piece(s), %u chunk(s) walked serially\n", ctr[0], ctr[1], ser);
                }
            }
            for (uint32_t k : ks) {
                if (po[k].flagged) seq.push_back(k);
                else { co[k] = ChainOut{po[k].ce_out, 0u, 0, 0u, -1}; }
            }
            if (getenv("BGP_VERBOSE")) fprintf(stderr, "[biodgpu] MD chain: %zu run(s) composed, %zu redone by the event loop\n", ks.size() - seq.size(), seq.size());
            if (getenv("BGP_VERBOSE") && nr <= 16) for (uint32_t k : ks) fprintf(stderr, "[biodgpu]   run %u: first %llu endrec %llu ce_in %u -> ce_out %u flagged %u nodes %u\n", k, (unsigned long long)cr[k].first, (unsigned long long)cr[k].endrec, cr[k].ce_in, po[k].ce_out, po[k].flagged, po[k].n_nodes);
            for (uint32_t k : seq) { const MRun& r = mr[k]; if (r.end_pos > r.start_pos) BGP_HIP(hipMemsetAsync(c->d_refbase.as<uint8_t>() + r.column_base, 'N', r.end_pos - r.start_pos, c->stream)); }
        } else seq = ks;
        if (seq.empty()) return BGP_OK;
        BGP_TRY(k6_rows(c, true));                             // the event loop is the authority: rows from the tags themselves
        for (uint32_t k = 0; k < nr; ++k) cr[k].active = 0;
        for (uint32_t k : seq) cr[k].active = 1;
        BGP_HIP(hipMemcpyAsync(st, cr.data(), sizeof(ChainRun) * nr, hipMemcpyHostToDevice, c->stream));
        {
            TimedRegion t(c, "k6_chain", 0);
            hipLaunchKernelGGL(k6_chain, dim3(nr), dim3(64), 0, c->stream, c->d_md_row.as<MdRow>(), (const ChainRun*)st, (ChainOut*)(st + off_out), c->d_md_iv.as<MdIv>(), skip_zero);
            BGP_HIP(hipGetLastError());
        }
        std::vector<ChainOut> got(nr);
        BGP_HIP(hipMemcpyAsync(got.data(), st + off_out, sizeof(ChainOut) * nr, hipMemcpyDeviceToHost, c->stream));
        BGP_HIP(hipStreamSynchronize(c->stream));
        uint64_t total = 0;
        std::vector<FillRun> fr;
        for (uint32_t k : seq) {
            co[k] = got[k];
            const MRun& r = mr[k];
            fr.push_back(FillRun{cr[k].iv_base, total, r.column_base, (uint32_t)r.start_pos, (uint32_t)r.end_pos, co[k].n_iv, 0u});
            total += co[k].n_iv;
        }
        if (total) {
            BGP_HIP(hipMemcpyAsync(st + off_fill, fr.data(), sizeof(FillRun) * fr.size(), hipMemcpyHostToDevice, c->stream));
            TimedRegion t(c, "k6_fill", n_pos);
            hipLaunchKernelGGL(k6_fill, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, c->d_stream, c->d_rec.as<bgp_record>(), c->d_md_iv.as<MdIv>(),
                               (const FillRun*)(st + off_fill), (uint32_t)fr.size(), total, c->d_refbase.as<uint8_t>());
            BGP_HIP(hipGetLastError());
            BGP_HIP(hipStreamSynchronize(c->stream));          // fr is a host vector: keep it alive until the copy is done
        }
        return BGP_OK;
    };
    std::vector<uint32_t> all(nr);
    for (uint32_t k = 0; k < nr; ++k) all[k] = k;
    BGP_TRY(solve(all, false));
    for (uint32_t round = 0; round <= nr; ++round) {           // check the carried-in guesses in order; a wrong one is re-solved, later runs wait for its real value
        uint32_t redo = nr;
        for (uint32_t k = 1; k < nr; ++k) {
            if (co[k - 1].err) break;                           // the sweep never gets past that run
            if (cr[k].ce_in != co[k - 1].ce_out) { cr[k].ce_in = co[k - 1].ce_out; redo = k; break; }
        }
        if (redo == nr) break;
        BGP_TRY(solve(std::vector<uint32_t>{redo}, true));
    }
    for (uint32_t k = 0; k < nr; ++k) if (co[k].err) {
        *md_err = co[k].err; *md_err_from = co[k].err_from;
        if (bt) { bt->err_last_run = k + 1 == nr ? 1u : 0u; bt->err_dep = co[k].err_dep; }
        break;
    }
    if (bt) {
        const ChainOut& last = co[nr - 1];
        bt->out_have = last.sv_have; bt->out_P = last.sv_P; bt->out_i = last.sv_i; bt->out_ci = last.sv_ci; bt->out_y = last.sv_y;
        bt->out_ce = last.sv_have ? last.sv_CE : cr[nr - 1].ce_in;      // no boundary in front of stop_at: the run begins afresh in the next batch, under the chunk_end_position it began with here
    }
    return BGP_OK;
}
