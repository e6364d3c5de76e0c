// synth_record.h -- deterministic synthetic coordinate-sorted BAM records (generator G1).
//
// One source for host and device: record i of a configuration is a pure function of
// (cfg, i) -- counter-based SplitMix64 draws, no sequential state -- so any shard can be generated
// independently (per rank, per thread, per lane) and the host (g++/hipcc host pass) and the gfx950
// kernel produce byte-identical streams by construction.
//
// Record layout written here is the BAM on-disk record BioD reads (bio/std/hts/bam/read.d:907-1003):
// block_size, 8 x int32 fixed fields, name\0, CIGAR u32[], 4-bit SEQ, QUAL, aux (one MD:Z tag).
// bin follows reg2bin (bio/std/hts/bam/bai/bin.d:82-92).
//
// Profiles (BASELINE.json configs 2, 3, 5):
//   0  M-only        l_seq fixed, CIGAR = <l_seq>M
//   1  mixed         70% M-only | 10% aS bM (a 1..30) | 8% xM yI zM (y 1..5) | 8% xM yD zM (y 1..20)
//                    | 4% xM yN zM (y 50..5000); query length = l_seq; never a leading/trailing I/D/N
//   2  long reads    l_seq uniform [len_min, len_max]; M runs 1..29 separated by I or D of 1..10
// Read bases follow a hashed per-contig reference with ~1% substitutions and ~0.1% N; quals 2..41.
// One deviation from SURVEY 8d, stated in DESIGN.md: positions are a stratified-uniform sample
// (read k of a contig lands in stride k of the contig, with hashed jitter) instead of
// "uniform then sorted", which keeps the stream sorted without a sort and makes records
// independently computable.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SYNTH_HD __host__ __device__ inline
#else
#define SYNTH_HD inline
#endif

struct bgp_synth_cfg {
    uint64_t seed;
    uint64_t n_reads;       // reads in this shard
    uint32_t n_ref;         // contigs in this shard
    uint32_t ref_len;       // length of every contig
    uint32_t len_min;       // l_seq (profiles 0/1: fixed = len_min)
    uint32_t len_max;
    uint32_t profile;       // 0 M-only, 1 mixed, 2 long
    uint32_t first_ref;     // global id of this shard's first contig; record ref_id = first_ref + local index
};

namespace synth {

SYNTH_HD uint64_t mix64(uint64_t z) {          // SplitMix64 finaliser
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
SYNTH_HD uint64_t draw(uint64_t s, uint64_t k) { return mix64(s + k * 0xD1342543DE82EF95ull); }

// 2-bit reference base of global contig c at position p (32 bases per hash)
SYNTH_HD uint32_t ref2(uint64_t seed, uint32_t c, uint32_t p) {
    uint64_t h = mix64(seed ^ 0x5EEDBA5Eull ^ ((uint64_t)c << 36) ^ (uint64_t)(p >> 5));
    return (uint32_t)(h >> ((p & 31) * 2)) & 3u;
}
SYNTH_HD uint32_t nt16_of2(uint32_t b2) { return 1u << b2; }     // A C G T -> 1 2 4 8
SYNTH_HD char char_of2(uint32_t b2) { return "ACGT"[b2]; }

SYNTH_HD uint32_t reg2bin(int32_t beg, int32_t end) {            // bai/bin.d:82-92
    if (end == beg) end = beg + 1;
    --end;
    if (beg >> 14 == end >> 14) return ((1 << 15) - 1) / 7 + (beg >> 14);
    if (beg >> 17 == end >> 17) return ((1 << 12) - 1) / 7 + (beg >> 17);
    if (beg >> 20 == end >> 20) return ((1 << 9) - 1) / 7 + (beg >> 20);
    if (beg >> 23 == end >> 23) return ((1 << 6) - 1) / 7 + (beg >> 23);
    if (beg >> 26 == end >> 26) return ((1 << 3) - 1) / 7 + (beg >> 26);
    return 0;
}

struct Read {                // everything about record i that does not need a base walk
    uint64_t s;              // per-read draw stream
    uint32_t contig_local, contig_global;
    uint64_t ic;             // index within contig
    int32_t pos;
    uint32_t l_seq, mapq, flag;
    uint32_t kind;           // profile 1: 0 M, 1 S+M, 2 M I M, 3 M D M, 4 M N M
    uint32_t a, b, c;        // op lengths for profiles 0/1
};

SYNTH_HD uint64_t reads_per_contig(const bgp_synth_cfg& g) { return (g.n_reads + g.n_ref - 1) / g.n_ref; }

SYNTH_HD uint32_t max_span(const bgp_synth_cfg& g) {
    if (g.profile == 0) return g.len_max;
    if (g.profile == 1) return g.len_max + 5000 + 32;
    return g.len_max + g.len_max / 2 + 64;            // deletions add at most ~ l_seq * 10/ (1+10) ... bounded well below 1.5x
}

SYNTH_HD Read make_read(const bgp_synth_cfg& g, uint64_t i) {
    Read r;
    const uint64_t per = reads_per_contig(g);
    r.contig_local = (uint32_t)(i / per);
    r.ic = i % per;
    r.contig_global = g.first_ref + r.contig_local;
    r.s = mix64(g.seed ^ mix64(((uint64_t)r.contig_global << 40) ^ r.ic));
    const uint64_t L = (uint64_t)g.ref_len - max_span(g);
    const uint64_t q = L / per, rem = L % per;
    const uint64_t base = r.ic * q + (r.ic * rem) / per;
    const uint64_t jit = q ? draw(r.s, 0) % q : 0;
    r.pos = (int32_t)(base + jit);
    const uint64_t d1 = draw(r.s, 1);
    r.mapq = (uint32_t)(d1 % 61);
    r.flag = ((d1 >> 8) & 1) ? 16u : 0u;
    r.l_seq = g.len_min + (g.len_max > g.len_min ? (uint32_t)((d1 >> 16) % (g.len_max - g.len_min + 1)) : 0);
    r.kind = 0; r.a = r.l_seq; r.b = 0; r.c = 0;
    if (g.profile == 1) {
        const uint64_t d2 = draw(r.s, 2);
        const uint32_t pct = (uint32_t)(d2 % 100);
        const uint32_t u = (uint32_t)(d2 >> 8), v = (uint32_t)(d2 >> 40);
        if (pct < 70) { r.kind = 0; }
        else if (pct < 80) { r.kind = 1; r.a = 1 + u % 30; r.b = r.l_seq - r.a; }
        else if (pct < 88) { r.kind = 2; r.b = 1 + u % 5; r.a = 1 + v % (r.l_seq - r.b - 1); r.c = r.l_seq - r.a - r.b; }
        else if (pct < 96) { r.kind = 3; r.b = 1 + u % 20; r.a = 1 + v % (r.l_seq - 1); r.c = r.l_seq - r.a; }
        else { r.kind = 4; r.b = 50 + u % 4951; r.a = 1 + v % (r.l_seq - 1); r.c = r.l_seq - r.a; }
    }
    return r;
}

// CIGAR op enumeration.  kind codes follow BAM: 0 M, 1 I, 2 D, 3 N, 4 S.
struct OpIter {
    const bgp_synth_cfg* g; const Read* r;
    uint32_t k;          // op index
    uint32_t q_left;     // query bases still to emit (profile 2)
    uint32_t op, len;
    bool done;
};
SYNTH_HD void op_next(OpIter& it);
SYNTH_HD OpIter op_begin(const bgp_synth_cfg& g, const Read& r) {
    OpIter it; it.g = &g; it.r = &r; it.k = 0; it.q_left = r.l_seq; it.op = 0; it.len = 0; it.done = false;
    it.k = (uint32_t)-1;
    op_next(it);
    return it;
}
SYNTH_HD void op_next(OpIter& it) {
    const Read& r = *it.r;
    ++it.k;
    if (it.g->profile != 2) {
        const uint32_t nops = r.kind == 0 ? 1 : (r.kind == 1 ? 2 : 3);
        if (it.k >= nops) { it.done = true; return; }
        // kind 1: S M | kind 2: M I M | kind 3: M D M | kind 4: M N M
        it.op = it.k == 0 ? (r.kind == 1 ? 4u : 0u) : (it.k == 1 && r.kind >= 2 ? r.kind - 1 : 0u);
        it.len = it.k == 0 ? r.a : (it.k == 1 ? r.b : r.c);
        return;
    }
    if (it.q_left == 0) { it.done = true; return; }
    const uint64_t d = draw(r.s, 16 + it.k);
    if ((it.k & 1) == 0) {                       // M run
        uint32_t len = 1 + (uint32_t)(d % 29);
        if (len > it.q_left) len = it.q_left;
        // keep room so the read never ends on an indel: if what is left after this run could not
        // hold an insertion plus one more match, swallow it
        if (it.q_left - len < 12) len = it.q_left;
        it.op = 0; it.len = len; it.q_left -= len;
    } else {
        uint32_t len = 1 + (uint32_t)((d >> 8) % 10);
        if (d & 1) { it.op = 1; it.len = len; it.q_left -= len; }   // I (q_left >= 12 here)
        else { it.op = 2; it.len = len; }                            // D
    }
}

// per-query-base draw: 16 bits, 4 bases per hash
SYNTH_HD uint32_t qdraw(const Read& r, uint32_t q) { return (uint32_t)(draw(r.s ^ 0xBA5EBA5Eull, q >> 2) >> ((q & 3) * 16)) & 0xFFFFu; }
SYNTH_HD uint32_t qual_of(const Read& r, uint32_t q) { return 2 + ((uint32_t)((draw(r.s ^ 0x9A1Full, q >> 3) >> ((q & 7) * 8)) & 0xFF) * 40 >> 8); }

// read base (nt16 code) for query index q aligned to reference 2-bit base rb (M op)
SYNTH_HD uint32_t m_base(const Read& r, uint32_t q, uint32_t rb) {
    const uint32_t d = qdraw(r, q), t = d & 1023;
    if (t < 10) return nt16_of2((rb + 1 + ((d >> 10) & 3) % 3) & 3);   // substitution, never the reference base
    if (t == 10) return 15;                                            // N
    return nt16_of2(rb);
}
SYNTH_HD uint32_t free_base(const Read& r, uint32_t q) { return nt16_of2((qdraw(r, q) >> 12) & 3); }   // inserted / clipped

// Sinks: CountSink only counts, ByteSink stores.
struct CountSink { uint64_t n = 0; SYNTH_HD void put(uint8_t) { ++n; } };
struct ByteSink { uint8_t* p; uint64_t n = 0; SYNTH_HD void put(uint8_t b) { p[n++] = b; } };

template <class S> SYNTH_HD void put_u32(S& s, uint32_t v) { s.put(v & 0xFF); s.put((v >> 8) & 0xFF); s.put((v >> 16) & 0xFF); s.put(v >> 24); }
template <class S> SYNTH_HD void put_dec(S& s, uint32_t v) {
    char buf[10]; int n = 0;
    do { buf[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) s.put((uint8_t)buf[--n]);
}

// MD:Z value (without the terminating NUL) for the read, samtools convention.
template <class S> SYNTH_HD void emit_md(const bgp_synth_cfg& g, const Read& r, S& s) {
    uint32_t run = 0, q = 0, rp = (uint32_t)r.pos;
    for (OpIter it = op_begin(g, r); !it.done; op_next(it)) {
        if (it.op == 0) {
            for (uint32_t k = 0; k < it.len; ++k, ++q, ++rp) {
                const uint32_t rb = ref2(g.seed, r.contig_global, rp);
                if (m_base(r, q, rb) == nt16_of2(rb)) ++run;
                else { put_dec(s, run); s.put((uint8_t)char_of2(rb)); run = 0; }
            }
        } else if (it.op == 2) {
            put_dec(s, run); s.put('^'); run = 0;
            for (uint32_t k = 0; k < it.len; ++k, ++rp) s.put((uint8_t)char_of2(ref2(g.seed, r.contig_global, rp)));
        } else if (it.op == 3) rp += it.len;
        else q += it.len;                         // I, S
    }
    put_dec(s, run);
}

// The whole record (with its block_size prefix) into sink `s`; `body_size` must be the value
// record_size() returned minus 4.
template <class S> SYNTH_HD void emit_record(const bgp_synth_cfg& g, uint64_t i, const Read& r, uint32_t body_size, uint32_t n_cigar, uint32_t ref_span, S& s) {
    put_u32(s, body_size);
    put_u32(s, r.contig_global);
    put_u32(s, (uint32_t)r.pos);
    put_u32(s, (reg2bin(r.pos, r.pos + (int32_t)ref_span) << 16) | (r.mapq << 8) | 11u);
    put_u32(s, (r.flag << 16) | n_cigar);
    put_u32(s, r.l_seq);
    put_u32(s, 0xFFFFFFFFu); put_u32(s, 0xFFFFFFFFu); put_u32(s, 0);      // next_refID, next_pos, tlen
    {   // name r%09u\0
        uint64_t id = ((uint64_t)r.contig_global * reads_per_contig(g) + r.ic) % 1000000000ull;
        char d[9];
        for (int k = 8; k >= 0; --k) { d[k] = (char)('0' + id % 10); id /= 10; }
        s.put('r');
        for (int k = 0; k < 9; ++k) s.put((uint8_t)d[k]);
        s.put(0);
    }
    for (OpIter it = op_begin(g, r); !it.done; op_next(it)) put_u32(s, (it.len << 4) | it.op);
    {   // 4-bit sequence
        uint32_t q = 0, rp = (uint32_t)r.pos, hi = 0;
        for (OpIter it = op_begin(g, r); !it.done; op_next(it)) {
            if (it.op == 2 || it.op == 3) { rp += it.len; continue; }
            for (uint32_t k = 0; k < it.len; ++k, ++q) {
                uint32_t code;
                if (it.op == 0) { code = m_base(r, q, ref2(g.seed, r.contig_global, rp)); ++rp; }
                else code = free_base(r, q);
                if (q & 1) s.put((uint8_t)((hi << 4) | code)); else hi = code;
            }
        }
        if (q & 1) s.put((uint8_t)(hi << 4));
    }
    for (uint32_t q = 0; q < r.l_seq; ++q) s.put((uint8_t)qual_of(r, q));
    s.put('M'); s.put('D'); s.put('Z');
    emit_md(g, r, s);
    s.put(0);
}

struct Shape { uint32_t n_cigar, ref_span, size; };   // size includes the 4-byte prefix

SYNTH_HD Shape record_shape(const bgp_synth_cfg& g, const Read& r) {
    Shape sh; sh.n_cigar = 0; sh.ref_span = 0;
    for (OpIter it = op_begin(g, r); !it.done; op_next(it)) { ++sh.n_cigar; if (it.op == 0 || it.op == 2 || it.op == 3) sh.ref_span += it.len; }
    CountSink c;
    emit_md(g, r, c);
    sh.size = 4 + 32 + 11 + 4 * sh.n_cigar + (r.l_seq + 1) / 2 + r.l_seq + 3 + (uint32_t)c.n + 1;
    return sh;
}

}  // namespace synth
