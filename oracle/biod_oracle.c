/*
 * biod_oracle.c -- CPU restatement of BioD's decode -> CIGAR walk -> pileup path (see biod_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: the parity oracle.  Never linked into the product library.
 * Parity pinned by tests/test_oracle_kats.py against the reference's own KATs and fixtures.
 * All citations are file:line in biod/BioD v0.2.4.  A little-endian host is assumed (as BioD's fast
 * path does, read.d:497-503).
 */
#include "biod_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ utils */

static inline uint32_t rd_u32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline int32_t rd_i32(const uint8_t* p) { int32_t v; memcpy(&v, p, 4); return v; }

#define GROW(ptr, cap, need, type)                                         \
    do {                                                                   \
        if ((need) > (cap)) {                                              \
            size_t ncap_ = (cap) ? (cap) * 2 : 1024;                       \
            while (ncap_ < (size_t)(need)) ncap_ *= 2;                     \
            type* np_ = (type*)realloc((ptr), ncap_ * sizeof(type));       \
            if (!np_) return ORC_ENOMEM;                                   \
            (ptr) = np_; (cap) = ncap_;                                    \
        }                                                                  \
    } while (0)

/* ------------------------------------------------------------------------------------------ cigar.d */

/* cigar.d:102-105 */
uint32_t orc_cigar_length(uint32_t raw) { return raw >> 4; }
/* cigar.d:109-111 */
char orc_cigar_type(uint32_t raw) { return "MIDNSHP=X????????"[raw & 0xF]; }
/* cigar.d:113-116: per op two bits, bit0 = query consuming, bit1 = reference consuming
 *                                  X  =  P  H  S  N  D  I  M                                */
static const uint32_t CIGAR_TYPE = 0x3C1A7u; /* 0b11_11_00_00_01_10_10_01_11 */
/* cigar.d:119-121 */
int orc_cigar_is_query_consuming(uint32_t raw) { return ((CIGAR_TYPE >> ((raw & 0xF) * 2)) & 1) != 0; }
/* cigar.d:124-126 */
int orc_cigar_is_reference_consuming(uint32_t raw) { return ((CIGAR_TYPE >> ((raw & 0xF) * 2)) & 2) != 0; }
/* cigar.d:129-131 */
int orc_cigar_is_match_or_mismatch(uint32_t raw) { return ((CIGAR_TYPE >> ((raw & 0xF) * 2)) & 3) == 3; }
/* cigar.d:134-136 */
int orc_cigar_is_clipping(uint32_t raw) { return ((raw & 0xF) >> 1) == 2; }

/* ------------------------------------------------------------------------------------------ base.d */

/* bio/core/base.d:85 */
char orc_base_code2char(uint32_t code) { return "=ACMGRSVTWYHKDBN"[code & 15]; }

/* bio/core/base.d:42-62, rebuilt from the IUPAC letters instead of a literal table: digits 0-3 are
 * A,C,G,T; '=' is 0; letters (either case) map to their nt16 code; everything else is N (15). */
uint32_t orc_base_char2code(uint8_t c) {
    static const char letters[] = "=ACMGRSVTWYHKDBN";
    if (c >= '0' && c <= '3') return 1u << (c - '0');
    if (c == '=') return 0;
    if (c >= 'a' && c <= 'z') c = (uint8_t)(c - 'a' + 'A');
    for (uint32_t k = 1; k < 16; ++k)
        if ((uint8_t)letters[k] == c) return k;
    return 15;
}

/* read.d:364-383 (use_first_4_bits == true, _index == 0) */
char orc_seq_base(const uint8_t* packed, uint32_t i) {
    uint8_t b = packed[i >> 1];
    return orc_base_code2char((i & 1) ? (b & 0xF) : (b >> 4));
}

/* ------------------------------------------------------------------------------------------ read.d */

/* readrange.d:118-173.  eof -> stop (:123-126); fewer than 4 bytes left -> readBlock returns 0 ->
 * stop (:139-149); body shorter than block_size -> readExact throws (:169). */
int64_t orc_frame(const uint8_t* stream, uint64_t len, uint64_t* rec_off, uint64_t cap) {
    uint64_t off = 0, n = 0;
    while (off < len) {
        if (len - off < 4) break;
        int32_t block_size = rd_i32(stream + off);
        if (block_size < 0) return ORC_EINVAL;
        if ((uint64_t)block_size > len - off - 4) return ORC_ETRUNC;
        if (rec_off && n < cap) rec_off[n] = off;
        ++n;
        off += 4 + (uint64_t)block_size;
    }
    return (int64_t)n;
}

/* read.d:907-1003 (fixed fields + offsets), read.d:255-262 (basesCovered), read.d:1380-1383 */
void orc_fields_at(const uint8_t* stream, uint64_t off, orc_fields* f) {
    const uint8_t* c = stream + off + 4; /* _chunk */
    f->block_size = rd_u32(stream + off);
    f->ref_id = rd_i32(c + 0);
    f->pos = rd_i32(c + 4);
    f->bin_mq_nl = rd_u32(c + 8);
    f->flag_nc = rd_u32(c + 12);
    f->l_seq = rd_i32(c + 16);
    f->next_ref_id = rd_i32(c + 20);
    f->next_pos = rd_i32(c + 24);
    f->tlen = rd_i32(c + 28);
    f->bin = f->bin_mq_nl >> 16;
    f->mapq = (f->bin_mq_nl >> 8) & 0xFF;
    f->l_read_name = f->bin_mq_nl & 0xFF;
    f->flag = f->flag_nc >> 16;
    f->n_cigar_op = f->flag_nc & 0xFFFF;
    f->name_off = off + 4 + 32;                                         /* read.d:984-986  */
    f->cigar_off = f->name_off + f->l_read_name;                        /* read.d:988-990  */
    f->seq_off = f->cigar_off + 4ull * f->n_cigar_op;                   /* read.d:992-994  */
    f->qual_off = f->seq_off + (uint64_t)((f->l_seq + 1) / 2);          /* read.d:996-998  */
    f->tags_off = f->qual_off + (uint64_t)f->l_seq;                     /* read.d:1001-1003 */
    int32_t covered = 0;
    if (!(f->flag & 0x4)) {                                             /* read.d:257-259  */
        uint32_t acc = 0;
        for (uint32_t k = 0; k < f->n_cigar_op; ++k) {
            uint32_t op = rd_u32(stream + f->cigar_off + 4ull * k);
            if (orc_cigar_is_reference_consuming(op)) acc += orc_cigar_length(op);
        }
        covered = (int32_t)acc;                                         /* read.d:261      */
    }
    f->bases_covered = covered;
    f->end_pos = (int32_t)((uint32_t)f->pos + (uint32_t)covered);       /* read.d:1382     */
}

/* A record is well formed when the variable-length sections fit inside block_size.  BioD performs no
 * such check (a malformed record makes a later slice throw RangeError); the oracle and the device
 * path both report EINVAL for it at decode time. */
static int record_well_formed(const orc_fields* f) {
    if (f->block_size < 32 || f->l_seq < 0 || f->l_read_name < 1) return 0;
    uint64_t need = 32ull + f->l_read_name + 4ull * f->n_cigar_op + (uint64_t)((f->l_seq + 1) / 2) + (uint64_t)f->l_seq;
    return need <= f->block_size;
}

int orc_decode_soa(const uint8_t* s, const uint64_t* rec_off, uint64_t n, orc_soa* o) {
    uint64_t so = 0, co = 0;
    for (uint64_t i = 0; i < n; ++i) {
        orc_fields f;
        orc_fields_at(s, rec_off[i], &f);
        if (!record_well_formed(&f)) return ORC_EINVAL;
        if (o->ref_id) o->ref_id[i] = f.ref_id;
        if (o->pos) o->pos[i] = f.pos;
        if (o->end_pos) o->end_pos[i] = f.end_pos;
        if (o->bin_mq_nl) o->bin_mq_nl[i] = f.bin_mq_nl;
        if (o->flag_nc) o->flag_nc[i] = f.flag_nc;
        if (o->l_seq) o->l_seq[i] = f.l_seq;
        if (o->seq_off) o->seq_off[i] = so;
        if (o->cigar_off) o->cigar_off[i] = co;
        if (o->seq)
            for (int32_t k = 0; k < f.l_seq; ++k) o->seq[so + k] = (uint8_t)orc_seq_base(s + f.seq_off, (uint32_t)k); /* read.d:438-440 */
        if (o->qual) memcpy(o->qual + so, s + f.qual_off, (size_t)f.l_seq);                                          /* read.d:468-470 */
        if (o->cigar) memcpy(o->cigar + co, s + f.cigar_off, 4ull * f.n_cigar_op);                                   /* read.d:225-228 */
        so += (uint64_t)f.l_seq;
        co += f.n_cigar_op;
    }
    if (o->seq_off) o->seq_off[n] = so;
    if (o->cigar_off) o->cigar_off[n] = co;
    return ORC_OK;
}

/* tagvalue.d:78-85,106-119 */
static int tag_sizeof(char t) {
    switch (t) {
        case 'A': case 'c': case 'C': return 1;
        case 's': case 'S': return 2;
        case 'i': case 'I': case 'f': return 4;
        default: return -1;
    }
}

/* read.d:1070-1087 (opIndex) with skipValue read.d:1219-1230 */
int orc_find_tag(const uint8_t* s, uint64_t rec_off, const char key[2], uint64_t* val_off, char* type) {
    orc_fields f;
    orc_fields_at(s, rec_off, &f);
    uint64_t end = rec_off + 4 + f.block_size;
    if (f.tags_off > end) return ORC_EINVAL;
    const uint8_t* t = s + f.tags_off;
    uint64_t tl = end - f.tags_off;
    if (tl < 4) return 0;                                 /* read.d:1073-1074 */
    uint64_t off = 0;
    while (off + 1 < tl) {                                /* read.d:1077      */
        int match = t[off] == (uint8_t)key[0] && t[off + 1] == (uint8_t)key[1];
        off += 2;
        if (off >= tl) return ORC_EINVAL;
        char ty = (char)t[off++];
        if (match) { *val_off = f.tags_off + off; *type = ty; return 1; }
        if (ty == 'Z' || ty == 'H') {                     /* read.d:1221-1222 */
            while (off < tl && t[off] != 0) ++off;
            if (off >= tl) return ORC_EINVAL;
            ++off;
        } else if (ty == 'B') {                           /* read.d:1223-1226 */
            if (off + 5 > tl) return ORC_EINVAL;
            int es = tag_sizeof((char)t[off]);
            if (es < 0) return ORC_EINVAL;
            uint32_t cnt = rd_u32(t + off + 1);
            off += 5 + (uint64_t)es * cnt;
        } else {                                          /* read.d:1227-1228 */
            int es = tag_sizeof(ty);
            if (es < 0) return ORC_EINVAL;
            off += (uint64_t)es;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ md/parse.d */

static int is_digit_(char c) { return c >= '0' && c <= '9'; }
static int is_upper_(char c) { return c >= 'A' && c <= 'Z'; }

typedef struct {               /* mdOperations.Result, md/parse.d:15-132 */
    const char* md; size_t b, e; /* the remaining slice _md = md[b..e]       */
    orc_md_op front, back;
    int rem;                   /* _rem: 255 = both caches live               */
} mdres;

static uint32_t to_uint_(const char* p, size_t n) { uint64_t v = 0; for (size_t i = 0; i < n; ++i) v = v * 10 + (uint64_t)(p[i] - '0'); return (uint32_t)v; }

static int md_cache_front(mdres* r) {            /* md/parse.d:85-109 */
    if (r->b >= r->e) return 0;
    const char* m = r->md;
    memset(&r->front, 0, sizeof r->front);
    if (m[r->b] == '^') {
        r->b++;
        size_t len = 0;
        while (r->b + len < r->e && is_upper_(m[r->b + len])) ++len;
        r->front.type = 2; r->front.del_off = (uint32_t)r->b; r->front.del_len = (uint32_t)len;
        r->b += len;
    } else if (is_digit_(m[r->b])) {
        size_t len = 0;
        while (r->b + len < r->e && is_digit_(m[r->b + len])) ++len;
        r->front.type = 0; r->front.match = to_uint_(m + r->b, len);
        r->b += len;
    } else {
        r->front.type = 1; r->front.mismatch = orc_base_code2char(orc_base_char2code((uint8_t)m[r->b])); /* Base16(c), md/operation.d:92-97 */
        r->b++;
    }
    return 1;
}

static int md_cache_back(mdres* r) {             /* md/parse.d:111-131 */
    if (r->b >= r->e) return 0;
    const char* m = r->md;
    memset(&r->back, 0, sizeof r->back);
    if (is_digit_(m[r->e - 1])) {
        size_t len = 0;
        while (r->e - len > r->b && is_digit_(m[r->e - 1 - len])) ++len;
        r->back.type = 0; r->back.match = to_uint_(m + r->e - len, len);
        r->e -= len;
    } else if (r->e - r->b == 1 || is_digit_(m[r->e - 2])) {
        r->back.type = 1; r->back.mismatch = orc_base_code2char(orc_base_char2code((uint8_t)m[r->e - 1]));
        r->e -= 1;
    } else {
        size_t len = 0;                            /* countUntil!"a == '^'"(retro(_md)) */
        while (r->e - len > r->b && m[r->e - 1 - len] != '^') ++len;
        r->back.type = 2; r->back.del_off = (uint32_t)(r->e - len); r->back.del_len = (uint32_t)len;
        r->e = (r->e - len > r->b) ? r->e - len - 1 : r->b;
    }
    return 1;
}

static void md_init(mdres* r, const char* md, size_t n) { /* md/parse.d:23-33 */
    r->md = md; r->b = 0; r->e = n; r->rem = 255;
    if (!md_cache_front(r)) r->rem = 0;
    else if (!md_cache_back(r)) { r->back = r->front; r->rem = 1; }
}
static int md_empty(const mdres* r) { return r->rem == 0; }
static void md_pop_front(mdres* r) {               /* md/parse.d:57-69 */
    if (r->b >= r->e) {
        if (r->rem == 255) { r->front = r->back; r->rem = 1; } else r->rem = 0;
    } else if (!md_cache_front(r)) r->rem = 0;
}
static void md_pop_back(mdres* r) {                /* md/parse.d:71-83 */
    if (r->b >= r->e) {
        if (r->rem == 255) { r->back = r->front; r->rem = 1; } else r->rem = 0;
    } else if (!md_cache_back(r)) r->rem = 0;
}
static int md_is_zero_match(const orc_md_op* o) { return o->type == 0 && o->match == 0; } /* md/parse.d:134-137 */

/* filterBidirectional!(not!isZeroMatch)(Result(md)), md/parse.d:139 (Phobos: the constructor trims
 * failing elements from both ends, popFront skips failing elements) */
static void mdf_init(mdres* r, const char* md, size_t n) {
    md_init(r, md, n);
    while (!md_empty(r) && md_is_zero_match(&r->front)) md_pop_front(r);
    while (!md_empty(r) && md_is_zero_match(&r->back)) md_pop_back(r);
}
static void mdf_pop_front(mdres* r) {
    do { md_pop_front(r); } while (!md_empty(r) && md_is_zero_match(&r->front));
}

int64_t orc_md_operations(const char* md, size_t md_len, orc_md_op* out, size_t cap) {
    mdres r;
    mdf_init(&r, md, md_len);
    size_t n = 0;
    while (!md_empty(&r)) {
        if (out && n < cap) out[n] = r.front;
        ++n;
        mdf_pop_front(&r);
    }
    return (int64_t)n;
}

/* ------------------------------------------------------------------------------------------ md/reconstruct.d */

/* A reconstructed reference chunk.  dna(read) is lazy in BioD; the oracle materialises it with the
 * same state machine (md/reconstruct.d:98-182) and remembers whether the lazy range would have hit
 * undefined behaviour past the materialised end (`bad`). */
typedef struct { char* s; size_t n, cap, i; int bad; } dnabuf;

static int dna_push(dnabuf* d, char c) {
    GROW(d->s, d->cap, d->n + 1, char);
    d->s[d->n++] = c;
    return ORC_OK;
}

/* the joined query sequence of reference-consuming query chunks, md/reconstruct.d:58-96,199 */
typedef struct { const uint8_t* s; const orc_fields* f; uint32_t op; uint32_t in_op; uint32_t q; int empty; } qseq_it;

static void qseq_settle(qseq_it* it) {
    while (it->op < it->f->n_cigar_op) {
        uint32_t raw = rd_u32(it->s + it->f->cigar_off + 4ull * it->op);
        if (orc_cigar_is_query_consuming(raw)) {
            if (orc_cigar_is_reference_consuming(raw) && it->in_op < orc_cigar_length(raw)) return;
            if (!orc_cigar_is_reference_consuming(raw)) it->q += orc_cigar_length(raw); /* chunk skipped by the filter */
        }
        it->op++; it->in_op = 0;
    }
    it->empty = 1;
}
static void qseq_init(qseq_it* it, const uint8_t* s, const orc_fields* f) { it->s = s; it->f = f; it->op = 0; it->in_op = 0; it->q = 0; it->empty = 0; qseq_settle(it); }
static char qseq_front(const qseq_it* it) { return orc_seq_base(it->s + it->f->seq_off, it->q); }
static void qseq_pop(qseq_it* it) { it->q++; it->in_op++; qseq_settle(it); }

/* dna(read), md/reconstruct.d:38-214 */
static int dna_read_build(const uint8_t* s, uint64_t rec_off, dnabuf* d) {
    orc_fields f;
    orc_fields_at(s, rec_off, &f);
    d->n = 0; d->i = 0; d->bad = 0;
    uint64_t voff = 0; char ty = 0;
    const char* md = ""; size_t mdn = 0;
    int found = orc_find_tag(s, rec_off, "MD", &voff, &ty);   /* md/reconstruct.d:184-188 */
    if (found < 0) return found;
    if (found == 1 && (ty == 'Z' || ty == 'H')) {
        md = (const char*)(s + voff);
        uint64_t end = rec_off + 4 + f.block_size;
        while (voff + mdn < end && md[mdn] != 0) ++mdn;
    }
    qseq_it q;
    qseq_init(&q, s, &f);
    mdres r;
    mdf_init(&r, md, mdn);
    if (md_empty(&r)) return ORC_OK;                          /* _fetchNextMdOperation: _empty = true */
    orc_md_op cur = r.front; mdf_pop_front(&r);
    uint32_t del_i = 0;
    for (;;) {
        /* front, md/reconstruct.d:118-127; popFront, :153-177 */
        if (cur.type == 1) {
            int rc = dna_push(d, cur.mismatch); if (rc) return rc;
            if (q.empty) { d->bad = 1; return ORC_OK; }       /* _qseqIsSuddenlyEmpty: range stalls */
            qseq_pop(&q);
        } else if (cur.type == 0) {
            if (q.empty) { d->bad = 1; return ORC_OK; }
            int rc = dna_push(d, qseq_front(&q)); if (rc) return rc;
            --cur.match;
            qseq_pop(&q);
            if (cur.match != 0) continue;
        } else {
            if (cur.del_len == 0) { d->bad = 1; return ORC_OK; } /* front of an empty deletion */
            /* deletion chars pass through Base16 as well, core/sequence.d:nucleotideSequence */
            int rc = dna_push(d, orc_base_code2char(orc_base_char2code((uint8_t)md[cur.del_off + del_i]))); if (rc) return rc;
            if (++del_i < cur.del_len) continue;
            del_i = 0;
        }
        if (md_empty(&r)) return ORC_OK;                      /* _fetchNextMdOperation */
        cur = r.front; mdf_pop_front(&r);
    }
}

int64_t orc_dna_read(const uint8_t* s, uint64_t rec_off, char* out, size_t cap) {
    dnabuf d; memset(&d, 0, sizeof d);
    int rc = dna_read_build(s, rec_off, &d);
    if (rc) { free(d.s); return rc; }
    size_t n = d.n < cap ? d.n : cap;
    if (out) memcpy(out, d.s, n);
    int64_t ret = (int64_t)d.n;
    free(d.s);
    return ret;
}

/* ------------------------------------------------------------------------------------------ read sources */

/* An input range of BamRead (the `R reads` of pileup.d) after filter!"a.basesCovered() > 0"
 * (pileup.d:481,510), optionally cut where ref_id changes (takeUntil, pileup.d:494).
 * Either pulls records through readNext-style framing (stream mode) or walks an explicit list of
 * record offsets (list mode, used by the chunked path). */
typedef struct {
    const uint8_t* s; uint64_t len;
    int list_mode;
    uint64_t off, ordinal;                       /* stream mode: next unread record */
    const uint64_t* offs; const uint64_t* ords; uint64_t n, i; /* list mode     */
    int empty, status;
    int limit_ref; int32_t ref_limit;
    orc_fields f; uint64_t cur_off, cur_ord;      /* front                          */
    int have_prev; int32_t prev_ref, prev_pos;    /* sortedness check               */
    uint64_t n_pulled;
} rsrc;

static void rsrc_advance(rsrc* r) {
    for (;;) {
        uint64_t off, ord;
        if (r->list_mode) {
            if (r->i >= r->n) { r->empty = 1; return; }
            off = r->offs[r->i]; ord = r->ords ? r->ords[r->i] : r->i; r->i++;
        } else {
            if (r->off >= r->len || r->len - r->off < 4) { r->empty = 1; return; } /* readrange.d:123-149 */
            int32_t bs = rd_i32(r->s + r->off);
            if (bs < 0) { r->status = ORC_EINVAL; r->empty = 1; return; }
            if ((uint64_t)bs > r->len - r->off - 4) { r->status = ORC_ETRUNC; r->empty = 1; return; }
            off = r->off; ord = r->ordinal++;
            r->off += 4 + (uint64_t)bs;
        }
        r->n_pulled++;
        orc_fields_at(r->s, off, &r->f);
        if (!record_well_formed(&r->f)) { r->status = ORC_EINVAL; r->empty = 1; return; }
        if (r->f.bases_covered > 0) {                        /* pileup.d:481,510 */
            if (r->limit_ref && r->f.ref_id != r->ref_limit) { r->empty = 1; return; } /* pileup.d:494 */
            if (r->have_prev && r->prev_ref == r->f.ref_id && r->f.pos < r->prev_pos) { r->status = ORC_EUNSORTED; r->empty = 1; return; }
            if (r->f.pos < 0) { r->status = ORC_EINVAL; r->empty = 1; return; }
            r->have_prev = 1; r->prev_ref = r->f.ref_id; r->prev_pos = r->f.pos;
            r->cur_off = off; r->cur_ord = ord;
            return;
        }
    }
}

static void rsrc_init_stream(rsrc* r, const uint8_t* s, uint64_t len) {
    memset(r, 0, sizeof *r); r->s = s; r->len = len; rsrc_advance(r);
}
static void rsrc_init_list(rsrc* r, const uint8_t* s, uint64_t len, const uint64_t* offs, const uint64_t* ords, uint64_t n) {
    memset(r, 0, sizeof *r); r->s = s; r->len = len; r->list_mode = 1; r->offs = offs; r->ords = ords; r->n = n; rsrc_advance(r);
}

/* ------------------------------------------------------------------------------------------ pileup.d */

typedef struct {             /* PileupRead!(EagerBamRead), pileup.d:86-230 + read.d:1378-1397 */
    uint64_t ordinal, off;
    int32_t ref_id, pos, end_position, l_seq;
    uint32_t n_cigar;
    uint64_t cigar_off, seq_off, qual_off;
    uint32_t cur_op_index, cur_op, cur_op_offset, query_offset;
} pread;

static inline uint32_t pr_cigar(const uint8_t* s, const pread* p, uint32_t k) { return rd_u32(s + p->cigar_off + 4ull * k); }

/* PileupRead ctor, pileup.d:175-192 */
static int pread_init(const uint8_t* s, const rsrc* src, pread* p) {
    const orc_fields* f = &src->f;
    p->ordinal = src->cur_ord; p->off = src->cur_off;
    p->ref_id = f->ref_id; p->pos = f->pos; p->end_position = f->end_pos; p->l_seq = f->l_seq;
    p->n_cigar = f->n_cigar_op; p->cigar_off = f->cigar_off; p->seq_off = f->seq_off; p->qual_off = f->qual_off;
    p->cur_op = 0; p->cur_op_offset = 0; p->query_offset = 0;
    for (p->cur_op_index = 0; p->cur_op_index < p->n_cigar; ++p->cur_op_index) {
        p->cur_op = pr_cigar(s, p, p->cur_op_index);
        if (orc_cigar_is_reference_consuming(p->cur_op)) {
            if (orc_cigar_type(p->cur_op) != 'N') break;
        } else if (orc_cigar_is_query_consuming(p->cur_op)) {
            p->query_offset += orc_cigar_length(p->cur_op);
        }
    }
    return p->cur_op_index < p->n_cigar ? ORC_OK : ORC_ECIGAR; /* pileup.d:191,224-228 */
}

/* incrementPosition, pileup.d:195-222 */
static int pread_increment(const uint8_t* s, pread* p) {
    ++p->cur_op_offset;
    if (orc_cigar_is_query_consuming(p->cur_op)) ++p->query_offset;
    if (p->cur_op_offset >= orc_cigar_length(p->cur_op)) {
        p->cur_op_offset = 0;
        for (++p->cur_op_index; p->cur_op_index < p->n_cigar; ++p->cur_op_index) {
            p->cur_op = pr_cigar(s, p, p->cur_op_index);
            if (orc_cigar_is_reference_consuming(p->cur_op)) break;
            if (orc_cigar_is_query_consuming(p->cur_op)) p->query_offset += orc_cigar_length(p->cur_op);
        }
        if (p->cur_op_index >= p->n_cigar) return ORC_ECIGAR;
    }
    return ORC_OK;
}

/* current_base, pileup.d:115-122.  *oob is set when BioD would index past the sequence. */
static inline uint8_t pread_base(const uint8_t* s, const pread* p, int* oob) {
    if (orc_cigar_is_query_consuming(p->cur_op) && orc_cigar_is_reference_consuming(p->cur_op)) {
        if (p->query_offset >= (uint32_t)p->l_seq) { *oob = 1; return 0xFF; }
        return (uint8_t)orc_seq_base(s + p->seq_off, p->query_offset);
    }
    return '-';
}
/* current_base_quality, pileup.d:127-134 */
static inline uint8_t pread_qual(const uint8_t* s, const pread* p, int* oob) {
    if (orc_cigar_is_query_consuming(p->cur_op) && orc_cigar_is_reference_consuming(p->cur_op)) {
        if (p->query_offset >= (uint32_t)p->l_seq) { *oob = 1; return 0xFF; }
        return s[p->qual_off + p->query_offset];
    }
    return 255;
}

typedef struct sweep sweep;
typedef int (*column_sink)(void* user, const sweep* sw);

struct sweep {               /* PileupRange (pileup.d:295-425) + PileupRangeUsingMdTag (:522-654) */
    const uint8_t* s;
    rsrc* reads;
    int skip_zero, use_md;
    pread* buf; size_t n, cap;           /* _read_buf                      */
    uint64_t position; int32_t ref_id;   /* _column._position/_ref_id      */
    size_t n_starting_here;
    char reference_base;
    /* PileupRangeUsingMdTag state, pileup.d:528-548 */
    dnabuf chunk; uint32_t chunk_end_position; pread next_provider; int has_next_provider;
    uint64_t prev_coverage; int32_t curr_ref_id;
    int status;
};

/* add(), pileup.d:315-317, and the MD override, pileup.d:563-596 */
static int sw_add(sweep* w) {
    int had_zero_coverage = w->prev_coverage == 0;
    GROW(w->buf, w->cap, w->n + 1, pread);
    int rc = pread_init(w->s, w->reads, &w->buf[w->n]);
    if (rc) return rc;
    w->n++;
    if (!w->use_md) return ORC_OK;
    const pread* r = &w->buf[w->n - 1];
    if (r->ref_id != w->curr_ref_id) {                         /* pileup.d:573-579 */
        w->curr_ref_id = r->ref_id;
        w->has_next_provider = 1; w->next_provider = *r;
        return ORC_OK;
    }
    if ((uint32_t)r->pos > w->chunk_end_position && !had_zero_coverage) return ORC_OK; /* pileup.d:583-585 */
    if ((uint32_t)r->end_position > w->chunk_end_position) {   /* pileup.d:588-595 */
        if (!w->has_next_provider) { w->has_next_provider = 1; w->next_provider = *r; }
        else if (r->end_position > w->next_provider.end_position) w->next_provider = *r;
    }
    return ORC_OK;
}

static int chunk_empty(const dnabuf* d) { return d->i >= d->n && !d->bad; }
static int chunk_front(const dnabuf* d, char* c) { if (d->i >= d->n) return ORC_EMD; *c = d->s[d->i]; return ORC_OK; }
static int chunk_pop(dnabuf* d) { if (d->i >= d->n) return ORC_EMD; d->i++; return ORC_OK; }

/* initNewReference(), pileup.d:399-424, and the MD override, pileup.d:598-611 */
static int sw_init_new_reference(sweep* w) {
    if (w->use_md) w->prev_coverage = 0;
    rsrc* rs = w->reads;
    w->position = (uint64_t)(int64_t)rs->f.pos;
    w->ref_id = rs->f.ref_id;
    size_t n = 1;
    int rc = sw_add(w); if (rc) return rc;
    rsrc_advance(rs);
    while (!rs->empty) {
        if (rs->f.ref_id == w->ref_id && (uint64_t)(int64_t)rs->f.pos == w->position) {
            rc = sw_add(w); if (rc) return rc;
            ++n;
            rsrc_advance(rs);
        } else break;
    }
    if (rs->status) return rs->status;
    w->n_starting_here = n;
    if (w->use_md) {
        if (w->has_next_provider) {
            rc = dna_read_build(w->s, w->next_provider.off, &w->chunk); if (rc) return rc;
            w->chunk_end_position = (uint32_t)w->next_provider.end_position;
            w->has_next_provider = 0;
            rc = chunk_front(&w->chunk, &w->reference_base); if (rc) return rc;
            rc = chunk_pop(&w->chunk); if (rc) return rc;
        } else w->reference_base = 'N';
    }
    return ORC_OK;
}

static int sw_empty(const sweep* w) { return w->reads->empty && w->n == 0; } /* pileup.d:340-342 */

/* PileupRange.popFront, pileup.d:345-397 */
static int sw_pop_front_base(sweep* w) {
    uint64_t pos = ++w->position;
    size_t survived = 0;
    pread* data = w->buf;
    for (size_t i = 0; i < w->n; ++i) {
        if ((uint64_t)(int64_t)data[i].end_position > pos) {
            if (survived < i) data[survived] = data[i];
            ++survived;
        }
    }
    for (size_t i = 0; i < survived; ++i) {
        int rc = pread_increment(w->s, &data[i]);
        if (rc) return rc;
    }
    w->n = survived;
    w->n_starting_here = 0;
    rsrc* rs = w->reads;
    if (!rs->empty) {
        if (rs->f.ref_id != w->ref_id && survived == 0) {
            int rc = sw_init_new_reference(w); if (rc) return rc;
        } else {
            size_t n = 0;
            while (!rs->empty && (uint64_t)(int64_t)rs->f.pos == pos && rs->f.ref_id == w->ref_id) {
                int rc = sw_add(w); if (rc) return rc;
                rsrc_advance(rs);
                ++n;
            }
            if (rs->status) return rs->status;
            w->n_starting_here = n;
            if (survived == 0 && n == 0 && w->skip_zero) {
                if (!rs->empty) { int rc = sw_init_new_reference(w); if (rc) return rc; }
            }
        }
    }
    return rs->status;
}

/* PileupRangeUsingMdTag.popFront, pileup.d:614-653 */
static int sw_pop_front(sweep* w) {
    if (!w->use_md) return sw_pop_front_base(w);
    if (!chunk_empty(&w->chunk)) {
        int rc = chunk_front(&w->chunk, &w->reference_base); if (rc) return rc;
        rc = chunk_pop(&w->chunk); if (rc) return rc;
    } else w->reference_base = 'N';
    w->prev_coverage = w->n;
    int rc = sw_pop_front_base(w); if (rc) return rc;
    if (chunk_empty(&w->chunk) && w->has_next_provider) {
        rc = dna_read_build(w->s, w->next_provider.off, &w->chunk); if (rc) return rc;
        w->chunk_end_position = (uint32_t)w->next_provider.end_position;
        w->has_next_provider = 0;
        uint64_t skip = w->position - (uint64_t)(int64_t)w->next_provider.pos; /* popFrontN, pileup.d:648 */
        for (uint64_t k = 0; k < skip && !chunk_empty(&w->chunk); ++k) { rc = chunk_pop(&w->chunk); if (rc) return rc; }
        rc = chunk_front(&w->chunk, &w->reference_base); if (rc) return rc;
        rc = chunk_pop(&w->chunk); if (rc) return rc;
    }
    return ORC_OK;
}

static int sw_construct(sweep* w, const uint8_t* s, rsrc* reads, int skip_zero, int use_md) { /* pileup.d:323-332,551-553 */
    memset(w, 0, sizeof *w);
    w->s = s; w->reads = reads; w->skip_zero = skip_zero; w->use_md = use_md;
    w->ref_id = -1; w->reference_base = 'N'; w->curr_ref_id = -1;
    if (reads->status) return reads->status;
    if (!reads->empty) return sw_init_new_reference(w);
    return ORC_OK;
}
static void sw_destroy(sweep* w) { free(w->buf); free(w->chunk.s); }

/* pileupColumns (pileup.d:509-519): every column of the sweep goes to the sink. */
static int run_pileup_columns(const uint8_t* s, rsrc* reads, int use_md, int skip_zero, column_sink sink, void* user) {
    sweep w;
    int rc = sw_construct(&w, s, reads, skip_zero, use_md);
    while (!rc && !sw_empty(&w)) {
        rc = sink(user, &w);
        if (!rc) rc = sw_pop_front(&w);
    }
    sw_destroy(&w);
    return rc;
}

/* makePileup -> pileupInstance (pileup.d:480-507,683-694) on an already filtered source. */
static int run_make_pileup(const uint8_t* s, rsrc* reads, int use_md, uint64_t start_from, uint64_t end_at,
                           int skip_zero, column_sink sink, void* user, int32_t* pileup_ref_id) {
    while (!reads->empty) {                                   /* pileup.d:482-489 */
        if ((uint64_t)(int64_t)(reads->f.pos + reads->f.bases_covered) < start_from) rsrc_advance(reads);
        else break;
    }
    if (reads->status) return reads->status;
    int32_t ref_id = -1;
    if (!reads->empty) ref_id = reads->f.ref_id;              /* pileup.d:490-493 */
    if (pileup_ref_id) *pileup_ref_id = ref_id;
    reads->limit_ref = 1; reads->ref_limit = ref_id;          /* pileup.d:494     */
    sweep w;
    int rc = sw_construct(&w, s, reads, skip_zero, use_md);
    while (!rc && !sw_empty(&w) && w.position < start_from) rc = sw_pop_front(&w);   /* pileup.d:497-504 */
    while (!rc && !sw_empty(&w) && !(w.position >= end_at)) {                        /* pileup.d:505     */
        rc = sink(user, &w);
        if (!rc) rc = sw_pop_front(&w);
    }
    sw_destroy(&w);
    return rc;
}

/* ------------------------------------------------------------------------------------------ sinks */

typedef struct {
    orc_pileup_result* r; int want_entries, soft_seq;
    size_t cap_cols, cap_off, cap_ent;
} collect;

static int base_class(uint8_t c) { switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; case '-': return 5; default: return 4; } }

static int collect_sink(void* user, const sweep* w) {
    collect* c = (collect*)user; orc_pileup_result* r = c->r;
    uint64_t k = r->n_columns;
    if (k + 1 > c->cap_cols) {
        size_t nc = c->cap_cols ? c->cap_cols * 2 : 4096;
        int32_t* a = (int32_t*)realloc(r->col_ref, nc * 4); if (!a) return ORC_ENOMEM; r->col_ref = a;
        uint64_t* b = (uint64_t*)realloc(r->col_pos, nc * 8); if (!b) return ORC_ENOMEM; r->col_pos = b;
        uint32_t* d = (uint32_t*)realloc(r->col_cov, nc * 4); if (!d) return ORC_ENOMEM; r->col_cov = d;
        uint32_t* e = (uint32_t*)realloc(r->col_nstart, nc * 4); if (!e) return ORC_ENOMEM; r->col_nstart = e;
        uint8_t* f = (uint8_t*)realloc(r->col_refbase, nc); if (!f) return ORC_ENOMEM; r->col_refbase = f;
        uint32_t* g = (uint32_t*)realloc(r->col_counts, nc * 24); if (!g) return ORC_ENOMEM; r->col_counts = g;
        c->cap_cols = nc;
    }
    r->col_ref[k] = w->ref_id; r->col_pos[k] = w->position; r->col_cov[k] = (uint32_t)w->n;
    r->col_nstart[k] = (uint32_t)w->n_starting_here; r->col_refbase[k] = (uint8_t)w->reference_base;
    uint32_t* cnt = r->col_counts + 6 * k; memset(cnt, 0, 24);
    if (c->want_entries) {
        GROW(r->col_entry_off, c->cap_off, k + 2, uint64_t);
        r->col_entry_off[k] = r->n_entries;
        GROW(r->entries, c->cap_ent, r->n_entries + w->n, orc_entry);
    }
    for (size_t i = 0; i < w->n; ++i) {
        const pread* p = &w->buf[i];
        int oob = 0;
        uint8_t b = pread_base(w->s, p, &oob), q = pread_qual(w->s, p, &oob);
        if (oob) c->soft_seq = 1;
        cnt[base_class(b)]++;
        if (c->want_entries) {
            orc_entry* e = &r->entries[r->n_entries++];
            e->read_ordinal = p->ordinal; e->cigar_op_index = p->cur_op_index; e->cigar_op_offset = p->cur_op_offset;
            e->query_offset = p->query_offset; e->cigar_op = p->cur_op; e->base = b; e->qual = q;
        }
    }
    if (c->want_entries) r->col_entry_off[k + 1] = r->n_entries;
    r->sum_coverage += w->n;
    if (w->n > r->max_coverage) r->max_coverage = (uint32_t)w->n;
    r->n_columns = k + 1;
    return ORC_OK;
}

int orc_pileup(const uint8_t* stream, uint64_t len, int mode, int use_md_tag, uint64_t start_from, uint64_t end_at,
               int skip_zero_coverage, int want_entries, orc_pileup_result* out) {
    memset(out, 0, sizeof *out);
    out->pileup_ref_id = -1;
    collect c; memset(&c, 0, sizeof c); c.r = out; c.want_entries = want_entries;
    rsrc reads; rsrc_init_stream(&reads, stream, len);
    int rc;
    if (mode == ORC_MODE_MAKE_PILEUP)
        rc = run_make_pileup(stream, &reads, use_md_tag, start_from, end_at, skip_zero_coverage, collect_sink, &c, &out->pileup_ref_id);
    else
        rc = run_pileup_columns(stream, &reads, use_md_tag, skip_zero_coverage, collect_sink, &c);
    if (!rc && c.soft_seq) rc = ORC_ESEQ;
    out->status = rc;
    return rc;
}

void orc_pileup_free(orc_pileup_result* r) {
    free(r->col_ref); free(r->col_pos); free(r->col_cov); free(r->col_nstart); free(r->col_refbase);
    free(r->col_counts); free(r->col_entry_off); free(r->entries);
    memset(r, 0, sizeof *r);
}

/* ------------------------------------------------------------------------------------------ dna(reads) */

/* dna(reads), md/reconstruct.d:271-400 */
int64_t orc_dna_reads(const uint8_t* s, uint64_t len, char* out, size_t cap) {
    rsrc reads; rsrc_init_stream(&reads, s, len);   /* filter basesCovered > 0, :398 */
    if (reads.status) return reads.status;
    size_t n = 0;
    if (reads.empty) return 0;
    dnabuf chunk; memset(&chunk, 0, sizeof chunk);
    int rc = dna_read_build(s, reads.cur_off, &chunk);            /* :284-285 */
    if (rc) { free(chunk.s); return rc; }
    uint64_t reference_pos = (uint64_t)(int64_t)reads.f.pos;       /* :286 */
    uint64_t bases_to_skip = 0;
    rsrc_advance(&reads);                                         /* :287 */
    int empty = 0;
    while (!empty) {
        char c;                                                   /* front, :294-299 */
        if (bases_to_skip > 0) c = 'N';
        else { rc = chunk_front(&chunk, &c); if (rc) break; }
        if (out && n < cap) out[n] = c;
        ++n;
        /* popFront, :307-386 */
        reference_pos += 1;
        if (bases_to_skip > 0) { --bases_to_skip; continue; }
        rc = chunk_pop(&chunk); if (rc) break;
        if (chunk_empty(&chunk)) {
            if (reads.empty) { empty = 1; break; }
            if ((uint64_t)(int64_t)reads.f.pos > reference_pos) { /* setSkipMode, :301-305 */
                uint64_t off = reads.cur_off; int32_t p = reads.f.pos;
                rsrc_advance(&reads);
                rc = dna_read_build(s, off, &chunk); if (rc) break;
                bases_to_skip = (uint64_t)(int64_t)p - reference_pos;
                continue;
            }
            uint64_t best_off = reads.cur_off; int32_t best_pos = reads.f.pos;
            uint64_t best_end = (uint64_t)(int64_t)(reads.f.bases_covered + reads.f.pos);
            int found_good = best_end > reference_pos;
            int done = 0;
            for (;;) {
                if (reads.empty) { if (!found_good) { empty = 1; done = 1; } break; }
                if ((uint64_t)(int64_t)reads.f.pos > reference_pos) {
                    if (!found_good) {
                        uint64_t off = reads.cur_off; int32_t p = reads.f.pos;
                        rsrc_advance(&reads);
                        rc = dna_read_build(s, off, &chunk);
                        bases_to_skip = (uint64_t)(int64_t)p - reference_pos;
                        done = 1;
                    }
                    break;
                }
                uint64_t end_pos = (uint64_t)(int64_t)(reads.f.bases_covered + reads.f.pos);
                if (end_pos > reference_pos) {
                    found_good = 1;
                    if (end_pos > best_end) { best_end = end_pos; best_off = reads.cur_off; best_pos = reads.f.pos; }
                }
                rsrc_advance(&reads);
            }
            if (rc) break;
            if (done) continue;
            rc = dna_read_build(s, best_off, &chunk); if (rc) break;
            uint64_t skip = reference_pos - (uint64_t)(int64_t)best_pos;     /* popFrontN, :382 */
            for (uint64_t k = 0; k < skip && !chunk_empty(&chunk); ++k) chunk_pop(&chunk);
        }
    }
    free(chunk.s);
    if (rc) return rc;
    if (reads.status) return reads.status;
    return (int64_t)n;
}

/* ------------------------------------------------------------------------------------------ pileupChunks */

typedef struct { uint64_t first, last; /* record index range [first, last] */ int32_t ref_id; } chunk_t;

typedef struct {
    const uint8_t* s; uint64_t len;
    uint64_t n; uint64_t* rec_off;
    int32_t* ref; int32_t* pos; int32_t* endp; int32_t* covered; int32_t* pmax; /* pmax: prefix max of end_pos over covering reads of the same ref run */
    chunk_t* chunks; size_t n_chunks;
} chunk_plan;

static void plan_free(chunk_plan* p) { free(p->rec_off); free(p->ref); free(p->pos); free(p->endp); free(p->covered); free(p->pmax); free(p->chunks); }

typedef struct { chunk_plan* p; uint64_t lo, hi; int rc; } fields_job;
static void* fields_worker(void* arg) {
    fields_job* j = (fields_job*)arg; chunk_plan* p = j->p;
    for (uint64_t i = j->lo; i < j->hi; ++i) {
        orc_fields f; orc_fields_at(p->s, p->rec_off[i], &f);
        if (!record_well_formed(&f)) { j->rc = ORC_EINVAL; return NULL; }
        p->ref[i] = f.ref_id; p->pos[i] = f.pos; p->endp[i] = f.end_pos; p->covered[i] = f.bases_covered;
    }
    return NULL;
}

/* chunksConsumingLessThan, splitter.d:35-101 (size_in_bytes = 4 + block_size, read.d:609-611).
 * Framing stays one serial pass (it is a pointer chase; BioD's reader thread does the same); the
 * per-record fields are computed on n_threads workers. */
static int plan_build(chunk_plan* p, const uint8_t* s, uint64_t len, uint64_t chunk_bytes, int n_threads) {
    memset(p, 0, sizeof *p); p->s = s; p->len = len;
    {
        size_t cap = 0; uint64_t off = 0, n = 0;
        while (off < len) {                                   /* readrange.d:118-173 */
            if (len - off < 4) break;
            int32_t bs = rd_i32(s + off);
            if (bs < 0) return ORC_EINVAL;
            if ((uint64_t)bs > len - off - 4) return ORC_ETRUNC;
            GROW(p->rec_off, cap, n + 2, uint64_t);
            p->rec_off[n++] = off;
            off += 4 + (uint64_t)bs;
        }
        p->n = n;
    }
    size_t a = (size_t)p->n + 1;
    if (!p->rec_off) p->rec_off = (uint64_t*)malloc(8);
    p->ref = (int32_t*)malloc(a * 4); p->pos = (int32_t*)malloc(a * 4);
    p->endp = (int32_t*)malloc(a * 4); p->covered = (int32_t*)malloc(a * 4); p->pmax = (int32_t*)malloc(a * 4);
    if (!p->rec_off || !p->ref || !p->pos || !p->endp || !p->covered || !p->pmax) return ORC_ENOMEM;
    {
        if (n_threads < 1) n_threads = 1;
        if ((uint64_t)n_threads > p->n / 4096 + 1) n_threads = (int)(p->n / 4096 + 1);
        pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
        fields_job* jobs = (fields_job*)calloc((size_t)n_threads, sizeof(fields_job));
        if (!th || !jobs) { free(th); free(jobs); return ORC_ENOMEM; }
        for (int t = 0; t < n_threads; ++t) {
            jobs[t].p = p; jobs[t].lo = p->n * (uint64_t)t / (uint64_t)n_threads; jobs[t].hi = p->n * (uint64_t)(t + 1) / (uint64_t)n_threads;
            pthread_create(&th[t], NULL, fields_worker, &jobs[t]);
        }
        int rc = ORC_OK;
        for (int t = 0; t < n_threads; ++t) { pthread_join(th[t], NULL); if (jobs[t].rc) rc = jobs[t].rc; }
        free(th); free(jobs);
        if (rc) return rc;
    }
    for (uint64_t i = 0; i < p->n; ++i) {
        int32_t prev = (i > 0 && p->ref[i - 1] == p->ref[i]) ? p->pmax[i - 1] : INT32_MIN;
        int32_t mine = p->covered[i] > 0 ? p->endp[i] : INT32_MIN;
        p->pmax[i] = prev > mine ? prev : mine;
    }
    size_t cap = 0;
    uint64_t i = 0;
    while (i < p->n) {                                        /* getNextChunk, splitter.d:65-91 */
        uint64_t first = i;
        uint64_t total = 4ull + rd_u32(s + p->rec_off[i]);
        ++i;
        while (total <= chunk_bytes && i < p->n) {
            if (p->ref[i] != p->ref[first]) break;
            total += 4ull + rd_u32(s + p->rec_off[i]);
            ++i;
        }
        GROW(p->chunks, cap, p->n_chunks + 1, chunk_t);
        p->chunks[p->n_chunks].first = first; p->chunks[p->n_chunks].last = i - 1; p->chunks[p->n_chunks].ref_id = p->ref[first];
        p->n_chunks++;
    }
    return ORC_OK;
}

/* [beg, end_pos) of PileupChunkRange's element for chunk j: _beg (pileup.d:932-939) and front()'s end_pos (:905-908). */
static void chunk_bounds(const chunk_plan* p, size_t j, int* same_prev_out, size_t* jp_out, uint64_t* beg_out, uint64_t* end_out) {
    const chunk_t* c = &p->chunks[j];
    int same_prev = 0; size_t jp = j;
    while (jp > 0) { --jp; if (p->chunks[jp].ref_id >= 0) { same_prev = p->chunks[jp].ref_id == c->ref_id; break; } }
    uint64_t beg = same_prev ? (uint64_t)(int64_t)p->pos[p->chunks[jp].last] : (uint64_t)(int64_t)p->pos[c->first]; /* pileup.d:932-939 */
    int32_t right_end = INT32_MIN;                                       /* computeRightEnd, pileup.d:870-872 */
    for (uint64_t i = c->first; i <= c->last; ++i) { int32_t e = p->pos[i] + p->covered[i]; if (e > right_end) right_end = e; }
    /* second exactness fix: BioD ends the last chunk of a reference at the right end of THAT chunk
     * (pileup.d:894,929), losing columns covered only by a longer read of an earlier chunk; the
     * oracle runs the last chunk to the maximum end_position of the whole reference run. */
    if (p->pmax[c->last] > right_end) right_end = p->pmax[c->last];
    uint64_t end_pos = (uint64_t)(int64_t)p->pos[c->last];               /* pileup.d:906 */
    size_t jn = j + 1;
    while (jn < p->n_chunks && p->chunks[jn].ref_id < 0) ++jn;           /* popFront skips ref_id < 0 chunks */
    if (jn >= p->n_chunks || p->chunks[jn].ref_id != p->ref[c->last]) end_pos = (uint64_t)(int64_t)right_end; /* pileup.d:907-908 */
    *same_prev_out = same_prev; *jp_out = jp; *beg_out = beg; *end_out = end_pos;
}

/* One PileupChunkRange element (pileup.d:859-987): the pileup of chunk j clipped to
 * [beg, end_pos).  The halo is EXACT (every earlier read of the same reference run with
 * end_position > beg) instead of BioD's 2 x median-read-length heuristic (pileup.d:944-985);
 * see SURVEY.md section 3.3 for why. */
static int chunk_run(const chunk_plan* p, size_t j, column_sink sink, void* user) {
    const chunk_t* c = &p->chunks[j];
    if (c->ref_id < 0) return ORC_OK;                                    /* pileup.d:886,926 */
    int same_prev; size_t jp; uint64_t beg, end_pos;
    chunk_bounds(p, j, &same_prev, &jp, &beg, &end_pos);
    /* exact halo: earlier reads of this reference run with end_position > beg */
    uint64_t run_lo = c->first;
    if (same_prev) {
        uint64_t lo = 0, hi = c->first;   /* first index in the run whose prefix-max end exceeds beg */
        uint64_t r0 = c->first; while (r0 > 0 && p->ref[r0 - 1] == c->ref_id) --r0;
        lo = r0;
        while (lo < hi) { uint64_t m = (lo + hi) / 2; if ((int64_t)p->pmax[m] > (int64_t)beg) hi = m; else lo = m + 1; }
        run_lo = lo;
    }
    uint64_t cnt = 0, capn = (c->last - run_lo + 1);
    uint64_t* offs = (uint64_t*)malloc((size_t)capn * 8); uint64_t* ords = (uint64_t*)malloc((size_t)capn * 8);
    if (!offs || !ords) { free(offs); free(ords); return ORC_ENOMEM; }
    for (uint64_t i = run_lo; i < c->first; ++i)
        if (p->covered[i] > 0 && (int64_t)p->endp[i] > (int64_t)beg) { offs[cnt] = p->rec_off[i]; ords[cnt] = i; ++cnt; }
    for (uint64_t i = c->first; i <= c->last; ++i) { offs[cnt] = p->rec_off[i]; ords[cnt] = i; ++cnt; }
    rsrc reads; rsrc_init_list(&reads, p->s, p->len, offs, ords, cnt);
    int rc = run_make_pileup(p->s, &reads, 0, beg, end_pos, 1, sink, user, NULL);   /* pileup.d:910-912 */
    free(offs); free(ords);
    return rc;
}

typedef struct { uint64_t n_columns, sum_coverage, checksum; uint32_t max_coverage; int soft; } bench_acc;

/* The fixed consumer of BASELINE.md section 2: touch coverage, n_starting_here and every read's
 * current_base / current_base_quality. */
static int bench_sink(void* user, const sweep* w) {
    bench_acc* a = (bench_acc*)user;
    uint64_t h = a->checksum;
    h = h * 1099511628211ull + (uint64_t)w->n * 31u + (uint64_t)w->n_starting_here;
    for (size_t i = 0; i < w->n; ++i) {
        int oob = 0;
        uint8_t b = pread_base(w->s, &w->buf[i], &oob), q = pread_qual(w->s, &w->buf[i], &oob);
        if (oob) a->soft = 1;
        h += ((uint64_t)b << 8) | q;
    }
    a->checksum = h;
    a->n_columns++; a->sum_coverage += w->n;
    if (w->n > a->max_coverage) a->max_coverage = (uint32_t)w->n;
    return ORC_OK;
}

typedef struct {
    const chunk_plan* plan; size_t next; pthread_mutex_t mu;
    int bench_mode; int want_entries;
    bench_acc* accs; orc_pileup_result* results; int* rcs;
} pool_job;

static void* pool_worker(void* arg) {
    pool_job* job = (pool_job*)arg;
    for (;;) {
        pthread_mutex_lock(&job->mu);
        size_t j = job->next++;
        pthread_mutex_unlock(&job->mu);
        if (j >= job->plan->n_chunks) return NULL;
        if (job->bench_mode) {
            job->rcs[j] = chunk_run(job->plan, j, bench_sink, &job->accs[j]);
        } else {
            collect c; memset(&c, 0, sizeof c); c.r = &job->results[j]; c.want_entries = job->want_entries;
            job->rcs[j] = chunk_run(job->plan, j, collect_sink, &c);
            if (!job->rcs[j] && c.soft_seq) job->rcs[j] = ORC_ESEQ;
        }
    }
}

static int pool_run(pool_job* job, int n_threads) {
    pthread_mutex_init(&job->mu, NULL);
    if (n_threads < 1) n_threads = 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
    if (!th) return ORC_ENOMEM;
    for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, pool_worker, job);
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    free(th);
    pthread_mutex_destroy(&job->mu);
    return ORC_OK;
}

int orc_pileup_bench(const uint8_t* stream, uint64_t len, int n_threads, uint64_t chunk_bytes, orc_bench_result* out) {
    memset(out, 0, sizeof *out);
    if (chunk_bytes == 0) chunk_bytes = 16384000;                      /* pileup.d:1011 */
    if (n_threads <= 1) {                                              /* cpu-serial */
        bench_acc a; memset(&a, 0, sizeof a);
        rsrc reads; rsrc_init_stream(&reads, stream, len);
        int rc = run_pileup_columns(stream, &reads, 0, 1, bench_sink, &a);
        out->n_columns = a.n_columns; out->sum_coverage = a.sum_coverage; out->max_coverage = a.max_coverage;
        out->checksum = a.checksum; out->n_reads = reads.n_pulled;
        if (!rc && a.soft) rc = ORC_ESEQ;
        out->status = rc;
        return rc;
    }
    chunk_plan plan;
    int rc = plan_build(&plan, stream, len, chunk_bytes, n_threads);
    if (rc) { plan_free(&plan); out->status = rc; return rc; }
    pool_job job; memset(&job, 0, sizeof job);
    job.plan = &plan; job.bench_mode = 1;
    job.accs = (bench_acc*)calloc(plan.n_chunks ? plan.n_chunks : 1, sizeof(bench_acc));
    job.rcs = (int*)calloc(plan.n_chunks ? plan.n_chunks : 1, sizeof(int));
    if (!job.accs || !job.rcs) { rc = ORC_ENOMEM; goto done; }
    rc = pool_run(&job, n_threads);
    for (size_t j = 0; j < plan.n_chunks && !rc; ++j) {
        if (job.rcs[j] && job.rcs[j] != ORC_ESEQ) rc = job.rcs[j];
        out->n_columns += job.accs[j].n_columns; out->sum_coverage += job.accs[j].sum_coverage;
        out->checksum += job.accs[j].checksum;
        if (job.accs[j].max_coverage > out->max_coverage) out->max_coverage = job.accs[j].max_coverage;
    }
    out->n_reads = plan.n;
done:
    free(job.accs); free(job.rcs); plan_free(&plan);
    out->status = rc;
    return rc;
}

static int append_result(orc_pileup_result* dst, const orc_pileup_result* src, int want_entries) {
    uint64_t n0 = dst->n_columns, n1 = src->n_columns, nt = n0 + n1;
    if (n1 == 0) return ORC_OK;
#define APP(field, type, mul)                                                                        \
    do { type* np = (type*)realloc(dst->field, (size_t)(nt * (mul)) * sizeof(type));                 \
         if (!np) return ORC_ENOMEM; dst->field = np;                                                \
         memcpy(dst->field + n0 * (mul), src->field, (size_t)(n1 * (mul)) * sizeof(type)); } while (0)
    APP(col_ref, int32_t, 1); APP(col_pos, uint64_t, 1); APP(col_cov, uint32_t, 1); APP(col_nstart, uint32_t, 1);
    APP(col_refbase, uint8_t, 1); APP(col_counts, uint32_t, 6);
#undef APP
    if (want_entries) {
        uint64_t* np = (uint64_t*)realloc(dst->col_entry_off, (size_t)(nt + 1) * 8); if (!np) return ORC_ENOMEM; dst->col_entry_off = np;
        for (uint64_t k = 0; k <= n1; ++k) dst->col_entry_off[n0 + k] = dst->n_entries + src->col_entry_off[k];
        orc_entry* ne = (orc_entry*)realloc(dst->entries, (size_t)(dst->n_entries + src->n_entries + 1) * sizeof(orc_entry)); if (!ne) return ORC_ENOMEM; dst->entries = ne;
        memcpy(dst->entries + dst->n_entries, src->entries, (size_t)src->n_entries * sizeof(orc_entry));
        dst->n_entries += src->n_entries;
    }
    dst->n_columns = nt; dst->sum_coverage += src->sum_coverage;
    if (src->max_coverage > dst->max_coverage) dst->max_coverage = src->max_coverage;
    return ORC_OK;
}

int orc_pileup_chunked(const uint8_t* stream, uint64_t len, uint64_t chunk_bytes, int n_threads, int want_entries, orc_pileup_result* out) {
    memset(out, 0, sizeof *out); out->pileup_ref_id = -1;
    if (chunk_bytes == 0) chunk_bytes = 16384000;
    chunk_plan plan;
    int rc = plan_build(&plan, stream, len, chunk_bytes, n_threads);
    if (rc) { plan_free(&plan); out->status = rc; return rc; }
    pool_job job; memset(&job, 0, sizeof job);
    job.plan = &plan; job.want_entries = want_entries;
    job.results = (orc_pileup_result*)calloc(plan.n_chunks ? plan.n_chunks : 1, sizeof(orc_pileup_result));
    job.rcs = (int*)calloc(plan.n_chunks ? plan.n_chunks : 1, sizeof(int));
    if (!job.results || !job.rcs) { rc = ORC_ENOMEM; goto done; }
    rc = pool_run(&job, n_threads);
    int soft = 0;
    for (size_t j = 0; j < plan.n_chunks; ++j) {
        if (!rc) {
            if (job.rcs[j] == ORC_ESEQ) soft = 1; else if (job.rcs[j]) rc = job.rcs[j];
            if (!rc) rc = append_result(out, &job.results[j], want_entries);
        }
        orc_pileup_free(&job.results[j]);
    }
    if (!rc && soft) rc = ORC_ESEQ;
done:
    free(job.results); free(job.rcs); plan_free(&plan);
    out->status = rc;
    return rc;
}

/* The chunks pileupChunks(reads, use_md, chunk_bytes) walks (chunks whose first read has ref_id < 0 are skipped,
 * pileup.d:886,926) with the [beg, end_pos) each one's pileup is clipped to -- start_from = 0, end_at = ulong.max. */
int64_t orc_chunk_plan(const uint8_t* stream, uint64_t len, uint64_t chunk_bytes, orc_chunk* out, uint64_t cap) {
    if (chunk_bytes == 0) chunk_bytes = 16384000;
    chunk_plan plan;
    int rc = plan_build(&plan, stream, len, chunk_bytes, 1);
    if (rc) { plan_free(&plan); return rc; }
    int64_t n = 0;
    int started = 0;
    for (size_t j = 0; j < plan.n_chunks; ++j) {
        if (plan.chunks[j].ref_id < 0) continue;
        int same_prev; size_t jp; uint64_t beg, end_pos;
        chunk_bounds(&plan, j, &same_prev, &jp, &beg, &end_pos);
        if (!started) {
            /* PileupChunkRange's constructor (pileup.d:878-900) walks on while _chunk_right_end > start_from does not hold
             * (:894-896; the int widens to ulong, sign and all): leading chunks of reads that cover nothing at position 0 are
             * never handed out -- and the first chunk kept has no _prev_chunk, whatever lay in front of it (:888). */
            const chunk_t* c = &plan.chunks[j];
            int32_t right_end = INT32_MIN;
            for (uint64_t i = c->first; i <= c->last; ++i) { int32_t e = plan.pos[i] + plan.covered[i]; if (e > right_end) right_end = e; }
            if (plan.pmax[c->last] > right_end) right_end = plan.pmax[c->last];
            if (!((uint64_t)(int64_t)right_end > 0ull)) continue;
            started = 1;
            beg = (uint64_t)(int64_t)plan.pos[c->first];
        }
        if ((uint64_t)n < cap) { out[n].first = plan.chunks[j].first; out[n].last = plan.chunks[j].last; out[n].ref_id = plan.chunks[j].ref_id; out[n].beg = beg; out[n].end = end_pos; }
        ++n;
    }
    plan_free(&plan);
    return n;
}
