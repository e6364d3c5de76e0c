/*
 * biod_oracle.h -- CPU restatement of BioD's BAM-record decode -> CIGAR walk -> pileup path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle: a plain-C restatement of the algorithm in
 * biod/BioD v0.2.4 (D source, cannot be compiled in this image -- no D compiler).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product library
 * (biod_amd/csrc -> libbiodgpu.so) never links, imports or calls anything in this directory.
 *
 * Parity is PINNED: tests/test_oracle_kats.py checks this file against every known-answer test the
 * reference holds for the path (P1-P5, R1-R3, C1, B1, M1-M3) and against the
 * reference's own fixture BAMs (tests/golden/).
 *
 * Every function cites the reference file:line (relative to the BioD repository root) it follows.
 */
#ifndef BIOD_ORACLE_H
#define BIOD_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes (negative = error); same numbering as include/biodgpu.h so tests can compare */
enum {
    ORC_OK = 0,
    ORC_EINVAL = -1,   /* bad argument / negative block_size                                   */
    ORC_ETRUNC = -2,   /* record body runs past the stream (readExact throws, readrange.d:169) */
    ORC_ECIGAR = -3,   /* cursor ran off the CIGAR (assertCigarIndexIsValid, pileup.d:224-228) */
    ORC_ENOMEM = -5,
    ORC_EMD = -7,      /* MD tag missing/short where BioD would read front of an empty range   */
    ORC_EUNSORTED = -8,/* input not coordinate sorted (BioD precondition, pileup.d:25-27)      */
    ORC_ESEQ = -9      /* query offset past l_seq (D RangeError in read.d:364-383)             */
};

/* ---------------------------------------------------------------- record model (read.d, cigar.d) */

/* Record framing, bio/std/hts/bam/readrange.d:118-173.  Fills rec_off[i] = byte offset of the i-th
 * record's 4-byte little-endian block_size prefix.  Returns the number of records (possibly larger
 * than cap; only the first cap offsets are stored) or a negative status. */
int64_t orc_frame(const uint8_t* stream, uint64_t len, uint64_t* rec_off, uint64_t cap);

/* Fixed fields of the record whose length prefix is at `off`, read.d:907-1003. */
typedef struct {
    int32_t ref_id, pos;            /* read.d:907-913                                 */
    uint32_t bin_mq_nl, flag_nc;    /* read.d:915-921                                 */
    int32_t l_seq, next_ref_id, next_pos, tlen; /* read.d:923-937                     */
    uint32_t bin, mapq, l_read_name, flag, n_cigar_op; /* read.d:958-972              */
    uint32_t block_size;
    uint64_t name_off, cigar_off, seq_off, qual_off, tags_off; /* stream offsets, read.d:984-1003 */
    int32_t bases_covered;          /* read.d:255-262                                 */
    int32_t end_pos;                /* EagerBamRead, read.d:1380-1383                 */
} orc_fields;
void orc_fields_at(const uint8_t* stream, uint64_t off, orc_fields* out);

/* CigarOperation, cigar.d:58-136 */
uint32_t orc_cigar_length(uint32_t raw);
char orc_cigar_type(uint32_t raw);
int orc_cigar_is_query_consuming(uint32_t raw);
int orc_cigar_is_reference_consuming(uint32_t raw);
int orc_cigar_is_match_or_mismatch(uint32_t raw);
int orc_cigar_is_clipping(uint32_t raw);

/* Base (nt16) <-> char, bio/core/base.d:42-62,85 */
char orc_base_code2char(uint32_t code);
uint32_t orc_base_char2code(uint8_t c);

/* SequenceResult.opIndex, read.d:364-383 : i-th base (as char) of a packed 4-bit sequence */
char orc_seq_base(const uint8_t* packed, uint32_t i);

/* Bulk SoA decode of n records (the product K1/K2 must reproduce byte for byte).
 * seq_off/cigar_off are exclusive prefix sums with n+1 entries.  Any pointer may be NULL. */
typedef struct {
    int32_t* ref_id; int32_t* pos; int32_t* end_pos;
    uint32_t* bin_mq_nl; uint32_t* flag_nc; int32_t* l_seq;
    uint64_t* seq_off;   /* n+1 */
    uint64_t* cigar_off; /* n+1 */
    uint8_t* seq;        /* chars, sum l_seq   */
    uint8_t* qual;       /* sum l_seq          */
    uint32_t* cigar;     /* sum n_cigar_op     */
} orc_soa;
int orc_decode_soa(const uint8_t* stream, const uint64_t* rec_off, uint64_t n, orc_soa* out);

/* Tag lookup, read.d:1070-1087 + skipValue read.d:1219-1230.  Returns 1 and sets *val_off (stream
 * offset of the first value byte) and *type when found, 0 when absent, negative on malformed aux. */
int orc_find_tag(const uint8_t* stream, uint64_t rec_off, const char key[2], uint64_t* val_off, char* type);

/* ---------------------------------------------------------------- MD path (md/parse.d, md/reconstruct.d) */

typedef struct { uint8_t type; /* 0 match, 1 mismatch, 2 deletion */ uint32_t match; char mismatch;
                 uint32_t del_off, del_len; /* slice of the md string */ } orc_md_op;
/* mdOperations(md), md/parse.d:13-142 (two-ended cache + filterBidirectional of zero matches) */
int64_t orc_md_operations(const char* md, size_t md_len, orc_md_op* out, size_t cap);
/* dna(read), md/reconstruct.d:38-214.  Returns length written (<= cap) or negative status. */
int64_t orc_dna_read(const uint8_t* stream, uint64_t rec_off, char* out, size_t cap);
/* dna(reads), md/reconstruct.d:271-400 over all records of a stream. */
int64_t orc_dna_reads(const uint8_t* stream, uint64_t len, char* out, size_t cap);

/* ---------------------------------------------------------------- pileup (pileup.d) */

typedef struct {             /* one (read, column) entry in canonical form (SURVEY 8 a-out) */
    uint64_t read_ordinal;   /* index of the record in the stream                            */
    uint32_t cigar_op_index; /* PileupRead._cur_op_index, pileup.d:164                       */
    uint32_t cigar_op_offset;/* pileup.d:170                                                 */
    uint32_t query_offset;   /* pileup.d:173                                                 */
    uint32_t cigar_op;       /* raw current op, pileup.d:167                                 */
    uint8_t base;            /* current_base, pileup.d:115-122                               */
    uint8_t qual;            /* current_base_quality, pileup.d:127-134                       */
} orc_entry;

typedef struct {
    int status;
    uint64_t n_columns;
    int32_t* col_ref;        /* PileupColumn.ref_id        pileup.d:257 */
    uint64_t* col_pos;       /* .position                  pileup.d:262 */
    uint32_t* col_cov;       /* .coverage                  pileup.d:252 */
    uint32_t* col_nstart;    /* _n_starting_here           pileup.d:241 */
    uint8_t* col_refbase;    /* .reference_base            pileup.d:245 */
    uint32_t* col_counts;    /* 6 per column: A,C,G,T,other,'-' (from current_base of every read) */
    uint64_t* col_entry_off; /* n_columns+1 (only when want_entries) */
    orc_entry* entries;
    uint64_t n_entries;
    uint64_t sum_coverage;
    uint32_t max_coverage;
    int32_t pileup_ref_id;   /* AbstractPileup.ref_id, pileup.d:455 (makePileup only) */
} orc_pileup_result;

enum { ORC_MODE_PILEUP_COLUMNS = 0, /* pileupColumns, pileup.d:509-519 */
       ORC_MODE_MAKE_PILEUP = 1     /* makePileup,    pileup.d:683-694 */ };

/* Serial sweep over a whole inflated record stream, records pulled through readNext-style framing. */
int orc_pileup(const uint8_t* stream, uint64_t len, int mode, int use_md_tag,
               uint64_t start_from, uint64_t end_at, int skip_zero_coverage,
               int want_entries, orc_pileup_result* out);
void orc_pileup_free(orc_pileup_result* r);

/* CPU-baseline consumers (bench.py cpu_baseline leg): walk every column, touch coverage,
 * n_starting_here and every read's current_base / current_base_quality.
 * n_threads == 1: serial pileupColumns sweep (cpu-serial).
 * n_threads  > 1: pileupChunks decomposition (pileup.d:859-1015, chunk_bytes default 16 384 000)
 *                 with the EXACT halo mapped over a pthread pool (cpu-parallel). */
typedef struct { uint64_t n_columns, sum_coverage, n_reads, checksum; uint32_t max_coverage; int status; } orc_bench_result;
int orc_pileup_bench(const uint8_t* stream, uint64_t len, int n_threads, uint64_t chunk_bytes,
                     orc_bench_result* out);

/* pileupChunks with BioD's own (heuristic) halo rule replaced by the exact one; emits the same
 * result structure as orc_pileup so tests can assert chunked == serial. */
int orc_pileup_chunked(const uint8_t* stream, uint64_t len, uint64_t chunk_bytes, int n_threads,
                       int want_entries, orc_pileup_result* out);

/* The chunk walk of pileupChunks (splitter.d:65-91, pileup.d:878-939): records [first, last] of every chunk it
 * yields a pileup for, and that pileup's column range [beg, end).  Returns the number of chunks (may exceed cap). */
typedef struct { uint64_t first, last; int32_t ref_id; int32_t _pad; uint64_t beg, end; } orc_chunk;
int64_t orc_chunk_plan(const uint8_t* stream, uint64_t len, uint64_t chunk_bytes, orc_chunk* out, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif
