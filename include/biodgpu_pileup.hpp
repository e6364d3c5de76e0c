// biodgpu_pileup.hpp -- header-only C++ mirror of BioD's pileup range API over the libbiodgpu C ABI.
//
// BioD (D) exposes input ranges: `foreach (column; makePileup(bam.reads, ...))` with
// empty/front/popFront (bio/std/hts/bam/pileup.d:335-397).  This header keeps that shape in C++ --
// same function names, parameter order and defaults as pileup.d:683-687 (makePileup) and pileup.d:509
// (pileupColumns) -- for hosts that cannot run D (this image has no D compiler).  The D module a BioD
// maintainer would add is shown in INTEGRATION.md; both are thin: every column comes from the HIP
// kernels behind include/biodgpu.h.
#pragma once
#include <cstdint>
#include <cstdio>
#include <limits>
#include <stdexcept>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "biodgpu.h"

namespace biodgpu {

// The D facade rethrows status codes as Exception (cf. ZlibException, bio/core/utils/zlib.d:247-273).
struct Error : std::runtime_error {
    int code;
    Error(int c, bgp_ctx* ctx) : std::runtime_error(std::string(bgp_strerror(c)) + " [" + (bgp_last_error(ctx) ? bgp_last_error(ctx) : "") + "]"), code(c) {}
};
inline int check(int rc, bgp_ctx* ctx = nullptr) { if (rc < 0) throw Error(rc, ctx); return rc; }

// `new BamReader(filename[, taskPool])`, bio/std/hts/bam/reader.d:127-138; `.reads` feeds the pileup.
class BamReader {
public:
    explicit BamReader(const std::string& filename, int n_threads = 0, int device = 0) {
        check(bgp_init(device, &ctx_));
        try { check(bgp_load_bam(ctx_, filename.c_str(), n_threads), ctx_); check(bgp_flush(ctx_), ctx_); }
        catch (...) { bgp_destroy(ctx_); throw; }
    }
    // reads given as inflated record bytes (what BamReadRange consumes, readrange.d:118-173)
    BamReader(const uint8_t* records, size_t len, int device = 0) {
        check(bgp_init(device, &ctx_));
        try { check(bgp_push_inflated(ctx_, records, len, 0, BGP_UOFFSET_UNKNOWN), ctx_); check(bgp_flush(ctx_), ctx_); }
        catch (...) { bgp_destroy(ctx_); throw; }
    }
    ~BamReader() { bgp_destroy(ctx_); }
    BamReader(const BamReader&) = delete;
    BamReader& operator=(const BamReader&) = delete;

    struct ReferenceSequenceInfo { std::string name; int32_t length; };       // referenceinfo.d
    std::vector<ReferenceSequenceInfo> reference_sequences() const {           // reader.d:189-199
        int32_t n = 0; check(bgp_n_ref(ctx_, &n), ctx_);
        std::vector<ReferenceSequenceInfo> v;
        for (int32_t i = 0; i < n; ++i) { const char* nm; int32_t len; check(bgp_ref_info(ctx_, i, &nm, &len), ctx_); v.push_back({nm, len}); }
        return v;
    }
    uint64_t num_reads() const { uint64_t n; check(bgp_num_records(ctx_, &n), ctx_); return n; }
    // windows cross the link as 8 bytes a position wherever every count fits a byte (bgp_set_compact_columns); PileupRange widens them:
    // the consumer sees the same numbers (pileup.d:236-290)
    BamReader& compact_transport(bool on = true) { check(bgp_set_compact_columns(ctx_, on ? 1 : 0), ctx_); return *this; }
    // reads.filter!(r => (r.flag & require) == require && !(r.flag & exclude) && r.mapping_quality >= min_mapq) in front of the pileup
    // (read.d:137-194): re-flushes what is staged with the predicate applied (bgp_set_read_filter + bgp_flush); reads that fail take part in nothing
    BamReader& filter(uint32_t flag_require, uint32_t flag_exclude, uint32_t min_mapping_quality, const std::string& filename, int n_threads = 0) {
        check(bgp_set_read_filter(ctx_, flag_require, flag_exclude, min_mapping_quality), ctx_);
        check(bgp_reset(ctx_), ctx_);
        check(bgp_load_bam(ctx_, filename.c_str(), n_threads), ctx_); check(bgp_flush(ctx_), ctx_);
        return *this;
    }
    BamReader& reads() { return *this; }                                       // reader.d:228-231
    // bam["chr1"][beg .. end] (reader.d opIndex -> reference.d:73-81): re-stages this reader with the reads that
    // overlap the region, found through the .bai next to the file (RandomAccessManager.getReads, :330-336).
    BamReader& fetch(const std::string& filename, const std::string& ref_name, uint32_t beg, uint32_t end, int n_threads = 0) {
        bgp_bai* bai = nullptr;
        check(bgp_bai_open(filename.c_str(), &bai), ctx_);                     // "BAM index file (.bai) must be provided"
        struct Close { bgp_bai* b; ~Close() { bgp_bai_close(b); } } guard{bai};
        int32_t id = -1; const auto refs = reference_sequences();
        for (size_t i = 0; i < refs.size(); ++i) if (refs[i].name == ref_name) id = (int32_t)i;
        if (id < 0) throw std::out_of_range("no such reference: " + ref_name);
        check(bgp_reset(ctx_), ctx_);
        check(bgp_load_bam_region(ctx_, filename.c_str(), bai, id, beg, end, n_threads), ctx_);
        check(bgp_flush(ctx_), ctx_);
        return *this;
    }
    uint64_t start_virtual_offset(uint64_t read_index) const { uint64_t v; check(bgp_record_voffset(ctx_, read_index, &v), ctx_); return v; }   // readrange.d:64-86
    // createIndex (reader.d:158-168 -> bai/indexing.d:353-366): the .bai of the file this reader was opened on.
    std::vector<uint8_t> index_bytes() const {
        const uint8_t* p = nullptr; uint64_t n = 0;
        check(bgp_bai_build(ctx_, &p, &n), ctx_);
        return std::vector<uint8_t>(p, p + n);
    }
    void createIndex(const std::string& bai_filename) const {
        const std::vector<uint8_t> b = index_bytes();
        std::FILE* f = std::fopen(bai_filename.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot write " + bai_filename);
        const size_t w = std::fwrite(b.data(), 1, b.size(), f);
        if (std::fclose(f) != 0 || w != b.size()) throw std::runtime_error("short write to " + bai_filename);
    }
    bgp_ctx* ctx() const { return ctx_; }

private:
    bgp_ctx* ctx_ = nullptr;
};

// A tag's value, tagvalue.d's Value as far as a reader needs it: found the way BamRead.opIndex finds it (read.d:1070-1087: a linear scan
// of the aux area from _tags_offset, two key bytes, a type byte, the value skipped by its type -- read.d:1219-1230), rendered by to_sam()
// the way Value.toSam does (tagvalue.d:461-505).  Views the host copy of the record.
struct TagValue {
    char type = 0;                 // 0: no such tag (Value(null), is_nothing); else the BAM type char: A c C s S i I f Z H B
    char elem = 0;                 // B: the elements' type char
    const uint8_t* p = nullptr;    // the value's bytes (Z / H: up to the NUL; B: the elements)
    uint32_t count = 0;            // B: elements; Z / H: chars
    bool is_nothing() const { return type == 0; }
    static size_t size_of(char t) { switch (t) { case 'A': case 'c': case 'C': return 1; case 's': case 'S': return 2; case 'i': case 'I': case 'f': return 4; default: return 0; } }   // tagvalue.d:78-85,106-119
    static std::string one(char t, const uint8_t* q) {
        char buf[48];
        switch (t) {
            case 'c': std::snprintf(buf, sizeof buf, "%d", (int)*(const int8_t*)q); break;
            case 'C': std::snprintf(buf, sizeof buf, "%u", (unsigned)*q); break;
            case 's': { int16_t v; std::memcpy(&v, q, 2); std::snprintf(buf, sizeof buf, "%d", (int)v); break; }
            case 'S': { uint16_t v; std::memcpy(&v, q, 2); std::snprintf(buf, sizeof buf, "%u", (unsigned)v); break; }
            case 'i': { int32_t v; std::memcpy(&v, q, 4); std::snprintf(buf, sizeof buf, "%d", v); break; }
            case 'I': { uint32_t v; std::memcpy(&v, q, 4); std::snprintf(buf, sizeof buf, "%u", v); break; }
            case 'f': { float v; std::memcpy(&v, q, 4); if (v != v) return "nan"; std::snprintf(buf, sizeof buf, "%g", (double)v); break; }
            default: buf[0] = 0;
        }
        return buf;
    }
    std::string to_sam() const {                                                                                            // tagvalue.d:468-505
        switch (type) {
            case 0: return "";
            case 'A': return std::string("A:") + (char)*p;
            case 'Z': return "Z:" + std::string((const char*)p, count);
            case 'H': return "H:" + std::string((const char*)p, count);
            case 'f': return "f:" + one('f', p);
            case 'B': { std::string o = std::string("B:") + elem; const size_t w = size_of(elem); for (uint32_t k = 0; k < count; ++k) o += "," + one(elem, p + k * w); return o; }
            default: return "i:" + one(type, p);                                                                          // every integer width is "i:" in SAM
        }
    }
};

// PileupRead, pileup.d:86-230 (a view into the window the range currently holds)
struct PileupRead {
    const bgp_entry* e; const bgp_records* recs;
    uint32_t read_ordinal() const { return e->read_ordinal; }
    uint32_t cigar_operation_offset() const { return e->cigar_op_offset; }     // pileup.d:99
    int query_offset() const { return (int)e->query_offset; }                  // pileup.d:146
    char current_base() const { return (char)e->base; }                        // pileup.d:115
    uint8_t current_base_quality() const { return e->qual; }                   // pileup.d:127
    const bgp_record& record() const { return recs->rec[e->read_ordinal]; }
    int ref_id() const { return record().ref_id; }
    int position() const { return record().pos; }
    int end_position() const { return record().end_pos; }
    uint8_t mapping_quality() const { return (record().bin_mq_nl >> 8) & 0xFF; }
    uint16_t flag() const { return (uint16_t)(record().flag_nc >> 16); }
    bool is_reverse_strand() const { return flag() & 0x10; }
    uint32_t cigar_operation() const { return recs->cigar[recs->cigar_off[e->read_ordinal] + e->cigar_op_index]; }   // raw len<<4|op, pileup.d:94
    const uint32_t* cigar_begin() const { return recs->cigar + recs->cigar_off[e->read_ordinal]; }
    const uint32_t* cigar_end() const { return recs->cigar + recs->cigar_off[e->read_ordinal + 1]; }
    uint16_t cigar_operation_index() const { return e->cigar_op_index; }       // cigar_before = [begin, begin+index), cigar_after = (begin+index, end)
    // The BamRead accessors PileupRead forwards with `alias _read this` (pileup.d:86-112), as BioD's consumers use them
    // (examples/read_bam_file.d:33-41, snpcallers/simple.d:24-28, maq.d:396-412): views of the host copy of the records.
    std::vector<uint32_t> cigar() const { return std::vector<uint32_t>(cigar_begin(), cigar_end()); }                    // read.d:225-228
    int sequence_length() const { return record().l_seq; }                                                                // read.d:923
    std::string sequence() const {                                                                                         // read.d:438-440, chars of base.d:85
        return std::string((const char*)recs->seq + recs->seq_off[e->read_ordinal], (size_t)(recs->seq_off[e->read_ordinal + 1] - recs->seq_off[e->read_ordinal]));
    }
    const uint8_t* base_qualities_begin() const { return recs->qual + recs->seq_off[e->read_ordinal]; }                  // read.d:468-470
    std::vector<uint8_t> base_qualities() const { return std::vector<uint8_t>(base_qualities_begin(), base_qualities_begin() + record().l_seq); }
    std::string name() const {                                                                                             // read.d:218-222: l_read_name - 1 chars behind the 32 fixed bytes
        if (!recs->stream) throw std::logic_error("read names need the host copy of the record stream (BGP_FETCH_STREAM)");
        return std::string((const char*)recs->stream + record().rec_off + 36, (size_t)(record().bin_mq_nl & 0xFF) - 1);
    }
    // read["RG"] (read.d:1070-1087): BioD's PileupRead is `alias read this` (pileup.d:89), so a consumer may ask column.reads[k] for a tag.
    // The aux area lies behind QUAL (read.d:1000-1003) and ends with the record (block_size).
    TagValue tag(const char* key) const {
        if (!recs->stream) throw std::logic_error("tags need the host copy of the record stream (BGP_FETCH_STREAM)");
        if (!key || std::strlen(key) != 2) throw std::invalid_argument("Key length must be 2");                           // read.d:1071
        const bgp_record& r = record();
        const uint8_t* rec0 = recs->stream + r.rec_off;
        uint32_t block_size; std::memcpy(&block_size, rec0, 4);
        const size_t l_name = r.bin_mq_nl & 0xFF, n_cig = r.flag_nc & 0xFFFF, l_seq = (size_t)r.l_seq;
        const uint8_t* t = rec0 + 36 + l_name + 4 * n_cig + (l_seq + 1) / 2 + l_seq;                                     // read.d:984-1003
        const uint8_t* end = rec0 + 4 + block_size;
        TagValue none;
        if (end - t < 4) return none;                                                                                      // read.d:1073-1074
        while (t + 1 < end) {                                                                                              // read.d:1077
            const bool hit = t[0] == (uint8_t)key[0] && t[1] == (uint8_t)key[1];
            t += 2;
            if (t >= end) break;
            const char ty = (char)*t++;
            TagValue v; v.type = ty; v.p = t;
            if (ty == 'Z' || ty == 'H') { const uint8_t* z = t; while (z < end && *z) ++z; v.count = (uint32_t)(z - t); t = z + 1; }          // read.d:1221-1222
            else if (ty == 'B') {                                                                                          // read.d:1223-1226
                if (end - t < 5) break;
                v.elem = (char)*t++; uint32_t n; std::memcpy(&n, t, 4); t += 4; v.p = t; v.count = n;
                t += (size_t)n * TagValue::size_of(v.elem);
            } else { const size_t w = TagValue::size_of(ty); if (!w) break; t += w; }                                      // read.d:1227-1229
            if (t > end) break;
            if (hit) return v;
        }
        return none;
    }
};

// PileupColumn, pileup.d:236-290
struct PileupColumn {
    int ref_id_; uint64_t position_; const bgp_column* c; const bgp_entry* ents; size_t n_ents; const bgp_records* recs; char refbase_; int snp_ = -1; const bgp_maq_call* maq_ = nullptr;
    size_t coverage() const { return c->coverage; }                            // pileup.d:252
    int ref_id() const { return ref_id_; }                                     // pileup.d:257
    uint64_t position() const { return position_; }                            // pileup.d:262
    char reference_base() const { return refbase_; }                           // pileup.d:245-249 ('N' unless use_md_tag)
    size_t n_starting_here() const { return c->n_starting_here; }
    std::vector<PileupRead> reads() const { std::vector<PileupRead> v; for (size_t i = 0; i < n_ents; ++i) v.push_back({ents + i, recs}); return v; }
    std::vector<PileupRead> reads_starting_here() const {                      // pileup.d:272-274
        std::vector<PileupRead> v; for (size_t i = n_ents - c->n_starting_here; i < n_ents; ++i) v.push_back({ents + i, recs}); return v;
    }
    // isSNP(column, settings), snpcallers/simple.d:15-41: BGP_SNP_NO / BGP_SNP_YES / BGP_SNP_TIE (BioD's answer would depend
    // on druntime's associative-array order); -1 unless PileupRange::callSNPs was used.  Columns the device could not
    // decide from the summary (ambiguity codes may hold the top count) are decided here from the per-read bases.
    int is_snp(int minimum_coverage = 5, int minimum_witnesses = 2) const {
        if (snp_ != BGP_SNP_NEEDS_ENTRIES || !ents) return snp_;
        if ((unsigned long long)coverage() < (unsigned long long)(long long)minimum_coverage) return BGP_SNP_NO;
        int cnt[256] = {0}, top = 0, n_top = 0;
        for (size_t i = 0; i < n_ents; ++i) if (ents[i].base != '-') ++cnt[ents[i].base];
        for (int b = 0; b < 256; ++b) top = cnt[b] > top ? cnt[b] : top;
        if (top == 0 || top < minimum_witnesses) return BGP_SNP_NO;
        for (int b = 0; b < 256; ++b) n_top += cnt[b] == top;
        if (cnt[(unsigned char)refbase_] != top) return BGP_SNP_YES;
        return n_top == 1 ? BGP_SNP_NO : BGP_SNP_TIE;
    }
    std::string bases() const { std::string s; for (size_t i = 0; i < n_ents; ++i) s.push_back((char)ents[i].base); return s; }   // pileup.d:277
    std::vector<uint8_t> base_qualities() const { std::vector<uint8_t> v; for (size_t i = 0; i < n_ents; ++i) v.push_back(ents[i].qual); return v; }   // pileup.d:282-284
    std::vector<uint8_t> read_qualities() const {                              // pileup.d:287-289: mapping quality of every read of the column
        std::vector<uint8_t> v; for (size_t i = 0; i < n_ents; ++i) v.push_back((uint8_t)((recs->rec[ents[i].read_ordinal].bin_mq_nl >> 8) & 0xFF)); return v;
    }
    // MaqSnpCaller.makeCall(column), snpcallers/maq.d:443-486, after PileupRange::callMaq: nullptr without it; a call whose
    // flags lack BGP_MAQ_CALLED is makeCall's null (no valid base).  genotype = base1 * 5 + base2 over "ACGTN".
    const bgp_maq_call* maq_call() const { return maq_; }
    bool is_maq_snp(float minimum_call_quality = 6.0f) const {                  // what findSNPs yields, maq.d:519-526
        return maq_ && (maq_->flags & BGP_MAQ_CALLED) && (maq_->flags & BGP_MAQ_VARIANT) && maq_->quality > minimum_call_quality;
    }
};

// The input range (empty/front/popFront, pileup.d:335-397) over device-computed windows.
class PileupRange {
public:
    PileupRange(BamReader& r, int mode, bool use_md_tag, uint64_t start_from, uint64_t end_at, bool skip_zero_coverage, bool want_entries, uint32_t window)
        : ctx_(r.ctx()), skip_zero_(skip_zero_coverage), want_entries_(want_entries), start_(start_from), end_(end_at), mode_(mode), use_md_(use_md_tag), window_(window) {
        if (want_entries) check(bgp_decode_fetch(ctx_, BGP_FETCH_ALL | BGP_FETCH_STREAM, &recs_), ctx_);
        restart();
    }
    // element `chunk` of pileupChunks (below): the chunk's columns, [start_position, end_position) of its reference
    PileupRange(BamReader& r, uint64_t chunk, bool want_entries)
        : ctx_(r.ctx()), skip_zero_(true), want_entries_(want_entries), start_(0), end_(0), mode_(BGP_MAKE_PILEUP), use_md_(false), window_(0), chunk_((int64_t)chunk) {
        if (want_entries) check(bgp_decode_fetch(ctx_, BGP_FETCH_ALL | BGP_FETCH_STREAM, &recs_), ctx_);
        restart();
    }
    uint64_t first_record() const { return ci_.first_record; }
    uint64_t last_record() const { return ci_.last_record; }
    bool empty() const { return done_; }
    PileupColumn front() const {
        const uint64_t a = want_entries_ ? w_.entry_off[k_] : 0, b = want_entries_ ? w_.entry_off[k_ + 1] : 0;
        return PileupColumn{w_.ref_id, w_.first_position + k_, w_.columns + k_, want_entries_ ? w_.entries + a : nullptr, (size_t)(b - a), &recs_,
                            w_.reference_base ? (char)w_.reference_base[k_] : 'N', w_.snp ? (int)w_.snp[k_] : -1, w_.maq ? w_.maq + k_ : nullptr};
    }
    void popFront() { ++k_; settle(); }
    // findSNPs' settings (simple.d:8-13,37-41); call right after construction with use_md_tag = true
    PileupRange& callSNPs(int minimum_coverage = 5, int minimum_witnesses = 2) {
        snp_ = true; snp_cov_ = minimum_coverage; snp_wit_ = minimum_witnesses;
        restart();                                                             // the first window was fetched without verdicts
        return *this;
    }
    // MaqSnpCaller's settings (maq.d:329-333): every column then carries the model's call
    PileupRange& callMaq(float depcorr = 0.17f, float eta = 0.03f, unsigned minimum_base_quality = 13) {
        maq_ = true; maq_depcorr_ = depcorr; maq_eta_ = eta; maq_min_bq_ = minimum_base_quality;
        restart();
        return *this;
    }
    uint64_t start_position() const { return start_; }                         // pileup.d:445
    uint64_t end_position() const { return end_; }                             // pileup.d:450
    int ref_id() const { return ref_id_; }                                     // pileup.d:455

    struct iterator {                                                          // for range-for, the `foreach` of D
        PileupRange* r;
        bool operator!=(const iterator&) const { return !r->empty(); }
        void operator++() { r->popFront(); }
        PileupColumn operator*() const { return r->front(); }
    };
    iterator begin() { return {this}; }
    iterator end() { return {this}; }

private:
    void restart() {
        if (chunk_ >= 0) {                                                     // a chunk of pileupChunks: select it, its windows follow
            check(bgp_pileup_chunk(ctx_, (uint64_t)chunk_, &ci_), ctx_);
            start_ = ci_.start_position; end_ = ci_.end_position; ref_id_ = ci_.ref_id;
            if (snp_) check(bgp_pileup_snp(ctx_, snp_cov_, snp_wit_), ctx_);   // (selecting a chunk starts a new iteration: the callers are set again)
            if (maq_) check(bgp_pileup_maq(ctx_, maq_depcorr_, maq_eta_, maq_min_bq_), ctx_);
            done_ = false;
            fetch(); settle();
            return;
        }
        check(bgp_pileup_begin(ctx_, mode_, use_md_, start_, end_, skip_zero_, want_entries_, window_), ctx_);
        if (snp_) check(bgp_pileup_snp(ctx_, snp_cov_, snp_wit_), ctx_);
        if (maq_) check(bgp_pileup_maq(ctx_, maq_depcorr_, maq_eta_, maq_min_bq_), ctx_);
        if (mode_ == BGP_MAKE_PILEUP) check(bgp_pileup_ref_id(ctx_, &ref_id_), ctx_);
        done_ = false;
        fetch(); settle();
    }
    void fetch() {
        const int rc = check(bgp_pileup_next_window(ctx_, &w_), ctx_); done_ = rc == BGP_EOF; k_ = 0;
        if (!done_ && !w_.columns && w_.columns8) {     // compact transport (bgp_set_compact_columns): a byte a count over the link, widened here -- the consumer sees pileup.d:236-290's numbers
            wide_.resize(w_.n_positions);
            for (uint32_t k = 0; k < w_.n_positions; ++k) { const bgp_column8& s = w_.columns8[k]; wide_[k] = bgp_column{s.coverage, s.n_starting_here, s.a, s.c, s.g, s.t, s.other, s.del}; }
            w_.columns = wide_.data();
        }
    }
    void settle() {            // skip zero-coverage positions the way PileupRange.popFront jumps over them, pileup.d:388-392
        while (!done_) {
            if (k_ >= w_.n_positions) { fetch(); continue; }
            if (skip_zero_ && w_.columns[k_].coverage == 0) { ++k_; continue; }
            return;
        }
    }
    bgp_ctx* ctx_; bool skip_zero_, want_entries_; uint64_t start_, end_;
    int mode_; bool use_md_; uint32_t window_; bool snp_ = false; int snp_cov_ = 5, snp_wit_ = 2;
    bool maq_ = false; float maq_depcorr_ = 0.17f, maq_eta_ = 0.03f; unsigned maq_min_bq_ = 13;
    bgp_columns w_{}; std::vector<bgp_column> wide_; bgp_records recs_{}; uint32_t k_ = 0; bool done_ = false; int32_t ref_id_ = -1;
    int64_t chunk_ = -1; bgp_chunk_info ci_{};
};

// makePileup(reads, use_md_tag=false, start_from=0, end_at=ulong.max, skip_zero_coverage=true), pileup.d:683-694
inline PileupRange makePileup(BamReader& reads, bool use_md_tag = false, uint64_t start_from = 0,
                              uint64_t end_at = std::numeric_limits<uint64_t>::max(), bool skip_zero_coverage = true,
                              bool want_entries = true, uint32_t window = 0) {
    return PileupRange(reads, BGP_MAKE_PILEUP, use_md_tag, start_from, end_at, skip_zero_coverage, want_entries, window);
}
// pileupColumns(reads, use_md_tag=false, skip_zero_coverage=true), pileup.d:509-519
inline PileupRange pileupColumns(BamReader& reads, bool use_md_tag = false, bool skip_zero_coverage = true,
                                 bool want_entries = true, uint32_t window = 0) {
    return PileupRange(reads, BGP_PILEUP_COLUMNS, use_md_tag, 0, std::numeric_limits<uint64_t>::max(), skip_zero_coverage, want_entries, window);
}

// pileupChunks(reads, use_md_tag=false, block_size=16_384_000, start_from=0, end_at=ulong.max), pileup.d:1011-1015:
// consecutive non-overlapping pileups.  Each element is positioned on its chunk when it is created (here: when the
// vector is walked with `for (auto& chunk : ...)` the elements were created in order; to revisit one, construct
// PileupRange(reads, k, want_entries) again) -- one ctx serves one chunk at a time.
class PileupChunks {
public:
    PileupChunks(BamReader& r, bool use_md_tag, uint64_t block_size, uint64_t start_from, uint64_t end_at, bool want_entries, uint32_t window)
        : r_(r), want_entries_(want_entries) {
        check(bgp_pileup_chunks_begin(r.ctx(), use_md_tag, block_size, start_from, end_at, want_entries, window, &n_), r.ctx());
    }
    uint64_t size() const { return n_; }
    PileupRange operator[](uint64_t k) const { return PileupRange(r_, k, want_entries_); }
    struct iterator {
        const PileupChunks* p; uint64_t k;
        bool operator!=(const iterator& o) const { return k != o.k; }
        void operator++() { ++k; }
        PileupRange operator*() const { return (*p)[k]; }
    };
    iterator begin() const { return {this, 0}; }
    iterator end() const { return {this, n_}; }
private:
    BamReader& r_; bool want_entries_; uint64_t n_ = 0;
};
inline PileupChunks pileupChunks(BamReader& reads, bool use_md_tag = false, uint64_t block_size = 16384000, uint64_t start_from = 0,
                                 uint64_t end_at = std::numeric_limits<uint64_t>::max(), bool want_entries = true, uint32_t window = 0) {
    return PileupChunks(reads, use_md_tag, block_size, start_from, end_at, want_entries, window);
}

// pileupColumns(new BamReader(filename).reads, use_md_tag) with the FILE streamed through the device in bounded memory (bgp_stream_bam_begin:
// reader threads -> ring of compressed chunks in HBM -> k_inflate -> a batch's kernels -> windows; BioD's ring of blocks in flight under a
// pulling consumer, inputstream.d:386-447,455-483).  The columns' summaries (coverage, counts, reference_base): `reads()` of a column is
// empty -- a consumer that wants the reads of a file that fits the device uses BamReader.  skip_zero_coverage as pileupColumns' default.
class StreamedPileup {
public:
    explicit StreamedPileup(const std::string& filename, bool use_md_tag = false, uint64_t batch_bytes = 0, uint32_t window = 0, bool compact_transport = true, int device = 0) {
        check(bgp_init(device, &ctx_));
        try {
            check(bgp_set_compact_columns(ctx_, compact_transport ? 1 : 0), ctx_);
            check(bgp_stream_bam_begin(ctx_, filename.c_str(), batch_bytes, 0, window), ctx_);
            if (use_md_tag) check(bgp_batch_set_md(ctx_, 1, 1), ctx_);
            fetch(); settle();
        } catch (...) { bgp_destroy(ctx_); throw; }
    }
    ~StreamedPileup() { bgp_destroy(ctx_); }
    StreamedPileup(const StreamedPileup&) = delete;
    StreamedPileup& operator=(const StreamedPileup&) = delete;
    bool empty() const { return done_; }
    PileupColumn front() const {
        return PileupColumn{w_.ref_id, w_.first_position + k_, w_.columns + k_, nullptr, 0, nullptr, w_.reference_base ? (char)w_.reference_base[k_] : 'N', -1, nullptr};
    }
    void popFront() { ++k_; settle(); }
    struct iterator {
        StreamedPileup* r;
        bool operator!=(const iterator&) const { return !r->empty(); }
        void operator++() { r->popFront(); }
        PileupColumn operator*() const { return r->front(); }
    };
    iterator begin() { return {this}; }
    iterator end() { return {this}; }
private:
    void fetch() {
        const int rc = check(bgp_batch_next_window(ctx_, &w_), ctx_); done_ = rc != BGP_OK; k_ = 0;
        if (!done_ && !w_.columns && w_.columns8) {
            wide_.resize(w_.n_positions);
            for (uint32_t k = 0; k < w_.n_positions; ++k) { const bgp_column8& s = w_.columns8[k]; wide_[k] = bgp_column{s.coverage, s.n_starting_here, s.a, s.c, s.g, s.t, s.other, s.del}; }
            w_.columns = wide_.data();
        }
    }
    void settle() { while (!done_) { if (k_ >= w_.n_positions) { fetch(); continue; } if (w_.columns[k_].coverage == 0) { ++k_; continue; } return; } }
    bgp_ctx* ctx_ = nullptr; bgp_columns w_{}; std::vector<bgp_column> wide_; uint32_t k_ = 0; bool done_ = false;
};

}  // namespace biodgpu
