// C++ rendering of the reference's own pileup checks through the range facade:
//   test/unittests.d:320-366 (makePileup -> chr1 only; pileupColumns -> 1470 + 1567 columns, first columns)
//   examples/make_pileup.d:20-30 (joined reads_starting_here == all reads)
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "biodgpu_pileup.hpp"

#define REQUIRE(x) do { if (!(x)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #x); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s ex1_header.bam illu_20_chunk.bam [bins.bam [copy_of_bins.bam [p1_records.bin [tags.bam]]]]\n", argv[0]); return 2; }
    using namespace biodgpu;
    {
        BamReader bam(argv[1], 2);
        auto refs = bam.reference_sequences();
        REQUIRE(refs.size() == 2 && refs[0].name == "chr1" && refs[1].name == "chr2");
        REQUIRE(bam.num_reads() == 3270);
        size_t n = 0;
        for (auto column : makePileup(bam.reads())) {                    // unittests.d:324-330
            REQUIRE(column.ref_id() == 0);
            for (auto& read : column.reads()) { REQUIRE(read.ref_id() == column.ref_id()); REQUIRE(!(read.flag() & 4)); }
            ++n;
        }
        REQUIRE(n == 1470);
        {   // the same sweep with the windows in compact transport (8 bytes a position over the link, widened by the range): same columns
            bam.compact_transport(true);
            size_t m = 0; uint64_t cov = 0, cov_wide = 0;
            for (auto column : makePileup(bam.reads())) { ++m; cov += column.coverage(); REQUIRE(column.coverage() == column.reads().size()); }
            bam.compact_transport(false);
            for (auto column : makePileup(bam.reads())) cov_wide += column.coverage();
            REQUIRE(m == 1470 && cov == cov_wide && cov > 0);
        }
        {   // reads.filter!(r => !r.is_reverse_strand) in front of the pileup (read.d:137-194): no column holds a reverse-strand read
            bam.filter(0, 0x10, 0, argv[1], 2);
            size_t seen = 0;
            for (auto column : pileupColumns(bam.reads())) for (auto& read : column.reads()) { REQUIRE(!(read.flag() & 0x10)); ++seen; }
            REQUIRE(seen > 0);
            bam.filter(0, 0, 0, argv[1], 2);
        }
        {   // the same file streamed through bounded device memory (bgp_stream_bam_begin), batches of two BGZF blocks: the same columns
            size_t m = 0; uint64_t cov = 0, cov_ref = 0;
            for (auto column : pileupColumns(bam.reads(), false, true, false)) cov_ref += column.coverage();
            for (auto column : StreamedPileup(argv[1], false, 128 << 10, 500)) { ++m; cov += column.coverage(); REQUIRE(column.reads().empty()); }
            REQUIRE(m == 1470 + 1567 && cov == cov_ref);
        }
        int expected[2] = {1470, 1567}, current = -1;
        for (auto column : pileupColumns(bam.reads())) {                 // unittests.d:334-366
            const int r = column.ref_id();
            --expected[r];
            if (r != current) {
                REQUIRE(r > current);
                if (r == 0) REQUIRE(column.position() == 99);
                if (r == 1) REQUIRE(column.position() == 0);
                current = r;
            }
            REQUIRE(column.coverage() == column.reads().size());
        }
        REQUIRE(expected[0] == 0 && expected[1] == 0);
        auto clipped = makePileup(bam.reads(), false, 500, 700, false);
        REQUIRE(clipped.start_position() == 500 && clipped.end_position() == 700 && clipped.ref_id() == 0);
        size_t k = 0; uint64_t last = 0;
        for (auto column : clipped) { REQUIRE(column.position() >= 500 && column.position() < 700); last = column.position(); ++k; }
        REQUIRE(k == 200 && last == 699);
    }
    {
        BamReader bam(argv[2], 2);                                       // make_pileup.d:20-30
        std::vector<uint32_t> starting;
        for (auto column : makePileup(bam.reads()))
            for (auto& read : column.reads_starting_here()) starting.push_back(read.read_ordinal());
        REQUIRE(starting.size() == 29);
        for (uint32_t i = 0; i < 29; ++i) REQUIRE(starting[i] == i);
        // make_pileup.d:16: makePileup(reads, true) -- reference_base rebuilt from the MD tags.  Bit-parity with
        // the oracle is tested through the Python binding; here: a base, and the one most reads agree with.
        size_t cols = 0, agree = 0;
        for (auto column : makePileup(bam.reads(), true)) {
            const char rb = column.reference_base();
            REQUIRE(rb == 'A' || rb == 'C' || rb == 'G' || rb == 'T' || rb == 'N');
            size_t same = 0;
            const std::string b = column.bases();
            for (char c : b) same += c == rb;
            agree += 2 * same > b.size();
            ++cols;
        }
        REQUIRE(cols > 100 && agree * 10 > cols * 9);
        for (auto column : makePileup(bam.reads(), false)) REQUIRE(column.reference_base() == 'N');      // pileup.d:245-249
        // snpcallers/simple.d:15-41 through the facade: every covered column gets a verdict; a column whose reads all
        // agree with the reference is never a SNP
        size_t verdicts = 0;
        auto snps = makePileup(bam.reads(), true);
        snps.callSNPs();
        for (auto column : snps) {
            const int v = column.is_snp();
            REQUIRE(v == BGP_SNP_NO || v == BGP_SNP_YES || v == BGP_SNP_TIE);
            bool all_ref = true;
            for (char c : column.bases()) all_ref = all_ref && c == column.reference_base();
            if (all_ref) REQUIRE(v == BGP_SNP_NO);
            ++verdicts;
        }
        REQUIRE(verdicts == cols);
        // snpcallers/maq.d:443-486 through the facade: a call for every covered column; columns whose valid bases all agree
        // with the reference call it homozygous reference, and no such column is one of findSNPs' SNPs
        size_t calls = 0;
        auto maq = makePileup(bam.reads(), true);
        maq.callMaq();
        for (auto column : maq) {
            const bgp_maq_call* m = column.maq_call();
            REQUIRE(m != nullptr);
            if (!(m->flags & BGP_MAQ_CALLED)) { REQUIRE(m->genotype == 0xFF); continue; }
            REQUIRE(m->genotype < 25); REQUIRE(m->quality >= 0.f);
            bool all_ref = true;
            for (char c : column.bases()) all_ref = all_ref && (c == column.reference_base() || c == '-');
            const char rb = column.reference_base();
            if (all_ref && (rb == 'A' || rb == 'C' || rb == 'G' || rb == 'T')) { REQUIRE(!(m->flags & BGP_MAQ_VARIANT)); REQUIRE(!column.is_maq_snp()); }
            ++calls;
        }
        REQUIRE(calls > 100);
        // pileup.d:1011-1015 through the facade: the chunks' columns, one chunk after the other, are pileupColumns'
        std::vector<uint64_t> whole, piece;
        for (auto column : pileupColumns(bam.reads(), false, true, false)) whole.push_back(((uint64_t)column.ref_id() << 40) ^ (column.position() << 8) ^ column.coverage());
        auto chunks = pileupChunks(bam.reads(), false, 2000, 0, std::numeric_limits<uint64_t>::max(), false);
        REQUIRE(chunks.size() >= 2);
        uint64_t prev_end = 0; int prev_ref = -1;
        for (auto chunk : chunks) {
            if (chunk.ref_id() == prev_ref) REQUIRE(chunk.start_position() == prev_end);      // the seam rule: a chunk begins where the previous one of its reference ended (pileup.d:905-913)
            prev_ref = chunk.ref_id();
            for (auto column : chunk) {
                REQUIRE(column.position() >= chunk.start_position() && column.position() < chunk.end_position());
                piece.push_back(((uint64_t)column.ref_id() << 40) ^ (column.position() << 8) ^ column.coverage());
            }
            prev_end = chunk.end_position();
        }
        REQUIRE(whole == piece);
        // mapping a caller over the chunks -- the reference's main use of pileupChunks: callMaq on an element holds
        size_t chunk_calls = 0;
        for (auto chunk : pileupChunks(bam.reads(), false, 2000)) {
            chunk.callMaq();
            for (auto column : chunk) { REQUIRE(column.maq_call() != nullptr); chunk_calls += (column.maq_call()->flags & BGP_MAQ_CALLED) != 0; }
        }
        REQUIRE(chunk_calls == calls);
    }
    if (argc > 3) {                                                      // unittests.d:145-185: bam["large"][beg .. end] == naive filter
        BamReader all(argv[3], 2);
        bgp_records recs; check(bgp_decode_fetch(all.ctx(), 0, &recs), all.ctx());
        const int regions[][2] = {{1400, 1500}, {10, 123}, {135, 1236}, {0, 3000}, {50000, 50100}, {950000, 950100}};
        BamReader bam(argv[3], 2);
        for (auto& rg : regions) {
            size_t naive = 0;
            for (uint64_t i = 0; i < recs.n; ++i)
                naive += recs.rec[i].ref_id == 2 && recs.rec[i].pos < rg[1] && recs.rec[i].end_pos > rg[0];
            REQUIRE(bam.fetch(argv[3], "large", (uint32_t)rg[0], (uint32_t)rg[1], 2).num_reads() == naive);
        }
    }
    if (argc > 4) {                                                      // createIndex, then the same region checks through the new index
        std::string copy = argv[4];                                      // a copy of bins.bam with no .bai beside it
        BamReader whole(copy, 2);
        whole.createIndex(copy + ".bai");
        BamReader all(copy, 2), bam(copy, 2);
        bgp_records recs; check(bgp_decode_fetch(all.ctx(), 0, &recs), all.ctx());
        const int regions[][2] = {{1400, 1500}, {0, 3000}, {50000, 50100}, {950000, 950100}};
        for (auto& rg : regions) {
            size_t naive = 0;
            for (uint64_t i = 0; i < recs.n; ++i)
                naive += recs.rec[i].ref_id == 2 && recs.rec[i].pos < rg[1] && recs.rec[i].end_pos > rg[0];
            REQUIRE(bam.fetch(copy, "large", (uint32_t)rg[0], (uint32_t)rg[1], 2).num_reads() == naive);
        }
    }
    if (argc > 5) {                                                      // the reference's own pileup unittest (pileup.d:699-826) through the facade, accessors included
        std::FILE* f = std::fopen(argv[5], "rb"); REQUIRE(f != nullptr);
        std::vector<uint8_t> st(1 << 16); st.resize(std::fread(st.data(), 1, st.size(), f)); std::fclose(f);
        BamReader bam(st.data(), st.size());
        const char* seqs[10] = {"ATTATGGACATTGTTTCCGTTATCATCATCATCATCATCATCATCATTATCATC", "GACATTGTTTCCGTTATCATCATCATCATCATCATCATCATCATCATCATCATC",
                                "ATTGTTTCCGTTATCATCATCATCATCATCATCATCATCATCATCATCATCACC", "TGTTTCCGTTATCATCATCATCATCATCATCATCATCATCATCATCATCACCAC",
                                "TCCGTTATCATCATCATCATCATCATCATCATCATCATCATCATCACCACCACC", "GTTATCATCATCATCATCATCATCATCATCATCATCATCATCATCGTCACCCTG",
                                "TCATCATCATCATAATCATCATCATCATCATCATCATCGTCACCCTGTGTTGAG", "TCATCATCATCGTCACCCTGTGTTGAGGACAGAAGTAATTTCCCTTTCTTGGCT",
                                "TCATCATCATCATCACCACCACCACCCTGTGTTGAGGACAGAAGTAATATCCCT", "CACCACCACCCTGTGTTGAGGACAGAAGTAATTTCCCTTTCTTGGCTGGTCACC"};   // pileup.d:709-718
        auto pileup = makePileup(bam.reads(), true, 796, 849, false);
        REQUIRE(pileup.front().position() == 796 && pileup.start_position() == 796 && pileup.end_position() == 849);   // pileup.d:766-768
        size_t seen = 0;
        for (auto column : pileup) {
            const auto reads = column.reads();
            REQUIRE(reads.size() == column.coverage());
            const auto bq = column.base_qualities(); const auto rq = column.read_qualities();
            REQUIRE(bq.size() == reads.size() && rq.size() == reads.size());
            for (size_t i = 0; i < reads.size(); ++i) {
                const auto& r = reads[i];
                REQUIRE(r.name() == "r" + std::to_string(r.read_ordinal()));                  // pileup.d:707
                REQUIRE(r.sequence() == seqs[r.read_ordinal()] && r.sequence_length() == 54);
                REQUIRE(r.base_qualities().size() == 54 && r.base_qualities()[0] == 0xFF);    // no qualities given: 0xFF, read.d:748
                REQUIRE(rq[i] == r.mapping_quality() && bq[i] == r.current_base_quality());
                if (r.current_base() != '-') REQUIRE(r.current_base() == r.sequence()[r.query_offset()]);   // pileup.d:115-122
                REQUIRE(r.cigar().size() == (size_t)(r.cigar_end() - r.cigar_begin()) && r.cigar()[r.cigar_operation_index()] == r.cigar_operation());
            }
            switch (column.position()) {                                                       // pileup.d:785-822
                case 796: REQUIRE(column.bases() == "CCCCCCAC"); ++seen; break;
                case 805: REQUIRE(column.bases() == "TCCCCCCCC"); ++seen; break;
                case 806: REQUIRE(column.bases() == "AAAAAAAGA"); ++seen; break;
                case 810: { const auto& r = reads[column.coverage() - 2]; REQUIRE((r.cigar_begin()[r.cigar_operation_index() + 1] & 15) == 2); ++seen; break; }    // cigar_after.front.type == 'D'
                case 817: { const auto& r = reads[column.coverage() - 2]; REQUIRE((r.cigar_begin()[r.cigar_operation_index() - 1] & 15) == 1); ++seen; break; }    // cigar_before.back.type == 'I'
                case 821: REQUIRE((reads[column.coverage() - 3].cigar_operation() & 15) == 2); REQUIRE(column.bases() == "AAGG-AA"); ++seen; break;
                case 826: REQUIRE(column.bases() == "CCCCCC"); ++seen; break;
                default: break;
            }
        }
        REQUIRE(seen == 7);
    }
    if (argc > 6) {
        // test/unittests.d:110-129 through column.reads: every read of tags.bam is named "tag_<key>:<the tag's value in SAM form>", and
        // alignment[key].toSam() must say the same.  BioD's PileupRead forwards to its BamRead (pileup.d:89); here the read of a COLUMN
        // is asked for its tag (read.d:1070-1087, tagvalue.d:461-505).
        BamReader bam(argv[6], 2);
        std::vector<char> seen(bam.num_reads(), 0);
        size_t checked = 0;
        for (auto column : pileupColumns(bam.reads()))
            for (auto& read : column.reads()) {
                if (seen[read.read_ordinal()]) continue;
                seen[read.read_ordinal()] = 1;
                const std::string name = read.name();
                REQUIRE(name.compare(0, 4, "tag_") == 0);
                const size_t colon = name.find(':', 4);
                REQUIRE(colon == 6);                                                     // two-letter keys
                const std::string key = name.substr(4, 2), want = name.substr(colon + 1);
                const TagValue v = read.tag(key.c_str());
                REQUIRE(!v.is_nothing());
                if (v.to_sam() != want) { std::fprintf(stderr, "tag %s: name says %s, value says %s\n", key.c_str(), want.c_str(), v.to_sam().c_str()); return 1; }
                REQUIRE(read.tag("??").is_nothing());                                    // "no such tag": Value(null)
                ++checked;
            }
        REQUIRE(checked > 300);                                                          // (every read of the file that covers a position: 417 reads, a few of them clipped to nothing)
        std::printf("tags through column.reads: %zu reads\n", checked);
    }
    std::puts("facade ok");
    return 0;
}
