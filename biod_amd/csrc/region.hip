// region.hip -- bam[ref][beg .. end] on the staged record table.
//
// Reference semantics: RandomAccessManager.BamReadFilter (bio/std/hts/bam/randomaccessmanager.d:366-462) over a
// range of reads sorted by (ref_id, position), for ONE region:
//   ref_id (as uint) above the region's  -> the range ends            (:398-402)
//   ref_id below                         -> read skipped              (:403-408)
//   position >= region.end               -> the range ends            (:410-431, last region)
//   position >  region.start             -> yielded                   (:433-435)
//   position + basesCovered <= start     -> skipped, else yielded     (:437-449)
// D compares the int position against the uint bounds as uint; the kernels do the same.  "The range ends" is a
// prefix property (first record that trips it), found with one atomicMin, so unsorted input gives BioD's answer
// too.  Survivors are compacted (stable) into a new record table with fresh SEQ / CIGAR arena offsets; rec_off
// keeps pointing into the staged stream, so K2/K3/K4/K5/K6 run on the subset unchanged.
#include "common.h"

#include <initializer_list>
#include <utility>

namespace {

struct RegArgs { const bgp_record* rec; uint64_t n; uint32_t ref, beg, end; unsigned long long* stop; uint64_t* keep; uint64_t* lseq; uint64_t* ncig; };

__global__ __launch_bounds__(256) void k_region_stop(RegArgs a) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool trip = false;
    if (i < a.n) {
        const bgp_record r = a.rec[i];
        const uint32_t rr = (uint32_t)r.ref_id;
        trip = rr > a.ref || (rr == a.ref && (uint32_t)r.pos >= a.end);
    }
    const uint64_t m = __ballot(trip);
    if (m && (threadIdx.x & 63) == (unsigned)(__ffsll((unsigned long long)m) - 1)) atomicMin(a.stop, (unsigned long long)i);
}

__global__ __launch_bounds__(256) void k_region_flags(RegArgs a) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const bgp_record r = a.rec[i];
    bool keep = false;
    if (i < *a.stop && (uint32_t)r.ref_id == a.ref)
        keep = (uint32_t)r.pos > a.beg || (uint32_t)r.end_pos > a.beg;       // end_pos = position + basesCovered
    a.keep[i] = keep ? 1u : 0u;
    a.lseq[i] = keep ? (uint64_t)(uint32_t)r.l_seq : 0u;
    a.ncig[i] = keep ? (uint64_t)(r.flag_nc & 0xFFFFu) : 0u;
}

__global__ __launch_bounds__(256) void k_region_scatter(const bgp_record* rec, uint64_t n, const uint64_t* keep, const uint64_t* idx, const uint64_t* soff, const uint64_t* coff,
                                                        bgp_record* rec2, uint64_t* soff2, uint64_t* coff2) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { const uint64_t k = idx[n]; soff2[k] = soff[n]; coff2[k] = coff[n]; return; }
    if (!keep[i]) return;
    const uint64_t k = idx[i];
    rec2[k] = rec[i]; soff2[k] = soff[i]; coff2[k] = coff[i];
}

}  // namespace

int k_region_filter(bgp_ctx* c, uint32_t ref, uint32_t beg, uint32_t end) {
    const uint64_t n = c->n_records;
    DevBuf& K = c->d_flt[0]; DevBuf& LS = c->d_flt[1]; DevBuf& NC = c->d_flt[2]; DevBuf& IDX = c->d_flt[3]; DevBuf& SO = c->d_flt[4]; DevBuf& CO = c->d_flt[5];
    for (DevBuf* b : {&K, &LS, &NC, &IDX, &SO, &CO}) BGP_TRY(b->reserve((n + 2) * 8));
    BGP_TRY(c->d_rec2.reserve((n + 1) * sizeof(bgp_record))); BGP_TRY(c->d_seq_off2.reserve((n + 1) * 8)); BGP_TRY(c->d_cigar_off2.reserve((n + 1) * 8));
    unsigned long long* stop = (unsigned long long*)(K.as<uint64_t>() + n + 1);
    const unsigned long long none = ~0ull;
    BGP_HIP(hipMemcpyAsync(stop, &none, 8, hipMemcpyHostToDevice, c->stream));
    uint64_t n_new = 0, sl = 0, sc = 0;
    if (n) {
        RegArgs a{c->d_rec.as<bgp_record>(), n, ref, beg, end, stop, K.as<uint64_t>(), LS.as<uint64_t>(), NC.as<uint64_t>()};
        const unsigned nb = (unsigned)((n + 255) / 256);
        TimedRegion t(c, "k_region_filter", n * 32);
        hipLaunchKernelGGL(k_region_stop, dim3(nb), dim3(256), 0, c->stream, a);
        hipLaunchKernelGGL(k_region_flags, dim3(nb), dim3(256), 0, c->stream, a);
        BGP_HIP(hipGetLastError());
        BGP_TRY(scan_exclusive_u64(c, K.as<uint64_t>(), IDX.as<uint64_t>(), n, &n_new));
        BGP_TRY(scan_exclusive_u64(c, LS.as<uint64_t>(), SO.as<uint64_t>(), n, &sl));
        BGP_TRY(scan_exclusive_u64(c, NC.as<uint64_t>(), CO.as<uint64_t>(), n, &sc));
        hipLaunchKernelGGL(k_region_scatter, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, c->stream, c->d_rec.as<bgp_record>(), n, K.as<uint64_t>(), IDX.as<uint64_t>(),
                           SO.as<uint64_t>(), CO.as<uint64_t>(), c->d_rec2.as<bgp_record>(), c->d_seq_off2.as<uint64_t>(), c->d_cigar_off2.as<uint64_t>());
        BGP_HIP(hipGetLastError());
        BGP_HIP(hipStreamSynchronize(c->stream));
    } else {
        BGP_HIP(hipMemsetAsync(c->d_seq_off2.p, 0, 8, c->stream)); BGP_HIP(hipMemsetAsync(c->d_cigar_off2.p, 0, 8, c->stream));
    }
    std::swap(c->d_rec, c->d_rec2); std::swap(c->d_seq_off, c->d_seq_off2); std::swap(c->d_cigar_off, c->d_cigar_off2);
    c->n_records = n_new; c->sum_lseq = sl; c->sum_ncig = sc;
    c->table_compacted = true;
    c->cig4_ok = false; c->rows_ok = false; c->lds_table = false; c->tilemax_ok = false;                 // the side array of first CIGAR ops is not compacted: K4's planner takes the ops from the stream
    c->piled = false; c->md_rows_ready = false; c->decoded_what = 0; c->it = bgp_ctx::Iter();
    return BGP_OK;
}

// ---- makePileup over a FILTERED range of reads: reads.filter!(pred) in front of the pileup (pileup.d:683-687 takes any input range of reads; the
// predicates callers write are the flag properties of read.d:137-194 -- is_duplicate, failed_quality_control, is_secondary_alignment ... -- and
// mapping_quality).  A read the filter drops is not in the range: it takes part in no column, holds no entry, provides no reference base.  Here
// the record table is marked instead of compacted: such a read "covers nothing" (end_position = position, the reads pileupInstance filters out
// itself, pileup.d:481,510), its plan row and class are cleared -- every kernel behind the table then leaves it out, and read ordinals stay the stream's.
namespace {
__global__ __launch_bounds__(256) void k_read_filter(bgp_record* rec, uint64_t n, uint32_t require, uint32_t exclude, uint32_t min_mapq, uint4* prow, uint8_t* rcls) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t bmn = rec[i].bin_mq_nl, fnc = rec[i].flag_nc;
    const uint32_t flag = fnc >> 16, mq = (bmn >> 8) & 0xFFu;
    if ((flag & require) == require && !(flag & exclude) && mq >= min_mapq) return;
    rec[i].end_pos = rec[i].pos;
    if (prow) { prow[i] = make_uint4(0u, 0u, 0u, 0u); rcls[i] = 0; }
}
}  // namespace
int read_filter_apply(bgp_ctx* c) {
    const uint64_t n = c->n_records;
    if (!c->flt_on || !n) return BGP_OK;
    TimedRegion t(c, "k_read_filter", n * 12);
    hipLaunchKernelGGL(k_read_filter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->d_rec.as<bgp_record>(), n, c->flt_require, c->flt_exclude, c->flt_min_mapq,
                       c->rows_ok ? c->d_prow.as<uint4>() : nullptr, c->rows_ok ? c->d_rcls.as<uint8_t>() : nullptr);
    BGP_HIP(hipGetLastError());
    c->tilemax_ok = false;                               // (K3 takes its tiles' maxima from the table as it is now)
    c->flt_applied = true;
    return BGP_OK;
}
