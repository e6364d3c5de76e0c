// bai_build.hip -- createIndex: the .bai of the staged file from the device record table.
//
// Reference semantics: IndexBuilder.put / finish (bio/std/hts/bam/bai/indexing.d:56-350) fed by
// readsWithProgress!withOffsets (indexing.d:353-366, readrange.d:51-68).  The builder is a sequential state machine
// over reads; its output has a closed form over the record table, which is what the kernels evaluate:
//
//   valid read        ref_id >= 0 and position >= 0                          (put returns early otherwise, :289-294)
//   sorted check      every read with ref_id != -1 against the last valid read before it            (:253-268)
//   bin run           maximal stretch of valid reads with one (ref_id, bin) -- updateChunks fires when either changes
//                     (:309-323,338-341).  Its chunk is [end offset of the last valid read before the run (start
//                     offset of the very first valid read), end offset of the run's last read)       (:221-246)
//   chunk merge       inside one bin, a chunk joins its predecessor when that one ends in the BGZF block this one
//                     begins in                                                                       (:230-244)
//   linear index      window w of a reference = start offset of the first valid read covering it; a read covers
//                     position/16384 .. (position + basesCovered - 1)/16384, one window when unmapped; holes are
//                     filled from the left when written                                          (:51-53,134-185)
//   pseudo-bin 37450  per written reference: [beg, end) offsets and mapped / unmapped counts over the reads with
//                     ref_id != -1 that arrive while that reference is current -- a read the builder skips
//                     counts towards the last valid read's reference (the first valid one if none came yet)
//                                                                                            (:118-132,197-217)
//   n_no_coor         reads with ref_id == -1                                                       (:119,347)
// Virtual offsets: a stream position inside block b is (coffset(b), offset in b); on a block boundary it belongs
// to the NEXT block, and past the last data block it is (end of that block, 0) (bgzf/inputstream.d:425-452,510-527).
//
// Device: validity + last-valid-before (max-scan), run heads (sum-scan), one 16-byte descriptor per run, the
// per-reference counters and the linear index by atomicMin.  Host: runs -> bins (stable counting sort per
// reference), the merge rule, serialisation.  Bins are written in ascending id order; the reference writes them
// in druntime's associative-array order (:193), which nothing pins.
#include "common.h"

#include <algorithm>

namespace {

struct VSeg { uint64_t staged, vo0; };                 // stream bytes from `staged` on have virtual offset vo0 + (u - staged)

struct BaiArgs {
    const bgp_record* rec; uint64_t n, stream_len;
    const VSeg* seg; uint32_t n_seg;                   // sorted by staged; the last entry is the end sentinel
    int32_t n_ref;
    uint64_t* lastv;                                   // max-scan of (valid ? i+1 : 0): last valid index + 1 at or before i
    uint64_t* head; const uint64_t* ord;               // run heads and their exclusive scan
    unsigned long long* mapped; unsigned long long* unmapped; unsigned long long* beg_vo; unsigned long long* end_vo;   // n_ref + 1 each (slot n_ref: before any valid read)
    unsigned int* wlen;                                // n_ref: _linear_index_write_length
    unsigned long long* misc;                          // [0] first unsorted record, [1] no_coord, [2] first offset of a read with ref_id != -1, [3] error bits, [4] end offset of the last valid read
    const uint64_t* lin_off; unsigned long long* lin;  // per-reference base into lin
    uint4* runs;                                       // {ref_id, bin, beg lo, beg hi}
};

constexpr uint32_t LIN_SIZE = 37449 - 4680 + 1;        // indexing.d:273

__device__ inline uint64_t vo_of(const BaiArgs& a, uint64_t u) {
    uint32_t lo = 0, hi = a.n_seg;                     // last segment with staged <= u
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (a.seg[mid].staged <= u) lo = mid; else hi = mid; }
    return a.seg[lo].vo0 + (u - a.seg[lo].staged);
}
__device__ inline uint64_t end_off(const BaiArgs& a, uint64_t i) { return i + 1 < a.n ? a.rec[i + 1].rec_off : a.stream_len; }
__device__ inline bool valid(const bgp_record& r) { return r.ref_id >= 0 && r.pos >= 0; }
__device__ inline uint32_t to_lin(int32_t p) { return p < 0 ? 0u : (uint32_t)p / 16384u; }

__global__ __launch_bounds__(256) void k_bai_valid(BaiArgs a) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n) a.head[i] = valid(a.rec[i]) ? i + 1 : 0;
}

__global__ __launch_bounds__(256) void k_bai_heads(BaiArgs a) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < a.n;
    bgp_record r{}; bgp_record p{}; uint64_t pv = 0;
    if (in) { r = a.rec[i]; pv = i ? a.lastv[i - 1] : 0; if (pv) p = a.rec[pv - 1]; }
    const bool v = in && valid(r), coord = in && r.ref_id != -1;
    if (coord && pv && !(p.ref_id < r.ref_id) && !(r.ref_id == p.ref_id && r.pos >= p.pos)) atomicMin(&a.misc[0], (unsigned long long)i);
    if (v && r.ref_id >= a.n_ref) atomicOr(&a.misc[3], 1ull);
    // metadata: wave-aggregated when the whole wave counts towards one reference
    const int32_t g = !coord ? -1 : v ? r.ref_id : pv ? p.ref_id : a.n_ref;
    const bool unm = (r.flag_nc >> 16) & 4u;
    const uint64_t mc = __ballot(coord), mn = __ballot(in && !coord), mu = __ballot(coord && unm);
    const int lane = threadIdx.x & 63;
    if (mn && lane == 0) atomicAdd(&a.misc[1], (unsigned long long)__popcll(mn));
    // one reference per wave is the rule: then one atomic per wave and counter (offsets grow with i, so the last
    // lane holds the wave's end offset); lanes go one by one only in a wave that straddles references
    uint32_t e1 = 0;                                       // end window + 1 of a valid read
    if (v && r.ref_id < a.n_ref) {
        const uint32_t b = to_lin(r.pos), e = unm ? b : to_lin(r.pos + (r.end_pos - r.pos) - 1);
        if (e >= LIN_SIZE) atomicOr(&a.misc[3], 2ull);                                          // a D RangeError (:154 or :169)
        else e1 = e + 1;
        if (pv && p.ref_id != r.ref_id) a.beg_vo[r.ref_id] = vo_of(a, end_off(a, pv - 1));    // dumpCurrentReference, :206-210
    }
    if (mc) {
        const int lead = __ffsll((unsigned long long)mc) - 1, tail = 63 - __clzll((unsigned long long)mc);
        const int32_t g0 = __shfl(g, lead);
        const bool uniform = __ballot(coord && g != g0) == 0;
        const bool ok = g >= 0 && g <= a.n_ref && (g < a.n_ref || !v);
        if (uniform) {
            uint32_t m1 = e1;
#pragma unroll
            for (int d = 32; d; d >>= 1) m1 = max(m1, (uint32_t)__shfl_xor((int)m1, d));
            if (lane == lead && ok) {
                atomicAdd(&a.mapped[g], (unsigned long long)__popcll(mc & ~mu)); atomicAdd(&a.unmapped[g], (unsigned long long)__popcll(mu));
                if (m1 && g < a.n_ref) atomicMax(&a.wlen[g], m1);
            }
            if (lane == tail && ok) atomicMax(&a.end_vo[g], (unsigned long long)vo_of(a, end_off(a, i)));
        } else if (coord && ok) {
            atomicAdd(unm ? &a.unmapped[g] : &a.mapped[g], 1ull);
            atomicMax(&a.end_vo[g], (unsigned long long)vo_of(a, end_off(a, i)));
            if (e1) atomicMax(&a.wlen[r.ref_id], e1);
        }
        if (lane == lead) atomicMin(&a.misc[2], (unsigned long long)vo_of(a, r.rec_off));      // the first such read of the file
    }
    if (in) a.head[i] = v && (!pv || p.ref_id != r.ref_id || (p.bin_mq_nl >> 16) != (r.bin_mq_nl >> 16));
    if (in && i == a.n - 1) { const uint64_t lv = a.lastv[i]; a.misc[4] = lv ? vo_of(a, end_off(a, lv - 1)) : 0; }
}

__global__ __launch_bounds__(256) void k_bai_runs(BaiArgs a) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const bgp_record r = a.rec[i];
    if (!valid(r)) return;
    const uint64_t start = vo_of(a, r.rec_off);
    if (a.head[i]) {
        const uint64_t pv = i ? a.lastv[i - 1] : 0;
        const uint64_t beg = pv ? vo_of(a, end_off(a, pv - 1)) : start;
        a.runs[a.ord[i]] = make_uint4((uint32_t)r.ref_id, r.bin_mq_nl >> 16, (uint32_t)beg, (uint32_t)(beg >> 32));
    }
    const bool unm = (r.flag_nc >> 16) & 4u;
    const uint32_t b = to_lin(r.pos), e = unm ? b : to_lin(r.pos + (r.end_pos - r.pos) - 1);
    unsigned long long* lin = a.lin + a.lin_off[r.ref_id];
    uint32_t pb = 1, pe = 0;                               // windows of the record right before, when it is a valid read of this reference
    if (i) {
        const bgp_record q = a.rec[i - 1];
        if (valid(q) && q.ref_id == r.ref_id) { pb = to_lin(q.pos); pe = ((q.flag_nc >> 16) & 4u) ? pb : to_lin(q.pos + (q.end_pos - q.pos) - 1); }
    }
    for (uint32_t w = b; w <= e && w != ~0u; ++w)          // first read to touch the window (:154-156); an earlier toucher next door wins anyway
        if (w < pb || w > pe) atomicMin(&lin[w], (unsigned long long)start);
}

template <class T> void put(std::vector<uint8_t>& o, T v) { const size_t at = o.size(); o.resize(at + sizeof v); memcpy(o.data() + at, &v, sizeof v); }

}  // namespace

int bai_build(bgp_ctx* c, std::vector<uint8_t>& out) {
    if (!c->bai_buildable) { c->last_error = "createIndex needs the records of one whole file (bgp_load_bam, then bgp_flush)"; return BGP_ESTATE; }
    const uint64_t n = c->n_records;
    const int32_t n_ref = (int32_t)c->refs.size();
    std::vector<VSeg> segs;
    for (const auto& s : c->stage_map) if (s.len) segs.push_back(VSeg{s.staged, (s.coff << 16) | s.skip});
    segs.push_back(VSeg{c->stream_len, c->stage_end_coff << 16});
    const size_t per = (size_t)n_ref + 1;
    DevBuf& LV = c->d_flt[0]; DevBuf& HD = c->d_flt[1]; DevBuf& OR = c->d_flt[2]; DevBuf& SG = c->d_flt[3]; DevBuf& CT = c->d_flt[4]; DevBuf& LN = c->d_flt[5];
    BGP_TRY(LV.reserve((n + 2) * 8)); BGP_TRY(HD.reserve((n + 2) * 8)); BGP_TRY(OR.reserve((n + 2) * 8));
    BGP_TRY(SG.reserve(segs.size() * sizeof(VSeg)));
    const size_t ct_bytes = (4 * per + 8) * 8 + per * 4;
    BGP_TRY(CT.reserve(ct_bytes));
    BGP_HIP(hipMemcpyAsync(SG.p, segs.data(), segs.size() * sizeof(VSeg), hipMemcpyHostToDevice, c->stream));
    BGP_HIP(hipMemsetAsync(CT.p, 0, ct_bytes, c->stream));
    BaiArgs a{};
    a.rec = c->d_rec.as<bgp_record>(); a.n = n; a.stream_len = c->stream_len; a.seg = SG.as<VSeg>(); a.n_seg = (uint32_t)segs.size(); a.n_ref = n_ref;
    a.lastv = LV.as<uint64_t>(); a.head = HD.as<uint64_t>(); a.ord = OR.as<uint64_t>();
    unsigned long long* ct = CT.as<unsigned long long>();
    a.mapped = ct; a.unmapped = ct + per; a.beg_vo = ct + 2 * per; a.end_vo = ct + 3 * per; a.misc = ct + 4 * per; a.wlen = (unsigned int*)(ct + 4 * per + 8);
    const unsigned long long none = ~0ull;
    BGP_HIP(hipMemcpyAsync(&a.misc[0], &none, 8, hipMemcpyHostToDevice, c->stream));
    BGP_HIP(hipMemcpyAsync(&a.misc[2], &none, 8, hipMemcpyHostToDevice, c->stream));
    std::vector<unsigned long long> h_ct(4 * per + 8); std::vector<unsigned int> h_wlen(per);
    std::vector<uint4> runs; std::vector<unsigned long long> lin; std::vector<uint64_t> lin_off(per + 1, 0);
    uint64_t n_runs = 0;
    if (n) {
        const unsigned nb = (unsigned)((n + 255) / 256);
        TimedRegion t(c, "k_bai_build", n * 32);
        hipLaunchKernelGGL(k_bai_valid, dim3(nb), dim3(256), 0, c->stream, a);
        BGP_HIP(hipGetLastError());
        BGP_TRY(scan_inclusive_max_u64(c, a.head, a.lastv, n));
        hipLaunchKernelGGL(k_bai_heads, dim3(nb), dim3(256), 0, c->stream, a);
        BGP_HIP(hipGetLastError());
        BGP_TRY(scan_exclusive_u64(c, a.head, OR.as<uint64_t>(), n, &n_runs));
        BGP_HIP(hipMemcpyAsync(h_ct.data(), ct, h_ct.size() * 8, hipMemcpyDeviceToHost, c->stream));
        BGP_HIP(hipMemcpyAsync(h_wlen.data(), a.wlen, per * 4, hipMemcpyDeviceToHost, c->stream));
        BGP_HIP(hipStreamSynchronize(c->stream));
        const unsigned long long* misc = h_ct.data() + 4 * per;
        if (misc[0] != none) { c->last_error = "BAM file is not coordinate-sorted (record " + std::to_string(misc[0]) + ")"; return BGP_EUNSORTED; }
        if (misc[3] & 1) { c->last_error = "a read refers to a reference the header does not list"; return BGP_EINVAL; }
        if (misc[3] & 2) { c->last_error = "a read lies beyond the 2^29 coordinate range of the BAI linear index"; return BGP_EINVAL; }
        for (int32_t r = 0; r < n_ref; ++r) lin_off[r + 1] = lin_off[r] + h_wlen[r];
        const uint64_t n_lin = lin_off[n_ref];
        BGP_TRY(LN.reserve((n_lin + per + 1) * 8 + (n_runs + 1) * sizeof(uint4) + 16));
        a.lin = LN.as<unsigned long long>(); a.lin_off = (const uint64_t*)(a.lin + n_lin);
        a.runs = (uint4*)(((uintptr_t)(a.lin + n_lin + per + 1) + 15) & ~(uintptr_t)15);
        BGP_HIP(hipMemsetAsync(a.lin, 0xFF, n_lin * 8, c->stream));
        BGP_HIP(hipMemcpyAsync((void*)a.lin_off, lin_off.data(), (per + 1) * 8, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_bai_runs, dim3(nb), dim3(256), 0, c->stream, a);
        BGP_HIP(hipGetLastError());
        runs.resize(n_runs); lin.resize(n_lin);
        if (n_runs) BGP_HIP(hipMemcpyAsync(runs.data(), a.runs, n_runs * sizeof(uint4), hipMemcpyDeviceToHost, c->stream));
        if (n_lin) BGP_HIP(hipMemcpyAsync(lin.data(), a.lin, n_lin * 8, hipMemcpyDeviceToHost, c->stream));
        BGP_HIP(hipStreamSynchronize(c->stream));
    } else {
        BGP_HIP(hipStreamSynchronize(c->stream));
    }
    const unsigned long long* mapped = h_ct.data(); const unsigned long long* unmapped = mapped + per;
    const unsigned long long* beg_vo = mapped + 2 * per; const unsigned long long* end_vo = mapped + 3 * per; const unsigned long long* misc = mapped + 4 * per;

    out.clear();
    out.insert(out.end(), {'B', 'A', 'I', 1});
    put<int32_t>(out, n_ref);
    std::vector<uint32_t> count(65537), order;
    struct Ch { uint64_t beg, end; };
    std::vector<Ch> chunks;
    size_t at = 0; bool first_written = true;
    for (int32_t r = 0; r < n_ref; ++r) {
        size_t to = at;
        while (to < runs.size() && (int32_t)runs[to].x == r) ++to;
        if (to == at) { put<int32_t>(out, 0); put<int32_t>(out, 0); continue; }             // writeEmptyReference, :114-117
        std::fill(count.begin(), count.end(), 0u);                                          // stable counting sort of the runs by bin
        for (size_t k = at; k < to; ++k) ++count[runs[k].y + 1];
        uint32_t n_bins = 0;
        for (uint32_t b = 0; b < 65536; ++b) { n_bins += count[b + 1] != 0; count[b + 1] += count[b]; }
        order.assign(to - at, 0);
        { std::vector<uint32_t> fill(count.begin(), count.end() - 1); for (size_t k = at; k < to; ++k) order[fill[runs[k].y]++] = (uint32_t)(k - at); }
        put<int32_t>(out, (int32_t)(n_bins + 1));
        for (uint32_t b = 0; b < 65536; ++b) {
            if (count[b] == count[b + 1]) continue;
            chunks.clear();
            for (uint32_t q = count[b]; q < count[b + 1]; ++q) {
                const size_t k = at + order[q];
                const uint64_t beg = (uint64_t)runs[k].z | ((uint64_t)runs[k].w << 32);
                const uint64_t end = k + 1 < runs.size() ? ((uint64_t)runs[k + 1].z | ((uint64_t)runs[k + 1].w << 32)) : misc[4];
                if (!chunks.empty() && (chunks.back().end >> 16) == (beg >> 16)) chunks.back().end = end;   // :230-244
                else chunks.push_back(Ch{beg, end});
            }
            put<uint32_t>(out, b); put<int32_t>(out, (int32_t)chunks.size());
            for (const Ch& ch : chunks) { put<uint64_t>(out, ch.beg); put<uint64_t>(out, ch.end); }
        }
        put<uint32_t>(out, 37450u); put<int32_t>(out, 2);                                   // :197-202
        put<uint64_t>(out, first_written ? misc[2] : beg_vo[r]);
        put<uint64_t>(out, first_written ? std::max(end_vo[r], end_vo[n_ref]) : end_vo[r]);
        put<uint64_t>(out, mapped[r] + (first_written ? mapped[n_ref] : 0));
        put<uint64_t>(out, unmapped[r] + (first_written ? unmapped[n_ref] : 0));
        first_written = false;
        put<int32_t>(out, (int32_t)h_wlen[r]);                                              // dumpCurrentLinearIndex, :164-185
        uint64_t last = 0;
        for (uint32_t w = 0; w < h_wlen[r]; ++w) {
            uint64_t v = lin[lin_off[r] + w];
            if (v == none) v = last; else last = v;
            put<uint64_t>(out, v);
        }
        at = to;
    }
    put<uint64_t>(out, misc[1]);
    return BGP_OK;
}
