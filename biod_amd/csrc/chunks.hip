// chunks.hip -- pileupChunks: the chunk walk of PileupChunkRange over the record table.
//
// Reference semantics.  chunksConsumingLessThan(reads, block_size) (bio/std/hts/bam/splitter.d:65-91): a chunk is its
// first read plus the following reads of the same ref_id, one after the other while the bytes taken so far
// (size_in_bytes = 4 + block_size of a record, read.d:609-611) do not exceed block_size -- the read that crosses the
// limit is still taken.  PileupChunkRange (pileup.d:859-987) skips chunks whose first read has ref_id < 0, clips chunk
// j's pileup to [beg, end): beg = position of the previous chunk's last read when that chunk is of the same reference,
// else of the chunk's own first read (:932-939); end = position of the chunk's last read, or the chunk's right end when
// the reference ends with it (:905-908).  Its constructor (:878-900) drops leading chunks that end at or before
// start_from and is empty when the first kept chunk begins at or after end_at.
//
// Two stated deviations, the same the oracle makes (oracle/biod_oracle.c: chunk_bounds): the reads a chunk's columns are
// counted from are ALL reads overlapping them (exact halo) where BioD keeps those of the previous chunk that start
// within 2 x a sampled median read length (pileup.d:944-985, randomSample: not even deterministic); and the last chunk
// of a reference runs to the maximum end_position of the reference so far, not of that chunk alone.  With them the
// chunks' columns concatenate to exactly pileupColumns' (tested).
//
// MI355X: record sizes are differences of rec_off, so "the first read that no longer fits" is a binary search, and the
// walk is ~ stream bytes / block_size dependent steps -- one thread; the chunks' right ends are one block reduction per
// chunk over the table (coalesced, one pass).  The columns themselves already exist (K4): a chunk is a view.
#include "common.h"

#include <algorithm>

namespace {

struct ChunkDev { uint64_t first, last; int32_t ref_id, pos_first, pos_last, right_end; uint64_t pmax_key; };   // pmax_key: K3's running maximum of (reference key << 32 | end_position) at the chunk's last read

__global__ __launch_bounds__(256) void k_ref_changes(const bgp_record* rec, uint64_t n, uint64_t* list, uint32_t* count, uint32_t cap) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (i >= n) return;
    if (rec[i].ref_id != rec[i - 1].ref_id) { const uint32_t s = atomicAdd(count, 1u); if (s < cap) list[s] = i; }
}

// one thread: getNextChunk after getNextChunk (splitter.d:65-91)
// `size_before`: when the record table was compacted (bam[ref][beg..end]: the reads the filter dropped still lie between
// the kept ones in the stream) the bytes of the reads in front of read i, as chunksConsumingLessThan adds them up over
// the range it is given; null: the table is the whole stream and rec_off is that sum
__global__ void k_chunk_cut(const bgp_record* rec, uint64_t n, uint64_t stream_len, const uint64_t* changes, uint32_t n_changes, uint64_t block_size,
                            ChunkDev* out, uint32_t cap, uint32_t* n_out, const uint64_t* size_before) {
    uint64_t f = 0; uint32_t k = 0, ci = 0;
    while (f < n) {
        while (ci < n_changes && changes[ci] <= f) ++ci;
        const uint64_t ref_end = ci < n_changes ? changes[ci] : n;             // first read of another ref_id
        // read i (> f) is taken iff the bytes of reads f .. i-1 do not exceed block_size: rec_off[i] - rec_off[f] <= block_size
        const uint64_t base = size_before ? size_before[f] : rec[f].rec_off;
        const uint64_t limit = block_size > ~0ull - base ? ~0ull : base + block_size;
        uint64_t lo = f + 1, hi = ref_end;                                      // first i in [lo, hi) with rec_off[i] > limit
        while (lo < hi) { const uint64_t m = (lo + hi) >> 1; if ((size_before ? size_before[m] : rec[m].rec_off) > limit) hi = m; else lo = m + 1; }
        if (k < cap) { ChunkDev c{}; c.first = f; c.last = lo - 1; c.ref_id = rec[f].ref_id; c.pos_first = rec[f].pos; c.pos_last = rec[lo - 1].pos; c.right_end = INT32_MIN; out[k] = c; }
        ++k;
        f = lo;
    }
    *n_out = k;
    (void)stream_len;
}

// computeRightEnd (pileup.d:870-872): max over the chunk's reads of position + basesCovered
__global__ __launch_bounds__(256) void k_record_sizes(const uint8_t* s, const bgp_record* rec, uint64_t n, uint64_t* size) {     // size_in_bytes, read.d:609-611
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) size[i] = 4ull + ld_u32(s + rec[i].rec_off);
}
__global__ __launch_bounds__(256) void k_chunk_right_end(const bgp_record* rec, ChunkDev* chunks, const uint64_t* pmax_end) {
    __shared__ int32_t ws[4];
    ChunkDev& c = chunks[blockIdx.x];
    if (threadIdx.x == 0) c.pmax_key = pmax_end[c.last];
    int32_t m = INT32_MIN;
    for (uint64_t i = c.first + threadIdx.x; i <= c.last; i += blockDim.x) {
        const uint4 x = *reinterpret_cast<const uint4*>(rec + i);               // ref_id, pos, end_pos, bin_mq_nl
        m = max(m, (int32_t)x.z);
    }
    for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_xor(m, d, 64));
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) c.right_end = max(max(ws[0], ws[1]), max(ws[2], ws[3]));
}

}  // namespace

int chunks_plan(bgp_ctx* c, uint64_t block_size, uint64_t start_from, uint64_t end_at) {
    c->chunks.clear();
    const uint64_t n = c->n_records;
    if (n == 0) return BGP_OK;
    if (block_size == 0) block_size = 16384000;                                // pileup.d:1011
    const bgp_record* rec = c->d_rec.as<bgp_record>();
    const uint32_t ch_cap = 1u << 20;
    BGP_TRY(c->d_flt[0].reserve((size_t)ch_cap * 8 + 64));
    uint64_t* d_changes = c->d_flt[0].as<uint64_t>();
    uint32_t* d_cnt = reinterpret_cast<uint32_t*>(d_changes + ch_cap);
    BGP_HIP(hipMemsetAsync(d_cnt, 0, 8, c->stream));
    hipLaunchKernelGGL(k_ref_changes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, rec, n, d_changes, d_cnt, ch_cap);
    uint32_t n_changes = 0;
    BGP_HIP(hipMemcpyAsync(&n_changes, d_cnt, 4, hipMemcpyDeviceToHost, c->stream));
    BGP_HIP(hipStreamSynchronize(c->stream));
    if (n_changes > ch_cap) { c->last_error = "pileupChunks: more than 2^20 changes of ref_id along the reads"; return BGP_EINVAL; }
    std::vector<uint64_t> changes(n_changes);
    if (n_changes) {
        BGP_HIP(hipMemcpyAsync(changes.data(), d_changes, (size_t)n_changes * 8, hipMemcpyDeviceToHost, c->stream));
        BGP_HIP(hipStreamSynchronize(c->stream));
        std::sort(changes.begin(), changes.end());
        BGP_HIP(hipMemcpyAsync(d_changes, changes.data(), (size_t)n_changes * 8, hipMemcpyHostToDevice, c->stream));
        BGP_HIP(hipStreamSynchronize(c->stream));
    }
    const uint64_t cap64 = c->stream_len / block_size + n_changes + 2;
    if (cap64 > (1ull << 22)) { c->last_error = "pileupChunks: block_size too small for this stream (more than 4 M chunks)"; return BGP_EINVAL; }   // the cut is a serial walk: a chunk is meant to be megabytes
    const uint64_t* d_size_before = nullptr;
    if (c->table_compacted) {                                                  // chunks of bam[ref][beg..end]: the sizes of the reads that are there
        BGP_TRY(c->d_flt[2].reserve((n + 2) * 8)); BGP_TRY(c->d_flt[3].reserve((n + 2) * 8));
        hipLaunchKernelGGL(k_record_sizes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->d_stream, rec, n, c->d_flt[2].as<uint64_t>());
        BGP_TRY(scan_exclusive_u64(c, c->d_flt[2].as<uint64_t>(), c->d_flt[3].as<uint64_t>(), n, nullptr));
        d_size_before = c->d_flt[3].as<uint64_t>();
    }
    const uint32_t cap = (uint32_t)cap64;
    BGP_TRY(c->d_flt[1].reserve((size_t)cap * sizeof(ChunkDev) + 64));
    ChunkDev* d_chunks = c->d_flt[1].as<ChunkDev>();
    hipLaunchKernelGGL(k_chunk_cut, dim3(1), dim3(1), 0, c->stream, rec, n, c->stream_len, d_changes, n_changes, block_size, d_chunks, cap, d_cnt + 1, d_size_before);
    uint32_t n_chunks = 0;
    BGP_HIP(hipMemcpyAsync(&n_chunks, d_cnt + 1, 4, hipMemcpyDeviceToHost, c->stream));
    BGP_HIP(hipStreamSynchronize(c->stream));
    if (n_chunks > cap) { c->last_error = "internal: chunk table overflow"; return BGP_ESTATE; }
    if (n_chunks) hipLaunchKernelGGL(k_chunk_right_end, dim3(n_chunks), dim3(256), 0, c->stream, rec, d_chunks, c->d_pmax_end.as<uint64_t>());
    BGP_HIP(hipGetLastError());
    std::vector<ChunkDev> raw(n_chunks);
    if (n_chunks) { BGP_HIP(hipMemcpyAsync(raw.data(), d_chunks, (size_t)n_chunks * sizeof(ChunkDev), hipMemcpyDeviceToHost, c->stream)); BGP_HIP(hipStreamSynchronize(c->stream)); }      // (on the kernels' stream: it is a non-blocking one, a plain hipMemcpy would not wait for k_chunk_right_end)

    // the maximum end_position of the reference so far (covering reads): the runs' prefix maxima K3 left in HBM, picked up
    // by k_chunk_right_end with the chunks (one copy for all of them)
    std::vector<int64_t> pmax(n_chunks, INT64_MIN);
    for (uint32_t j = 0; j < n_chunks; ++j) {
        const uint64_t rk = raw[j].ref_id < 0 ? 0x7FFFFFFFull : (uint64_t)(uint32_t)(raw[j].ref_id + 1);
        if ((raw[j].pmax_key >> 32) == rk) pmax[j] = (int64_t)(uint32_t)raw[j].pmax_key;
    }
    std::vector<int32_t> prev_mapped(n_chunks, -1), next_mapped(n_chunks, -1);    // nearest chunk of mapped reads on either side
    for (uint32_t j = 1; j < n_chunks; ++j) prev_mapped[j] = raw[j - 1].ref_id >= 0 ? (int32_t)(j - 1) : prev_mapped[j - 1];
    for (int64_t j = (int64_t)n_chunks - 2; j >= 0; --j) next_mapped[j] = raw[j + 1].ref_id >= 0 ? (int32_t)(j + 1) : next_mapped[j + 1];
    // PileupChunkRange: constructor (pileup.d:878-900), front (:903-913), popFront (:915-939)
    bool started = false;
    for (uint32_t j = 0; j < n_chunks; ++j) {
        const ChunkDev& ch = raw[j];
        if (ch.ref_id < 0) continue;                                           // :886, :926
        int64_t right_end = std::max<int64_t>(ch.right_end, pmax[j]);
        int32_t prev = prev_mapped[j]; const int32_t next = next_mapped[j];
        if (!started) {
            const uint64_t beg0 = (uint64_t)(int64_t)ch.pos_first;
            if (beg0 >= end_at) break;                                         // :889-892: the range is empty
            if (!((uint64_t)right_end > start_from)) continue;                 // :895: _chunk_right_end > start_from (D widens the int to ulong, sign and all)
            started = true;
            prev = -1;                                                         // _prev_chunk is empty for the first chunk kept
        }
        const bool same_prev = prev >= 0 && raw[prev].ref_id == ch.ref_id;
        const uint64_t beg = same_prev ? (uint64_t)(int64_t)raw[prev].pos_last : (uint64_t)(int64_t)ch.pos_first;
        uint64_t end = (uint64_t)(int64_t)ch.pos_last;
        // front(): _chunks.front is the very next chunk of the splitter, unmapped ones included; the oracle looks at the next mapped one
        if (next < 0 || raw[next].ref_id != ch.ref_id) end = (uint64_t)right_end;
        bgp_chunk_info ci{};
        ci.ref_id = ch.ref_id; ci.first_record = ch.first; ci.last_record = ch.last;
        ci.start_position = std::max(beg, start_from); ci.end_position = std::min(end, end_at);
        c->chunks.push_back(ci);
    }
    return BGP_OK;
}
