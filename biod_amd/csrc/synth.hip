// synth.hip -- synthetic coordinate-sorted BAM record streams, on the host and straight into HBM.
// Both paths instantiate the same functions from synth_record.h, so they agree byte for byte.
// Bench/test support (generator G1); not part of the timed hot path.
#include "common.h"
#include "synth_record.h"

namespace {

bgp_synth_cfg to_cfg(const bgp_synth_cfg_c* c) {
    bgp_synth_cfg g;
    g.seed = c->seed; g.n_reads = c->n_reads; g.n_ref = c->n_ref ? c->n_ref : 1; g.ref_len = c->ref_len;
    g.len_min = c->len_min; g.len_max = c->len_max < c->len_min ? c->len_min : c->len_max; g.profile = c->profile; g.first_ref = c->first_ref;
    return g;
}

bool cfg_ok(const bgp_synth_cfg& g) {
    if (g.profile > 2 || g.len_min == 0 || g.n_reads == 0) return false;
    if (g.profile == 1 && g.len_min < 40) return false;
    if (g.profile == 2 && g.len_min < 64) return false;
    return (uint64_t)g.ref_len > (uint64_t)synth::max_span(g) + 1;
}

__global__ __launch_bounds__(256) void k_sizes(bgp_synth_cfg g, uint64_t* sizes) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.n_reads) return;
    const synth::Read r = synth::make_read(g, i);
    sizes[i] = synth::record_shape(g, r).size;
}

__global__ __launch_bounds__(256) void k_write(bgp_synth_cfg g, const uint64_t* offs, uint8_t* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.n_reads) return;
    const synth::Read r = synth::make_read(g, i);
    const synth::Shape sh = synth::record_shape(g, r);
    synth::ByteSink s; s.p = out + offs[i]; s.n = 0;
    synth::emit_record(g, i, r, sh.size - 4, sh.n_cigar, sh.ref_span, s);
}

}  // namespace

int synth_host(const bgp_synth_cfg_c* cfg, uint64_t first, uint64_t count, uint8_t* out, uint64_t cap, uint64_t* bytes) {
    const bgp_synth_cfg g = to_cfg(cfg);
    if (!cfg_ok(g) || first > g.n_reads || count > g.n_reads - first) return BGP_EINVAL;
    uint64_t off = 0;
    for (uint64_t i = first; i < first + count; ++i) {
        const synth::Read r = synth::make_read(g, i);
        const synth::Shape sh = synth::record_shape(g, r);
        if (out) {
            if (off + sh.size > cap) return BGP_ENOMEM;
            synth::ByteSink s; s.p = out + off; s.n = 0;
            synth::emit_record(g, i, r, sh.size - 4, sh.n_cigar, sh.ref_span, s);
            if (s.n != sh.size) return BGP_EINVAL;
        }
        off += sh.size;
    }
    *bytes = off;
    return BGP_OK;
}

int synth_device(bgp_ctx* c, const bgp_synth_cfg_c* cfg, uint64_t* bytes) {
    const bgp_synth_cfg g = to_cfg(cfg);
    if (!cfg_ok(g)) return BGP_EINVAL;
    const uint64_t n = g.n_reads;
    BGP_TRY(c->d_pmax_end.reserve((n + 1) * 8));
    BGP_TRY(c->d_pmax_pos.reserve((n + 1) * 8));
    uint64_t* sizes = c->d_pmax_end.as<uint64_t>();
    uint64_t* offs = c->d_pmax_pos.as<uint64_t>();
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_sizes, dim3(nb), dim3(256), 0, c->stream, g, sizes);
    uint64_t total = 0;
    BGP_TRY(scan_exclusive_u64(c, sizes, offs, n, &total));
    BGP_TRY(c->d_synth.reserve(total + 64));
    hipLaunchKernelGGL(k_write, dim3(nb), dim3(256), 0, c->stream, g, offs, c->d_synth.as<uint8_t>());
    BGP_HIP(hipGetLastError());
    BGP_HIP(hipStreamSynchronize(c->stream));
    c->d_stream = c->d_synth.as<uint8_t>(); c->stream_len = total; c->stream_on_device = true;
    if (bytes) *bytes = total;
    return BGP_OK;
}
