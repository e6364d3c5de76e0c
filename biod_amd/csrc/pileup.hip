// pileup.hip -- kernels K3 (reference runs, halo search keys), K4 (per-position column summary)
// and K5 (per-column read entries for a window).
//
// Reference semantics (bio/std/hts/bam/pileup.d).  PileupRange.popFront (:345-397) sweeps one
// reference position at a time keeping the reads with position <= pos < end_position in file order;
// PileupRead (:175-222) steps a per-read cursor (cigar op index, offset in op, query offset).  Both
// are serial only in form: the state of read R at column p is a closed function of (R, p - R.position):
//   * ops before the first reference-consuming non-N op only add their query length (S, I) (:180-189);
//   * from there every reference-consuming op k owns max(len_k, 1) consecutive columns (a zero-length
//     op still holds the cursor for one column, :196-203); cstart[k] = exclusive sum of those;
//   * column offset r lies in op k with cstart[k] <= r < cstart[k+1]; cigar_operation_offset =
//     r - cstart[k]; query_offset = qstart[k] + (op k consumes query ? r - cstart[k] : 0) (:199-201,
//     :215-217), qstart[k] = query bases consumed by all earlier ops;
//   * the read is in the column while r < basesCovered (end_position > pos, :352);
//   * current_base = sequence[query_offset] for M/=/X, '-' otherwise (:115-122).
// A column is the set of such reads, in file order (:351-359 stable compaction, :377-386 appends).
//
// MI355X design.  K4 is a histogram-shaped scatter (integer, HBM / LDS-bound, no MFMA).  The route nearly every window takes is
// PLAN + ACCUMULATE (k4_plan / k4_acc2, further down): one pass over the reads turns every read into 8-byte descriptors in the
// slabs of the 1024-position windows it touches, and a persistent workgroup per range of windows adds the described bases to
// byte-wide counters in LDS -- a piece of a descriptor per lane, eight positions a step, the class of a base found for four bases
// at once by v_perm_b32 -- and flushes 32 contiguous bytes a position.  Long reads are walked from per-window CIGAR checkpoints
// (k4_ck_build) inside the same kernel.  What the plan does not take (odd CIGARs, windows too deep) goes to the GENERAL kernel
// (k4_tile, round 1's K4): one workgroup owns a window with its counters in LDS and is handed the contiguous record range that
// can touch it (exact halo: first record whose running max end_position exceeds the window start -- SURVEY.md 3.3 -- from two
// prefix-maximum arrays built by K3), and walks a read one of three ways: LANE = READ for ordinary short reads, LANE = OP for long
// reads (64 CIGAR ops per coalesced load, wave prefix scan, resumed from the checkpoints), lanes along the read for the odd rest.
#include "common.h"

#include <algorithm>
#include <cstdio>

namespace {

// Window geometry.  A window's life is a chain of dependent memory round trips (candidate rows -> record rows ->
// CIGAR + SEQ) that does not shrink with the window, so the biggest window LDS can hold wins: measured on cfg3
// (50 M reads) 2048 positions / 512 threads (2 workgroups per CU) 10.0 ms, 1024 / 256 (4 per CU) 10.6 ms,
// 4096 / 1024 (148 KB of LDS, one workgroup of 16 waves per CU) 8.9 ms.
#ifndef BGP_K4_W
#define BGP_K4_W 1024
#define BGP_K4_T 1024
#define BGP_K4_T16 512
#define BGP_K4_HB 512
#endif
constexpr int W = BGP_K4_W;            // reference positions per window of the plan / accumulate path (k4_plan / k4_acc), power of two
constexpr int WO = 4096;               // ... of the general kernel when a stream is left to it entirely (long reads): the biggest window LDS holds wins there
constexpr int HB = BGP_K4_HB;          // depth-histogram bins kept in LDS
constexpr int K4_T = BGP_K4_T;         // threads of a workgroup on the 32-bit layout
constexpr int K4_T16 = BGP_K4_T16;     // ... on the 16-bit layout (two workgroups per CU)
constexpr int cls_shift_of(int w) { return w == 4096 ? 14 : w == 2048 ? 13 : w == 1024 ? 12 : w == 512 ? 11 : -1; }   // log2(bytes of one class array)
static_assert(cls_shift_of(W) > 0 && cls_shift_of(WO) > 0, "windows must be 512, 1024, 2048 or 4096 positions");
constexpr uint32_t RUNS_FIRST = 1u << 16;   // entries of the run table to begin with (a stream with more reference runs grows it and repeats K3)
#ifndef BGP_DEEP_WINDOW
#define BGP_DEEP_WINDOW 65535
#endif
constexpr uint32_t DEEP_WINDOW = BGP_DEEP_WINDOW; // candidate records from which a window takes K4's 32-bit LDS layout (0: every window, for tests)


struct RunDev { int32_t ref_id; uint32_t pad; uint64_t first_record, end_record, start_pos, end_pos, column_base, tile_base; };

struct K3Args { const bgp_record* rec; uint64_t n; uint64_t* key_end; uint64_t* key_pos; uint32_t* status; /* [3] any long read */ uint32_t run_cap; uint32_t gap; /* a run also begins behind this many positions without coverage (0: never) */ };

constexpr int FAST_LSEQ = 160;         // reads up to 20 dwords of packed SEQ and 4 CIGAR ops take K4's lane = read path
constexpr int FAST_SEG = 255;          // longest read the planner turns into SEG descriptors (a descriptor's length field: 8 bits; <= 8 staging chunks)
// Reads that get per-window CIGAR checkpoints and, on the accumulate path, a row of class bytes: more than four ops, or longer
// than a SEG descriptor holds.  (A read of 161..255 bases with up to four ops needs neither: the planner takes it as SEGs,
// and in a window left to the general kernel it walks from its start -- so a stream of 250-base reads costs no walk.)
__device__ __forceinline__ bool wants_ckpt(uint32_t n_cig, uint32_t l_seq) { return n_cig > 4 || l_seq > (uint32_t)FAST_SEG; }

// K3 is one two-pass inclusive max-scan over two keys derived on the fly from the record table
// (reference key << 32 | end_position, ... | position; 0 for reads with basesCovered() == 0):
// pass 1 reduces 2048-record tiles, a single workgroup scans the tile maxima, pass 2 rescans each
// tile with its carry-in, writes both prefix-maximum arrays and -- because the running maximum of the
// position key just before record i is exactly "the last covering read before i" -- checks sortedness
// (makePileup's precondition, pileup.d:25-27), finds the first read of every reference run
// (pileup.d:399-424) and notes whether any long read exists.
constexpr int K3_T = 256, K3_ITEMS = 8, K3_TILE = K3_T * K3_ITEMS;

__device__ __forceinline__ K3Key k3_max(K3Key a, K3Key b) { K3Key r; r.e = a.e > b.e ? a.e : b.e; r.p = a.p > b.p ? a.p : b.p; return r; }

__device__ __forceinline__ K3Key k3_key_of(const bgp_record& r) { return ::k3_key_of(r.ref_id, r.pos, r.end_pos); }
static_assert(K3_TILE_RECORDS == 256 * 8, "K1 leaves its tile maxima for K3's tiles");

// exclusive prefix (max) of per-thread values across the workgroup + workgroup total
__device__ __forceinline__ K3Key k3_block_excl(K3Key mine, K3Key* total, K3Key* lds /* K3_T/64 */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    K3Key inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        K3Key o; o.e = __shfl_up(inc.e, d, 64); o.p = __shfl_up(inc.p, d, 64);
        if (lane >= d) inc = k3_max(inc, o);
    }
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    K3Key carry{0, 0}, tot{0, 0};
#pragma unroll
    for (int k = 0; k < K3_T / 64; ++k) { if (k < w) carry = k3_max(carry, lds[k]); tot = k3_max(tot, lds[k]); }
    *total = tot;
    K3Key ex; ex.e = __shfl_up(inc.e, 1, 64); ex.p = __shfl_up(inc.p, 1, 64);
    if (lane == 0) { ex.e = 0; ex.p = 0; }
    __syncthreads();
    return k3_max(carry, ex);
}

// Layout: a tile is 2048 records = 4 waves x 8 rows x 64 lanes; lane = record inside a row, so every
// load of the 32-byte rows is a coalesced 2 KiB per wave (a thread owning 8 consecutive rows' worth of
// records would stride 256 B between lanes).  Scan order = record order: rows of a wave in sequence,
// waves in sequence.
__device__ __forceinline__ K3Key k3_wave_incl(K3Key v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        K3Key o; o.e = __shfl_up(v.e, d, 64); o.p = __shfl_up(v.p, d, 64);
        if (lane >= d) v = k3_max(v, o);
    }
    return v;
}

__global__ __launch_bounds__(K3_T) void k3_reduce(K3Args a, K3Key* bsum) {
    __shared__ K3Key lds[K3_T / 64];
    const uint64_t base = (uint64_t)blockIdx.x * K3_TILE + threadIdx.x;
    K3Key acc{0, 0};
#pragma unroll
    for (int k = 0; k < K3_ITEMS; ++k) { const uint64_t i = base + (uint64_t)k * K3_T; if (i < a.n) acc = k3_max(acc, k3_key_of(a.rec[i])); }
    K3Key tot;
    (void)k3_block_excl(acc, &tot, lds);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(K3_T) void k3_bsum(K3Key* bsum, uint64_t nb) {      // single workgroup: exclusive scan of the tile maxima
    __shared__ K3Key lds[K3_T / 64];
    K3Key carry{0, 0};
    for (uint64_t base = 0; base < nb; base += K3_TILE) {                        // a thread takes K3_ITEMS consecutive maxima: 24 rounds of two barriers for 10^8 records (one per round: 191, 0.2 ms)
        const uint64_t i0 = base + (uint64_t)threadIdx.x * K3_ITEMS;
        K3Key v[K3_ITEMS], mine{0, 0};
#pragma unroll
        for (int k = 0; k < K3_ITEMS; ++k) { v[k] = K3Key{0, 0}; if (i0 + k < nb) v[k] = bsum[i0 + k]; }
#pragma unroll
        for (int k = 0; k < K3_ITEMS; ++k) mine = k3_max(mine, v[k]);
        K3Key tot;
        K3Key run = k3_max(carry, k3_block_excl(mine, &tot, lds));
#pragma unroll
        for (int k = 0; k < K3_ITEMS; ++k) { if (i0 + k < nb) bsum[i0 + k] = run; run = k3_max(run, v[k]); }
        carry = k3_max(carry, tot);
    }
}

__global__ __launch_bounds__(K3_T) void k3_down(K3Args a, const K3Key* bsum, RunDev* runs) {
    __shared__ K3Key lds[K3_T / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t wbase = (uint64_t)blockIdx.x * K3_TILE + (uint64_t)w * (K3_ITEMS * 64) + lane;
    K3Key key[K3_ITEMS]; int32_t ref[K3_ITEMS]; uint32_t flg[K3_ITEMS];           // flg: bit0 long read, bit1 bad coordinates
    K3Key mine{0, 0};
#pragma unroll
    for (int k = 0; k < K3_ITEMS; ++k) {
        const uint64_t i = wbase + (uint64_t)k * 64;
        key[k] = K3Key{0, 0}; ref[k] = 0; flg[k] = 0;
        if (i < a.n) {
            const bgp_record r = a.rec[i];
            key[k] = k3_key_of(r); ref[k] = r.ref_id;
            flg[k] = (wants_ckpt(r.flag_nc & 0xFFFFu, (uint32_t)r.l_seq) ? 1u : 0u) | ((r.pos < 0 || r.end_pos < r.pos) ? 2u : 0u);
            mine = k3_max(mine, key[k]);
        }
    }
    // maxima over every record before this wave's first: tile carry-in + the waves in front
    K3Key wtot = mine;
    for (int d = 32; d >= 1; d >>= 1) { K3Key o; o.e = __shfl_xor(wtot.e, d, 64); o.p = __shfl_xor(wtot.p, d, 64); wtot = k3_max(wtot, o); }
    if (lane == 0) lds[w] = wtot;
    __syncthreads();
    K3Key carry = bsum[blockIdx.x];
    for (int k = 0; k < w; ++k) carry = k3_max(carry, lds[k]);
    uint32_t n_long = 0;                                                   // long covering reads of this wave (wave-uniform)
#pragma unroll
    for (int k = 0; k < K3_ITEMS; ++k) {
        const uint64_t i = wbase + (uint64_t)k * 64;
        const K3Key inc = k3_wave_incl(key[k], lane);
        K3Key before; before.e = __shfl_up(inc.e, 1, 64); before.p = __shfl_up(inc.p, 1, 64);
        if (lane == 0) before = K3Key{0, 0};
        before = k3_max(before, carry);                                       // maxima over every record before record i
        n_long += (uint32_t)__popcll(__ballot(i < a.n && key[k].p && (flg[k] & 1u)));
        if (i < a.n) {
            if (key[k].p) {                                                   // a covering read
                if (flg[k] & 2u) atomicOr(&a.status[0], ST_EINVAL);
                if (key[k].p < before.p) atomicOr(&a.status[0], ST_EUNSORTED); // pileup.d:25-27
                // first read of a reference -- or of an ISLAND of it: nothing covers the a.gap positions in front of the read.  BioD's sweep
                // jumps over such a stretch (initNewReference, pileup.d:388-392,399-424) or walks it column by column
                // (skip_zero_coverage = false); here the dense columns of a run would otherwise span it (two reads 2^31 apart: 64 GB)
                const bool island = a.gap && (before.p >> 32) == (key[k].p >> 32) && (uint64_t)(uint32_t)before.e + a.gap <= (uint64_t)(uint32_t)key[k].p;
                if ((before.p >> 32) != (key[k].p >> 32) || island) {
                    const uint32_t slot = atomicAdd(&a.status[2], 1u);
                    if (slot < a.run_cap) { RunDev rd{}; rd.ref_id = ref[k]; rd.first_record = i; rd.start_pos = (uint64_t)(uint32_t)key[k].p; runs[slot] = rd; }
                }
            }
            const K3Key out = k3_max(before, key[k]);
            a.key_end[i] = out.e; a.key_pos[i] = out.p;
        }
        K3Key last; last.e = __shfl(inc.e, 63, 64); last.p = __shfl(inc.p, 63, 64);
        carry = k3_max(carry, last);
    }
    if (n_long && lane == 0) atomicAdd(&a.status[3], n_long);                 // [3]: long covering reads of the stream
}

__global__ void k3_run_ends(RunDev* runs, uint32_t n_runs, const uint64_t* pmax_end, uint64_t n) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_runs) return;
    const uint64_t last = runs[j].end_record - 1;
    runs[j].end_pos = (uint64_t)(uint32_t)pmax_end[last];
}

// The run table finished on the device when it is small (nearly always: a run per reference and per island of coverage): the runs
// in file order, where each one ends, its columns and windows in front of it -- the host did this between three round trips
// (table down, sorted table up, ends down, bases up: ~0.15 ms of a 4.5 ms step on an eighth of cfg3).
constexpr uint32_t RUNS_ON_DEVICE = 1024;
__global__ __launch_bounds__(1024) void k3_finish(RunDev* runs, const uint32_t* status, uint32_t run_cap, const uint64_t* pmax_end, uint64_t n, uint32_t w) {
    __shared__ uint64_t s_first[RUNS_ON_DEVICE], s_span[RUNS_ON_DEVICE];
    const uint32_t n_runs = status[2];
    if (n_runs == 0 || n_runs > RUNS_ON_DEVICE || n_runs > run_cap) return;       // (the host's own loop takes a larger table)
    const uint32_t j = threadIdx.x;
    RunDev my{};
    if (j < n_runs) { my = runs[j]; s_first[j] = my.first_record; }
    __syncthreads();
    uint32_t rank = 0;
    if (j < n_runs) for (uint32_t k = 0; k < n_runs; ++k) rank += s_first[k] < my.first_record ? 1u : 0u;
    __syncthreads();
    if (j < n_runs) s_first[rank] = my.first_record;
    __syncthreads();
    if (j < n_runs) {
        my.end_record = rank + 1 < n_runs ? s_first[rank + 1] : n;
        my.end_pos = (uint64_t)(uint32_t)pmax_end[my.end_record - 1];
        s_span[rank] = my.end_pos - my.start_pos;
    }
    __syncthreads();
    if (j < n_runs) {
        uint64_t cb = 0, tb = 0;
        for (uint32_t k = 0; k < rank; ++k) { cb += s_span[k]; tb += (s_span[k] + w - 1) / w; }
        my.column_base = cb; my.tile_base = tb;
        runs[rank] = my;
    }
}

struct K4Args {
    const uint8_t* s; const bgp_record* rec; const uint64_t* pmax_end; const uint64_t* pmax_pos;
    const RunDev* runs; uint32_t n_runs; uint64_t n_tiles; uint64_t stream_len;
    bgp_column* cols; unsigned long long* hist; unsigned long long* pstats; /* [0] n_columns [1] sum_cov [2] max_cov */
    uint32_t* status; const struct TileB* tiles; unsigned long long* tile_stats;
    const uint32_t* deep; const uint32_t* n_deep;       // windows with too many candidate records for the 16-bit layout
    const uint64_t* ck_off; const struct Ckpt* ck;      // long-read checkpoints (null when the stream has no long read)
};

__device__ __forceinline__ uint64_t lower_bound_gt(const uint64_t* a, uint64_t lo, uint64_t hi, uint64_t key) {   // first i in [lo,hi) with a[i] > key
    while (lo < hi) { const uint64_t m = (lo + hi) >> 1; if (a[m] > key) hi = m; else lo = m + 1; }
    return lo;
}

__device__ __forceinline__ uint32_t base_class(uint32_t code) {   // nt16 code -> A,C,G,T,other
    return (uint32_t)(0x4444444344424104ull >> (code * 4)) & 0xFu;  // 1->0 2->1 4->2 8->3 else 4
}

// One thread per window: the record range that can touch it (exact halo).  Done here, massively
// parallel, because the two ~25-step dependent binary searches are pure latency.
struct TileB { uint64_t i_lo, i_hi, col_out; uint32_t T0, Tend, run_start, pad; };   // everything a window needs in one 40-byte (scalar) load; pad = 1: deep window
// mode 0: every window, geometry and record range (two binary searches through the halo keys); 1: geometry only -- the accumulate path of a
// stream of short reads never asks which records can touch a window, its plan knows (cfg3: 0.16 -> 0.03 ms) --; 2: the record ranges of the
// windows on `list` (what the accumulate path left for the general kernel, which does scan them); 3: geometry and the range of plan ROWS the
// accumulate path reads for the window (k12_lds wrote them, common.h: bgp_plan_row)
// first i in [lo, hi) with a[i] > key, a = one of K3's prefix-maximum arrays over n records, through the exclusive maxima of its 2048-record tiles
// (bs[j].p: what k3_bsum left; 16 steps in an array the L2 holds, then 11 inside one tile's 16 KB) instead of 27 steps through 800 MB
__device__ __forceinline__ uint64_t lower_bound_gt_pos2(const uint64_t* a, const K3Key* bs, uint64_t nt, uint64_t n, uint64_t lo, uint64_t hi, uint64_t key) {
    uint64_t tl = 0, th = nt;
    while (tl < th) { const uint64_t m = (tl + th) >> 1; if (bs[m].p > key) th = m; else tl = m + 1; }      // first tile in front of which something exceeds the key
    const uint64_t j = tl ? tl - 1 : 0;                                                                       // ... the tile before it holds the first record that does (or none does)
    const uint64_t g = lower_bound_gt(a, j * K3_TILE_RECORDS, min(n, (j + 1) * (uint64_t)K3_TILE_RECORDS), key);
    return min(max(g, lo), hi);
}
__global__ __launch_bounds__(256) void k4_bounds(const uint64_t* pmax_end, const uint64_t* pmax_pos, const RunDev* runs, uint32_t n_runs, uint64_t n_tiles, TileB* out, uint32_t w, int mode, const uint32_t* list, const uint32_t* n_list, const K3Key* bsum = nullptr, uint64_t n_rec = 0) {
    uint64_t tile = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (mode == 2) { if (tile >= *n_list) return; tile = list[tile]; }
    if (tile >= n_tiles) return;
    uint32_t lo_r = 0, hi_r = n_runs;
    while (hi_r - lo_r > 1) { const uint32_t m = (lo_r + hi_r) >> 1; if (runs[m].tile_base <= tile) lo_r = m; else hi_r = m; }
    const RunDev run = runs[lo_r];
    const uint32_t T0 = (uint32_t)(run.start_pos + (tile - run.tile_base) * (uint64_t)w);
    const uint32_t Tend = (uint32_t)min((uint64_t)T0 + w, run.end_pos);
    const uint64_t rk = ref_key(run.ref_id) << 32;
    TileB b;
    b.i_lo = b.i_hi = 0;
    if (mode == 3) {         // the rows k4_acc2 takes the window's ordinary reads from: every record that starts less than BGP_ROW_REACH positions in front of the window or inside it (the stream is sorted)
        const uint64_t nt = (n_rec + K3_TILE_RECORDS - 1) / K3_TILE_RECORDS;
        b.i_lo = T0 >= BGP_ROW_REACH ? lower_bound_gt_pos2(pmax_pos, bsum, nt, n_rec, run.first_record, run.end_record, rk | (T0 - BGP_ROW_REACH)) : run.first_record;
        b.i_hi = lower_bound_gt_pos2(pmax_pos, bsum, nt, n_rec, b.i_lo, run.end_record, rk | (Tend - 1));
    } else if (mode != 1) {
        b.i_lo = lower_bound_gt(pmax_end, run.first_record, run.end_record, rk | T0);             // first record whose running max end_position exceeds T0
        b.i_hi = lower_bound_gt(pmax_pos, run.first_record, run.end_record, rk | (Tend - 1));     // first record starting at/after Tend
    }
    b.col_out = run.column_base + (T0 - run.start_pos); b.T0 = T0; b.Tend = Tend; b.run_start = (uint32_t)run.start_pos;
    b.pad = mode != 3 && b.i_hi - b.i_lo >= DEEP_WINDOW ? 1u : 0u;     // bit 0: a count per position and class is bounded by the candidates, 16 bits are enough below this (mode 3's range is not the candidates': k4_wscan lists a window of too many rows, and mode 2 then judges it)
    if (mode == 2) b.pad |= out[tile].pad & 2u;           // (bit 1, "not for the accumulate path", is set by k4_wscan, which also builds the list of such windows)
    out[tile] = b;
}

// ---- long reads: CIGAR checkpoints ------------------------------------------------------------
// A read with thousands of CIGAR ops overlaps many windows; walking its ops from the start in every
// window is O(n_cigar) dependent loads per visit.  One pre-pass walks each long read ONCE, 64 ops per
// coalesced load with a wave prefix scan over the ops' column / query lengths (the closed form of
// pileup.d:175-222), and records for every window the read overlaps the cursor state
// (op index, columns before it, query bases before it) of the op that owns the window's first column.
struct Ckpt { uint32_t k, cs, qs, pad; };

// inclusive prefix sum over the 64 lanes with DPP: shifts inside the rows of 16, then the rows' totals carried over (six instructions;
// with __shfl_up it was eighteen, and the long reads' walk and row writer do three of these per chunk of 64 ops)
#define BGP_DPP(v, ctrl, rmask) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rmask, 0xF, false)
__device__ __forceinline__ uint32_t wave_incl_u32(uint32_t v, int) {
    v += BGP_DPP(v, 0x111, 0xF); v += BGP_DPP(v, 0x112, 0xF); v += BGP_DPP(v, 0x114, 0xF); v += BGP_DPP(v, 0x118, 0xF);
    v += BGP_DPP(v, 0x142, 0xA); v += BGP_DPP(v, 0x143, 0xC);
    return v;
}

__device__ __forceinline__ uint32_t rl(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }

// column / query length an op contributes, given whether the cursor has started (pileup.d:180-189,196-217)
__device__ __forceinline__ void op_lens(uint32_t op, bool active, bool started, uint32_t* rlen, uint32_t* qlen) {
    const uint32_t len = op >> 4, qc = cig_qc(op), rc = cig_rc(op);
    *rlen = 0; *qlen = 0;
    if (!active) return;
    if (started) { const uint32_t elen = len ? len : 1u; if (rc) { *rlen = elen; if (qc) *qlen = elen; } else if (qc) *qlen = len; }
    else if (qc && !rc) *qlen = len;                               // leading S / I; a leading N contributes nothing
}

__device__ __forceinline__ uint32_t run_of_record(const RunDev* runs, uint32_t n_runs, uint64_t i) {
    uint32_t lo = 0, hi = n_runs;
    while (hi - lo > 1) { const uint32_t m = (lo + hi) >> 1; if (runs[m].first_record <= i) lo = m; else hi = m; }
    return lo;
}

__global__ __launch_bounds__(256) void k4_ck_count(const bgp_record* rec, uint64_t n, const RunDev* runs, uint32_t n_runs, uint64_t* n_ck, uint32_t W) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bgp_record r = rec[i];
    uint64_t c = 0;
    if (r.end_pos != r.pos && wants_ckpt(r.flag_nc & 0xFFFFu, (uint32_t)r.l_seq)) {
        const RunDev& run = runs[run_of_record(runs, n_runs, i)];
        c = ((uint64_t)(uint32_t)r.end_pos - 1 - run.start_pos) / W - ((uint64_t)(uint32_t)r.pos - run.start_pos) / W + 1;
    }
    n_ck[i] = c;
}

// WPR: a wave per RECORD (a stream of long reads: with a lane per record a wave walks its 64 reads one after the other --
// cfg5's 200 k reads are 3125 waves of 2000 serial steps on a device with 8192 wave slots); otherwise a lane per record
// and the wave walks the few long reads among its 64.
template <bool WPR> __global__ __launch_bounds__(256) void k4_ck_build(const uint8_t* s, uint64_t stream_len, const bgp_record* rec, uint64_t n, const RunDev* runs, uint32_t n_runs,
                                                   const uint64_t* ck_off, Ckpt* ck, uint32_t W_) {
    const int64_t W = (int64_t)W_;
    const int lane = threadIdx.x & 63;
    const uint64_t wave0 = WPR ? (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6) : (uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63);   // the wave's (first) record
    const uint64_t i = WPR ? wave0 : wave0 + lane;
    bgp_record r{};
    bool lng = false;
    if (i < n) { r = rec[i]; lng = r.end_pos != r.pos && wants_ckpt(r.flag_nc & 0xFFFFu, (uint32_t)r.l_seq); }
    unsigned long long m = WPR ? (lng ? 1ull : 0ull) : __ballot(lng);
    while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        const uint64_t ri = wave0 + src;
        const uint32_t pos = (uint32_t)__shfl(r.pos, src, 64), covered = (uint32_t)__shfl(r.end_pos, src, 64) - pos;
        const uint32_t n_cig = (uint32_t)__shfl((int)r.flag_nc, src, 64) & 0xFFFFu, l_name = (uint32_t)__shfl((int)r.bin_mq_nl, src, 64) & 0xFFu;
        const uint64_t rec_off = ((uint64_t)(uint32_t)__shfl((int)(r.rec_off >> 32), src, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)r.rec_off, src, 64);
        const uint8_t* cg = s + rec_off + 36 + l_name;
        const RunDev& run = runs[run_of_record(runs, n_runs, ri)];
        const uint32_t wsh = (uint32_t)__builtin_ctz(W_);                                       // (windows are powers of two: no 64-bit division per op)
        const uint64_t t_first = ((uint64_t)pos - run.start_pos) >> wsh;
        const uint32_t n_ck = (uint32_t)(ck_off[ri + 1] - ck_off[ri]);
        Ckpt* out = ck + ck_off[ri];
        // window j (j >= 1) starts at column offset D + j*W of this read (D <= 0)
        const int64_t D = (int64_t)(run.start_pos + t_first * W) - (int64_t)pos;
        const uint32_t mD = (uint32_t)(-D);
        uint32_t cs = 0, qs = 0; bool started = false;
        // what the accumulate path asks of a read (k4_plan: "the mould"): the first reference-consuming op is M/=/X (so the
        // read's first column is a base of it: no leading N, D), no zero-length reference-consuming op, no M base past SEQ
        bool plain = true;
        const uint32_t l_seq_r = (uint32_t)__shfl(r.l_seq, src, 64);
        // The walk is a chain of chunks per read: 256 ops a chunk, FOUR consecutive ops a lane (one 16-byte load; a lane adds up its
        // four before the wave's scan -- with an op a lane the two scans and the ballots of a chunk were paid for 64 ops: 1.83 ms
        // for cfg5's 6 * 10^8 ops, bound by instruction issue), and two chunks on their way while one is scanned (a chunk's scan is a
        // fraction of a microsecond: waiting out an HBM round trip per chunk was 2.6 ms).
        constexpr int CK_AHEAD = 2;
        const uint64_t cg_off = rec_off + 36 + l_name;
        auto load4 = [&](uint32_t first) {                                // ops first .. first + 3 of the read (zeros behind its last op)
            uint4 v = make_uint4(0, 0, 0, 0);
            if (first < n_cig) {
                if (cg_off + 4ull * first + 16 <= stream_len) __builtin_memcpy(&v, cg + 4ull * first, 16);      // (behind the CIGAR lies the read's SEQ: the bytes exist)
                else { v.x = ld_u32(cg + 4ull * first); if (first + 1 < n_cig) v.y = ld_u32(cg + 4ull * first + 4); if (first + 2 < n_cig) v.z = ld_u32(cg + 4ull * first + 8); if (first + 3 < n_cig) v.w = ld_u32(cg + 4ull * first + 12); }
            }
            return v;
        };
        uint4 opq[CK_AHEAD];
#pragma unroll
        for (int q = 0; q < CK_AHEAD; ++q) opq[q] = load4(256u * q + 4u * lane);
        for (uint32_t c0 = 0; c0 < n_cig && cs < covered; c0 += 256) {
            const uint4 o4 = opq[0];
#pragma unroll
            for (int q = 0; q + 1 < CK_AHEAD; ++q) opq[q] = opq[q + 1];
            opq[CK_AHEAD - 1] = load4(c0 + 256u * CK_AHEAD + 4u * lane);
            const uint32_t k0 = c0 + 4u * (uint32_t)lane;                     // this lane's first op
            const uint32_t op[4] = {o4.x, o4.y, o4.z, o4.w};
            bool act[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) act[t] = k0 + t < n_cig;
            // the first reference-consuming non-N op starts the cursor (pileup.d:180-189)
            uint32_t fl = 256;                                              // its place in the chunk (256: not in this chunk)
            if (!started) {
                uint32_t f0 = 256;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const unsigned long long st = __ballot(act[t] && cig_rc(op[t]) && (op[t] & 15u) != 3u), anyrc = __ballot(act[t] && cig_rc(op[t]));
                    if (st) fl = min(fl, 4u * (uint32_t)(__ffsll((long long)st) - 1) + t);
                    if (anyrc) f0 = min(f0, 4u * (uint32_t)(__ffsll((long long)anyrc) - 1) + t);
                }
                if (f0 < 256u) {      // a leading N, or a first column that is a D
                    const uint32_t sub = f0 & 3u, opsel = sub == 0 ? op[0] : sub == 1 ? op[1] : sub == 2 ? op[2] : op[3];
                    if (f0 != fl || !cig_qc((uint32_t)__shfl((int)opsel, (int)(f0 >> 2), 64))) plain = false;
                }
            }
            uint32_t rlen[4], qlen[4], rt = 0, qt = 0, qcm[4], badm = 0;
            if (started) {      // (every chunk but a read's first: masks and arithmetic where op_lens' cases were exec-mask bookkeeping)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const uint32_t o = op[t], len = o >> 4, tb = 0x3C1A7u >> ((o & 15u) * 2);
                    const uint32_t am = act[t] ? ~0u : 0u, rcm = am & (0u - ((tb >> 1) & 1u)), qm = am & (0u - (tb & 1u));
                    const uint32_t elen = max(len, 1u);
                    rlen[t] = rcm & elen; qlen[t] = qm & ((rcm & elen) | (~rcm & len)); qcm[t] = qm;
                    badm |= rcm & (len == 0u ? ~0u : 0u);
                    rt += rlen[t]; qt += qlen[t];
                }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    op_lens(op[t], act[t], 4u * (uint32_t)lane + t >= fl, &rlen[t], &qlen[t]); rt += rlen[t]; qt += qlen[t];
                    qcm[t] = act[t] && cig_qc(op[t]) ? ~0u : 0u;
                    badm |= act[t] && cig_rc(op[t]) && (op[t] >> 4) == 0 ? ~0u : 0u;
                }
            }
            const uint32_t ir = wave_incl_u32(rt, lane), iq = wave_incl_u32(qt, lane);
            uint32_t cs_b = cs + ir - rt, qs_b = qs + iq - qt;
            bool bad = badm != 0u;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bad |= rlen[t] && qcm[t] && qs_b + qlen[t] > l_seq_r;
                if (rlen[t]) {
                    if (cs_b == 0 && n_ck) { Ckpt e{k0 + t, 0u, qs_b, 0u}; out[0] = e; }      // window of the read's first column
                    // the windows that start inside the op's columns [cs_b, cs_b + rlen): window j starts at column D + j * W, so they are
                    // j = ceil(u / W) .. ceil((u + rlen) / W) - 1 with u = cs_b - D (32 bits: D <= 0, a read's columns stay below 2^31)
                    const uint32_t u = cs_b + mD, j_lo = max(1u, (u + W_ - 1u) >> wsh), j_hi = min(n_ck, (u + rlen[t] + W_ - 1u) >> wsh);
                    for (uint32_t j = j_lo; j < j_hi; ++j) { Ckpt e{k0 + t, cs_b, qs_b, 0u}; out[j] = e; }
                }
                cs_b += rlen[t]; qs_b += qlen[t];
            }
            if (__ballot(bad)) plain = false;
            cs += __shfl(ir, 63, 64); qs += __shfl(iq, 63, 64);
            started = started || fl < 256u;
        }
        if (lane == 0 && n_ck) out[0].pad = (plain && cs >= covered) ? 0u : 1u;      // the verdict rides on the read's first checkpoint (k4_plan reads it)
    }
}

// ---- K4: per-position column summary ----------------------------------------------------------
// LDS tile (u32 words): cnt[5][W] (A,C,G,T,other) | deld[W+4] ('-' as a difference array: +1 where a
// D/N op starts, -1 where it ends, summed at flush) | nst[W] (n_starting_here) | hist[HB] | list[LCAP] |
// slist[LCAP]; cnt and nst are 16 bits wide in the layout ordinary windows use (struct Lay below).
//
// Per window: (1) SCAN the candidate record range one workgroup-width at a time (one coalesced 16-byte read
// per lane, four rounds requested before the first is looked at), count n_starting_here and compact the
// indices of the reads that really overlap the window into an LDS list; (2) PROCESS the list 64 reads per
// wave with LANE = READ: a lane loads its read's CIGAR ops and whole packed SEQ up front (five unaligned
// 16-byte loads, all in flight together), turns its <= 4 ops into window-clipped query intervals, and then
// all lanes step the same packed-SEQ dword -- 8 bases classified with word-parallel bit tricks, one ds_add
// per base; a D/N op is two adds on the difference array.  Reads that do not fit that mould (longer than
// 160 bases, more than 4 CIGAR ops, zero-length or leading-N ops, CIGAR longer than SEQ) go to a second list
// dealt to all waves: long reads resume from per-window CIGAR checkpoints with LANE = OP (pile_read_long),
// the odd rest takes the fully general lanes-along-the-read walk (pile_read_coop), both the closed form of
// pileup.d:175-222; (3) FLUSH: block scan of the difference array, 32 contiguous bytes per position.

constexpr int LDS_PAD = 8;               // words in front of the tile (see k4_tile's hot loop)
// Two layouts of the window in LDS.  P16: the five class counters and n_starting_here are 16 bits wide, two
// positions to a word (an add is 1 << 16 * (position & 1); a half never carries because a window whose candidate
// range holds 65 535 records or more is not given to this layout) -- half the LDS, so two workgroups share a CU and
// one window's memory waits overlap the other's arithmetic.  P32: plain 32-bit counters, for those deep windows.
template <bool P16_, int T_, int WIN_> struct Lay {
    static constexpr int WIN = WIN_;                              // reference positions of the window
    static constexpr int CLS_SHIFT = cls_shift_of(WIN_);
    static constexpr bool P16 = P16_;
    static constexpr int T = T_;
    static constexpr int CNT = P16_ ? 5 * WIN_ / 2 : 5 * WIN_;          // word offsets inside the tile
    static constexpr int DELD = CNT;
    static constexpr int NST = DELD + WIN_ + 4;
    static constexpr int HIST = NST + (P16_ ? WIN_ / 2 : WIN_);
    static constexpr int LIST = HIST + HB;
    static constexpr int LCAP = 2 * T_;                            // a scan round adds at most T entries
    static constexpr int SLIST = LIST + LCAP;                      // second list: reads for the wave-cooperative paths
    static constexpr int WORDS = SLIST + LCAP + LDS_PAD;
    static constexpr int CSH = P16_ ? CLS_SHIFT - 1 : CLS_SHIFT;   // log2(bytes of one class array)
    __device__ __forceinline__ static void add_cls(uint32_t* lds, uint32_t cls, uint32_t col) {
        const uint32_t i = cls * WIN_ + col;
        if (P16_) atomicAdd(&lds[i >> 1], 1u << ((i & 1u) * 16)); else atomicAdd(&lds[i], 1u);
    }
    __device__ __forceinline__ static void add_nst(uint32_t* lds, uint32_t col) {
        if (P16_) atomicAdd(&lds[NST + (col >> 1)], 1u << ((col & 1u) * 16)); else atomicAdd(&lds[NST + col], 1u);
    }
    __device__ __forceinline__ static uint32_t get_cls(const uint32_t* lds, uint32_t cls, uint32_t col) {
        const uint32_t i = cls * WIN_ + col;
        return P16_ ? (lds[i >> 1] >> ((i & 1u) * 16)) & 0xFFFFu : lds[i];
    }
    __device__ __forceinline__ static uint32_t get_nst(const uint32_t* lds, uint32_t col) {
        return P16_ ? (lds[NST + (col >> 1)] >> ((col & 1u) * 16)) & 0xFFFFu : lds[NST + col];
    }
};
using L16 = Lay<true, K4_T16, W>;          // the general kernel on the accumulate path's tiling (windows that path leaves behind)
using L32 = Lay<false, K4_T, W>;
using L16o = Lay<true, K4_T16, WO>;        // ... on its own tiling (streams of long reads)
using L32o = Lay<false, K4_T, WO>;


// packed SEQ dwords [w0, w0+64) of a read; lane i holds dword w0+i (bytes past the stream end read as 0)
__device__ __forceinline__ uint32_t load_seq_words(const uint8_t* s, uint64_t stream_len, uint64_t seq_off, uint32_t n_bytes, uint32_t w0, int lane) {
    const uint32_t b = (w0 + lane) * 4;
    if (b >= n_bytes) return 0;
    const uint64_t o = seq_off + b;
    if (o + 4 <= stream_len) return ld_u32(s + o);
    uint32_t v = 0;
    for (uint32_t k = 0; k < 4 && o + k < stream_len; ++k) v |= (uint32_t)s[o + k] << (8 * k);
    return v;
}

// General path: one read against the window, executed by a whole wave with lanes along the read.
// All arguments are wave-uniform.
template <class L> __device__ __forceinline__ void pile_read_coop(const K4Args& a, uint32_t* lds, uint32_t T0, uint32_t Tend, int lane,
                                               uint32_t pos, uint32_t covered, uint32_t n_cig, uint32_t l_seq, uint64_t cig_off) {
    const uint64_t seq_off = cig_off + 4ull * n_cig;
    const uint32_t seq_bytes = (l_seq + 1) / 2;
    uint32_t win0 = 0;
    uint32_t words = load_seq_words(a.s, a.stream_len, seq_off, seq_bytes, 0, lane);
    const uint32_t need_hi = min(covered, Tend - pos);                    // column offsets [need_lo, need_hi) fall into the window
    const uint32_t need_lo = T0 > pos ? T0 - pos : 0;
    uint32_t cs = 0, qs = 0, k = 0;
    bool started = false;
    for (; k < n_cig && cs < need_hi; ++k) {
        const uint32_t op = ld_u32(a.s + cig_off + 4ull * k);
        const uint32_t len = op >> 4, qc = cig_qc(op), rc = cig_rc(op);
        if (!started) {                                                   // PileupRead ctor, pileup.d:180-189
            if (rc) { if ((op & 15u) != 3u) started = true; else continue; }
            else { if (qc) qs += len; continue; }
        }
        if (!rc) { if (qc) qs += len; continue; }                         // pileup.d:215-217
        const uint32_t elen = len ? len : 1u;                             // pileup.d:196-203
        const uint32_t lo = max(cs, need_lo), hi = min(cs + elen, need_hi);
        if (lo < hi) {
            if (qc) {                                                     // M, =, X : sequence[query_offset], pileup.d:117-118
                for (uint32_t r0 = lo; r0 < hi; r0 += 64) {
                    const uint32_t qa = qs + (r0 - cs);                   // query offset of lane 0 in this step
                    const uint32_t qb = qa + min(hi - r0, 64u) - 1;
                    if ((qa >> 3) < win0 || (qb >> 3) >= win0 + 64) {     // slide the 512-base window
                        win0 = qa >> 3;
                        words = load_seq_words(a.s, a.stream_len, seq_off, seq_bytes, win0, lane);
                    }
                    const uint32_t q = qa + lane;
                    const uint32_t wv = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((q >> 3) - win0) << 2), (int)words);
                    if (r0 + lane < hi) {
                        uint32_t cls = 4;
                        if (q < l_seq) {
                            const uint32_t byte = (wv >> (((q >> 1) & 3u) * 8)) & 0xFFu;
                            cls = base_class((q & 1u) ? (byte & 0xFu) : (byte >> 4));   // read.d:368-373
                        } else atomicOr(a.status, ST_ESEQ);
                        L::add_cls(lds, cls, pos + r0 + lane - T0);
                    }
                }
            } else if (lane == 0) {                                       // D, N : '-' over [lo, hi), pileup.d:119-121
                atomicAdd(&lds[L::DELD + (pos + lo - T0)], 1u);
                atomicAdd(&lds[L::DELD + (pos + hi - T0)], 0xFFFFFFFFu);
            }
        }
        if (qc) qs += elen;
        cs += elen;
    }
    if (k == n_cig && cs < need_hi && lane == 0) atomicOr(a.status, ST_ECIGAR);   // assertCigarIndexIsValid, pileup.d:224-228
}

// Long read against the window: resume from the window's checkpoint and take the CIGAR 64 ops at a
// time with LANE = OP.  A wave prefix scan over the ops' column / query lengths gives every lane its
// op's first column and first query base; D/N ops are two adds on the difference array per lane; the
// M/=/X ops of the chunk are then stepped together (step t = base t of every lane's op), their bases
// looked up with ds_bpermute in a 1024-base window of packed SEQ held two dwords per lane.
// An op longer than 64 columns (rare in indel-dense long reads) is handed to the lanes-along-the-read
// walk of pile_read_coop's inner loop.
template <class L> __device__ __forceinline__ void pile_read_long(const K4Args& a, uint32_t* lds, uint32_t T0, uint32_t Tend, int lane,
                                               uint32_t pos, uint32_t covered, uint32_t n_cig, uint32_t l_seq, uint64_t cig_off, Ckpt ck) {
    const uint64_t seq_off = cig_off + 4ull * n_cig;
    const uint32_t seq_bytes = (l_seq + 1) / 2;
    const uint32_t need_hi = min(covered, Tend - pos), need_lo = T0 > pos ? T0 - pos : 0;
    if (ck.k == 0xFFFFFFFFu) { if (lane == 0 && need_lo < need_hi) atomicOr(a.status, ST_ECIGAR); return; }   // cursor ran off before this window
    uint32_t cs = ck.cs, qs = ck.qs;
    uint32_t c0 = ck.k;
    for (; c0 < n_cig && cs < need_hi; c0 += 64) {
        const bool active = c0 + lane < n_cig;
        const uint32_t op = active ? ld_u32(a.s + cig_off + 4ull * (c0 + lane)) : 0u;
        uint32_t rlen, qlen;
        op_lens(op, active, true, &rlen, &qlen);
        const uint32_t ir = wave_incl_u32(rlen, lane), iq = wave_incl_u32(qlen, lane);
        const uint32_t cs_b = cs + ir - rlen, qs_b = qs + iq - qlen;
        const uint32_t lo = max(cs_b, need_lo), hi = min(cs_b + rlen, need_hi);
        const bool in_win = rlen && lo < hi;
        const bool is_m = cig_qc(op) != 0;
        if (in_win && !is_m) {                                                   // D, N: '-' over [lo, hi), pileup.d:119-121
            atomicAdd(&lds[L::DELD + (pos + lo - T0)], 1u);
            atomicAdd(&lds[L::DELD + (pos + hi - T0)], 0xFFFFFFFFu);
        }
        const uint32_t mlen = (in_win && is_m) ? hi - lo : 0u;
        uint32_t maxlen = mlen;
        for (int d = 32; d >= 1; d >>= 1) maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, d, 64));
        if (maxlen) {
            const uint32_t q_first = qs_b + (lo - cs_b);                           // query offset of this lane's first base in the window
            // query span of the chunk's M bases
            uint32_t qmin = mlen ? q_first : 0xFFFFFFFFu, qmax = mlen ? q_first + mlen - 1 : 0u;
            for (int d = 32; d >= 1; d >>= 1) { qmin = min(qmin, (uint32_t)__shfl_xor((int)qmin, d, 64)); qmax = max(qmax, (uint32_t)__shfl_xor((int)qmax, d, 64)); }
            if (maxlen <= 64 && (qmax >> 3) - (qmin >> 3) < 128) {
                const uint32_t win0 = qmin >> 3;
                const uint32_t wa = load_seq_words(a.s, a.stream_len, seq_off, seq_bytes, win0, lane);
                const uint32_t wb = load_seq_words(a.s, a.stream_len, seq_off, seq_bytes, win0 + 64, lane);
                for (uint32_t t = 0; t < maxlen; ++t) {
                    const uint32_t q = q_first + t;
                    const uint32_t idx = (q >> 3) - win0;
                    const uint32_t va = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((idx & 63u) << 2), (int)wa);
                    const uint32_t vb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((idx & 63u) << 2), (int)wb);
                    if (t < mlen) {
                        uint32_t cls = 4;
                        if (q < l_seq) {
                            const uint32_t wv = idx < 64 ? va : vb;
                            const uint32_t byte = (wv >> (((q >> 1) & 3u) * 8)) & 0xFFu;
                            cls = base_class((q & 1u) ? (byte & 0xFu) : (byte >> 4));       // read.d:368-373
                        } else atomicOr(a.status, ST_ESEQ);
                        L::add_cls(lds, cls, pos + lo + t - T0);
                    }
                }
            } else {
                // long M ops: one op at a time with lanes along the read
                unsigned long long mm = __ballot(mlen != 0);
                while (mm) {
                    const int o = __ffsll((long long)mm) - 1;
                    mm &= mm - 1;
                    const uint32_t olo = rl(lo, o), ohi = rl(hi, o), oq = rl(q_first, o);
                    uint32_t win0 = 0xFFFFFFFFu, words = 0;
                    for (uint32_t r0 = olo; r0 < ohi; r0 += 64) {
                        const uint32_t qa = oq + (r0 - olo), qb = qa + min(ohi - r0, 64u) - 1;
                        if (win0 == 0xFFFFFFFFu || (qa >> 3) < win0 || (qb >> 3) >= win0 + 64) { win0 = qa >> 3; words = load_seq_words(a.s, a.stream_len, seq_off, seq_bytes, win0, lane); }
                        const uint32_t q = qa + lane;
                        const uint32_t wv = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((q >> 3) - win0) << 2), (int)words);
                        if (r0 + lane < ohi) {
                            uint32_t cls = 4;
                            if (q < l_seq) { const uint32_t byte = (wv >> (((q >> 1) & 3u) * 8)) & 0xFFu; cls = base_class((q & 1u) ? (byte & 0xFu) : (byte >> 4)); }
                            else atomicOr(a.status, ST_ESEQ);
                            L::add_cls(lds, cls, pos + r0 + lane - T0);
                        }
                    }
                }
            }
        }
        cs += __shfl(ir, 63, 64); qs += __shfl(iq, 63, 64);
    }
    if (c0 >= n_cig && cs < need_hi && lane == 0) atomicOr(a.status, ST_ECIGAR);      // assertCigarIndexIsValid, pileup.d:224-228
}

template <class L> __global__ __launch_bounds__(L::T) void k4_tile(K4Args a, int deep_pass) {
    constexpr int K4_T = L::T, LIST = L::LIST, SLIST = L::SLIST, LCAP = L::LCAP, HIST = L::HIST;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_raw[];
    uint32_t* const lds = lds_raw + LDS_PAD;      // counters of masked-off bases may land up to 7 words below cnt[0][0]
    __shared__ uint32_t s_n;           // entries in the LDS list
    __shared__ uint32_t s_nslow;       // entries in the second list
    __shared__ uint32_t wsum[K4_T / 64];
    // The general kernel: it runs only the windows the accumulate path (k4_plan / k4_acc below) leaves on the list --
    // windows touched by a long read or an odd CIGAR, windows too deep for a descriptor slab.  A window whose
    // candidate range could overflow 16-bit counters takes the 32-bit layout, every other one the 16-bit layout.
  for (uint64_t it = blockIdx.x; it < *a.n_deep; it += gridDim.x) {
    (void)deep_pass;
    const uint64_t tile = a.deep[it];
    const TileB tb = a.tiles[tile];                      // window geometry + record range, precomputed by k4_bounds
    if (L::P16 != ((tb.pad & 1u) == 0u)) continue;       // (uniform: no barrier has been passed yet in this iteration)
    const uint32_t T0 = tb.T0, Tend = tb.Tend;
    for (int i = threadIdx.x; i < LIST; i += K4_T) lds[i] = 0;
    const uint64_t i_lo = tb.i_lo, i_hi = tb.i_hi;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t* list = lds + LIST;
    uint32_t* slist = lds + SLIST;

    for (uint64_t scan = i_lo; scan < i_hi;) {
        if (threadIdx.x == 0) { s_n = 0; s_nslow = 0; }
        __syncthreads();
        // ---- scan: compact the reads that overlap the window
        for (bool more = true; more;) {
            // four rounds of candidate rows are requested before the first is looked at: one HBM round trip, not four
            uint4 xs[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint64_t i = scan + (uint64_t)r * K4_T + threadIdx.x;
                xs[r] = make_uint4(0, 0, 0, 0);
                if (i < i_hi) xs[r] = *reinterpret_cast<const uint4*>(a.rec + i);   // ref_id, pos, end_pos, bin_mq_nl
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (!more) break;
                const uint64_t i = scan + threadIdx.x;
                const uint4 x = xs[r];
                const bool valid = i < i_hi && x.z != x.y && x.z > T0 && x.y < Tend;   // basesCovered() > 0 and overlaps the window
                if (valid && x.y >= T0) L::add_nst(lds, x.y - T0);      // n_starting_here, pileup.d:386,422
                const unsigned long long m = __ballot(valid);
                uint32_t slot = 0;
                if (lane == 0 && m) slot = atomicAdd(&s_n, (uint32_t)__popcll(m));
                slot = rl(slot, 0);
                if (valid) list[slot + __popcll(m & ((1ull << lane) - 1))] = (uint32_t)(i - i_lo);
                scan += K4_T;
                __syncthreads();
                const uint32_t sn = s_n;                              // every wave must take the same decision: read the counter ...
                __syncthreads();                                      // ... before a faster wave's next round adds to it
                if (scan >= i_hi || sn > LCAP - K4_T) more = false;
            }
        }
        const uint32_t n_list = s_n;
        // ---- process: 64 listed reads per wave, lane = read
#ifdef BGP_K4_NOPROC
        for (uint32_t b0 = n_list; b0 < n_list; b0 += (K4_T / 64) * 64) {
#else
        for (uint32_t b0 = (uint32_t)wave * 64; b0 < n_list; b0 += (K4_T / 64) * 64) {
#endif
            const bool have = b0 + lane < n_list;
            uint32_t pos = 0, covered = 0, n_cig = 0, l_seq = 0, o0 = 0, o1 = 0, o2 = 0, o3 = 0;
            uint64_t cig_off = 0;
            uint4 w0 = {}, w1 = {}, w2 = {}, w3 = {}, w4 = {};
            bool fast = false;
            uint32_t qsum = 0;
            if (have) {
                const uint4* rp = reinterpret_cast<const uint4*>(a.rec + i_lo + list[b0 + lane]);
                const uint4 x = rp[0], y = rp[1];             // ref_id,pos,end_pos,bin_mq_nl | flag_nc,l_seq,rec_off
                pos = x.y; covered = x.z - x.y; n_cig = y.x & 0xFFFFu; l_seq = y.y;
                cig_off = (((uint64_t)y.w << 32) | y.z) + 36 + (x.w & 0xFFu);
                const uint64_t seq_off = cig_off + 4ull * n_cig;
                fast = n_cig >= 1 && n_cig <= 4 && l_seq <= FAST_LSEQ && seq_off + 80 <= a.stream_len;
                if (fast) {
                    uint4 c4; __builtin_memcpy(&c4, a.s + cig_off, 16);   // <= 4 ops in one load (SEQ follows: in bounds)
                    o0 = c4.x;
                    if (n_cig > 1) o1 = c4.y;
                    if (n_cig > 2) o2 = c4.z;
                    if (n_cig > 3) o3 = c4.w;
                    const uint8_t* sp = a.s + seq_off;
                    __builtin_memcpy(&w0, sp, 16); __builtin_memcpy(&w1, sp + 16, 16); __builtin_memcpy(&w2, sp + 32, 16);
                    __builtin_memcpy(&w3, sp + 48, 16); __builtin_memcpy(&w4, sp + 64, 16);
                    // the mould: no zero-length reference-consuming op, first reference-consuming op is not N,
                    // the CIGAR does not consume more query than SEQ holds
                    bool seen_ref = false, ok = true;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t op = k == 0 ? o0 : k == 1 ? o1 : k == 2 ? o2 : o3;
                        if ((uint32_t)k < n_cig) {
                            const uint32_t len = op >> 4;
                            if (cig_rc(op)) { if (len == 0 || (!seen_ref && (op & 15u) == 3u)) ok = false; seen_ref = true; }
                            if (cig_qc(op)) qsum += len;
                        }
                    }
                    fast = ok && qsum <= l_seq;
                }
            }
            // lane-parallel walk: all lanes step the same query index q.  Each CIGAR op slot k of a lane
            // becomes (if it is M/=/X) one window-clipped query interval [sq_k, sq_k + sl_k) whose bases
            // land on LDS word sb_k + q; D/N ops are two adds on the difference array, done here, once.
            uint32_t maxq = fast ? qsum : 0;
            for (int d = 32; d >= 1; d >>= 1) maxq = max(maxq, (uint32_t)__shfl_xor((int)maxq, d, 64));
            if (maxq || __ballot(fast)) {
                const uint32_t need_hi = min(covered, Tend - pos), need_lo = T0 > pos ? T0 - pos : 0;
                const uint32_t colbase = pos - T0;                         // wraps for pos < T0; sums below are exact mod 2^32
                // slots are COMPACT: slot m = the m-th M/=/X interval of the lane that touches the window
                // (ordinary reads have one or two), so the hot loop below tests as few slots as possible
                uint32_t sq[4] = {0, 0, 0, 0}, sl[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0};
                {
                    uint32_t q = 0, p = 0, m = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t op = k == 0 ? o0 : k == 1 ? o1 : k == 2 ? o2 : o3;
                        if (fast && (uint32_t)k < n_cig) {
                            const uint32_t len = op >> 4, qc = cig_qc(op), rc = cig_rc(op);
                            if (rc) {
                                const uint32_t lo = max(p, need_lo), hi = min(p + len, need_hi);   // columns of this op inside the window
                                if (lo < hi) {
                                    if (qc) {                                                       // pileup.d:117-118
                                        const uint32_t vq = q + (lo - p), vl = hi - lo, vb = (colbase + p - q) * 4;
#pragma unroll
                                        for (int j = 0; j <= k; ++j) if (m == (uint32_t)j) { sq[j] = vq; sl[j] = vl; sb[j] = vb; }
                                        ++m;
                                    } else { atomicAdd(&lds[L::DELD + colbase + lo], 1u); atomicAdd(&lds[L::DELD + colbase + hi], 0xFFFFFFFFu); }   // pileup.d:119-121
                                }
                                p += len;
                            }
                            if (qc) q += len;
                        }
                    }
                }
                // The hot loop works a packed-SEQ dword (8 bases) at a time, wi wave-uniform.  Per dword and lane:
                // the 8 nt16 codes become 8 class nibbles with word-parallel bit tricks, the dword is given to ONE
                // interval slot -- the last one that starts at or before its last base -- whose clip against the
                // dword is an 8-bit validity mask, and every base then costs three VALU ops (class extract,
                // address, validity bit) and one ds_add (invalid bases add 0 within 7 words of a valid counter:
                // the tile's LDS starts 8 words into the allocation for that).  The bases a dword holds of an
                // EARLIER slot (an indel inside the dword) are done afterwards, one extra dword step per slot.
                const bool use1 = __ballot(sl[1] != 0) != 0, use2 = __ballot(sl[2] != 0) != 0, use3 = __ballot(sl[3] != 0) != 0;
                const uint32_t nwords = (maxq + 7) >> 3;
                auto dword_step = [&](uint32_t cur, uint32_t q0, uint32_t vq, uint32_t vl, uint32_t vb) {
                    // nibbles into base order (SEQ keeps the first base of a byte in the high nibble, read.d:368-373)
                    const uint32_t x = ((cur & 0x0F0F0F0Fu) << 4) | ((cur >> 4) & 0x0F0F0F0Fu);
                    // nt16 -> A,C,G,T,other per nibble: one-hot codes 1,2,4,8 -> 0,1,2,3 (= b1 + 2 b2 + 3 b3), anything else -> 4
                    const uint32_t b0 = x & 0x11111111u, b1 = (x >> 1) & 0x11111111u, b2 = (x >> 2) & 0x11111111u, b3 = (x >> 3) & 0x11111111u;
                    const uint32_t e = (b0 + b1 + b2 + b3) ^ 0x11111111u;                   // nibble == 0 <=> exactly one bit set
                    const uint32_t f = (e | (e >> 1) | (e >> 2)) & 0x11111111u;             // 1 where the code is not one-hot
                    const uint32_t m7 = f * 7u;
                    const uint32_t cw = ((b1 + (b2 << 1) + b3 * 3u) & ~m7) | (f << 2);
                    // bases [lo, hi) of the dword lie in the slot's query interval [vq, vq + vl)
                    const int32_t d0 = (int32_t)(vq - q0), d1 = (int32_t)(vq + vl - q0);
                    const uint32_t lo = (uint32_t)min(max(d0, 0), 8), hi = (uint32_t)min(max(d1, 0), 8);
                    const uint32_t vm = hi > lo ? (((1u << (hi - lo)) - 1u) << lo) : 0u;
                    const uint32_t bq = vm ? vb + q0 * 4u : 0u;                             // byte address of cnt[0][column of base 0]
                    char* const l8 = reinterpret_cast<char*>(lds);
                    if (L::P16) {
                        // two positions to a word.  With C the column of base 0: base b sits in word (C >> 1) + ((b + (C & 1)) >> 1),
                        // half (C + b) & 1 -- the word offset is an immediate for even b and immediate + 4 * (C & 1) for odd b,
                        // the half alternates; both depend on the lane only through C's parity, fixed once per dword
                        const int32_t w0b = ((int32_t)bq >> 1) & ~3;                      // byte offset of word (C >> 1)
                        const uint32_t par4 = bq & 4u;                                    // 4 * (C & 1)
                        const uint32_t sh_even = par4 << 2, sh_odd = sh_even ^ 16u;       // 16 * half
                        char* const pe = l8 + w0b;
                        char* const po = pe + par4;
#pragma unroll
                        for (int b = 0; b < 8; ++b) {
                            const uint32_t cls = __builtin_amdgcn_ubfe(cw, 4 * b, 3);
                            atomicAdd(reinterpret_cast<uint32_t*>(((b & 1) ? po : pe) + (int32_t)(cls << L::CSH) + 4 * (b >> 1)), __builtin_amdgcn_ubfe(vm, b, 1) << ((b & 1) ? sh_odd : sh_even));
                        }
                    } else {
#pragma unroll
                        for (int b = 0; b < 8; ++b) {
                            const uint32_t cls = __builtin_amdgcn_ubfe(cw, 4 * b, 3);
                            atomicAdd(reinterpret_cast<uint32_t*>(l8 + (int32_t)((cls << L::CLS_SHIFT) + bq) + 4 * b), __builtin_amdgcn_ubfe(vm, b, 1));
                        }
                    }
                };
                // the slot a dword is given to only ever moves forward: keep the current one and a queue of the next ones
                // (an unused slot never starts), advance when the next one starts at or before the dword's last base
                uint32_t cq = sq[0], cl = sl[0], cb = sb[0];
                uint32_t n1q = sl[1] ? sq[1] : 0xFFFFFFFFu, n1l = sl[1], n1b = sb[1];
                uint32_t n2q = sl[2] ? sq[2] : 0xFFFFFFFFu, n2l = sl[2], n2b = sb[2];
                uint32_t n3q = sl[3] ? sq[3] : 0xFFFFFFFFu, n3l = sl[3], n3b = sb[3];
                for (uint32_t wi = 0; wi < nwords; ++wi) {
                    uint32_t cur;                                           // dword wi of this lane's packed SEQ (wi is wave-uniform)
                    switch (wi) {
                        case 0: cur = w0.x; break; case 1: cur = w0.y; break; case 2: cur = w0.z; break; case 3: cur = w0.w; break;
                        case 4: cur = w1.x; break; case 5: cur = w1.y; break; case 6: cur = w1.z; break; case 7: cur = w1.w; break;
                        case 8: cur = w2.x; break; case 9: cur = w2.y; break; case 10: cur = w2.z; break; case 11: cur = w2.w; break;
                        case 12: cur = w3.x; break; case 13: cur = w3.y; break; case 14: cur = w3.z; break; case 15: cur = w3.w; break;
                        case 16: cur = w4.x; break; case 17: cur = w4.y; break; case 18: cur = w4.z; break; default: cur = w4.w; break;
                    }
                    const uint32_t q0 = wi * 8, qlast = q0 + 7;
                    if (use1) {                                             // as many advances per dword as a wave can need: 1, 2 or 3
                        const bool t = n1q <= qlast;
                        cq = t ? n1q : cq; cl = t ? n1l : cl; cb = t ? n1b : cb;
                        if (use2) { n1q = t ? n2q : n1q; n1l = t ? n2l : n1l; n1b = t ? n2b : n1b; n2q = t ? n3q : n2q; n2l = t ? n3l : n2l; n2b = t ? n3b : n2b; n3q = t ? 0xFFFFFFFFu : n3q; }
                        else n1q = t ? 0xFFFFFFFFu : n1q;
                    }
                    if (use2) {
                        const bool t = n1q <= qlast;
                        cq = t ? n1q : cq; cl = t ? n1l : cl; cb = t ? n1b : cb;
                        n1q = t ? n2q : n1q; n1l = t ? n2l : n1l; n1b = t ? n2b : n1b; n2q = t ? 0xFFFFFFFFu : n2q;
                    }
                    if (use3) { const bool t = n1q <= qlast; cq = t ? n1q : cq; cl = t ? n1l : cl; cb = t ? n1b : cb; n1q = t ? 0xFFFFFFFFu : n1q; }
#ifndef BGP_K4_NOQ
                    dword_step(cur, q0, cq, cl, cb);
#endif
                }
                // what a slot's first dword holds of the slot before it
                auto fixup = [&](uint32_t sqm, uint32_t slm, uint32_t pq, uint32_t pl, uint32_t pb) {
                    const uint32_t wf = slm ? sqm >> 3 : 0u;
                    // lane-varying pick of one of the 20 preloaded dwords: a select tree on the index BITS (written
                    // as element compares the compiler turns it into a scratch-indexed vector extract)
                    const bool k0 = wf & 1u, k1 = wf & 2u;
                    auto pick4 = [&](const uint4& w) { const uint32_t a = k0 ? w.y : w.x, b = k0 ? w.w : w.z; return k1 ? b : a; };
                    const uint32_t p0 = pick4(w0), p1 = pick4(w1), p2 = pick4(w2), p3 = pick4(w3), p4 = pick4(w4);
                    const uint32_t g = wf >> 2;
                    const uint32_t lo01 = (g & 1u) ? p1 : p0, lo23 = (g & 1u) ? p3 : p2;
                    const uint32_t cur = g >= 4 ? p4 : ((g & 2u) ? lo23 : lo01);
                    dword_step(cur, wf * 8, pq, slm ? pl : 0u, pb);
                };
                if (use1) fixup(sq[1], sl[1], sq[0], sl[0], sb[0]);
                if (use2) fixup(sq[2], sl[2], sq[1], sl[1], sb[1]);
                if (use3) fixup(sq[3], sl[3], sq[2], sl[2], sb[2]);
            }
            // everything else, one read at a time with lanes along the read
            // everything else goes to a second list so that ALL waves share it (a window's long reads
            // would otherwise sit in one batch and be walked by one wave while seven idle)
            const bool is_slow = have && !fast;
            const unsigned long long sm = __ballot(is_slow);
            if (sm) {
                uint32_t slot = 0;
                if (lane == 0) slot = atomicAdd(&s_nslow, (uint32_t)__popcll(sm));
                slot = rl(slot, 0);
                if (is_slow) slist[slot + __popcll(sm & ((1ull << lane) - 1))] = list[b0 + lane];
            }
        }
        __syncthreads();
        // one read at a time per wave, whole wave on the read
        const uint32_t n_slow = s_nslow;
        for (uint32_t j = (uint32_t)wave; j < n_slow; j += K4_T / 64) {
            const uint64_t ri = i_lo + slist[j];
            const uint4* rp = reinterpret_cast<const uint4*>(a.rec + ri);
            const uint4 x = rp[0], y = rp[1];
            const uint32_t rpos = rl(x.y, 0), rcov = rl(x.z - x.y, 0), rncig = rl(y.x & 0xFFFFu, 0), rlseq = rl(y.y, 0);
            const uint64_t co = (((uint64_t)rl(y.w, 0) << 32) | rl(y.z, 0)) + 36 + rl(x.w & 0xFFu, 0);
            if (a.ck && wants_ckpt(rncig, rlseq)) {
                const uint64_t wj = (uint64_t)(T0 - tb.run_start) / L::WIN - ((uint64_t)rpos - tb.run_start) / L::WIN;   // window index within the read
                const Ckpt ck = a.ck[a.ck_off[ri] + wj];
                pile_read_long<L>(a, lds, T0, Tend, lane, rpos, rcov, rncig, rlseq, co, ck);
            } else
                pile_read_coop<L>(a, lds, T0, Tend, lane, rpos, rcov, rncig, rlseq, co);
        }
        __syncthreads();
    }
    // ---- flush: prefix-sum the '-' difference array, then 32 contiguous bytes per position
    const uint32_t np = Tend - T0;
    {
        constexpr int PER = L::WIN / K4_T;
        const uint32_t p0 = threadIdx.x * PER;
        uint32_t loc[PER]; uint32_t run_sum = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) { run_sum += lds[L::DELD + p0 + j]; loc[j] = run_sum; }
        uint32_t inc = run_sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t carry = 0;
        for (int w = 0; w < wave; ++w) carry += wsum[w];
        const uint32_t ex = carry + inc - run_sum;
#pragma unroll
        for (int j = 0; j < PER; ++j) lds[L::DELD + p0 + j] = ex + loc[j];
        __syncthreads();
    }
    bgp_column* out = a.cols + tb.col_out;
    uint64_t ncol = 0, scov = 0; uint32_t mx = 0;
    for (uint32_t p = threadIdx.x; p < np; p += K4_T) {
        const uint32_t A = L::get_cls(lds, 0, p), C = L::get_cls(lds, 1, p), G = L::get_cls(lds, 2, p), Tt = L::get_cls(lds, 3, p), O = L::get_cls(lds, 4, p), D = lds[L::DELD + p], S = L::get_nst(lds, p);
        const uint32_t cov = A + C + G + Tt + O + D;
        uint4* o4 = reinterpret_cast<uint4*>(out + p);
#ifndef BGP_K4_NOFLUSH
        o4[0] = make_uint4(cov, S, A, C);
        o4[1] = make_uint4(G, Tt, O, D);
#else
        if (cov == 0xFFFFFFFFu) o4[0] = make_uint4(cov, S, A, C);
#endif
        ncol += cov > 0; scov += cov; mx = max(mx, cov);
        if (cov < HB) atomicAdd(&lds[HIST + cov], 1u);
        else atomicAdd(&a.hist[min(cov, (uint32_t)BGP_DEPTH_BINS - 1)], 1ull);
    }
    // per-window totals go to a per-window slot (same-address global atomics from every wave cost
    // more than the whole rest of the kernel); k4_stats folds the slots afterwards
    uint64_t nst = 0;
    for (uint32_t p = threadIdx.x; p < np; p += K4_T) nst += L::get_nst(lds, p);
    for (int d = 32; d >= 1; d >>= 1) { ncol += __shfl_down(ncol, d, 64); scov += __shfl_down(scov, d, 64); nst += __shfl_down(nst, d, 64); mx = max(mx, (uint32_t)__shfl_down(mx, d, 64)); }
    __shared__ unsigned long long part[K4_T / 64][4];
    if (lane == 0) { part[wave][0] = ncol; part[wave][1] = scov; part[wave][2] = mx; part[wave][3] = nst; }
    __syncthreads();
    if (threadIdx.x < 4) {
        unsigned long long v = 0;
        for (int w = 0; w < K4_T / 64; ++w) v = threadIdx.x == 2 ? max(v, part[w][2]) : v + part[w][threadIdx.x];
        a.tile_stats[tile * 4 + threadIdx.x] = v;
    }
    __syncthreads();
#ifndef BGP_K4_NOHIST
    for (int b = threadIdx.x; b < HB; b += K4_T) { const uint32_t h = lds[HIST + b]; if (h) atomicAdd(&a.hist[b], (unsigned long long)h); }
#endif
    __syncthreads();                                     // the deep pass reuses the tile's LDS for its next window
  }
}

// a stream left to the general kernel entirely: every window is on its list
__global__ __launch_bounds__(256) void k4_all_listed(uint32_t* deep, uint32_t* n_deep, uint32_t n_tiles) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_tiles) deep[i] = i;
    if (i == 0) *n_deep = n_tiles;
}

// folds the per-window totals: [0] columns with coverage, [1] sum of coverage, [2] max coverage, [4] reads used
__global__ __launch_bounds__(256) void k4_stats(const unsigned long long* ts, uint64_t n_tiles, unsigned long long* pstats) {
    unsigned long long c = 0, s = 0, m = 0, u = 0;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_tiles; t += (uint64_t)gridDim.x * blockDim.x) {
        c += ts[t * 4]; s += ts[t * 4 + 1]; m = max(m, ts[t * 4 + 2]); u += ts[t * 4 + 3];
    }
    for (int d = 32; d >= 1; d >>= 1) { c += __shfl_down(c, d, 64); s += __shfl_down(s, d, 64); u += __shfl_down(u, d, 64); m = max(m, (unsigned long long)__shfl_down(m, d, 64)); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&pstats[0], c); atomicAdd(&pstats[1], s); atomicMax(&pstats[2], m); atomicAdd(&pstats[4], u); }
}

// ---------------------------------------------------------------------------------------------- K4: plan + accumulate
//
// The same closed form (file header) without a window ever scanning a candidate range.  Two kernels:
//
//   k4_plan   one lane per READ, one pass over the record table and the first-four-ops side array K1 leaves
//             (the ~1000-row halo a window had to scan because of N-skip reads is gone).  An ordinary
//             read (1..4 ops, l_seq <= 255, first reference-consuming op M/=/X, nothing zero-length) becomes 8-byte
//             DESCRIPTORS appended to the slabs of the windows it touches: one SEG per (M/=/X op, window) --
//             where the bases are in the stream, where they land in the window -- and one DEL per (D/N op, edge
//             window); windows an N/D op covers completely get +1 through a difference array over windows.  A long read
//             whose CIGAR walk (k4_ck_build) found it well formed marks the windows it touches (k4_acc2 walks its ops there,
//             from the window's checkpoint).  Any other read marks the windows it overlaps for the general kernel
//             (k4_tile above).
//   k4_acc2   (further down) takes the windows' slabs a PIECE of a descriptor per lane and adds the bases to byte-wide
//             counters in LDS, eight positions a step.  Its predecessors -- round 2's k4_exec (class bytes staged in LDS, every
//             POSITION gathering its bases: 10.3 ms on cfg3) and round 3's k4_acc (lanes along the reads, an LDS atomic a base:
//             8.6 ms) -- are gone; LABNOTES.md keeps their measurements.
//
// Descriptor (64 bits): [1:0] kind (0 none, 1 SEG, 2 DEL)
//   SEG  [2] the read starts here (n_starting_here)  [13:3] first position in the window  [21:14] bases (1..255)
//        [22] the first base is the LOW nibble of its byte  [63:23] stream offset of that byte
//   DEL  [13:3] first position  [25:14] one past the last position (<= W)
//   A long read (more than four ops or 255 bases) whose CIGAR walk found it well formed gets no descriptors: it marks the windows it
//   touches, and k4_acc2 walks its ops there, 64 a step from the window's checkpoint, an op a lane.
// sum / max over the 64 lanes of a wave with DPP (row reductions, then row_bcast15 / row_bcast31); result in every lane
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    v += BGP_DPP(v, 0xB1, 0xF); v += BGP_DPP(v, 0x4E, 0xF); v += BGP_DPP(v, 0x141, 0xF); v += BGP_DPP(v, 0x140, 0xF);
    v += BGP_DPP(v, 0x142, 0xA); v += BGP_DPP(v, 0x143, 0xC);
    return rl(v, 63);
}
// inclusive prefix sum over the 64 lanes: shifts inside the rows of 16, then the rows' totals carried over
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t v) {
    v += BGP_DPP(v, 0x111, 0xF); v += BGP_DPP(v, 0x112, 0xF); v += BGP_DPP(v, 0x114, 0xF); v += BGP_DPP(v, 0x118, 0xF);
    v += BGP_DPP(v, 0x142, 0xA); v += BGP_DPP(v, 0x143, 0xC);
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = max(v, BGP_DPP(v, 0xB1, 0xF)); v = max(v, BGP_DPP(v, 0x4E, 0xF)); v = max(v, BGP_DPP(v, 0x141, 0xF)); v = max(v, BGP_DPP(v, 0x140, 0xF));
    v = max(v, BGP_DPP(v, 0x142, 0xA)); v = max(v, BGP_DPP(v, 0x143, 0xC));
    return rl(v, 63);
}

#ifndef BGP_PLAN_ABL
#define BGP_PLAN_ABL 0                        // timing experiments on k4_plan: 1 no '-' window atomics, 2 no far-slot atomics, 4 no descriptor stores
#endif
constexpr uint32_t SLAB_CAP = W;              // descriptors a window's slab holds (~3 x the ordinary count at 30-fold coverage); more: the window goes to k4_tile

struct PlanArgs {
    const bgp_record* rec; const uint4* cig4; const uint8_t* s; uint64_t stream_len, n;
    const RunDev* runs; uint32_t n_runs; uint64_t n_tiles;
    unsigned long long* slab; uint32_t slab_cap; uint32_t* wcount; int32_t* wdel; int32_t* wslow;
    const uint64_t* ck_off; const Ckpt* ck;       // long reads' checkpoints on this tiling (null: none, or long reads are left to the general kernel)
    uint32_t* wlong;                              // ... and the windows a well-formed long read touches (k4_acc2 walks its CIGAR there, from the window's checkpoint)
    int batches;                                  // batches of 64 reads a wave plans
    const uint8_t* rcls;                          // k4_plan_rest: the records' classes (k12_lds); null for k4_plan
};

#ifndef BGP_PLAN_NB
#define BGP_PLAN_NB 4
#endif
constexpr int PLAN_NB = BGP_PLAN_NB;          // batches of 64 reads a wave plans, the next batch's rows requested before the current one is worked on
// One batch: lane `lane` plans record ii (x, y: its row; c4: its first four ops); lanes with !in work on a record a second time and keep
// nothing.  i0 = the record of lane 0 (the smallest of the batch).  The wave's run is cached in r_* (fresh: look it up).
template <bool SPARSE> __device__ __forceinline__ void plan_batch(const PlanArgs& a, int lane, bool in, uint64_t ii, uint64_t i0, uint4 x, uint4 y, uint4 c4, bool fresh, uint64_t& r_end, uint64_t& r_start_pos, uint64_t& r_tile_base, uint32_t* wlds = nullptr) {
    // Straight-line, selects instead of joins, descriptors in 32-bit halves: written with a branch per case and per op this
    // kernel spent 600 VALU instructions a wave, four in ten of them register moves where the branches met -- and VALU
    // issue is what bounds it (SQ counters: 78 % of its time).
    constexpr uint32_t SH = (uint32_t)__builtin_ctz((unsigned)W);             // log2(W)
    unsigned long long dsc[8]; uint32_t win[8];
    // the reference run of a record: reads are sorted, so a wave nearly always lies inside one run -- the first lane's
    // run is looked up once (a binary search of dependent loads) and only lanes beyond its end search for themselves
    uint64_t w_start_pos, w_tile_base;                    // (the wave's run; lanes beyond its end look theirs up below)
    if (fresh || i0 >= r_end) { const RunDev rd = a.runs[run_of_record(a.runs, a.n_runs, i0)]; r_end = rd.end_record; r_start_pos = rd.start_pos; r_tile_base = rd.tile_base; }
    w_start_pos = r_start_pos; w_tile_base = r_tile_base;
    const bool covered = in && x.z != x.y;                // basesCovered() > 0, pileup.d:481,510
    if (covered && ii >= r_end) { const RunDev& rd = a.runs[run_of_record(a.runs, a.n_runs, ii)]; w_start_pos = rd.start_pos; w_tile_base = rd.tile_base; }
    const uint32_t rel0 = x.y - (uint32_t)w_start_pos;
    const uint32_t tb = (uint32_t)w_tile_base;
    const uint32_t wf0 = tb + (rel0 >> SH), wl = tb + ((x.z - 1 - (uint32_t)w_start_pos) >> SH);
    const uint32_t n_cig = y.x & 0xFFFFu, l_seq = y.y;
    const uint64_t cig_off = (((uint64_t)y.w << 32) | y.z) + 36 + (x.w & 0xFFu);
    const uint64_t seq_off = cig_off + 4ull * n_cig;
    // (k4_plan_rest, behind plan rows: reads of up to BGP_PLAN_MAX_OPS ops, four at a time, the ops behind the fourth from the stream -- such a stream
    // has no other way for them: framing has vouched that nothing in it wants checkpoints)
    constexpr uint32_t MAX_OPS = SPARSE ? BGP_PLAN_MAX_OPS : 4u;
    constexpr int NG = SPARSE ? (int)(BGP_PLAN_MAX_OPS / 4u) : 1;
    bool simple = covered && n_cig >= 1 && n_cig <= MAX_OPS && l_seq >= 1 && l_seq <= (uint32_t)FAST_SEG && seq_off + (l_seq + 1) / 2 + 16 <= a.stream_len;   // (staging reads 16 bytes per lane)
    uint32_t ops[4];
    if (a.cig4) { ops[0] = c4.x; ops[1] = c4.y; ops[2] = c4.z; ops[3] = c4.w; }
    else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { ops[k] = 0; if (simple && (uint32_t)k < n_cig) ops[k] = ld_u32(a.s + cig_off + 4 * k); }
    }
    {   // the mould: no zero-length reference-consuming op, the first reference-consuming op is M/=/X (so the read's first
        // column is its first staged base), the CIGAR does not consume more query than SEQ holds
        bool seen_ref = false, bad = false; uint32_t qsum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t op = ops[k], len = op >> 4;
            const bool on = (uint32_t)k < n_cig, rc = on && cig_rc(op), qc = on && cig_qc(op);
            bad |= rc && (len == 0 || (!seen_ref && !qc));
            seen_ref |= rc;
            qsum += qc ? len : 0u;
        }
        for (int g = 1; g < NG; ++g) {                    // (SPARSE only) the ops behind the fourth
            const bool more = simple && n_cig > 4u * (uint32_t)g;
            if (!__builtin_amdgcn_ballot_w64(more)) break;
            uint4 og = make_uint4(0u, 0u, 0u, 0u);
            if (more) __builtin_memcpy(&og, a.s + cig_off + 16ull * (uint64_t)g, 16);      // (inside the record: the CIGAR lies in front of SEQ, and seq_off + 16 <= stream_len)
            const uint32_t o4[4] = {og.x, og.y, og.z, og.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t op = o4[k], len = op >> 4;
                const bool on = more && 4u * (uint32_t)g + (uint32_t)k < n_cig, rc = on && cig_rc(op), qc = on && cig_qc(op);
                bad |= rc && (len == 0 || (!seen_ref && !qc));
                seen_ref |= rc;
                qsum += qc ? len : 0u;
            }
        }
        simple = simple && !bad && qsum <= l_seq;
    }
    if (covered && !simple) {
        if (a.ck && wants_ckpt(n_cig, l_seq) && a.ck[a.ck_off[ii]].pad == 0) {
            // a long read that fits the mould (k4_ck_build walked its CIGAR): no descriptors -- k4_acc2 walks its ops window by window
            for (uint32_t w = wf0; w <= wl; ++w) atomicOr(&a.wlong[w], 1u);
        } else {                                          // left to the general kernel, with every window it overlaps
            atomicAdd(&a.wslow[wf0], 1); atomicAdd(&a.wslow[wl + 1], -1);
        }
    }
    const uint32_t wf = simple ? wf0 : 0xFFFFFFFFu;
    bool any = simple;
    uint32_t p = 0, q = 0;                                // columns / query bases before the op (file header: cstart, qstart)
    for (int g = 0; g < NG; ++g) {                        // (one round; SPARSE: a round per four ops)
    if (g) {
        any = simple && n_cig > 4u * (uint32_t)g;
        if (!__builtin_amdgcn_ballot_w64(any)) break;
        uint4 og = make_uint4(0u, 0u, 0u, 0u);
        if (any) __builtin_memcpy(&og, a.s + cig_off + 16ull * (uint64_t)g, 16);
        ops[0] = og.x; ops[1] = og.y; ops[2] = og.z; ops[3] = og.w;
    }
    {
        const uint32_t n_ops = simple ? n_cig : 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t op = ops[k], len = op >> 4;
            const bool on = 4u * (uint32_t)g + (uint32_t)k < n_ops, qc = on && cig_qc(op), rc = on && cig_rc(op);
            const uint32_t a0 = rel0 + p, a1 = a0 + len;                             // run-relative columns [a0, a1) of the op
            const uint32_t w0 = a0 >> SH, w1 = (a1 - 1) >> SH;
            const uint32_t lo0 = a0 & (uint32_t)(W - 1), n0 = min(len, (uint32_t)W - lo0);
            const bool two = rc && w1 > w0;
            // M, =, X (pileup.d:117-118; len <= 255 < W: at most two windows): where the bases are; D, N ('-', :119-121): where the op ends
            const uint32_t q1 = q + n0;
            const uint64_t s0 = (seq_off + (q >> 1)) << 23, s1 = (seq_off + (q1 >> 1)) << 23;
            const uint32_t seg0 = 1u | (p == 0 ? 4u : 0u) | (lo0 << 3) | (n0 << 14) | ((q & 1u) << 22) | (uint32_t)s0;
            const uint32_t seg1 = 1u | ((len - n0) << 14) | ((q1 & 1u) << 22) | (uint32_t)s1;
            const uint32_t del0 = 2u | (lo0 << 3) | ((lo0 + n0) << 14), del1 = 2u | ((a1 - (w1 << SH)) << 14);
            dsc[2 * k] = qc ? ((unsigned long long)(uint32_t)(s0 >> 32) << 32) | seg0 : (unsigned long long)del0;
            dsc[2 * k + 1] = qc ? ((unsigned long long)(uint32_t)(s1 >> 32) << 32) | seg1 : (unsigned long long)del1;
            win[2 * k] = rc ? tb + w0 : 0xFFFFFFFFu;
            win[2 * k + 1] = two ? tb + w1 : 0xFFFFFFFFu;
#if !(BGP_PLAN_ABL & 1)
            if (two && !qc && w1 - w0 >= 2) { atomicAdd(&a.wdel[tb + w0 + 1], 1); atomicAdd(&a.wdel[tb + w1], -1); }   // windows covered end to end
#endif
            p += rc ? len : 0u;
            q += qc ? len : 0u;
        }
    }
    const unsigned long long am = __ballot(any);
    if (!am) { if (SPARSE) continue; return; }
    if (SPARSE) {
        // The batch is the few reads in a thousand that k4_plan_rest lists -- N skips mostly: their descriptors spread over a dozen windows
        // (where the reads start, and up to a skip's length further on), a few lanes each.  Claimed one by one that was 1.8 x 10^7
        // returning atomics on cfg3 (1.0 ms); the lanes count themselves into the 32 windows from the batch's first one on in LDS, and
        // one global atomic a window claims the slots of all of them.
        uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < 8; ++j) mn = min(mn, win[j]);
        const uint32_t A0 = ~wave_max_u32(~mn);
        if (lane < 32) wlds[lane] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        uint32_t rank[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            rank[j] = 0;
            if (win[j] == 0xFFFFFFFFu) continue;
            const uint32_t off = win[j] - A0;
            if (off < 32u) rank[j] = atomicAdd(&wlds[off], 1u);
            else rank[j] = atomicAdd(&a.wcount[win[j]], 1u);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        if (lane < 32) { const uint32_t cw = wlds[lane]; wlds[32 + lane] = cw ? atomicAdd(&a.wcount[A0 + (uint32_t)lane], cw) : 0u; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (win[j] == 0xFFFFFFFFu) continue;
            const uint32_t off = win[j] - A0;
            const uint32_t slot = off < 32u ? wlds[32 + off] + rank[j] : rank[j];
            if (slot < a.slab_cap) a.slab[(uint64_t)win[j] * a.slab_cap + slot] = dsc[j];
        }
        __builtin_amdgcn_wave_barrier();
        continue;
    }
    // slots in the windows' slabs.  Reads are sorted, so nearly every descriptor of a wave goes to the wave's first
    // window A or to A + 1: those two are claimed with one atomic each for the whole wave; the few others (second
    // halves of N-skip reads) claim theirs one by one.  All atomics are in flight together.
    const uint32_t A = (uint32_t)__shfl((int)wf, __ffsll((long long)am) - 1, 64);
    uint32_t pk = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) pk += (win[j] == A ? 1u : 0u) + (win[j] == A + 1 ? 0x10000u : 0u);
    const uint32_t inc = wave_incl_u32(pk, lane);
    const uint32_t tot = (uint32_t)__shfl((int)inc, 63, 64);
    uint32_t base0 = 0, base1 = 0;
    if (lane == 0) {
        if (tot & 0xFFFFu) base0 = atomicAdd(&a.wcount[A], (BGP_PLAN_ABL & 4) ? 0u : tot & 0xFFFFu);
        if (tot >> 16) base1 = atomicAdd(&a.wcount[A + 1], (BGP_PLAN_ABL & 4) ? 0u : tot >> 16);
    }
    uint32_t far_slot[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { far_slot[j] = 0;
#if !(BGP_PLAN_ABL & 2)
        if (win[j] != 0xFFFFFFFFu && win[j] != A && win[j] != A + 1) far_slot[j] = atomicAdd(&a.wcount[win[j]], (BGP_PLAN_ABL & 4) ? 0u : 1u);
#endif
    }
    uint32_t o0 = rl(base0, 0) + ((inc - pk) & 0xFFFFu), o1 = rl(base1, 0) + ((inc - pk) >> 16);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (win[j] == 0xFFFFFFFFu) continue;
        const uint32_t slot = win[j] == A ? o0++ : win[j] == A + 1 ? o1++ : far_slot[j];
#if !(BGP_PLAN_ABL & 4)
        if (slot < a.slab_cap) a.slab[(uint64_t)win[j] * a.slab_cap + slot] = dsc[j];
#else
        if (slot == 0xFFFFFFF0u) a.slab[(uint64_t)win[j] * a.slab_cap] = dsc[j];
#endif
    }
    }      // (rounds of four ops)
}

__global__ __launch_bounds__(256) void k4_plan(PlanArgs a) {
    const int lane = threadIdx.x & 63;
    // A wave's time was its chain of round trips -- rows, then the slots' atomics, then the stores -- with a wave slot held all the
    // while (77 % of its cycles waiting): a wave takes PLAN_NB batches and asks for the next one's rows before it works on this one's.
    const int nb_ = a.batches;                                             // (PLAN_NB; 1 for a stream of so few reads that the waves would not fill the device)
    const uint64_t first = ((uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (uint64_t)(64 * nb_);
    if (first >= a.n) return;
    uint4 nx, ny, nc4 = make_uint4(0, 0, 0, 0);
    { const uint64_t ii = min(first + (uint64_t)lane, a.n - 1); const uint4* rp = reinterpret_cast<const uint4*>(a.rec + ii); nx = rp[0]; ny = rp[1]; if (a.cig4) nc4 = a.cig4[ii]; }
    uint64_t r_end = 0, r_start_pos = 0, r_tile_base = 0;
    for (int it = 0; it < nb_; ++it) {
        const uint64_t i = first + (uint64_t)(64 * it) + (uint64_t)lane;
        const uint64_t i0 = i - lane;
        if (i0 >= a.n) break;
        const uint4 x = nx, y = ny, c4 = nc4;                 // ref_id,pos,end_pos,bin_mq_nl | flag_nc,l_seq,rec_off | the first four CIGAR ops
        if (it + 1 < nb_ && i0 + 64 < a.n) { const uint64_t jn = min(i + 64, a.n - 1); const uint4* rp = reinterpret_cast<const uint4*>(a.rec + jn); nx = rp[0]; ny = rp[1]; if (a.cig4) nc4 = a.cig4[jn]; }
        const bool in = i < a.n;
        plan_batch<false>(a, lane, in, in ? i : a.n - 1, i0, x, y, c4, it == 0, r_end, r_start_pos, r_tile_base);
    }
}

// The planner behind k12_lds's plan rows (a.rcls: a class byte a record, common.h): ordinary reads are taken from their rows by
// k4_acc2 itself, and only what is left -- reads that reach further than a row says (N skips), odd CIGARs, long reads: a few in a
// hundred -- is planned here.  A wave looks at 1024 class bytes (sixteen a lane, one load), lists the records of class 2 in LDS in
// file order and plans them 64 at a time with plan_batch: the rows of 96 reads in a hundred are never loaded.
constexpr int PLAN_SPAN = 1024;
__global__ __launch_bounds__(256) void k4_plan_rest(PlanArgs a) {
    __shared__ uint16_t lst[4][PLAN_SPAN];
    __shared__ uint32_t wl[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t base = ((uint64_t)blockIdx.x * 4 + (uint64_t)wv) * PLAN_SPAN;
    if (base >= a.n) return;
    const uint64_t o = base + 16ull * (uint64_t)lane;
    uint32_t cw[4] = {0u, 0u, 0u, 0u};
    if (o < a.n) __builtin_memcpy(cw, a.rcls + o, 16);        // (the array has sixteen bytes of room behind the last record)
    uint32_t m = 0;                                              // bit j: record o + j is class 2
#pragma unroll
    for (int j = 0; j < 16; ++j) if (((cw[j >> 2] >> (8 * (j & 3))) & 3u) == 2u && o + (uint64_t)j < a.n) m |= 1u << j;
    const uint32_t cnt = (uint32_t)__builtin_popcount(m);
    const uint32_t inc = wave_incl_u32(cnt, lane), total = rl(inc, 63);
    if (!total) return;
    {
        uint32_t at = inc - cnt, mm = m;
        while (mm) { const int j = __builtin_ctz(mm); mm &= mm - 1; lst[wv][at++] = (uint16_t)(16 * lane + j); }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint64_t r_end = 0, r_start_pos = 0, r_tile_base = 0;
    for (uint32_t k = 0; k < total; k += 64) {
        const bool in = k + (uint32_t)lane < total;
        const uint64_t ii = base + lst[wv][in ? k + (uint32_t)lane : total - 1u], i0 = base + lst[wv][k];
        const uint4* rp = reinterpret_cast<const uint4*>(a.rec + ii);
        const uint4 x = rp[0], y = rp[1];
        const uint4 c4 = a.cig4 ? a.cig4[ii] : make_uint4(0, 0, 0, 0);
        plan_batch<true>(a, lane, in, ii, i0, x, y, c4, k == 0, r_end, r_start_pos, r_tile_base, wl[wv]);
    }
}

// Prefix sums over windows of the two difference arrays (whole-window '-' levels, slow-read marks), then the verdict
// per window -- bit 1 of TileB.pad = "not for the accumulate path" -- and the list of such windows.  Three small launches:
// per-block totals, their exclusive scan by one workgroup, the prefix proper.
__device__ __forceinline__ void wscan_block(int32_t d, int32_t sl, int32_t* id_out, int32_t* is_out, int32_t* pd, int32_t* ps, int32_t* td, int32_t* ts, int32_t (*ws)[16]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int32_t id = d, is = sl;
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) { const int32_t od = __shfl_up(id, k, 64), os = __shfl_up(is, k, 64); if (lane >= k) { id += od; is += os; } }
    if (lane == 63) { ws[0][w] = id; ws[1][w] = is; }
    __syncthreads();
    int32_t a = 0, b = 0, c = 0, e = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { if (k < w) { a += ws[0][k]; b += ws[1][k]; } c += ws[0][k]; e += ws[1][k]; }
    *id_out = id; *is_out = is; *pd = a; *ps = b; *td = c; *ts = e;
    __syncthreads();
}
__global__ __launch_bounds__(1024) void k4_wscan_tot(const int32_t* wdel, const int32_t* wslow, uint64_t n_tiles, int32_t* btot) {
    __shared__ int32_t ws[2][16];
    const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
    int32_t id, is, pd, ps, td, ts;
    wscan_block(i < n_tiles ? wdel[i] : 0, i < n_tiles ? wslow[i] : 0, &id, &is, &pd, &ps, &td, &ts, ws);
    if (threadIdx.x == 0) { btot[2 * blockIdx.x] = td; btot[2 * blockIdx.x + 1] = ts; }
}
__global__ __launch_bounds__(1024) void k4_wscan_mid(int32_t* btot, uint32_t nb) {
    __shared__ int32_t ws[2][16];
    int32_t cd = 0, cs = 0;
    for (uint32_t base = 0; base < nb; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const int32_t d = i < nb ? btot[2 * i] : 0, sl = i < nb ? btot[2 * i + 1] : 0;
        int32_t id, is, pd, ps, td, ts;
        wscan_block(d, sl, &id, &is, &pd, &ps, &td, &ts, ws);
        if (i < nb) { btot[2 * i] = cd + pd + id - d; btot[2 * i + 1] = cs + ps + is - sl; }     // exclusive
        cd += td; cs += ts;
    }
}
constexpr uint32_t ROW_CAP = 8192;            // plan rows a window of the accumulate path reads at most (32 rounds); more: the general kernel's
__global__ __launch_bounds__(1024) void k4_wscan(int32_t* wdel, const int32_t* wslow, const uint32_t* wcount, TileB* tiles, uint64_t n_tiles, const int32_t* btot, uint32_t* deep, uint32_t* n_deep, uint32_t slab_cap, uint32_t row_cap) {
    __shared__ int32_t ws[2][16];
    const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
    int32_t id, is, pd, ps, td, ts;
    wscan_block(i < n_tiles ? wdel[i] : 0, i < n_tiles ? wslow[i] : 0, &id, &is, &pd, &ps, &td, &ts, ws);
    if (i < n_tiles) {
        wdel[i] = btot[2 * blockIdx.x] + pd + id;
        const bool rows_over = row_cap && tiles[i].i_hi - tiles[i].i_lo > (uint64_t)row_cap;
        const uint32_t pad = (tiles[i].pad & 1u) | ((btot[2 * blockIdx.x + 1] + ps + is > 0 || wcount[i] > slab_cap || rows_over) ? 2u : 0u);
        tiles[i].pad = pad;
        if (pad) deep[atomicAdd(n_deep, 1u)] = (uint32_t)i;
    }
}

struct GArgs {
    const uint8_t* s; const unsigned long long* slab; uint32_t slab_cap; const uint32_t* wcount; const int32_t* wdel; const TileB* tiles; uint64_t n_tiles;
    bgp_column* cols; unsigned long long* hist; unsigned long long* tile_stats; uint32_t* deep; uint32_t* n_deep;
    const uint32_t* wlong; const bgp_record* rec; const uint64_t* ck_off; const Ckpt* ck;      // long reads (null: none): windows they touch, the record table, their per-window checkpoints
    uint64_t stream_len;
    const uint4* prow;                            // ROWS: the records' plan rows (k12_lds; common.h: bgp_plan_row); a window's are [tiles[w].i_lo, i_hi)
};

// ---------------------------------------------------------------------------------------------- K4, accumulate path: lane = piece, byte planes (round 4)
//
// Four executors have run on k4_plan's descriptors or their predecessors (DESIGN.md / LABNOTES.md keep the measurements):
// round 1 let every BASE find its counter with lanes = reads (k4_tile, still the general kernel: 15.5 ms on cfg3); round 2 made
// every POSITION gather its bases out of class bytes staged in LDS (k4_exec: 10.3 ms); round 3 went back to atomics with lanes
// along the reads, eight positions a lane (k4_acc: 8.6 ms) -- ~80 VALU instructions per lane-chunk of eight positions: a byte map and a
// 16-byte LDS read to find the chunk's descriptor, the limits of the chunk, one table lookup, one address and one LDS atomic PER
// BASE, and its counters said instruction issue is what binds it.  k4_acc2 runs the same plan with a quarter of the instructions:
//
//   * a LANE owns a PIECE of a SEG descriptor -- up to 11 chunks of eight window-aligned positions -- for the piece's whole life:
//     its packed bases are in registers (three 16-byte loads from the chunk-aligned origin), the loop over the chunks is unrolled,
//     every register index and every LDS offset is static, and nothing is looked up per chunk;
//   * no base is handled alone.  The four even and the four odd positions of a chunk are two words of nibble codes, one byte a
//     position; v_perm_b32 with the codes as SELECTORS turns a word into the four "is A" bytes (0 / 1), another into "is C" ...:
//     eight words of byte-wide increments per chunk, added to LDS by eight ds_add_u32 whose addresses differ by constants
//     (class x even/odd planes of 136 words).  Positions outside the piece get selector 0 ('=': in no plane) by the AND that
//     isolates the nibbles; the three codes v_perm cannot index (13..15: it answers 0xFF) are forced to 12 (answers 0) first;
//   * "any other base" is not counted: coverage comes from a difference array (+1 / -1 per SEG descriptor, as '-' always did),
//     other = coverage - A - C - G - T.  The same array says how deep the window is: the byte-wide counters hold 255, a window
//     deeper than that is handed to the general kernel (k4_tile) by the flush -- nothing was written for it -- which at the
//     reads' ~150 bases is past the slab's 1024 descriptors anyway;
//   * lanes next to each other hold pieces a few positions apart: the same or neighbouring words, i.e. the same LDS banks
//     (scratch/r4/lds_plane_bench.hip: 93 clocks a chunk-step against 34 for distinct consecutive words).  The counters exist
//     FOUR times, copy = (lane >> 1) & 3, a copy 8 banks further: measured 34 clocks (two copies: 9.3 ms instead of 8.0 on cfg3).
//     The flush adds the copies up;
//   * a window's descriptors become pieces in a pool in LDS, 512 descriptors a round: per-descriptor atomics ('-', starts, coverage),
//     a 150-base read two pieces of 10 chunks; long pieces fill the pool from its front, short ones from its back, so that a batch of 64
//     from either end runs about as long in every lane; the loads of a wave's next batch are in flight while it works on the current one;
//   * long reads come as the same descriptors, an op each (k4_long_desc): pieces of one or two chunks from the pool's short end.
//
// What bounds it now (cfg3, `BGP_ACC2_ABL`): with the bases read from a cache-resident stretch and no column stores 6.65 ms, with the
// loads and stores but no accumulate 7.8 ms, both 8.0 ms -- the 41 GB it moves (24.6 GB of 128-byte lines around 75-byte packed
// sequences, 16 GB of columns) at 5.2 TB/s and the LDS (57 % busy) are about equally in the way.
//
// Same descriptors, same columns: every pileup test checks them against the oracle.
constexpr int AT = 256;                       // threads of a k4_acc2 workgroup
constexpr int ANW = AT / 64;                  // waves
constexpr int ABPW = (W / 64) / ANW;          // 64-position blocks a wave flushes
#ifndef BGP_ACC2_MINW
#define BGP_ACC2_MINW 4
#endif
#ifndef BGP_ACC2_ABL
#define BGP_ACC2_ABL 0                        // timing experiments (wrong columns): 1 bases from a cache-resident stretch, 2 no column stores
#endif
#ifndef BGP_ACC2_COPIES
#define BGP_ACC2_COPIES 4
#endif
#ifndef BGP_ACC2_NT_COLS
#define BGP_ACC2_NT_COLS 0                    // 1: the columns leave with non-temporal stores
#endif
#ifndef BGP_ACC2_DEPTH
#define BGP_ACC2_DEPTH 1                      // batches whose loads are in flight while a wave works on one
#endif
#ifndef BGP_ACC2_CPSH
#define BGP_ACC2_CPSH 1                       // a lane's copy: (lane >> this) & (copies - 1)
#endif
constexpr int P_NCP = BGP_ACC2_COPIES;        // private copies of the counters (2 or 4)
constexpr int P_CH = 11;                      // chunks of a piece at most (12 dwords loaded: chunk j of an odd-nibble piece wants dword j + 1 too)
constexpr int P_PLW = W / 8 + 8;              // words of a plane (class x even/odd positions): one per chunk, + 8 so that ...
constexpr int P_CPW = 8 * P_PLW + 8;          // ... a plane and a copy both start 8 banks after the one before
constexpr int P_RD = 512;                     // descriptors of a round (two a thread) ...
constexpr int P_POOL = 3 * P_RD;              // ... and the pieces they can make (a SEG descriptor: <= 255 bases + 7 = 33 chunks = 3 pieces)
constexpr int P_HB = 128;                     // depth-histogram bins kept in LDS
constexpr int P_SW = 8 * (W / 8 + 8);         // words of a staging array of the flush: position 8m + k at k x 136 + m
struct A2Lay {
    static constexpr int CNT = 0;                               // P_NCP copies x P_CPW x u32
    static constexpr int COVD = CNT + P_NCP * P_CPW * 4;        // (W + 8) x u32: SEG coverage, difference array
    static constexpr int DELD = COVD + (W + 8) * 4;             // (W + 8) x u32: '-', difference array
    static constexpr int NST = DELD + (W + 8) * 4;              // W x u16
    static constexpr int HIST = NST + W * 2;                    // P_HB x u32 (kept for the workgroup's whole range)
    static constexpr int POOL = (HIST + P_HB * 4 + 15) & ~15;   // P_POOL x u64: the round's pieces; the flush's two staging arrays lie over it
    static constexpr int MISC = POOL + P_POOL * 8;
    static constexpr int BYTES = MISC + 256;
};
static_assert(P_POOL * 8 >= 2 * P_SW * 4, "the flush stages a window's sums where the pool is");
static_assert(BGP_ACC2_MINW * A2Lay::BYTES <= 160 * 1024, "k4_acc2 workgroups per CU");
static_assert(W == 1024 && AT == 256 && P_RD == 2 * AT, "k4_acc2 geometry: a thread per (chunk, even/odd) in the flush, two descriptors a thread and round");

// one word of nibble codes (a byte a position; 0 where the position is not the piece's) -> its increments of the A, C, G, T planes
__device__ __forceinline__ void planes_of_codes(uint32_t sel, uint32_t& pa, uint32_t& pc, uint32_t& pg, uint32_t& pt) {
    const uint32_t bad = __builtin_amdgcn_perm(0u, 0u, sel);                       // 0xFF where the code is 13..15 (v_perm: "selector >= 13 -> 0xFF")
    sel = (sel & ~bad) | (bad & 0x0C0C0C0Cu);                                     // ... those become 12 ("-> 0x00"): in no plane
    pa = __builtin_amdgcn_perm(0u, 0x00000100u, sel);                             // code 1
    pc = __builtin_amdgcn_perm(0u, 0x00010000u, sel);                             // code 2
    pg = __builtin_amdgcn_perm(0x00000001u, 0u, sel);                             // code 4
    pt = __builtin_amdgcn_perm(0u, 0x00008000u, sel) & 0x01010101u;               // code 8 selects "the sign of table byte 1": 0xFF (code 1 itself: 0x80)
}

// a batch of the accumulate: a piece per lane.  Entry: [41:0] nibble address of the piece's chunk-aligned origin in the stream, [48:42] the origin's
// chunk of the window, [51:49] first position of the piece in its first chunk, [58:52] one past its last position, counted from the origin (0: no piece)
struct A2Batch { uint4 q0, q1, q2; uint32_t eh, par; };
__device__ __forceinline__ A2Batch acc2_load(const uint8_t* s, unsigned long long e) {
    A2Batch b;
    const uint64_t no = e & ((1ull << 42) - 1);
    b.eh = (uint32_t)(e >> 32); b.par = (uint32_t)no & 1u;
    const uint32_t tl = (b.eh >> 20) & 127u;
#if BGP_ACC2_ABL & 1
    const uint8_t* b0 = s + ((no >> 1) & 0x3FFFFFull);          // timing experiment: every load from the stream's first 4 MiB
#elif BGP_ACC2_ABL & 16
    const uint8_t* b0 = s + (((no >> 1) * 33ull) >> 6);          // timing experiment: the traffic of a dense arena of bases (a read every ~150 bytes instead of every ~290)
#else
    const uint8_t* b0 = s + (no >> 1);
#endif
    const uint32_t need = (b.par + tl + 1u) >> 1;                                  // bytes from b0 on (branch-free: a dword nobody needs is loaded from b0 again)
    __builtin_memcpy(&b.q0, b0, 16);
    __builtin_memcpy(&b.q1, b0 + (need > 16u ? 16 : 0), 16);
    __builtin_memcpy(&b.q2, b0 + (need > 32u ? 32 : 0), 16);
    return b;
}
__device__ __forceinline__ void acc2_work(const A2Batch& b, uint32_t* cbase0) {
#if BGP_ACC2_ABL & 4
    if (b.q0.x == 0x12345678u && b.q1.y == 0x9ABCDEF0u && b.q2.z == 77u) cbase0[0] = 1;      // timing experiment: the loads, not the work
    return;
#endif
    const uint32_t m0 = (b.eh >> 10) & 127u, s = (b.eh >> 17) & 7u, tl = (b.eh >> 20) & 127u, par = b.par;
    const uint32_t R[12] = {b.q0.x, b.q0.y, b.q0.z, b.q0.w, b.q1.x, b.q1.y, b.q1.z, b.q1.w, b.q2.x, b.q2.y, b.q2.z, b.q2.w};
    const uint32_t jl = (tl - 1u) >> 3, rr = tl - 8u * jl;                        // the piece's last chunk and its positions there (1..8)
    const uint32_t vt_e = 0x0F0F0F0Fu >> (8u * (4u - ((rr + 1u) >> 1)));
    const uint32_t vt_o = (uint32_t)(0x0F0F0F0Full >> (8u * (4u - (rr >> 1))));
    const uint32_t vf_e = (uint32_t)(0x0F0F0F0Full << (8u * ((s + 1u) >> 1))), vf_o = 0x0F0F0F0Fu << (8u * (s >> 1));
    uint32_t* const cb = cbase0 + m0;
#pragma unroll
    for (int j = 0; j < P_CH; ++j) {
        if (!__builtin_amdgcn_ballot_w64(tl > 8u * (uint32_t)j)) break;
        if (tl > 8u * (uint32_t)j) {
            const uint32_t D = R[j], Dn = R[j + 1];
            // even positions 8j, 8j+2 .. / odd ones: high / low nibbles of the four bytes -- or, for a piece whose origin is a low
            // nibble, the low nibbles of these four bytes / the high ones of the next four (read.d:368-373)
            const uint32_t es = par ? D : D >> 4, os = par ? __builtin_amdgcn_alignbit(Dn, D, 12) : D;
            uint32_t vme = (uint32_t)j == jl ? vt_e : 0x0F0F0F0Fu, vmo = (uint32_t)j == jl ? vt_o : 0x0F0F0F0Fu;
            if (j == 0) { vme &= vf_e; vmo &= vf_o; }
            uint32_t pa, pc, pg, pt;
            planes_of_codes(es & vme, pa, pc, pg, pt);
            atomicAdd(cb + j + 0 * P_PLW, pa); atomicAdd(cb + j + 2 * P_PLW, pc); atomicAdd(cb + j + 4 * P_PLW, pg); atomicAdd(cb + j + 6 * P_PLW, pt);
            planes_of_codes(os & vmo, pa, pc, pg, pt);
            atomicAdd(cb + j + 1 * P_PLW, pa); atomicAdd(cb + j + 3 * P_PLW, pc); atomicAdd(cb + j + 5 * P_PLW, pg); atomicAdd(cb + j + 7 * P_PLW, pt);
        }
    }
}

// A long read's ops are ~15 bases each: its pieces are cut at three chunks (24 positions: one 16-byte load), and three of them are
// in flight per wave -- a piece's bases arrive while the two before it are accumulated.
constexpr int P_CHS = 3;
struct A2Small { uint4 q0; uint32_t eh, par; };
__device__ __forceinline__ A2Small acc2_load_small(const uint8_t* s, uint64_t stream_len, unsigned long long e) {
    A2Small b;
    const uint64_t no = e & ((1ull << 42) - 1), bo = no >> 1;
    b.eh = (uint32_t)(e >> 32); b.par = (uint32_t)no & 1u;
    if (bo + 16 <= stream_len) __builtin_memcpy(&b.q0, s + bo, 16);
    else {                                          // the stream's last few bases (a long read without tags behind its QUAL... or an attached buffer of exactly `len` bytes): byte by byte, nothing behind the end
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        for (uint32_t k = 0; k < 16u && bo + k < stream_len; ++k) w[k >> 2] |= (uint32_t)s[bo + k] << (8u * (k & 3u));
        b.q0 = make_uint4(w[0], w[1], w[2], w[3]);
    }
    return b;
}
__device__ __forceinline__ void acc2_work_small(const A2Small& b, uint32_t* cbase0) {
#if BGP_ACC2_ABL & 32
    if (b.q0.x == 0x12345678u && b.q0.y == 0x9ABCDEF0u && b.q0.z == 77u) cbase0[0] = 1;      // timing experiment: a long read's loads and bookkeeping, not its accumulate
    return;
#endif
    const uint32_t m0 = (b.eh >> 10) & 127u, s = (b.eh >> 17) & 7u, tl = (b.eh >> 20) & 127u, par = b.par;
    if (!__builtin_amdgcn_ballot_w64(tl != 0)) return;
    const uint32_t R[4] = {b.q0.x, b.q0.y, b.q0.z, b.q0.w};
    const uint32_t jl = (tl - 1u) >> 3, rr = tl - 8u * jl;
    const uint32_t vt_e = 0x0F0F0F0Fu >> (8u * (4u - ((rr + 1u) >> 1)));
    const uint32_t vt_o = (uint32_t)(0x0F0F0F0Full >> (8u * (4u - (rr >> 1))));
    const uint32_t vf_e = (uint32_t)(0x0F0F0F0Full << (8u * ((s + 1u) >> 1))), vf_o = 0x0F0F0F0Fu << (8u * (s >> 1));
    uint32_t* const cb = cbase0 + m0;
#pragma unroll
    for (int j = 0; j < P_CHS; ++j) {
        if (!__builtin_amdgcn_ballot_w64(tl > 8u * (uint32_t)j)) break;
        if (tl > 8u * (uint32_t)j) {
            const uint32_t D = R[j], Dn = R[j + 1];
            const uint32_t es = par ? D : D >> 4, os = par ? __builtin_amdgcn_alignbit(Dn, D, 12) : D;
            uint32_t vme = (uint32_t)j == jl ? vt_e : 0x0F0F0F0Fu, vmo = (uint32_t)j == jl ? vt_o : 0x0F0F0F0Fu;
            if (j == 0) { vme &= vf_e; vmo &= vf_o; }
            uint32_t pa, pc, pg, pt;
            planes_of_codes(es & vme, pa, pc, pg, pt);
            atomicAdd(cb + j + 0 * P_PLW, pa); atomicAdd(cb + j + 2 * P_PLW, pc); atomicAdd(cb + j + 4 * P_PLW, pg); atomicAdd(cb + j + 6 * P_PLW, pt);
            planes_of_codes(os & vmo, pa, pc, pg, pt);
            atomicAdd(cb + j + 1 * P_PLW, pa); atomicAdd(cb + j + 3 * P_PLW, pc); atomicAdd(cb + j + 5 * P_PLW, pg); atomicAdd(cb + j + 7 * P_PLW, pt);
        }
    }
}

// LONG: the stream holds long reads (the phase that walks their CIGARs is compiled in: it costs registers -- 128 and a few spilled
// against 102 -- that a stream of short reads need not pay for: cfg3 8.8 against 8.0 ms)
// ROWS: the window's ordinary reads come as k12_lds's plan rows, a row a thread and round -- its one or two M runs are cut at the window
// here and become pieces like a SEG descriptor's -- and the slab holds only what k4_plan_rest planned (a few descriptors a window).
template <bool LONG, bool ROWS> __global__ __launch_bounds__(AT, BGP_ACC2_MINW) void k4_acc2(GArgs a) {
    static_assert(!(LONG && ROWS), "a stream with long reads is planned read by read");
    using L = A2Lay;
    extern __shared__ __attribute__((aligned(16))) uint8_t alds[];
    uint32_t* const cnt = reinterpret_cast<uint32_t*>(alds + L::CNT);
    uint32_t* const covd = reinterpret_cast<uint32_t*>(alds + L::COVD);
    uint32_t* const deld = reinterpret_cast<uint32_t*>(alds + L::DELD);
    uint32_t* const nst = reinterpret_cast<uint32_t*>(alds + L::NST);
    uint32_t* const hist = reinterpret_cast<uint32_t*>(alds + L::HIST);
    uint32_t* const misc = reinterpret_cast<uint32_t*>(alds + L::MISC);      // [0] long pieces of the round (from the pool's front), [1] short ones (from its back), [2] next batch; [8..11] '-' totals of the waves, [12..15] coverage totals, [16..19] deepest SEG coverage
    unsigned long long* const pool = reinterpret_cast<unsigned long long*>(alds + L::POOL);
    uint32_t* const stg = reinterpret_cast<uint32_t*>(alds + L::POOL);        // [0, P_SW) A | C << 16, [P_SW, 2 P_SW) G | T << 16
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint64_t nbt = a.n_tiles, per = (nbt + gridDim.x - 1) / gridDim.x;
    const uint64_t tile_end = min(nbt, ((uint64_t)blockIdx.x + 1) * per);
    uint64_t tile = (uint64_t)blockIdx.x * per;
    if (tile >= tile_end) return;
    for (int i = t; i < L::POOL / 4; i += AT) cnt[i] = 0;                      // counters, difference arrays, starts (a window leaves them zeroed again), hist
    if (t < 32) misc[t] = 0;
    struct Pre { uint4 ii; uint2 col; uint4 tt; uint32_t n; int32_t wd; unsigned long long d0, d1; uint4 r0, r1; };   // the next window's header and first round of descriptors, a window ahead, in vector registers: left to the compiler the loads became scalars with an s_waitcnt right behind the "prefetch" -- which, gfx9 counting loads and stores with ONE in-order counter, also stood for the previous window's column stores
    auto prefetch_hdr = [&](uint64_t w, Pre& p) {
        const char* tp = reinterpret_cast<const char*>(a.tiles + w);
        __builtin_memcpy(&p.ii, tp, 16);
        __builtin_memcpy(&p.col, tp + 16, 8);
        __builtin_memcpy(&p.tt, tp + 24, 16);
        p.n = a.wcount[w]; p.wd = a.wdel[w];
    };
    auto prefetch_pay = [&](uint64_t w, Pre& p) {       // ROWS: the window's first round of rows lies where its header says -- the header is asked for TWO windows ahead, the rows one
        const unsigned long long* sl = a.slab + w * (uint64_t)a.slab_cap;      // (the slab is there whatever the count says; slots behind the count hold anything)
        p.d0 = sl[t];
        if (!ROWS) p.d1 = sl[AT + t];
        else {
            p.d1 = 0ull;
            const uint64_t lo = ((uint64_t)p.ii.y << 32) | p.ii.x, hi = ((uint64_t)p.ii.w << 32) | p.ii.z;
            p.r0 = p.r1 = make_uint4(0u, 0u, 0u, 0u);
            if (lo + (uint64_t)t < hi) p.r0 = a.prow[lo + (uint64_t)t];
            if (lo + (uint64_t)(AT + t) < hi) p.r1 = a.prow[lo + (uint64_t)(AT + t)];
        }
    };
    Pre pre;
    uint4 ii2 = make_uint4(0u, 0u, 0u, 0u);          // ROWS: the record range of the window after next (the rows of the next window are asked for a window ahead, and where they lie has to be known by then)
    prefetch_hdr(tile, pre);
    if (ROWS && tile + 1 < tile_end) __builtin_memcpy(&ii2, reinterpret_cast<const char*>(a.tiles + tile + 1), 16);
    prefetch_pay(tile, pre);
    uint32_t r_ncol = 0, r_nsum = 0, r_mx = 0; unsigned long long r_scov = 0;
    uint32_t* const cbase0 = cnt + ((lane >> BGP_ACC2_CPSH) & (P_NCP - 1)) * P_CPW;
    uint32_t rounds = 0;
    __syncthreads();
    for (; tile < tile_end; ++tile) {
        asm volatile("; window header settled" : "+v"(pre.col.x), "+v"(pre.col.y), "+v"(pre.tt.x), "+v"(pre.tt.y), "+v"(pre.tt.z), "+v"(pre.tt.w), "+v"(pre.n), "+v"(pre.wd));
        const uint64_t col_out = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)pre.col.y) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)pre.col.x);
        const uint32_t T0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)pre.tt.x), Tend = (uint32_t)__builtin_amdgcn_readfirstlane((int)pre.tt.y);
        const uint32_t pad = (uint32_t)__builtin_amdgcn_readfirstlane((int)pre.tt.w);
        const uint32_t n_desc = (uint32_t)__builtin_amdgcn_readfirstlane((int)pre.n);
        const int32_t wd = __builtin_amdgcn_readfirstlane(pre.wd);
        const uint64_t i_lo = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)pre.ii.y) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)pre.ii.x);
        const uint64_t i_hi = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)pre.ii.w) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)pre.ii.z);
        const uint32_t run_start = (uint32_t)__builtin_amdgcn_readfirstlane((int)pre.tt.z);
        unsigned long long dd[2] = {pre.d0, pre.d1};
        uint4 rw = pre.r0; const uint4 pre_r1 = pre.r1;
        if (!ROWS) { if (tile + 1 < tile_end) { prefetch_hdr(tile + 1, pre); prefetch_pay(tile + 1, pre); } }
        else if (tile + 1 < tile_end) {                      // (the record range of tile + 1 has been on its way for a window: its rows can be asked for; and the range of tile + 2)
            prefetch_hdr(tile + 1, pre);
            pre.ii = ii2;
            prefetch_pay(tile + 1, pre);
            if (tile + 2 < tile_end) __builtin_memcpy(&ii2, reinterpret_cast<const char*>(a.tiles + tile + 2), 16);
        }
        if (pad) continue;                                   // on the general kernel's list
        const unsigned long long* slab = a.slab + tile * (uint64_t)a.slab_cap;
        const uint32_t n_rows = ROWS ? (uint32_t)(i_hi - i_lo) : 0u;      // (k4_wscan lists a window of more than ROW_CAP rows for the general kernel)
        const uint32_t n_items = ROWS ? max(n_rows, 2u * n_desc) : n_desc;      // (ROWS: a round is 512 rows and 256 descriptors of the slab)
        bool ovf = false;
        for (uint32_t rb = 0; rb < ((BGP_ACC2_ABL & 8) ? 0u : n_items); rb += (uint32_t)P_RD) {      // (8: timing experiment, the flush alone)
            // ---- a round, first half: P_RD descriptors, two a thread -- ROWS: a plan row (its two M runs cut at the window: two descriptors) and a
            // descriptor of the slab a thread.  Per-descriptor atomics ('-', starts, coverage); a SEG descriptor is cut
            // into pieces of equal length, long pieces (9 chunks or more) fill the pool from its front, short ones from its back: a batch of 64
            // from either end runs about as long in every lane
            uint4 rw2 = make_uint4(0u, 0u, 0u, 0u);
            if (!ROWS) { if (rb) { dd[0] = slab[rb + t]; dd[1] = slab[rb + AT + t]; } }
            else {
                if (rb) { rw = make_uint4(0u, 0u, 0u, 0u); if (rb + t < n_rows) rw = a.prow[i_lo + rb + t]; rw2 = make_uint4(0u, 0u, 0u, 0u); if (rb + AT + t < n_rows) rw2 = a.prow[i_lo + rb + AT + t]; if (rb / 2 + t < n_desc) dd[0] = slab[rb / 2 + t]; }
                else rw2 = pre_r1;
            }
            uint32_t* const mc = misc + 4 * (rounds & 1u);       // the round's three counters; the other set is zeroed while this one is in use
            ++rounds;
            // one or two SEG descriptors (d1: 0 or the second run of the same row) become pieces behind ONE scan of the wave
            auto emit = [&](unsigned long long d0, unsigned long long d1, bool two) {
                uint32_t lo_[2], len_[2], sft_[2], npc_[2], cpp_[2]; bool lng_[2]; uint64_t na_[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const unsigned long long d = q ? d1 : d0;
                    const uint32_t kind = (q && !two) ? 0u : (uint32_t)d & 3u;
                    const uint32_t lo = ((uint32_t)d >> 3) & 0x7FFu, f14 = (uint32_t)(d >> 14);
                    const uint32_t len = f14 & 0xFFu, nib = (f14 >> 8) & 1u;
                    if (kind == 2) { atomicAdd(&deld[lo], 1u); atomicAdd(&deld[f14 & 0xFFFu], 0xFFFFFFFFu); }       // D / N inside the window: '-' from lo up to the op's end (pileup.d:119-121)
                    const bool seg = kind == 1;
                    if (seg) {
                        if (d & 4u) atomicAdd(&nst[lo >> 1], 1u << ((lo & 1u) * 16));                              // n_starting_here, pileup.d:386,422
                        atomicAdd(&covd[lo], 1u); atomicAdd(&covd[lo + len], 0xFFFFFFFFu);                       // M / = / X: in the column from lo on (pileup.d:117-118, :352)
                    }
                    const uint32_t sft = lo & 7u;
                    const uint32_t nch = seg ? (sft + len + 7u) >> 3 : 0u;
                    const uint32_t npc = (nch + (uint32_t)P_CH - 1u) / (uint32_t)P_CH;            // 0..3
                    const uint32_t cpp = npc ? (nch + npc - 1u) / npc : 0u;                       // chunks a piece: the pieces of a descriptor are equally long
                    lo_[q] = lo; len_[q] = len; sft_[q] = sft; npc_[q] = npc; cpp_[q] = cpp; lng_[q] = cpp >= 9u;
                    na_[q] = 2ull * (d >> 23) + nib - sft;                                          // nibble address of the descriptor's chunk-aligned origin
                }
                const uint32_t pk0 = lng_[0] ? npc_[0] : npc_[0] << 16, pk1 = lng_[1] ? npc_[1] : npc_[1] << 16, pk = pk0 + pk1;
                const uint32_t inc = wave_scan_u32(pk), tot = rl(inc, 63);
                uint32_t bl = 0, bs = 0;
                if (lane == 0) { if (tot & 0xFFFFu) bl = atomicAdd(&mc[0], tot & 0xFFFFu); if (tot >> 16) bs = atomicAdd(&mc[1], tot >> 16); }
                bl = rl(bl, 0); bs = rl(bs, 0);
                uint32_t ex = inc - pk;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (q && !two) break;
                    const uint32_t slot0 = lng_[q] ? bl + (ex & 0xFFFFu) : (uint32_t)P_POOL - 1u - (bs + (ex >> 16));
                    for (uint32_t k = 0; k < npc_[q]; ++k) {
                        const uint32_t c0 = k * cpp_[q];
                        const uint32_t tl = min(sft_[q] + len_[q] - 8u * c0, 8u * cpp_[q]);
                        const unsigned long long e = (na_[q] + 8u * c0) | ((unsigned long long)(((lo_[q] >> 3) + c0) | ((k ? 0u : sft_[q]) << 7) | (tl << 10)) << 42);
                        const uint32_t at = lng_[q] ? slot0 + k : slot0 - k;
                        if (!ROWS || at < (uint32_t)P_POOL) pool[at] = e;      // (ROWS: a round can make more pieces than the pool holds -- reads of 250 bases in two runs next to a full slab; the window then goes to the general kernel, below)
                    }
                    ex += q ? 0u : pk0;
                }
            };
            if (!ROWS) {
#pragma unroll
                for (int h = 0; h < 2; ++h) emit(rb + h * AT + t < n_desc ? dd[h] : 0ull, 0ull, false);
            } else {
                // a plan row a thread and half-round: its runs inside [T0, Tend) as SEG descriptors, its D / N stretch as two entries of the '-' difference array
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint4 r = h ? rw2 : rw;
                    const bool on = (r.y >> 28) != 0u;
                    const uint32_t len0 = r.y & 0xFFu, len1 = (r.y >> 8) & 0xFFu, r1 = (r.y >> 16) & 0x1FFu;
                    const uint64_t zw = ((uint64_t)r.w << 32) | r.z, na0 = zw & ((1ull << 42) - 1ull);
                    const uint32_t dq = (uint32_t)(zw >> 42) & 0xFFu;
                    unsigned long long dr[2] = {0ull, 0ull};
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const uint32_t st = q ? r.x + r1 : r.x, len = q ? len1 : len0;
                        const uint32_t c_lo = max(st, T0), c_hi = min(st + len, Tend);
                        if (on && c_lo < c_hi) {
                            const uint64_t nb = na0 + (q ? dq : 0u) + (c_lo - st);
                            dr[q] = (unsigned long long)(1u | (q == 0 && c_lo == st ? 4u : 0u) | ((c_lo - T0) << 3) | ((c_hi - c_lo) << 14) | (((uint32_t)nb & 1u) << 22)) | ((nb >> 1) << 23);
                        }
                    }
                    if (on && ((r.y >> 25) & 1u)) {
                        const uint32_t c_lo = max(r.x + len0, T0), c_hi = min(r.x + r1, Tend);
                        if (c_lo < c_hi) { atomicAdd(&deld[c_lo - T0], 1u); atomicAdd(&deld[c_hi - T0], 0xFFFFFFFFu); }      // pileup.d:119-121
                    }
                    if (h == 0 || rb + AT + 64u * (uint32_t)wave < n_rows) emit(dr[0], dr[1], true);      // (the second half-round: only the waves that hold rows)
                }
                if (rb / 2 + 64u * (uint32_t)wave < n_desc) emit(rb / 2 + t < n_desc ? dd[0] : 0ull, 0ull, false);      // what k4_plan_rest planned: a handful of descriptors, the first wave's
            }
            __syncthreads();
            if (ROWS && mc[0] + mc[1] > (uint32_t)P_POOL) ovf = true;      // the pool's two ends have met: what it holds is not the round's pieces
            // ---- second half: batches of 64 pieces, a piece a lane; the loads of a wave's next batch are in flight while it works on the current one
            {
                if (t < 3) misc[4 * (rounds & 1u) + t] = 0;       // (the next round's: nobody has touched them since the barrier that ended the round before this one)
                const uint32_t n_l = ovf ? 0u : mc[0], n_s = ovf ? 0u : mc[1];
                const uint32_t nb_l = (n_l + 63u) >> 6, nb = nb_l + ((n_s + 63u) >> 6);
                auto entry = [&](uint32_t b) {
                    unsigned long long e = 0;
                    if (b < nb_l) { const uint32_t i = b * 64u + (uint32_t)lane; if (i < n_l) e = pool[i]; }
                    else if (b < nb) { const uint32_t i = (b - nb_l) * 64u + (uint32_t)lane; if (i < n_s) e = pool[(uint32_t)P_POOL - 1u - i]; }
                    return e;
                };
                auto draw = [&]() { uint32_t b = 0; if (lane == 0) b = atomicAdd(&mc[2], 1u); return (uint32_t)__builtin_amdgcn_readfirstlane((int)b); };
                uint32_t b = draw();
                if (b < nb) {
#if BGP_ACC2_DEPTH == 2
                    A2Batch cur = acc2_load(a.s, entry(b));
                    b = draw();
                    if (b >= nb) acc2_work(cur, cbase0);
                    else {
                        A2Batch nxt = acc2_load(a.s, entry(b));
                        for (;;) {
                            b = draw();
                            if (b >= nb) break;
                            const A2Batch nx2 = acc2_load(a.s, entry(b));
                            acc2_work(cur, cbase0);
                            cur = nxt; nxt = nx2;
                        }
                        acc2_work(cur, cbase0);
                        acc2_work(nxt, cbase0);
                    }
#else
                    A2Batch cur = acc2_load(a.s, entry(b));
                    for (;;) {
                        b = draw();
                        if (b >= nb) break;
                        const A2Batch nxt = acc2_load(a.s, entry(b));
                        acc2_work(cur, cbase0);
                        cur = nxt;
                    }
                    acc2_work(cur, cbase0);
#endif
                }
            }
            __syncthreads();                                 // every add of the round is in; the pool is free
            if (ovf) break;
        }
        const uint32_t np = Tend - T0;
        if (LONG && a.wlong[tile]) {
            // ---- long reads.  Their checkpoint for this window (k4_ck_build) is the op that owns the window's first column with the
            // columns and query bases in front of it; from there 64 ops a step, a LANE per op: two wave scans give the op's first column
            // and query base (the closed form of pileup.d:175-222), and the op is accumulated as it stands -- an M / = / X op is a piece
            // (or several) like any other, a D / N op two entries of the '-' difference array.  Nothing is written for a long read but its
            // checkpoints: rounds 2 and 3 put every column of every long read into an arena of class bytes first (k4_rows_build 6.3 ms +
            // k4_acc<ROWS> 3.6 ms on cfg5, 26 GB of traffic for it).
            // A read is a chain of dependent loads (row -> checkpoint -> ops -> bases), so the window's candidates are looked at by all
            // threads at once and the long reads that touch it listed in LDS (where the pool was); then a wave per listed read, the NEXT
            // read's ops requested before the current one is scanned and the current one's bases while the one before is accumulated.
            struct LRead { unsigned long long cg, nib; int32_t shift; uint32_t k, cs, qs, n_cig, pad; };
            constexpr int LCAND = 96, LPB = 4 * 64;               // candidates a round; a wave's ring of pieces (a step deals out three an op at most, up to 63 wait from the step before)
            static_assert(sizeof(LRead) == 40 && LCAND * sizeof(LRead) + ANW * LPB * 8 <= (size_t)P_POOL * 8, "the window's long reads and the waves' pieces are listed where the pool is");
            LRead* const lst = reinterpret_cast<LRead*>(alds + L::POOL);
            unsigned long long* const pcs = reinterpret_cast<unsigned long long*>(alds + L::POOL + LCAND * sizeof(LRead)) + wave * LPB;
            constexpr uint32_t SH = (uint32_t)__builtin_ctz((unsigned)W);
            for (uint64_t cb = i_lo; cb < i_hi; cb += LCAND) {
                uint32_t* const mc = misc + 4 * (rounds & 1u);       // (a round like the descriptors': mc[0] counts the listed reads; the other set is zeroed meanwhile)
                ++rounds;
                const uint64_t i = cb + (uint64_t)t;
                if (t < LCAND && i < i_hi) {
                    const bgp_record r = a.rec[i];
                    const uint64_t cko = a.ck_off[i];                 // (asked for with the row, not behind it)
                    const uint32_t n_cig = r.flag_nc & 0xFFFFu, l_name = r.bin_mq_nl & 0xFFu;
                    if (r.end_pos != r.pos && wants_ckpt(n_cig, (uint32_t)r.l_seq) && (uint32_t)r.pos < Tend && (uint32_t)r.end_pos > T0) {
                        const uint32_t j = ((T0 - run_start) >> SH) - (((uint32_t)r.pos - run_start) >> SH);
                        const Ckpt ck = a.ck[cko + j];
                        const uint64_t cig = r.rec_off + 36 + l_name;
                        lst[atomicAdd(&mc[0], 1u)] = LRead{(unsigned long long)(a.s + cig), 2ull * (cig + 4ull * n_cig), (int32_t)((int64_t)(uint32_t)r.pos - (int64_t)T0), ck.k, ck.cs, ck.qs, n_cig, 0u};
                    }
                }
                __syncthreads();
                if (t < 3) misc[4 * (rounds & 1u) + t] = 0;
                const uint32_t n_l = mc[0];
                A2Small pn0, pn1; pn0.eh = pn1.eh = 0; pn0.par = pn1.par = 0; pn0.q0 = pn1.q0 = make_uint4(0, 0, 0, 0);      // the two pieces whose bases are on their way
                // Pieces wait in a ring until 64 of them are there: a read's window is ~34 M ops, half a batch -- taken read by read the
                // accumulate ran with half its lanes.  What is left when the round's reads are through leaves as one last batch.
                uint32_t r_head = 0, r_pend = 0;
                // (TWO ops a lane: a window of 1024 positions is ~68 ops of a read with an indel every ~15 bases -- one step of 128 instead of
                // two of 64 --, and as M and I / D alternate nearly every lane then holds exactly one M op: the accumulate runs with full waves)
                auto ops_of = [&](const LRead& E, uint32_t c0) {
                    uint2 o = make_uint2(0u, 0u);
                    const uint32_t k0 = c0 + 2u * (uint32_t)lane;
                    const uint8_t* q = reinterpret_cast<const uint8_t*>(E.cg) + 4ull * k0;
                    if (k0 + 1 < E.n_cig) __builtin_memcpy(&o, q, 8);
                    else if (k0 < E.n_cig) o.x = ld_u32(q);
                    return o;
                };
                constexpr int LQ = 8;                                 // reads whose first 128 ops a wave asks for before it scans the first
                for (uint32_t e0 = (uint32_t)wave; e0 < n_l; e0 += ANW * LQ) {
                uint2 opq[LQ];
#pragma unroll
                for (int q = 0; q < LQ; ++q) { const uint32_t e = e0 + ANW * q; opq[q] = make_uint2(0u, 0u); if (e < n_l) { const LRead Eq = lst[e]; opq[q] = ops_of(Eq, Eq.k); } }
#pragma unroll
                for (int q = 0; q < LQ; ++q) {
                    const uint32_t e = e0 + ANW * q;
                    if (e >= n_l) break;
                    const LRead cur = lst[e]; uint2 op2 = opq[q];
                    uint32_t cs = cur.cs, qs = cur.qs;
                    for (uint32_t c0 = cur.k; c0 < cur.n_cig; c0 += 128) {
                        if (c0 != cur.k) op2 = ops_of(cur, c0);
                        const uint32_t k0 = c0 + 2u * (uint32_t)lane;
                        uint32_t rl0, ql0, rl1, ql1;
                        {   // op_lens with the cursor started, by masks and arithmetic (its cases were exec-mask bookkeeping)
                            const uint32_t l0 = op2.x >> 4, t0 = 0x3C1A7u >> ((op2.x & 15u) * 2), a0 = k0 < cur.n_cig ? ~0u : 0u, r0 = a0 & (0u - ((t0 >> 1) & 1u)), e0 = max(l0, 1u);
                            rl0 = r0 & e0; ql0 = a0 & (0u - (t0 & 1u)) & ((r0 & e0) | (~r0 & l0));
                            const uint32_t l1 = op2.y >> 4, t1 = 0x3C1A7u >> ((op2.y & 15u) * 2), a1 = k0 + 1 < cur.n_cig ? ~0u : 0u, r1 = a1 & (0u - ((t1 >> 1) & 1u)), e1 = max(l1, 1u);
                            rl1 = r1 & e1; ql1 = a1 & (0u - (t1 & 1u)) & ((r1 & e1) | (~r1 & l1));
                        }
                        const uint32_t ir = wave_scan_u32(rl0 + rl1), iq = wave_scan_u32(ql0 + ql1);
                        const uint32_t csb[2] = {cs + ir - rl0 - rl1, cs + ir - rl1}, qsb[2] = {qs + iq - ql0 - ql1, qs + iq - ql1};
                        const uint32_t rln[2] = {rl0, rl1}, opv[2] = {op2.x, op2.y};
                        int32_t sa0[2]; uint32_t sp[2], se[2];                              // per op: window position of its column 0; the stretch of the window still to be accumulated
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int32_t a0 = cur.shift + (int32_t)csb[h], a1 = a0 + (int32_t)rln[h];      // the op's columns in window positions
                            const uint32_t lo = (uint32_t)max(a0, 0), hi = (uint32_t)min(a1, (int32_t)np);
                            const bool in_w = rln[h] && a1 > 0 && a0 < (int32_t)np;
                            const bool isM = cig_qc(opv[h]) != 0;
                            if (in_w) {
                                if (isM) {
                                    if (csb[h] == 0 && a0 >= 0) atomicAdd(&nst[lo >> 1], 1u << ((lo & 1u) * 16));       // n_starting_here, pileup.d:386,422
                                    atomicAdd(&covd[lo], 1u); atomicAdd(&covd[hi], 0xFFFFFFFFu);
                                } else { atomicAdd(&deld[lo], 1u); atomicAdd(&deld[hi], 0xFFFFFFFFu); }      // '-' while the op lasts (pileup.d:119-121)
                            }
                            sa0[h] = a0; sp[h] = lo; se[h] = in_w && isM ? hi : lo;
                        }
                        // the ops' pieces (three chunks at most: one 16-byte load), dealt out to the lanes: an op of 15 bases is one piece, the
                        // odd long one several -- handled by the op's own lane the wave ran as long as its longest op while most lanes sat out
                        while (!(BGP_ACC2_ABL & 64) && __ballot(sp[0] < se[0] || sp[1] < se[1])) {      // (64: timing experiment, the ops' scan alone)
                            uint32_t npc = 0;
                            {
                                uint32_t q0 = sp[0], q1 = sp[1];
                                while (npc < 3u && (q0 < se[0] || q1 < se[1])) {
                                    if (q0 < se[0]) q0 += min(se[0] - q0, 8u * (uint32_t)P_CHS - (q0 & 7u)); else q1 += min(se[1] - q1, 8u * (uint32_t)P_CHS - (q1 & 7u));
                                    ++npc;
                                }
                            }
                            const uint32_t ipc = wave_scan_u32(npc), tot = rl(ipc, 63);
                            uint32_t at = r_head + r_pend + ipc - npc;
                            for (uint32_t k = 0; k < npc; ++k) {
                                const int h = sp[0] < se[0] ? 0 : 1;
                                const uint32_t p = sp[h];
                                const uint32_t n = min(se[h] - p, 8u * (uint32_t)P_CHS - (p & 7u));
                                const uint64_t nb = cur.nib + qsb[h] + (uint32_t)((int32_t)p - sa0[h]) - (p & 7u);
                                pcs[at++ & (uint32_t)(LPB - 1)] = nb | ((unsigned long long)((p >> 3) | ((p & 7u) << 7) | (((p & 7u) + n) << 10)) << 42);
                                sp[h] = p + n;
                            }
                            r_pend += tot;
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            while (r_pend >= 64u) {
                                const unsigned long long en = pcs[(r_head + (uint32_t)lane) & (uint32_t)(LPB - 1)];
                                const A2Small bt = acc2_load_small(a.s, a.stream_len, en);
                                acc2_work_small(pn0, cbase0);
                                pn0 = pn1; pn1 = bt;
                                r_head += 64u; r_pend -= 64u;
                            }
                            __builtin_amdgcn_wave_barrier();
                        }
                        cs += rl(ir, 63); qs += rl(iq, 63);
                        if (cur.shift + (int32_t)cs >= (int32_t)np) break;                              // the next op begins behind the window
                    }
                }
                }
                if (r_pend) {                                    // the pieces still waiting: a last, partial batch
                    const unsigned long long en = (uint32_t)lane < r_pend ? pcs[(r_head + (uint32_t)lane) & (uint32_t)(LPB - 1)] : 0ull;
                    const A2Small bt = acc2_load_small(a.s, a.stream_len, en);
                    acc2_work_small(pn0, cbase0);
                    pn0 = pn1; pn1 = bt;
                }
                acc2_work_small(pn0, cbase0); acc2_work_small(pn1, cbase0);
                __syncthreads();                                 // the list is free, every add is in
            }
        }
        // ---- flush, first half: the two difference arrays summed up (a wave scans its own positions with DPP, the waves before it enter
        // as one number each); the four copies of the counters added up by a thread per (chunk, even / odd) and staged a position at a time
        uint32_t dl[ABPW], cv[ABPW], sw[ABPW];
        uint32_t run_d = 0, run_c = 0, mxs = 0;
#pragma unroll
        for (int bb = 0; bb < ABPW; ++bb) {
            const uint32_t p = 64 * (wave * ABPW + bb) + lane;
            const uint32_t e = deld[p], f = covd[p];
            deld[p] = 0; covd[p] = 0;
            const uint32_t sd = wave_scan_u32(e), sc = wave_scan_u32(f);
            dl[bb] = run_d + sd; cv[bb] = run_c + sc;
            run_d += rl(sd, 63); run_c += rl(sc, 63);
            sw[bb] = nst[p >> 1];
            if ((p & 1u) == 0) nst[p >> 1] = 0;             // (two lanes share a word: both have read it -- the wave runs in lockstep)
        }
        if (t == 0) { deld[W] = 0; covd[W] = 0; }           // (an op / a segment that ends with the window leaves its -1 here)
        if (lane == 0) { misc[8 + wave] = run_d; misc[12 + wave] = run_c; }
        {
            const uint32_t m = (uint32_t)t >> 1, eo = (uint32_t)t & 1u;
            uint32_t s16[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // class x (bytes 0 and 2 | 1 and 3 of the word): two 16-bit sums each
#pragma unroll
            for (int c = 0; c < P_NCP; ++c) {
#pragma unroll
                for (int cl = 0; cl < 4; ++cl) {
                    uint32_t* wp = cnt + c * P_CPW + (2 * cl + (int)eo) * P_PLW + m;
                    const uint32_t w = *wp; *wp = 0;
                    s16[2 * cl] += w & 0x00FF00FFu; s16[2 * cl + 1] += (w >> 8) & 0x00FF00FFu;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {                     // position 8m + eo + 2i: byte i of the words
                const int h = i & 1, sh = (i >> 1) * 16;
                const uint32_t at = (eo + 2u * (uint32_t)i) * (uint32_t)(P_SW / 8) + m;
                stg[at] = ((s16[0 + h] >> sh) & 0xFFFFu) | (((s16[2 + h] >> sh) & 0xFFFFu) << 16);
                stg[P_SW + at] = ((s16[4 + h] >> sh) & 0xFFFFu) | (((s16[6 + h] >> sh) & 0xFFFFu) << 16);
            }
        }
        {   // the waves' carries are needed for the deepest SEG coverage, which decides whether anything is written: a barrier of its own
            __syncthreads();
            uint32_t carry_d = (uint32_t)wd, carry_c = 0;
            for (int k = 0; k < wave; ++k) { carry_d += misc[8 + k]; carry_c += misc[12 + k]; }
#pragma unroll
            for (int bb = 0; bb < ABPW; ++bb) { dl[bb] += carry_d; cv[bb] += carry_c; mxs = max(mxs, cv[bb]); }
            mxs = wave_max_u32(mxs);
            if (lane == 0) misc[16 + wave] = mxs;
            __syncthreads();
            mxs = max(max(misc[16], misc[17]), max(misc[18], misc[19]));
        }
        if (mxs > 255u || ovf) {                             // a byte-wide counter may have wrapped: the window is the general kernel's (rare: deeper than 255 and still within the slab)
            if (t == 0) a.deep[atomicAdd(a.n_deep, 1u)] = (uint32_t)tile;
            __syncthreads();                                 // (the staging is not read: the next window's pool may overwrite it)
            continue;
        }
        bgp_column* out = a.cols + col_out;
        uint32_t ncol = 0, scov = 0, nsum = 0, mx = 0;
#pragma unroll
        for (int bb = 0; bb < ABPW; ++bb) {
            const uint32_t p = 64 * (wave * ABPW + bb) + lane;
            const uint32_t at = (p & 7u) * (uint32_t)(P_SW / 8) + (p >> 3);
            const uint32_t qx = stg[at], qy = stg[P_SW + at];
            if (p < np) {
                const uint32_t cA = qx & 0xFFFFu, cC = qx >> 16, cG = qy & 0xFFFFu, cT = qy >> 16;
                const uint32_t D = dl[bb], S = (sw[bb] >> ((p & 1u) * 16)) & 0xFFFFu;
                const uint32_t cov = cv[bb] + D, cO = cv[bb] - (cA + cC + cG + cT);
                uint4* o4 = reinterpret_cast<uint4*>(out + p);
#if BGP_ACC2_ABL & 2
                if (cov == 0xFFFFFFF0u)                      // timing experiment: no column stores
#endif
                {
#if BGP_ACC2_NT_COLS
                nt_store16(o4, make_uint4(cov, S, cA, cC));      // (written once, read by nobody inside the step)
                nt_store16(o4 + 1, make_uint4(cG, cT, cO, D));
#else
                o4[0] = make_uint4(cov, S, cA, cC);
                o4[1] = make_uint4(cG, cT, cO, D);
#endif
                }
                ncol += cov > 0; scov += cov; mx = max(mx, cov); nsum += S;
#if !(BGP_ACC2_ABL & 128)                            // (128: timing experiment, no depth histogram)
                if (cov < (uint32_t)P_HB) atomicAdd(&hist[cov], 1u);
                else atomicAdd(&a.hist[min(cov, (uint32_t)BGP_DEPTH_BINS - 1)], 1ull);
#endif
            }
        }
        r_ncol += ncol; r_nsum += nsum; r_mx = max(r_mx, mx); r_scov += scov;
        __syncthreads();                                     // the staging is read: the next window's pool may use its place
    }
    {   // the range's totals: wave reductions with DPP, then one atomic per wave and total (the slots were zeroed by the host)
        const uint32_t w_ncol = wave_sum_u32(r_ncol), w_nsum = wave_sum_u32(r_nsum), w_mx = wave_max_u32(r_mx);
        const uint32_t s0 = wave_sum_u32((uint32_t)r_scov & 0xFFFFu), s1 = wave_sum_u32((uint32_t)(r_scov >> 16) & 0xFFFFu), s2 = wave_sum_u32((uint32_t)(r_scov >> 32) & 0xFFFFu), s3 = wave_sum_u32((uint32_t)(r_scov >> 48));
        if (lane == 0) {
            unsigned long long* ts = a.tile_stats + (a.n_tiles + blockIdx.x) * 4;
            if (w_ncol) atomicAdd(&ts[0], (unsigned long long)w_ncol);
            const unsigned long long sc = ((unsigned long long)s3 << 48) + ((unsigned long long)s2 << 32) + ((unsigned long long)s1 << 16) + s0;
            if (sc) atomicAdd(&ts[1], sc);
            if (w_mx) atomicMax(&ts[2], (unsigned long long)w_mx);
            if (w_nsum) atomicAdd(&ts[3], (unsigned long long)w_nsum);
        }
    }
    __syncthreads();
    for (int b = t; b < P_HB; b += AT) { const uint32_t h = hist[b]; if (h) atomicAdd(&a.hist[b], (unsigned long long)h); }
}

// ---------------------------------------------------------------------------------------------- K5

struct K5Args {
    const uint8_t* s; const bgp_record* rec; const uint64_t* pmax_end; const uint64_t* pmax_pos;
    RunDev run; uint64_t pos_begin, pos_end; const bgp_column* cols; uint64_t* cov64; const uint64_t* col_off; bgp_entry* entries;
    const uint64_t* ck_off; const Ckpt* ck; uint32_t ck_w;      // K4's long-read checkpoints (null / 0: none): per read and window, the cursor at the window's first column
    const uint4* cig4;                                          // K1's side array of first CIGAR ops, or null (a compacted table)
    uint64_t stream_len;
};

__global__ __launch_bounds__(256) void k5_cov(K5Args a) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.pos_end - a.pos_begin) return;
    a.cov64[k] = a.cols[a.run.column_base + (a.pos_begin - a.run.start_pos) + k].coverage;
}

// the cursor of read r at column offset rr, walked from a cursor that stands at or before it: op k, cs columns and qs query
// bases in front of it (pileup.d:175-222 in closed form, see the file header)
__device__ __forceinline__ bgp_entry entry_walk(const uint8_t* s, const bgp_record& r, uint32_t k, uint32_t cs, uint32_t qs, bool started, uint32_t rr, uint32_t ordinal) {
    const uint32_t l_name = r.bin_mq_nl & 0xFFu, n_cig = r.flag_nc & 0xFFFFu;
    const uint8_t* cg = s + r.rec_off + 36 + l_name;
    const uint8_t* sq = cg + 4ull * n_cig;
    const uint8_t* ql = sq + (uint32_t)((r.l_seq + 1) / 2);
    bgp_entry e; e.read_ordinal = ordinal; e.cigar_op_offset = 0; e.query_offset = 0; e.cigar_op_index = 0xFFFF; e.base = 0xFF; e.qual = 0xFF;
    for (; k < n_cig; ++k) {
        const uint32_t op = ld_u32(cg + 4 * k);
        const uint32_t len = op >> 4, qc = cig_qc(op), rc = cig_rc(op);
        if (!started) {
            if (rc) { if ((op & 15u) != 3u) started = true; else continue; }
            else { if (qc) qs += len; continue; }
        }
        if (!rc) { if (qc) qs += len; continue; }
        const uint32_t elen = len ? len : 1u;
        if (rr < cs + elen) {
            e.cigar_op_index = (uint16_t)k; e.cigar_op_offset = rr - cs;
            e.query_offset = qs + (qc ? rr - cs : 0u);
            if (qc) {
                if (e.query_offset < (uint32_t)r.l_seq) {
                    const uint32_t byte = sq[e.query_offset >> 1];
                    const uint32_t code = (e.query_offset & 1u) ? (byte & 0xFu) : (byte >> 4);
                    const uint64_t lo = 0x565352474D43413Dull, hi = 0x4E42444B48595754ull;   // "=ACMGRSVTWYHKDBN", base.d:85
                    e.base = (uint8_t)(((code & 8u) ? hi : lo) >> ((code & 7u) * 8));
                    e.qual = ql[e.query_offset];
                }
            } else { e.base = '-'; e.qual = 255; }                                         // pileup.d:119-121,131-133
            return e;
        }
        cs += elen; if (qc) qs += elen;
    }
    return e;
}

// where the walk for column offset rr of a read starts: its first op, or -- a long read -- the cursor K4 recorded for the
// window the column lies in (op index, columns before it, query bases before it; the op owns a column, so the cursor has
// started): O(ops inside a window) per entry instead of O(ops of the read)
__device__ __forceinline__ void entry_cursor(const bgp_record& r, uint32_t rr, uint32_t ordinal, const K5Args& a, uint32_t* k, uint32_t* cs, uint32_t* qs, bool* started) {
    const uint32_t n_cig = r.flag_nc & 0xFFFFu;
    *k = 0; *cs = 0; *qs = 0; *started = false;
    if (a.ck && wants_ckpt(n_cig, (uint32_t)r.l_seq)) {
        const uint64_t wj = ((uint64_t)(uint32_t)r.pos + rr - a.run.start_pos) / a.ck_w - ((uint64_t)(uint32_t)r.pos - a.run.start_pos) / a.ck_w;
        const Ckpt c = a.ck[a.ck_off[ordinal] + wj];
        if (c.k != 0xFFFFFFFFu) { *k = c.k; *cs = c.cs; *qs = c.qs; *started = true; }
        else *k = n_cig;                                                     // the cursor ran off the CIGAR before this window (pileup.d:224-228)
    }
}

// closed-form cursor of read r at column offset rr (see file header)
[[maybe_unused]] __device__ __forceinline__ bgp_entry entry_of(const uint8_t* s, const bgp_record& r, uint32_t rr, uint32_t ordinal, const K5Args& a) {
    uint32_t k, cs, qs; bool started;
    entry_cursor(r, rr, ordinal, a, &k, &cs, &qs, &started);
    return entry_walk(s, r, k, cs, qs, started, rr, ordinal);
}

// entry_of for a read of at most four CIGAR ops that are already in registers (the same closed form, no CIGAR load)
__device__ __forceinline__ bgp_entry entry_of_ops(const uint8_t* s, const bgp_record& r, const uint32_t (&ops)[4], uint32_t rr, uint32_t ordinal) {
    const uint32_t l_name = r.bin_mq_nl & 0xFFu, n_cig = r.flag_nc & 0xFFFFu;
    const uint8_t* sq = s + r.rec_off + 36 + l_name + 4ull * n_cig;
    const uint8_t* ql = sq + (uint32_t)((r.l_seq + 1) / 2);
    bgp_entry e; e.read_ordinal = ordinal; e.cigar_op_offset = 0; e.query_offset = 0; e.cigar_op_index = 0xFFFF; e.base = 0xFF; e.qual = 0xFF;
    uint32_t cs = 0, qs = 0; bool started = false, hit = false, hit_qc = false;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        if (k >= n_cig || hit) continue;
        const uint32_t op = ops[k], len = op >> 4, qc = cig_qc(op), rc = cig_rc(op);
        if (!started) {
            if (rc) { if ((op & 15u) != 3u) started = true; else continue; }
            else { if (qc) qs += len; continue; }
        }
        if (!rc) { if (qc) qs += len; continue; }
        const uint32_t elen = len ? len : 1u;
        if (rr < cs + elen) {
            e.cigar_op_index = (uint16_t)k; e.cigar_op_offset = rr - cs;
            e.query_offset = qs + (qc ? rr - cs : 0u);
            hit = true; hit_qc = qc != 0;
        } else { cs += elen; if (qc) qs += elen; }
    }
    if (hit) {
        if (hit_qc) {
            if (e.query_offset < (uint32_t)r.l_seq) {
                const uint32_t byte = sq[e.query_offset >> 1];
                const uint32_t code = (e.query_offset & 1u) ? (byte & 0xFu) : (byte >> 4);
                const uint64_t lo = 0x565352474D43413Dull, hi = 0x4E42444B48595754ull;   // "=ACMGRSVTWYHKDBN", base.d:85
                e.base = (uint8_t)(((code & 8u) ? hi : lo) >> ((code & 7u) * 8));
                e.qual = ql[e.query_offset];
            }
        } else { e.base = '-'; e.qual = 255; }                                         // pileup.d:119-121,131-133
    }
    return e;
}

// The cursor of a read of at most three ops in closed form, without a branch (pileup.d:175-222; entry_of_ops above is the same walk
// written as one): every op that OWNS columns -- consumes reference, and is not an N in front of the first such op (the reference's
// leading-N quirk) -- is a STOP: its first column cs, its length (a zero-length op owns one column), the query bases in front of it,
// its index and whether it consumes query.  The cursor at column rr stands on the last stop with cs <= rr; it has run off the
// CIGAR when rr lies behind that stop's columns (pileup.d:224-228).  (All of it by value: the optimizer turns a select between
// values it finds in memory -- array elements, variables a lambda captured by reference -- into an indexed load from scratch.)
struct Stop { uint32_t cs, len, qb, ix; };
struct Stops { Stop a, b, c; };
__device__ __forceinline__ Stops stops_of(uint32_t n_cig, uint32_t op0, uint32_t op1, uint32_t op2) {
    Stops r; uint32_t cs = 0, qs = 0; bool started = false;
#define BGP_STOP(K, OP, OUT) { \
        const uint32_t len = (OP) >> 4; \
        const bool on = (K) < n_cig, qc = on && cig_qc(OP), rc = on && cig_rc(OP); \
        const bool stop = rc && (started || ((OP) & 15u) != 3u); \
        started |= stop; \
        const uint32_t elen = len ? len : 1u; \
        OUT.cs = stop ? cs : 0xFFFFFFFFu; OUT.len = elen; OUT.qb = qs; OUT.ix = (K) | (qc ? 0x10000u : 0u); \
        cs += stop ? elen : 0u; \
        qs += qc ? (stop ? elen : (rc ? 0u : len)) : 0u; }
    BGP_STOP(0u, op0, r.a) BGP_STOP(1u, op1, r.b) BGP_STOP(2u, op2, r.c)
#undef BGP_STOP
    return r;
}
struct CursorAt { uint32_t ix, off, qo; bool hit, hqc; };
__device__ __forceinline__ CursorAt cursor_at(const Stops st, uint32_t rr) {
    const bool c2 = rr >= st.c.cs, c1 = rr >= st.b.cs, c0 = rr >= st.a.cs;
    const uint32_t cs = c2 ? st.c.cs : c1 ? st.b.cs : st.a.cs, len = c2 ? st.c.len : c1 ? st.b.len : st.a.len;
    const uint32_t qb = c2 ? st.c.qb : c1 ? st.b.qb : st.a.qb, ix = c2 ? st.c.ix : c1 ? st.b.ix : st.a.ix;
    CursorAt r; r.off = rr - cs; r.hit = (c0 || c1 || c2) && r.off < len; r.hqc = (ix >> 16) != 0; r.ix = ix & 0xFFFFu;
    r.qo = qb + (r.hqc ? r.off : 0u);
    return r;
}
// One wave per 64 consecutive COLUMNS, lane = column: the candidate rows of the block are scanned once (64 rows a step,
// coalesced), every row that touches the block is handed to all lanes in file order, and a lane whose column the read
// covers appends the read's entry to its own column -- a counter a lane, no ranks, no ballots per entry.  (With a wave per
// column every column scanned the whole candidate range for itself: ~1000 rows on cfg3, where a read with an N skip keeps
// every later read a candidate -- 32 KB of rows for ~50 entries, 200 GB/s of entries.)
#ifndef BGP_K5R
#define BGP_K5R 4
#endif
#ifndef BGP_K5_MINW
#define BGP_K5_MINW 4                         // waves a SIMD the register allocation leaves room for
#endif
#ifndef BGP_K5_ABL
#define BGP_K5_ABL 0                          // timing experiments: 1 searches only, 2 no entries of listed reads, 3 their entries computed but not stored
#endif
constexpr int K5R = BGP_K5R;            // entries of a column that leave together (a power of two)
static_assert(sizeof(bgp_entry) == 16, "an entry is one 16-byte store");
// CK: the stream holds long reads (K4 left checkpoints): the route that settles a read's ops by the wave is compiled in.  (A stream without them
// gets an instantiation without that route's registers.)
template <bool CK> __global__ __launch_bounds__(256, BGP_K5_MINW) void k5_fill_blk(K5Args a) {
    const uint64_t np = a.pos_end - a.pos_begin;
    const uint64_t col0 = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    const int lane = threadIdx.x & 63;
    if (col0 >= np) return;
    const uint64_t p0 = a.pos_begin + col0, p1 = min(p0 + 64, a.pos_end);           // the block's positions [p0, p1)
    const uint64_t p = p0 + lane;
    const bool valid = p < p1;
    const uint64_t rk = ref_key(a.run.ref_id) << 32;
    const uint64_t i_lo = lower_bound_gt(a.pmax_end, a.run.first_record, a.run.end_record, rk | (uint32_t)p0);          // first record whose running maximum of end_position exceeds p0
    const uint64_t i_hi = lower_bound_gt(a.pmax_pos, a.run.first_record, a.run.end_record, rk | (uint32_t)(p1 - 1));    // first record that starts behind the block
    uint64_t w = valid ? a.col_off[col0 + lane] : 0;
#if BGP_K5_ABL == 1
    if (i_lo + w == 0xFFFFFFFFFFFFull) a.entries[0].qual = 1; return;
#endif
    // Two ways to the same entries, by the read.  An ordinary read (up to four CIGAR ops, which came with its row: K1's side array) is
    // listed in LDS; when 64 are listed -- or a read of the other kind comes up, or the rows run out -- a LANE takes a listed READ and the
    // wave goes through the block's columns: the lanes whose reads cover the column work out their entries, and since lanes are in file
    // order the entries of a column leave as one run of consecutive 16-byte stores (a counter a column, kept by the column's lane).
    // Round 2 and 3 handed every read to all lanes (lane = column) and wrote each column's entries eight at a time from a ring in LDS:
    // 64 different lines per store instruction, 2.24 ms for 2.1 * 10^8 entries of cfg3.  That way is kept for the reads that need a
    // CIGAR walk (long reads: the cursor at the block's first column is found ONCE for the 64 lanes, 64 ops a step from the window's
    // checkpoint); the ring is flushed where the two ways alternate, so a column's entries stay in file order.
    // a wave's LDS: the ring (K5R x 64 entries), behind it the listed reads -- row (two 16-byte halves) and first three CIGAR ops with the
    // ordinal, a read a slot; while a list is flushed (the ring is empty then, and every lane holds its read in registers) the same
    // words hold the reads' bases and qualities, word k of lane l at [k][l]
    __shared__ uint4 s_wave[4][K5R + 3][64];
    static_assert(K5R + 3 >= 7, "27 words a lane of staged bases and qualities");
    uint4 (*ring)[64] = s_wave[threadIdx.x >> 6];
    uint4 (*list)[64] = s_wave[threadIdx.x >> 6] + K5R;
    uint32_t* const stg = reinterpret_cast<uint32_t*>(s_wave[threadIdx.x >> 6]) + lane;      // word k of this lane: stg[64 * k]
    uint32_t cnt = 0, n_list = 0;
    uint32_t pf_op = 0; int pf_src = -1; uint64_t pf_base = ~0ull;      // the ops of the NEXT checkpointed read of the step, asked for while this one is worked on
    auto flush_ring = [&]() {
        for (uint32_t k = 0; k < (cnt & (K5R - 1)); ++k) { const uint4 x = ring[k][lane]; __builtin_memcpy(a.entries + w + k, &x, 16); }
        w += cnt & (K5R - 1); cnt = 0;
    };
    auto flush_list = [&]() {
        if (!n_list) return;
#if BGP_K5_ABL == 2
        n_list = 0; return;
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // Lane = (read, stretch of columns): G lanes a read, each with 64 / G of the block's columns -- G = 1 with more than 32 reads
        // listed, 8 with eight or fewer (the handful left over when 64 were listed and flushed: eight column steps instead of 64).
        // The lanes that stand on the same column in a step are those of one stretch (lane mod G), in file order.
        const uint32_t lg = n_list > 32u ? 0u : n_list > 16u ? 1u : n_list > 8u ? 2u : 3u, G = 1u << lg, CW = 64u >> lg;
        const uint32_t ri = (uint32_t)lane >> lg, cg = (uint32_t)lane & (G - 1u);
        const bool have = ri < n_list;
        uint4 x = make_uint4(0, 0, 0, 0), y = x, z = x;
        if (have) { x = list[0][ri]; y = list[1][ri]; z = list[2][ri]; }
        const uint32_t pos = x.y, end_pos = x.z, n_cig = y.x & 0xFFFFu, l_seq = y.y, ord = z.w;      // (a listed read has at most three ops: the slot's fourth word is its ordinal)
        const uint64_t rec_off = ((uint64_t)y.w << 32) | y.z;
        const uint64_t sq_off = rec_off + 36 + (x.w & 0xFFu) + 4ull * n_cig;
        const uint8_t* const sq = a.s + sq_off;
        const uint8_t* const ql = sq + ((l_seq + 1) >> 1);
        const Stops st = stops_of(n_cig, z.x, z.y, z.z);
        // the read's columns of the block that fall into this lane's stretch [j0, j0 + CW)
        const uint32_t j0 = cg * CW;
        const uint32_t r_lo = have ? (uint32_t)(max(p0, (uint64_t)pos) - p0) : 64u, r_hi = have ? (uint32_t)(min(p1, (uint64_t)end_pos) - p0) : 0u;
        const uint32_t c_lo = max(r_lo, j0), c_hi = min(r_hi, j0 + CW);
        const uint32_t rr0 = (uint32_t)p0 - pos;                               // column j of the block is column rr0 + j of the read
        const unsigned long long pat = lg == 0u ? ~0ull : lg == 1u ? 0x5555555555555555ull : lg == 2u ? 0x1111111111111111ull : 0x0101010101010101ull;
        const unsigned long long gmask = pat << cg;                            // the lanes of this lane's stretch
        const unsigned long long cmask = pat << ((uint32_t)lane >> (6u - lg)); // ... and, the lane taken as a COLUMN (w is kept a column a lane), of the stretch that column lies in
        // The bases and qualities a lane's entries need are STAGED in LDS before the first entry leaves: 72 query bases from the
        // read's query offset at the lane's first column on (query offsets only grow from there) -- 9 words of nibbles, 18 of
        // qualities -- so that the column steps hold no load and wait for none (on this chip a wait for a load
        // is also one for the stores issued since: vmcnt counts both; with the loads inside the loop, a group of columns ahead, the
        // kernel spent 0.4 of its 1.4 ms on cfg3 draining its own stores).  An entry whose query offset lies outside the staged 72
        // (more than eight inserted bases inside the block) loads its two bytes in a second pass over the lane's columns.
        const CursorAt a0 = cursor_at(st, rr0 + min(c_lo, 63u));
        const uint32_t qa = a0.qo & ~1u;                                        // staged: query bases [qa, qa + 72): a stretch's 64 and room for insertions
        const bool want = have && c_lo < c_hi && a0.hit && qa < l_seq;
        const bool safe = want && sq_off + ((l_seq + 1) >> 1) + qa + 72 <= a.stream_len;
        bool slow = want && !safe;                         // this lane has an entry whose bases are not among the staged: its entries are written again below
        {
            uint4 n0 = make_uint4(0, 0, 0, 0), n1 = n0, q0 = n0, q1 = n0, q2 = n0, q3 = n0; uint32_t n2 = 0; uint2 q4 = make_uint2(0, 0);
            if (BGP_K5_ABL != 4 && safe) {
                const uint8_t* sp = sq + (qa >> 1); const uint8_t* qp = ql + qa;
                __builtin_memcpy(&n0, sp, 16); __builtin_memcpy(&n1, sp + 16, 16); __builtin_memcpy(&n2, sp + 32, 4);
                __builtin_memcpy(&q0, qp, 16); __builtin_memcpy(&q1, qp + 16, 16); __builtin_memcpy(&q2, qp + 32, 16); __builtin_memcpy(&q3, qp + 48, 16); __builtin_memcpy(&q4, qp + 64, 8);
            }
            __builtin_amdgcn_wave_barrier();               // (every lane has its read out of the list: the words are free)
            stg[64 * 0] = n0.x; stg[64 * 1] = n0.y; stg[64 * 2] = n0.z; stg[64 * 3] = n0.w; stg[64 * 4] = n1.x; stg[64 * 5] = n1.y; stg[64 * 6] = n1.z; stg[64 * 7] = n1.w; stg[64 * 8] = n2;
            stg[64 * 9] = q0.x; stg[64 * 10] = q0.y; stg[64 * 11] = q0.z; stg[64 * 12] = q0.w; stg[64 * 13] = q1.x; stg[64 * 14] = q1.y; stg[64 * 15] = q1.z; stg[64 * 16] = q1.w;
            stg[64 * 17] = q2.x; stg[64 * 18] = q2.y; stg[64 * 19] = q2.z; stg[64 * 20] = q2.w; stg[64 * 21] = q3.x; stg[64 * 22] = q3.y; stg[64 * 23] = q3.z; stg[64 * 24] = q3.w;
            stg[64 * 25] = q4.x; stg[64 * 26] = q4.y;
        }
        // one step: every stretch its column JJ; the entries of the reads that cover a column leave as one run of stores, in file order
#define BGP_K5_COLUMN(JJ, STAGED) { \
            const uint32_t j_ = j0 + (JJ); \
            const bool cov_all = j_ >= c_lo && j_ < c_hi; \
            const bool cov = cov_all && (STAGED || slow); \
            const unsigned long long m = __ballot(cov_all); \
            if (m) { \
                const unsigned long long gm = m & gmask; \
                const uint32_t gn = (uint32_t)__popcll(gm); \
                const uint64_t wj = (((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute((int)(j_ << 2), (int)(uint32_t)(w >> 32)) << 32) | (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j_ << 2), (int)(uint32_t)w)) - (STAGED ? 0u : gn); \
                if (cov) { \
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(gm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)gm, 0u)); \
                    const CursorAt at = cursor_at(st, rr0 + j_); \
                    const bool ins = at.hit && at.hqc && at.qo < l_seq; \
                    uint32_t sb, qv; \
                    if (STAGED) { /* nibble word t >> 3: the base's byte is (t & 7) >> 1 (the even base the high nibble); quality word 9 + (t >> 2) */ \
                        const uint32_t t = at.qo - qa; \
                        slow |= BGP_K5_ABL != 4 && ins && t >= 72u; \
                        const uint32_t tt = min(t, 71u); \
                        sb = stg[64u * (tt >> 3)] >> (8u * ((tt >> 1) & 3u)); \
                        qv = stg[64u * (9u + (tt >> 2))] >> (8u * (tt & 3u)); \
                    } else { sb = ins ? sq[at.qo >> 1] : 0u; qv = ins ? ql[at.qo] : 0u; } \
                    const uint32_t code = (sb >> ((at.qo & 1u) ? 0u : 4u)) & 15u; \
                    const uint32_t c8 = code & 7u;      /* "=ACMGRSV" "TWYHKDBN" (base.d:85), a byte picked by v_perm */ \
                    const uint32_t blo = __builtin_amdgcn_perm(0x56535247u, 0x4D43413Du, c8), bhi = __builtin_amdgcn_perm(0x4E42444Bu, 0x48595754u, c8); \
                    const uint32_t bq = (((code & 8u) ? bhi : blo) & 0xFFu) | ((qv & 0xFFu) << 8); \
                    /* no hit (the cursor ran off the CIGAR): index 0xFFFF, offsets 0, base and quality 0xFF; D / N: '-' and 255 */ \
                    /* (pileup.d:119-121,131-133); a query offset behind SEQ: 0xFF, 0xFF */ \
                    const uint32_t tail = !at.hit ? 0xFFFFu : !at.hqc ? (0xFF00u | (uint32_t)'-') : ins ? bq : 0xFFFFu; \
                    uint4 v; \
                    v.x = ord; v.y = at.hit ? at.off : 0u; v.z = at.hit ? at.qo : 0u; v.w = (at.hit ? at.ix : 0xFFFFu) | (tail << 16); \
                    K5_STORE(a.entries + wj + rank, v) \
                } \
                /* the lane as a column: this step was its column's if it is column JJ of its stretch */ \
                if (STAGED && ((uint32_t)lane & (CW - 1u)) == (JJ)) w += (uint64_t)__popcll(m & cmask); \
            } }
#if BGP_K5_ABL == 3
#define K5_STORE(P, V) if (V.x == 0xFFFFFFF3u && V.w == 77u) __builtin_memcpy(P, &V, 16);
#else
#define K5_STORE(P, V) __builtin_memcpy(P, &V, 16);
#endif
#pragma unroll 1
        for (uint32_t jj = 0; jj < CW; ++jj) BGP_K5_COLUMN(jj, true)
        if (BGP_K5_ABL != 5 && __ballot(slow)) {
#pragma unroll 1
            for (uint32_t jj = 0; jj < CW; ++jj) BGP_K5_COLUMN(jj, false)
        }
#undef K5_STORE
#undef BGP_K5_COLUMN
        n_list = 0;
        __builtin_amdgcn_wave_barrier();
    };
    for (uint64_t base = i_lo; base < i_hi; base += 64) {
        const uint64_t i = base + lane;
        bgp_record r{};
        bool hit = false;
        uint4 c4 = make_uint4(0, 0, 0, 0);
        if (i < i_hi) { r = a.rec[i]; hit = r.end_pos != r.pos && (uint64_t)(uint32_t)r.pos < p1 && (uint64_t)(uint32_t)r.end_pos > p0; if (hit && a.cig4) c4 = a.cig4[i]; }
        // a long read's checkpoint for the window its first column of the block lies in, asked for here by the read's own lane: 64 rows'
        // two dependent loads at once, not two round trips for every long read in its turn
        Ckpt ckr{0xFFFFFFFFu, 0u, 0u, 0u};
        const bool ckd = CK && hit && a.ck && wants_ckpt(r.flag_nc & 0xFFFFu, (uint32_t)r.l_seq);
        if (ckd) {
            const uint64_t rr0 = max(p0, (uint64_t)(uint32_t)r.pos) - (uint32_t)r.pos;
            const uint64_t wj = ((uint64_t)(uint32_t)r.pos + rr0 - a.run.start_pos) / a.ck_w - ((uint64_t)(uint32_t)r.pos - a.run.start_pos) / a.ck_w;
            ckr = a.ck[a.ck_off[i] + wj];
        }
        const bool small_r = hit && a.cig4 && (r.flag_nc & 0xFFFFu) <= 3u;      // (three ops: the list's third slot carries them and the read's ordinal)
        unsigned long long mh = __ballot(hit);
        const unsigned long long ms = __ballot(small_r);
        while (mh) {
            // the run of listable reads in front of the next read of the other kind (all of them, if there is none)
            const unsigned long long big = mh & ~ms;
            const int stop = big ? __ffsll((long long)big) - 1 : 64;
            const unsigned long long runm = stop >= 64 ? mh : (mh & ((1ull << stop) - 1));
            if (runm) {
                if (cnt & (K5R - 1)) flush_ring(); else cnt = 0;
                unsigned long long todo = runm;
                while (todo) {
                    const uint32_t room = 64u - n_list, have_n = (uint32_t)__popcll(todo);
                    // the first `room` of them
                    unsigned long long take = todo;
                    if (have_n > room) { unsigned long long t2 = todo; for (uint32_t k = 0; k < room; ++k) t2 &= t2 - 1; take = todo & ~t2; }
                    if ((take >> lane) & 1ull) {
                        const uint32_t slot = n_list + (uint32_t)__popcll(take & ((1ull << lane) - 1));
                        list[0][slot] = make_uint4((uint32_t)r.ref_id, (uint32_t)r.pos, (uint32_t)r.end_pos, r.bin_mq_nl);
                        list[1][slot] = make_uint4(r.flag_nc, (uint32_t)r.l_seq, (uint32_t)r.rec_off, (uint32_t)(r.rec_off >> 32));
                        list[2][slot] = make_uint4(c4.x, c4.y, c4.z, (uint32_t)i);
                    }
                    n_list += (uint32_t)__popcll(take);
                    todo &= ~take;
                    if (n_list == 64u) flush_list();
                }
                mh &= ~runm;
            }
            if (stop < 64) {
                flush_list();
                const int src = stop;
                mh &= ~(1ull << src);
                bgp_record q;                                                           // the read, the same in every lane (scalar registers)
                q.ref_id = __builtin_amdgcn_readlane(r.ref_id, src); q.pos = __builtin_amdgcn_readlane(r.pos, src); q.end_pos = __builtin_amdgcn_readlane(r.end_pos, src);
                q.bin_mq_nl = (uint32_t)__builtin_amdgcn_readlane((int)r.bin_mq_nl, src); q.flag_nc = (uint32_t)__builtin_amdgcn_readlane((int)r.flag_nc, src);
                q.l_seq = __builtin_amdgcn_readlane(r.l_seq, src);
                q.rec_off = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(r.rec_off >> 32), src) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)r.rec_off, src);
                const bool small = a.cig4 && (q.flag_nc & 0xFFFFu) <= 4u;           // its ops came with the row (K1's side array): no CIGAR load per entry
                const uint32_t ops[4] = {(uint32_t)__builtin_amdgcn_readlane((int)c4.x, src), (uint32_t)__builtin_amdgcn_readlane((int)c4.y, src), (uint32_t)__builtin_amdgcn_readlane((int)c4.z, src), (uint32_t)__builtin_amdgcn_readlane((int)c4.w, src)};
                // any other read: the cursor at the block's first column of the read, found ONCE for the 64 lanes (for a long read from
                // the window's checkpoint: every lane walking there by itself was ~35 dependent CIGAR loads per entry on cfg5)
                uint32_t bk = 0, bcs = 0, bqs = 0; bool bst = false;
                const bool mine = valid && (uint64_t)(uint32_t)q.pos <= p && (uint64_t)(uint32_t)q.end_pos > p;      // the read covers this lane's column
                // A read walked from its window's checkpoint: WHICH op owns each of the block's columns is settled by the wave, 64 ops a step
                // with a lane an op -- two scans give every op its first column and query base, an op that owns columns of the block leaves its
                // lane number at the first of them (LDS, 64 words), a running maximum over the columns hands every column its op, three
                // ds_bpermute its data.  (Every lane then walking from the block's first column by itself was a CIGAR load per op and
                // lane: cfg5, an op every ~10 columns, 3.5 ms for 1.2 * 10^8 entries.)
                bool res = false; uint32_t r_ix = 0xFFFFu, r_off = 0, r_qo = 0; bool r_hit = false, r_qc = false;
                // (a read is two round trips -- its ops, then its bases -- and the reads of a block come one after the other: the first 64 ops of
                // the next checkpointed read of this step are asked for before this read's are looked at)
                const bool have_pf = CK && pf_src == src && pf_base == base;
                const uint32_t my_pf = pf_op;
                if (CK) {
                    const unsigned long long rest = mh & ~ms;                      // (mh no longer holds src)
                    pf_src = -1;
                    if (rest) {
                        const int ns = __ffsll((long long)rest) - 1;
                        if (__builtin_amdgcn_readlane((int)ckd, ns)) {
                            const uint32_t kn = (uint32_t)__builtin_amdgcn_readlane((int)ckr.k, ns);
                            if (kn != 0xFFFFFFFFu) {
                                const uint64_t ro = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(r.rec_off >> 32), ns) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)r.rec_off, ns);
                                const uint32_t ncn = (uint32_t)__builtin_amdgcn_readlane((int)r.flag_nc, ns) & 0xFFFFu, lnn = (uint32_t)__builtin_amdgcn_readlane((int)r.bin_mq_nl, ns) & 0xFFu;
                                const uint32_t kk = kn + (uint32_t)lane;
                                pf_op = kk < ncn ? ld_u32(a.s + ro + 36 + lnn + 4ull * kk) : 0u;
                                pf_src = ns; pf_base = base;
                            }
                        }
                    }
                }
                if (!small) {
                    const uint32_t rr0 = (uint32_t)(max(p0, (uint64_t)(uint32_t)q.pos) - (uint32_t)q.pos);
                    if (CK && __builtin_amdgcn_readlane((int)ckd, src)) {      // entry_cursor, from what the read's lane loaded with its row
                        const uint32_t k_ = (uint32_t)__builtin_amdgcn_readlane((int)ckr.k, src);
                        if (k_ != 0xFFFFFFFFu) { bk = k_; bcs = (uint32_t)__builtin_amdgcn_readlane((int)ckr.cs, src); bqs = (uint32_t)__builtin_amdgcn_readlane((int)ckr.qs, src); bst = true; }
                        else bk = q.flag_nc & 0xFFFFu;                    // the cursor ran off the CIGAR before this window (pileup.d:224-228)
                    }
                    if (CK && bst) {
                        const uint32_t n_cig = q.flag_nc & 0xFFFFu;
                        const uint8_t* cg = a.s + q.rec_off + 36 + (q.bin_mq_nl & 0xFFu);
                        const uint32_t c0 = (uint32_t)(max(p0, (uint64_t)(uint32_t)q.pos) - p0);      // the read's first column of the block
                        const uint32_t rrc = (uint32_t)(p - (uint32_t)q.pos);                          // this lane's column, counted from the read's first
                        uint32_t* const mk = reinterpret_cast<uint32_t*>(list);                         // (the list is empty here: flush_list above)
                        bool first_step = true;
                        while (bk < n_cig && __ballot(mine && !res)) {
                            const uint32_t kk = bk + (uint32_t)lane;
                            const uint32_t op = first_step && have_pf ? my_pf : (kk < n_cig ? ld_u32(cg + 4ull * kk) : 0u);
                            first_step = false;
                            const bool on = kk < n_cig, qc = on && cig_qc(op), rc = on && cig_rc(op);
                            const uint32_t len = op >> 4, elen = len ? len : 1u;
                            const uint32_t cl = rc ? elen : 0u, ql = qc ? (rc ? elen : len) : 0u;
                            const uint32_t ic = wave_incl_u32(cl, lane), iq = wave_incl_u32(ql, lane);
                            const uint32_t s_k = bcs + ic - cl, q_k = bqs + iq - ql;                   // the op's first column; the query bases in front of it
                            mk[lane] = 0u;
                            __builtin_amdgcn_wave_barrier();
                            if (cl && s_k + cl > rr0) { const uint32_t at = max(s_k, rr0) - rr0 + c0; if (at < 64u) mk[at] = (uint32_t)lane + 1u; }
                            __builtin_amdgcn_wave_barrier();
                            uint32_t own = mk[lane];
#pragma unroll
                            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)own, d, 64); if (lane >= d) own = max(own, o); }
                            const int ol = (int)own - 1;
                            const uint32_t so = (uint32_t)__shfl((int)s_k, ol < 0 ? 0 : ol, 64), qo = (uint32_t)__shfl((int)q_k, ol < 0 ? 0 : ol, 64), ow = (uint32_t)__shfl((int)op, ol < 0 ? 0 : ol, 64);
                            const uint32_t olen = (ow >> 4) ? (ow >> 4) : 1u;
                            if (mine && !res && own && rrc >= so && rrc - so < olen) {
                                res = true; r_hit = true; r_ix = bk + (uint32_t)ol; r_off = rrc - so; r_qc = cig_qc(ow) != 0; r_qo = qo + (r_qc ? r_off : 0u);
                            }
                            __builtin_amdgcn_wave_barrier();
                            bcs += (uint32_t)__builtin_amdgcn_readlane((int)ic, 63); bqs += (uint32_t)__builtin_amdgcn_readlane((int)iq, 63);
                            bk = min(bk + 64u, n_cig);
                        }
                    }
                }
                if (CK && !small && bst) {
                    if (mine) {      // the entry as entry_walk writes it (a column no op owns: the cursor ran off the CIGAR, pileup.d:224-228)
                        bgp_entry e; e.read_ordinal = (uint32_t)(base + (uint64_t)src); e.cigar_op_index = (uint16_t)(r_hit ? r_ix : 0xFFFFu);
                        e.cigar_op_offset = r_hit ? r_off : 0u; e.query_offset = r_hit ? r_qo : 0u; e.base = 0xFF; e.qual = 0xFF;
                        if (r_hit && !r_qc) { e.base = '-'; e.qual = 255; }
                        else if (r_hit && r_qo < (uint32_t)q.l_seq) {
                            const uint8_t* sq = a.s + q.rec_off + 36 + (q.bin_mq_nl & 0xFFu) + 4ull * (q.flag_nc & 0xFFFFu);
                            const uint32_t byte = sq[r_qo >> 1], code = (r_qo & 1u) ? (byte & 0xFu) : (byte >> 4);
                            const uint64_t lo = 0x565352474D43413Dull, hi = 0x4E42444B48595754ull;   // "=ACMGRSVTWYHKDBN", base.d:85
                            e.base = (uint8_t)(((code & 8u) ? hi : lo) >> ((code & 7u) * 8));
                            e.qual = sq[(uint32_t)((q.l_seq + 1) / 2) + r_qo];
                        }
                        uint4 v; __builtin_memcpy(&v, &e, 16);
                        ring[cnt & (K5R - 1)][lane] = v;
                        ++cnt;
                        if ((cnt & (K5R - 1)) == 0) {
#pragma unroll
                            for (int k = 0; k < K5R; ++k) { const uint4 x = ring[k][lane]; __builtin_memcpy(a.entries + w + k, &x, 16); }
                            w += K5R;
                        }
                    }
                } else
                if (valid && (uint64_t)(uint32_t)q.pos <= p && (uint64_t)(uint32_t)q.end_pos > p) {
                    bgp_entry e;
                    if (small) e = entry_of_ops(a.s, q, ops, (uint32_t)(p - (uint32_t)q.pos), (uint32_t)(base + (uint64_t)src));
                    else e = entry_walk(a.s, q, bk, bcs, bqs, bst, (uint32_t)(p - (uint32_t)q.pos), (uint32_t)(base + (uint64_t)src));
                    uint4 v; __builtin_memcpy(&v, &e, 16);
                    ring[cnt & (K5R - 1)][lane] = v;
                    ++cnt;
                    if ((cnt & (K5R - 1)) == 0) {
#pragma unroll
                        for (int k = 0; k < K5R; ++k) { const uint4 x = ring[k][lane]; __builtin_memcpy(a.entries + w + k, &x, 16); }
                        w += K5R;
                    }
                }
            }
        }
    }
    flush_list();
    flush_ring();
}

// one wave per column: stream-compact the overlapping reads in file order
#ifdef BGP_EXPERIMENTS
__global__ __launch_bounds__(256) void k5_fill(K5Args a) {
    const uint64_t col = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (col >= a.pos_end - a.pos_begin) return;
    const uint64_t p = a.pos_begin + col;
    const uint64_t rk = ref_key(a.run.ref_id) << 32;
    const uint64_t i_lo = lower_bound_gt(a.pmax_end, a.run.first_record, a.run.end_record, rk | (uint32_t)p);
    const uint64_t i_hi = lower_bound_gt(a.pmax_pos, a.run.first_record, a.run.end_record, rk | (uint32_t)p);
    uint64_t w = a.col_off[col];
    for (uint64_t base = i_lo; base < i_hi; base += 64) {
        const uint64_t i = base + lane;
        bool alive = false; bgp_record r{};
        if (i < i_hi) { r = a.rec[i]; alive = r.end_pos != r.pos && (uint64_t)(uint32_t)r.pos <= p && (uint64_t)(uint32_t)r.end_pos > p; }
        const unsigned long long m = __ballot(alive);
        if (alive) {
            const uint32_t rank = __popcll(m & ((1ull << lane) - 1));
            a.entries[w + rank] = entry_of(a.s, r, (uint32_t)(p - (uint32_t)r.pos), (uint32_t)i, a);
        }
        w += __popcll(m);
    }
}
#endif

}  // namespace

int k34_pileup(bgp_ctx* c) {
    const uint64_t n = c->n_records;
    prefetch_cancel(c);                                        // (a window of the previous result may still be on its way to the host)
    c->runs.clear();
    memset(&c->pinfo, 0, sizeof c->pinfo);
    BGP_TRY(c->d_pstats.reserve(256));
    BGP_TRY(c->d_depth_hist.reserve((size_t)BGP_DEPTH_BINS * 8));
    { const FillItem f[3] = {{c->d_pstats.p, 256, 0u}, {c->d_depth_hist.p, (uint64_t)BGP_DEPTH_BINS * 8, 0u}, {c->d_status.p, 64, 0u}}; BGP_TRY(fill_many(c, f, 3)); }
    if (n == 0) { BGP_HIP(hipStreamSynchronize(c->stream)); c->pinfo.device_depth_hist = c->d_depth_hist.as<uint64_t>(); return BGP_OK; }
    if (n >= (1ull << 32)) return BGP_EINVAL;                  // read_ordinal is 32 bit
    BGP_TRY(c->d_pmax_end.reserve(n * 8)); BGP_TRY(c->d_pmax_pos.reserve(n * 8));
    const unsigned nb = (unsigned)((n + 255) / 256);
    uint32_t* status = c->d_status.as<uint32_t>();
    uint32_t st[4] = {0, 0, 0, 0};
    constexpr uint32_t FIRST_RUNS = 16;
    RunDev first_runs[FIRST_RUNS];
    int rc = BGP_OK;
    for (uint32_t run_cap = std::max<uint32_t>(RUNS_FIRST, (uint32_t)(c->d_runs.cap / sizeof(RunDev)));;) {
        BGP_TRY(c->d_runs.reserve(sizeof(RunDev) * (size_t)run_cap));
        {
            TimedRegion t(c, "k3_keys_scan", n * (c->tilemax_ok ? 32 + 16 : 32 + 16 + 32));      // (one pass over the table when K1 left the tile maxima)
            static const uint32_t run_gap = getenv("BGP_RUN_GAP") ? (uint32_t)atoll(getenv("BGP_RUN_GAP")) : 65536u;      // (tests: small islands)
            K3Args a{c->d_rec.as<bgp_record>(), n, c->d_pmax_end.as<uint64_t>(), c->d_pmax_pos.as<uint64_t>(), status, run_cap, run_gap};
            const uint64_t nt = (n + K3_TILE - 1) / K3_TILE;
            BGP_TRY(c->d_k3_bsum.reserve((nt + 2) * sizeof(K3Key))); c->k3_bsum_n = n;      // (a buffer of its own: k4_bounds searches it after other scans have used d_scan_tmp)
            K3Key* bsum = c->d_k3_bsum.as<K3Key>();
            if (c->tilemax_ok)      // K1 left the tiles' maxima behind (it had every row in registers): no first pass over the table
                BGP_HIP(hipMemcpyAsync(bsum, c->d_tilemax.p, nt * sizeof(K3Key), hipMemcpyDeviceToDevice, c->stream));
            else
                hipLaunchKernelGGL(k3_reduce, dim3((unsigned)nt), dim3(K3_T), 0, c->stream, a, bsum);
            hipLaunchKernelGGL(k3_bsum, dim3(1), dim3(K3_T), 0, c->stream, bsum, nt);
            hipLaunchKernelGGL(k3_down, dim3((unsigned)nt), dim3(K3_T), 0, c->stream, a, bsum, c->d_runs.as<RunDev>());
        }
        hipLaunchKernelGGL(k3_finish, dim3(1), dim3(1024), 0, c->stream, c->d_runs.as<RunDev>(), status, run_cap, c->d_pmax_end.as<uint64_t>(), n, (uint32_t)W);
        { const MailItem it[2] = {{status, st, 16}, {c->d_runs.p, first_runs, sizeof first_runs}}; BGP_TRY(mail_fetch(c, it, 2)); }      // (the table has room for RUNS_FIRST entries at least)
        rc = bgp_status_from_bits(st[0]);
        if (rc < 0 && rc != BGP_ESEQ) return rc;
        if (st[2] <= run_cap) break;
        run_cap = st[2] + (st[2] >> 3) + 16;            // more reference runs than the table holds (assemblies with 10^5..10^6 contigs): once more, with room
        BGP_HIP(hipMemsetAsync(c->d_status.p, 0, 64, c->stream));
    }
    const uint32_t n_runs = st[2];
    std::vector<RunDev> runs(n_runs);
    const bool finished_on_device = n_runs <= RUNS_ON_DEVICE;
    if (finished_on_device) {                                  // k3_finish left the table sorted and complete
        if (n_runs <= FIRST_RUNS) memcpy(runs.data(), first_runs, sizeof(RunDev) * n_runs);
        else BGP_HIP(hipMemcpy(runs.data(), c->d_runs.p, sizeof(RunDev) * n_runs, hipMemcpyDeviceToHost));
    } else {                                                   // (a table of more than RUNS_ON_DEVICE runs: by the host)
        BGP_HIP(hipMemcpy(runs.data(), c->d_runs.p, sizeof(RunDev) * n_runs, hipMemcpyDeviceToHost));
        std::sort(runs.begin(), runs.end(), [](const RunDev& x, const RunDev& y) { return x.first_record < y.first_record; });
        for (uint32_t j = 0; j < n_runs; ++j) runs[j].end_record = j + 1 < n_runs ? runs[j + 1].first_record : n;
        BGP_HIP(hipMemcpyAsync(c->d_runs.p, runs.data(), sizeof(RunDev) * n_runs, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k3_run_ends, dim3((n_runs + 63) / 64), dim3(64), 0, c->stream, c->d_runs.as<RunDev>(), n_runs, c->d_pmax_end.as<uint64_t>(), n);
        BGP_HIP(hipMemcpyAsync(runs.data(), c->d_runs.p, sizeof(RunDev) * n_runs, hipMemcpyDeviceToHost, c->stream));
        BGP_HIP(hipStreamSynchronize(c->stream));
    }
    // A stream of long reads (a quarter of its reads or more) is left to the general kernel entirely, on its own tiling of
    // 4096 positions; every other stream is tiled for the accumulate path, the general kernel taking the windows that leaves.
#ifdef BGP_EXPERIMENTS
    static const bool long_rows = !(getenv("BGP_LONG_ROWS") && atoi(getenv("BGP_LONG_ROWS")) == 0);
#else
    constexpr bool long_rows = true;
#endif
        // long reads through the accumulate path as rows in reference coordinates (0: the general kernel, as before)
    if (c->rows_ok) st[3] = 0;      // (behind plan rows: framing has vouched that no read has more than BGP_PLAN_MAX_OPS ops or 255 bases -- k4_plan_rest plans them all, nothing wants checkpoints)
    const bool long_stream = !long_rows && (uint64_t)st[3] * 4 >= n && st[3] > 0;
    const uint32_t tw = long_stream ? (uint32_t)WO : (uint32_t)W;
    uint64_t col_base = 0, tile_base = 0;
    for (auto& r : runs) {
        r.column_base = col_base; r.tile_base = tile_base;
        const uint64_t span = r.end_pos - r.start_pos;
        col_base += span; tile_base += (span + tw - 1) / tw;
    }
    if (tile_base >= (1ull << 31)) return BGP_EINVAL;
    c->runs.resize(n_runs);
    for (uint32_t j = 0; j < n_runs; ++j) {
        bgp_run& o = c->runs[j]; const RunDev& r = runs[j];
        o.ref_id = r.ref_id; o._pad = 0; o.first_record = r.first_record; o.end_record = r.end_record;
        o.start_pos = r.start_pos; o.end_pos = r.end_pos; o.column_base = r.column_base;
    }
    {   // columns are dense over every reference's [first position, largest end_position): 32 bytes per position, covered or not
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (col_base + 1) * sizeof(bgp_column) > c->d_columns.cap + free_b) {
            uint64_t widest = 0; int32_t wref = 0;
            for (const auto& r : runs) if (r.end_pos - r.start_pos > widest) { widest = r.end_pos - r.start_pos; wref = r.ref_id; }
            char msg[256];
            snprintf(msg, sizeof msg, "dense columns need %.1f GB for %llu reference positions (widest: reference %d spans %llu positions) and %.1f GB are free: "
                     "pile up region by region (bgp_load_bam_region / bgp_filter_region) or in batches (bgp_batch_*)",
                     (double)(col_base * 32) / 1e9, (unsigned long long)col_base, wref, (unsigned long long)widest, (double)free_b / 1e9);
            c->last_error = msg;
            return BGP_ENOMEM;
        }
    }
    BGP_TRY(c->d_columns.reserve((col_base + 1) * sizeof(bgp_column)));
    if (n_runs) {
        if (!finished_on_device || tw != (uint32_t)W) BGP_HIP(hipMemcpyAsync(c->d_runs.p, runs.data(), sizeof(RunDev) * n_runs, hipMemcpyHostToDevice, c->stream));      // (k3_finish wrote the same bases)
        K4Args a;
        a.s = c->d_stream; a.rec = c->d_rec.as<bgp_record>(); a.pmax_end = c->d_pmax_end.as<uint64_t>(); a.pmax_pos = c->d_pmax_pos.as<uint64_t>();
        a.runs = c->d_runs.as<RunDev>(); a.n_runs = n_runs; a.n_tiles = tile_base; a.stream_len = c->stream_len;
        a.cols = c->d_columns.as<bgp_column>(); a.hist = (unsigned long long*)c->d_depth_hist.p; a.pstats = (unsigned long long*)c->d_pstats.p; a.status = status;
        const size_t lds16 = (size_t)(long_stream ? L16o::WORDS : L16::WORDS) * 4, lds32 = (size_t)(long_stream ? L32o::WORDS : L32::WORDS) * 4;
        uint64_t used = 0;   // algorithmic bytes: record table + CIGAR + packed SEQ per read, 32 B per position
        used = c->n_records * 32 + c->sum_ncig * 4 + (c->sum_lseq + 1) / 2 + col_base * 32;
        BGP_TRY(c->d_tiles.reserve(tile_base * sizeof(TileB)));
        bool lite_bounds = false, use_rows = false;
        BGP_TRY(c->d_deep.reserve((tile_base + 2) * 4));                 // [0] count, [1..] window indices
        BGP_HIP(hipMemsetAsync(c->d_deep.p, 0, 4, c->stream));
        a.n_deep = c->d_deep.as<uint32_t>(); a.deep = c->d_deep.as<uint32_t>() + 1;
        a.tiles = c->d_tiles.as<TileB>();
        {
            TimedRegion tb(c, "k4_bounds", tile_base * sizeof(TileB));
            // (a stream with long reads: k4_acc2 walks them from the window's record range; a stream left to the general kernel: every window is its)
            lite_bounds = !long_stream && !st[3];
            // k12_lds left plan rows: the accumulate path takes a window's ordinary reads from them (a sorted stream of short reads; anything else is planned read by read)
            static const bool rows_env = !(getenv("BGP_K4_ROWS") && atoi(getenv("BGP_K4_ROWS")) == 0);
            use_rows = rows_env && c->rows_ok && lite_bounds && rc == BGP_OK && c->k3_bsum_n == n;      // (the scanned tile maxima the row ranges are searched through must be this table's)
            hipLaunchKernelGGL(k4_bounds, dim3((unsigned)((tile_base + 255) / 256)), dim3(256), 0, c->stream, a.pmax_end, a.pmax_pos, a.runs, n_runs, tile_base, c->d_tiles.as<TileB>(), tw, use_rows ? 3 : lite_bounds ? 1 : 0, (const uint32_t*)nullptr, (const uint32_t*)nullptr, c->d_k3_bsum.as<K3Key>(), n);      // (K3's tile maxima, scanned: k3_bsum's own buffer, made for these n records)
        }
        constexpr uint64_t WG_SLOTS = 65536;            // k4_acc's workgroups leave their ranges' totals behind the windows' slots
        BGP_TRY(c->d_col_lo.reserve((tile_base + WG_SLOTS) * 32));
        a.tile_stats = (unsigned long long*)c->d_col_lo.p;
        a.ck = nullptr; a.ck_off = nullptr;
        bool have_rows = false;                        // long reads take the accumulate path (k4_acc2 walks their ops from the windows' checkpoints)
        if (st[3]) {                                   // the stream holds long reads: build their per-window CIGAR checkpoints (the planner reads their verdicts)
            bool rows = false;
            {
            TimedRegion tc(c, "k4_checkpoints", c->sum_ncig * 4);
            BGP_TRY(c->d_ck_off.reserve((n + 2) * 8));
            uint64_t* ck_off = c->d_ck_off.as<uint64_t>();
            rows = long_rows && !long_stream;               // (the accumulate path takes the long reads: k4_acc2 walks their ops from these checkpoints)
            hipLaunchKernelGGL(k4_ck_count, dim3(nb), dim3(256), 0, c->stream, a.rec, n, a.runs, n_runs, ck_off, tw);
            uint64_t total_ck = 0;
            BGP_TRY(scan_exclusive_u64(c, ck_off, ck_off, n, &total_ck));
            BGP_TRY(c->d_ck.reserve((total_ck + 1) * sizeof(Ckpt)));
            BGP_HIP(hipMemsetAsync(c->d_ck.p, 0xFF, (total_ck + 1) * sizeof(Ckpt), c->stream));
            if ((uint64_t)st[3] * 8 >= n)             // mostly long reads: a wave each
                hipLaunchKernelGGL(k4_ck_build<true>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, c->stream, a.s, a.stream_len, a.rec, n, a.runs, n_runs, ck_off, c->d_ck.as<Ckpt>(), tw);
            else
                hipLaunchKernelGGL(k4_ck_build<false>, dim3(nb), dim3(256), 0, c->stream, a.s, a.stream_len, a.rec, n, a.runs, n_runs, ck_off, c->d_ck.as<Ckpt>(), tw);
            a.ck = c->d_ck.as<Ckpt>(); a.ck_off = ck_off;
            }
            have_rows = rows;
        }
        if (long_stream) {
            hipLaunchKernelGGL(k4_all_listed, dim3((unsigned)((tile_base + 255) / 256)), dim3(256), 0, c->stream, c->d_deep.as<uint32_t>() + 1, c->d_deep.as<uint32_t>(), (uint32_t)tile_base);
        } else {
        // the plan: descriptor slabs per window, from one pass over the reads
        const uint32_t slab_cap = SLAB_CAP;
        if (have_rows) { BGP_TRY(c->d_wlong.reserve((tile_base + 1) * 4)); BGP_HIP(hipMemsetAsync(c->d_wlong.p, 0, (tile_base + 1) * 4, c->stream)); }
        BGP_TRY(c->d_slab.reserve(tile_base * (size_t)slab_cap * 8));
        BGP_TRY(c->d_wcount.reserve((tile_base + 1) * 4)); BGP_TRY(c->d_wdel.reserve((tile_base + 2) * 4)); BGP_TRY(c->d_wslow.reserve((tile_base + 2) * 4));
        BGP_TRY(c->d_col_lo.reserve((tile_base + 65536) * 32));          // (k4_acc's totals: zeroed here with the planner's arrays, one launch)
        { const FillItem f[4] = {{c->d_wcount.p, (tile_base + 1) * 4, 0u}, {c->d_wdel.p, (tile_base + 2) * 4, 0u}, {c->d_wslow.p, (tile_base + 2) * 4, 0u}, {c->d_col_lo.p, (tile_base + 65536) * 32, 0u}}; BGP_TRY(fill_many(c, f, 4)); }
        if (!use_rows) BGP_TRY(ensure_cig4(c));      // (the planner over every read wants the first ops beside the rows it reads, not 290 bytes apart in the stream)
        {
            TimedRegion tp(c, use_rows ? "k4_plan_rest" : "k4_plan", use_rows ? c->n_records : c->n_records * 48);      // (what it reads of every record: the class byte | the row and the first ops)
            PlanArgs pa;
            pa.rec = a.rec; pa.cig4 = c->cig4_ok ? c->d_cig4.as<uint4>() : nullptr; pa.s = a.s; pa.stream_len = a.stream_len; pa.n = n;
            pa.runs = a.runs; pa.n_runs = n_runs; pa.n_tiles = tile_base;
            pa.slab = (unsigned long long*)c->d_slab.p; pa.slab_cap = slab_cap; pa.wcount = c->d_wcount.as<uint32_t>(); pa.wdel = c->d_wdel.as<int32_t>(); pa.wslow = c->d_wslow.as<int32_t>();
            pa.ck = have_rows ? a.ck : nullptr; pa.ck_off = a.ck_off; pa.wlong = have_rows ? c->d_wlong.as<uint32_t>() : nullptr;
            pa.batches = n >= (4ull << 20) ? PLAN_NB : 1;
            pa.rcls = use_rows ? c->d_rcls.as<uint8_t>() : nullptr;
            if (use_rows) hipLaunchKernelGGL(k4_plan_rest, dim3((unsigned)((n + 4ull * PLAN_SPAN - 1) / (4ull * PLAN_SPAN))), dim3(256), 0, c->stream, pa);
            else hipLaunchKernelGGL(k4_plan, dim3((unsigned)((n + 256ull * pa.batches - 1) / (256ull * pa.batches))), dim3(256), 0, c->stream, pa);
            const unsigned wb = (unsigned)((tile_base + 1023) / 1024);
            BGP_TRY(c->d_scan_tmp.reserve((size_t)wb * 8 + 16));
            int32_t* btot = c->d_scan_tmp.as<int32_t>();
            hipLaunchKernelGGL(k4_wscan_tot, dim3(wb), dim3(1024), 0, c->stream, pa.wdel, pa.wslow, tile_base, btot);
            hipLaunchKernelGGL(k4_wscan_mid, dim3(1), dim3(1024), 0, c->stream, btot, wb);
            hipLaunchKernelGGL(k4_wscan, dim3(wb), dim3(1024), 0, c->stream, pa.wdel, pa.wslow, pa.wcount, c->d_tiles.as<TileB>(), tile_base, btot, c->d_deep.as<uint32_t>() + 1, c->d_deep.as<uint32_t>(), slab_cap, use_rows ? ROW_CAP : 0u);
        }
        }
        c->ck_w = a.ck ? tw : 0;                     // K5 resumes long reads from these checkpoints
        // (timers named after the kernels they bracket: K4's algorithmic bytes are booked on the accumulate launch -- on the general kernel's
        // where that takes every window)
        if (long_stream) BGP_HIP(hipMemsetAsync(a.tile_stats, 0, (tile_base + WG_SLOTS) * 32, c->stream));      // the accumulate path adds its totals wave by wave (zeroed with the planner's arrays above)
        if (!long_stream) {
            TimedRegion t(c, "k4_acc2", used);
            GArgs g;
            g.s = a.s; g.slab = (const unsigned long long*)c->d_slab.p; g.slab_cap = SLAB_CAP; g.wcount = c->d_wcount.as<uint32_t>(); g.wdel = c->d_wdel.as<int32_t>(); g.tiles = a.tiles; g.n_tiles = tile_base;
            g.wlong = have_rows ? c->d_wlong.as<uint32_t>() : nullptr; g.rec = a.rec; g.ck_off = a.ck_off; g.ck = a.ck;
            g.cols = a.cols; g.hist = a.hist; g.tile_stats = a.tile_stats; g.deep = c->d_deep.as<uint32_t>() + 1; g.n_deep = c->d_deep.as<uint32_t>();
            // 1280 workgroups are resident (five of 30 KB a CU); handing the device a dozen times as many, each with a shorter range of
            // windows, lets it balance the ranges' unequal work at the tail (cfg3: 9.1 ms with 4096, 8.55 with 16384, the same with 65536)
            static const unsigned g_env = getenv("BGP_G_BLOCKS") ? (unsigned)atoi(getenv("BGP_G_BLOCKS")) : 0u;
            const unsigned g_blocks = std::min(65536u, g_env ? g_env : 16384u);
            g.prow = use_rows ? c->d_prow.as<uint4>() : nullptr; g.stream_len = a.stream_len;
            static const size_t lds_pad = getenv("BGP_ACC2_LDS_PAD") ? (size_t)atoi(getenv("BGP_ACC2_LDS_PAD")) : 0;      // (occupancy experiments: LDS the kernel never touches, so that fewer workgroups fit a CU)
            if (have_rows) hipLaunchKernelGGL((k4_acc2<true, false>), dim3((unsigned)std::min<uint64_t>(tile_base, g_blocks)), dim3(AT), (size_t)A2Lay::BYTES + lds_pad, c->stream, g);
            else if (use_rows) hipLaunchKernelGGL((k4_acc2<false, true>), dim3((unsigned)std::min<uint64_t>(tile_base, g_blocks)), dim3(AT), (size_t)A2Lay::BYTES + lds_pad, c->stream, g);
            else hipLaunchKernelGGL((k4_acc2<false, false>), dim3((unsigned)std::min<uint64_t>(tile_base, g_blocks)), dim3(AT), (size_t)A2Lay::BYTES + lds_pad, c->stream, g);
        }
        // windows the accumulate path left on the list (odd CIGARs, too deep): the general kernel
        {
        TimedRegion t(c, "k4_tile", long_stream ? used : 0);
        const unsigned dgrid = (unsigned)std::min<uint64_t>(tile_base, 2048);
        if (lite_bounds) hipLaunchKernelGGL(k4_bounds, dim3((unsigned)((tile_base + 255) / 256)), dim3(256), 0, c->stream, a.pmax_end, a.pmax_pos, a.runs, n_runs, tile_base, c->d_tiles.as<TileB>(), tw, 2, (const uint32_t*)(c->d_deep.as<uint32_t>() + 1), (const uint32_t*)c->d_deep.as<uint32_t>());
        if (long_stream) {
            hipLaunchKernelGGL(k4_tile<L16o>, dim3(dgrid), dim3(L16o::T), lds16, c->stream, a, 1);
            hipLaunchKernelGGL(k4_tile<L32o>, dim3(std::min(dgrid, 512u)), dim3(L32o::T), lds32, c->stream, a, 1);
        } else {
            hipLaunchKernelGGL(k4_tile<L16>, dim3(dgrid), dim3(L16::T), lds16, c->stream, a, 1);
            hipLaunchKernelGGL(k4_tile<L32>, dim3(std::min(dgrid, 512u)), dim3(L32::T), lds32, c->stream, a, 1);
        }
        }
        { TimedRegion t(c, "k4_stats", 0);
        hipLaunchKernelGGL(k4_stats, dim3(32), dim3(256), 0, c->stream, a.tile_stats, tile_base + WG_SLOTS, a.pstats); }
        BGP_HIP(hipGetLastError());
    }
    if (getenv("BGP_VERBOSE") && n_runs) {
        uint32_t nd = 0;
        BGP_HIP(hipMemcpyAsync(&nd, c->d_deep.p, 4, hipMemcpyDeviceToHost, c->stream));
        BGP_HIP(hipStreamSynchronize(c->stream));
        fprintf(stderr, "[biodgpu] K4: %u of %llu windows left to the general kernel\n", nd, (unsigned long long)tile_base);
    }
    uint64_t ps[16];
    { const MailItem it[2] = {{c->d_pstats.p, ps, 128}, {status, st, 16}}; BGP_TRY(mail_fetch(c, it, 2)); }
    rc = bgp_status_from_bits(st[0]);
    if (rc < 0 && rc != BGP_ESEQ) return rc;
    bgp_pileup_info& pi = c->pinfo;
    pi.n_runs = n_runs; pi.runs = c->runs.data(); pi.n_positions = col_base;
    pi.n_columns = ps[0]; pi.sum_coverage = ps[1]; pi.max_coverage = (uint32_t)ps[2]; pi.n_reads_used = ps[4];
    pi.soft_status = rc == BGP_ESEQ ? BGP_ESEQ : BGP_OK;
    pi.device_columns = c->d_columns.as<bgp_column>(); pi.device_depth_hist = c->d_depth_hist.as<uint64_t>();
    return BGP_OK;
}

namespace {
__global__ void k_first_survivor(const uint64_t* pmax_end, const bgp_record* rec, uint64_t first, uint64_t end, uint64_t key, uint64_t* out) {
    // first record of the run whose running max end_position reaches `key`: it is the read that raised the max
    uint64_t lo = first, hi = end;
    while (lo < hi) { const uint64_t m = (lo + hi) >> 1; if (pmax_end[m] >= key) hi = m; else lo = m + 1; }
    out[0] = lo < end ? (uint64_t)(uint32_t)rec[lo].pos : ~0ull;
    out[1] = lo < end ? lo : ~0ull;
}
}  // namespace

namespace {
__global__ void k_batch_cut(const uint64_t* pmax_end, const uint64_t* pmax_pos, const bgp_record* rec, uint64_t n, uint64_t first, uint64_t end, uint64_t* out) {
    const uint64_t key = pmax_pos[n - 1];                        // (reference key << 32 | position) of the last covering read; 0: none
    uint64_t lo = first, hi = end;
    if (key) { while (lo < hi) { const uint64_t m = (lo + hi) >> 1; if (pmax_end[m] > key) hi = m; else lo = m + 1; } }
    out[0] = key; out[1] = lo; out[2] = key && lo < end ? rec[lo].rec_off : 0;
}
}  // namespace

// batched staging: the position of the stream's last covering read and the first record of its run whose running maximum
// of end_position exceeds it -- the exact halo of that position: every column before it is final, every read that
// matters from there on comes at or after that record
int batch_cut(bgp_ctx* c, uint64_t* last_key, uint64_t* carry_record, uint64_t* carry_off) {
    *last_key = 0; *carry_record = 0; *carry_off = 0;
    if (c->runs.empty() || !c->n_records) return BGP_OK;
    const bgp_run& r = c->runs.back();
    uint64_t* out = c->d_pstats.as<uint64_t>() + 8;
    hipLaunchKernelGGL(k_batch_cut, dim3(1), dim3(1), 0, c->stream, c->d_pmax_end.as<uint64_t>(), c->d_pmax_pos.as<uint64_t>(), c->d_rec.as<bgp_record>(), c->n_records, r.first_record, r.end_record, out);
    uint64_t h[3] = {0, 0, 0};
    BGP_HIP(hipMemcpyAsync(h, out, 24, hipMemcpyDeviceToHost, c->stream));
    BGP_HIP(hipStreamSynchronize(c->stream));
    *last_key = h[0]; *carry_record = h[1]; *carry_off = h[2];
    return BGP_OK;
}

// position of the first read of run `run_idx` with position + basesCovered >= start_from (pileup.d:482-489)
int k34_first_survivor(bgp_ctx* c, uint32_t run_idx, uint64_t start_from, uint64_t* pos, uint64_t* record) {
    const bgp_run& r = c->runs[run_idx];
    *record = ~0ull;
    if (start_from > 0x7FFFFFFFull) { *pos = ~0ull; return BGP_OK; }
    const uint64_t key = (((uint64_t)(r.ref_id < 0 ? 0x7FFFFFFFu : (uint32_t)(r.ref_id + 1))) << 32) | (uint32_t)start_from;
    uint64_t* out = c->d_pstats.as<uint64_t>() + 6;
    hipLaunchKernelGGL(k_first_survivor, dim3(1), dim3(1), 0, c->stream, c->d_pmax_end.as<uint64_t>(), c->d_rec.as<bgp_record>(), r.first_record, r.end_record, key, out);
    uint64_t h[2] = {~0ull, ~0ull};
    BGP_HIP(hipMemcpyAsync(h, out, 16, hipMemcpyDeviceToHost, c->stream));
    BGP_HIP(hipStreamSynchronize(c->stream));
    *pos = h[0]; *record = h[1];
    return BGP_OK;
}

int k5_entries(bgp_ctx* c, uint32_t run_idx, uint64_t pos_begin, uint64_t pos_end, uint64_t* n_entries) {
    const bgp_run& r = c->runs[run_idx];
    const uint64_t np = pos_end - pos_begin;
    *n_entries = 0;
    if (np == 0) return BGP_OK;
    BGP_TRY(c->d_entry_off.reserve((np + 1) * 8)); BGP_TRY(c->d_col_lo.reserve((np + 1) * 8));
    K5Args a;
    a.s = c->d_stream; a.rec = c->d_rec.as<bgp_record>(); a.pmax_end = c->d_pmax_end.as<uint64_t>(); a.pmax_pos = c->d_pmax_pos.as<uint64_t>();
    a.run.ref_id = r.ref_id; a.run.first_record = r.first_record; a.run.end_record = r.end_record; a.run.start_pos = r.start_pos; a.run.end_pos = r.end_pos;
    a.run.column_base = r.column_base; a.run.tile_base = 0; a.run.pad = 0;
    a.pos_begin = pos_begin; a.pos_end = pos_end; a.cols = c->d_columns.as<bgp_column>(); a.cov64 = c->d_col_lo.as<uint64_t>();
    a.col_off = c->d_entry_off.as<uint64_t>(); a.entries = nullptr;
    a.ck_w = c->ck_w; a.ck = c->ck_w ? c->d_ck.as<Ckpt>() : nullptr; a.ck_off = c->ck_w ? c->d_ck_off.as<uint64_t>() : nullptr;
    BGP_TRY(ensure_cig4(c));                             // (behind k12_lds's plan rows the side array is made now, from the CIGAR arena)
    a.cig4 = c->cig4_ok ? c->d_cig4.as<uint4>() : nullptr;
    a.stream_len = c->stream_len;
    uint64_t total = 0;
    {
        TimedRegion t(c, "k5_cov", np * 40);           // a 32-byte column read and an 8-byte offset written per position, the scan on top
        hipLaunchKernelGGL(k5_cov, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, c->stream, a);
        BGP_TRY(scan_exclusive_u64(c, a.cov64, c->d_entry_off.as<uint64_t>(), np, &total));
    }
    BGP_TRY(c->d_entries.reserve((total + 1) * sizeof(bgp_entry)));
    a.entries = c->d_entries.as<bgp_entry>();
    {
        TimedRegion t(c, "k5_fill", total * 16);
#ifdef BGP_EXPERIMENTS
        static const bool per_column = getenv("BGP_K5_PER_COLUMN") && atoi(getenv("BGP_K5_PER_COLUMN")) != 0;      // (the first version: a wave per column)
        if (per_column) hipLaunchKernelGGL(k5_fill, dim3((unsigned)((np + 3) / 4)), dim3(256), 0, c->stream, a);
        else
#endif
        if (a.ck) hipLaunchKernelGGL(k5_fill_blk<true>, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, c->stream, a);
        else hipLaunchKernelGGL(k5_fill_blk<false>, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, c->stream, a);
    }
    BGP_HIP(hipGetLastError());
    *n_entries = total;
    return BGP_OK;
}
