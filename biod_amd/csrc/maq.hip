// maq.hip -- kernel K8: the MAQ consensus model per pileup column (second half).
//
// Reference semantics: bio/std/hts/snpcallers/maq.d (itself after samtools' errmod.c):
//   ErrorModelCoefficients(depcorr, eta)         :73-118   tables fk[n], beta[q][n][k], lhet[n][k]
//   computeLikelihoods(read_bases)               :133-240  bases sorted by quality, walked from the best one down;
//                                                          fsum / bsum per base, then ten genotype scores (float)
//   GenotypeLikelihoodInfo(dict)                 :245-279  genotypes in rising score order (insertion sort, stable in
//                                                          genotype-code order)
//   MaqSnpCaller.genotypeLikelihoodInfo/makeCall :388-486  which reads of a column count (base quality, not '-'),
//                                                          quality = min(base quality, mapping quality), the call
// findSNPs (:488-541) is then "is_variant && quality > minimum_call_quality", a comparison the facades make.
//
// Three places where the reference's answer is not a function of its input, and what is done instead (stated in
// include/biodgpu.h; the oracle, oracle/maq_oracle.py, does the same):
//   * bases are sorted with std.algorithm.sort, which is unstable, and the sums depend on the order among equal
//     qualities when both strands of a base are present (:148-168): here the sort is STABLE (file order);
//   * a column with more than 255 valid bases is subsampled with randomSample (:141-143): here the first 255 in file
//     order are taken, and the call carries a flag;
//   * the tables are built in 80-bit `real` (:106-118) -- done on the host in long double, the same libm -- and the
//     heterozygous scores subtract in `real` before rounding to float (:203-207) -- done in double here: one float ulp
//     of difference is possible where a score sits on a rounding boundary.
// The reference holds no test, fixture or expected output for this module: parity is oracle-exact, reference-unpinned.
//
// MI355X shape: one wave per column.  Lanes pick the column's valid entries (ballot compaction, file order kept), rank
// them by (quality, arrival) all at once -- every lane counts how many elements come before its own, the elements
// broadcast from LDS -- and one lane then walks the sorted bases: the sums are sequential by definition (the k-th
// occurrence of a base weighs fk(k), and floating-point addition is kept in the reference's order).
#include "common.h"

#include <cmath>

namespace {

constexpr int MAQ_T = 256;                    // four columns per workgroup
constexpr int MAQ_MAXB = 255;

struct MaqArgs {
    const bgp_entry* entries; const uint64_t* entry_off; const bgp_record* rec; const uint8_t* refbase; uint64_t np;
    const double* fk; const double* beta; const double* lhet; uint32_t min_bq;
    bgp_maq_call* out;
};

__device__ __forceinline__ uint32_t base16_code(uint32_t ch) {          // Base(char), bio/core/base.d:41-63: "=ACMGRSVTWYHKDBN", digits 0-3 = ACGT, anything else N
    switch (ch | 0x20u) {                                                // (letters: either case)
        case 'a': return 1; case 'c': return 2; case 'g': return 4; case 't': return 8;
        case 'm': return 3; case 'r': return 5; case 's': return 6; case 'v': return 7; case 'w': return 9; case 'y': return 10;
        case 'h': return 11; case 'k': return 12; case 'd': return 13; case 'b': return 14;
        default: break;
    }
    if (ch == '=') return 0;
    if (ch >= '0' && ch <= '3') return 1u << (ch - '0');
    return 15;
}
__device__ __forceinline__ uint32_t base5_code(uint32_t ch) {           // Base5(char), base.d:188-213
    switch (ch | 0x20u) { case 'a': return 0; case 'c': return 1; case 'g': return 2; case 't': return 3; default: return 4; }
}

struct MaqCoopLds { uint8_t q[256], c[256], sq[256], sc[256]; double fs[16], bs[16]; uint32_t w[32], cn[16]; };

// One column by a whole wave (any depth): lanes pick and rank, lane 0 walks.  The first version of K8 ran every column
// this way; k8_maq64 below keeps it for columns with more than 64 valid bases.
__device__ void maq_column_coop(const MaqArgs& a, uint64_t p, int lane, MaqCoopLds& L) {
    uint8_t* const s_q_ = L.q; uint8_t* const s_c_ = L.c; uint8_t* const s_sq_ = L.sq; uint8_t* const s_sc_ = L.sc;
    double* const s_fs_ = L.fs; double* const s_bs_ = L.bs; uint32_t* const s_w_ = L.w; uint32_t* const s_cn_ = L.cn;
    const uint64_t e0 = a.entry_off[p], e1 = a.entry_off[p + 1];
    // ---- the column's valid bases, in file order (maq.d:400-412): base quality >= minimum, not '-'
    uint32_t n_valid = 0;
    for (uint64_t base = e0; base < e1; base += 64) {
        const uint64_t i = base + lane;
        bool valid = false; uint32_t q = 0, code = 0;
        if (i < e1) {
            const bgp_entry en = a.entries[i];
            valid = en.qual >= a.min_bq && en.base != '-';
            if (valid) {
                const bgp_record r = a.rec[en.read_ordinal];
                q = min((uint32_t)en.qual, (r.bin_mq_nl >> 8) & 0xFFu);                              // min(current_base_quality, mapping_quality), :406
                code = base16_code(en.base) * 2u + ((r.flag_nc >> 20) & 1u);                         // BaseWithStrand(base, is_reverse_strand), :32-34 (flag 0x10)
            }
        }
        const unsigned long long m = __ballot(valid);
        const uint32_t idx = n_valid + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (valid && idx < (uint32_t)MAQ_MAXB) { s_q_[idx] = (uint8_t)q; s_c_[idx] = (uint8_t)code; }
        n_valid += (uint32_t)__popcll(m);
    }
    const uint32_t ref5 = a.refbase ? base5_code(a.refbase[p]) : 4u;                                  // Base5(column.reference_base), :455-459 ('N' without reference bases)
    if (n_valid == 0) {                                                                               // no genotype likelihoods: makeCall returns null, :436-439,451
        if (lane == 0) { bgp_maq_call o; o.quality = 0.f; o.genotype = 0xFF; o.reference_base5 = (uint8_t)ref5; o.flags = 0; o.n_bases = 0; a.out[p] = o; }
        return;
    }
    const uint32_t n = min(n_valid, (uint32_t)MAQ_MAXB);
    // ---- stable rising sort by quality (:146): every element's rank counted directly
    for (uint32_t i = lane; i < n; i += 64) {
        const uint32_t qi = s_q_[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; ++j) { const uint32_t qj = s_q_[j]; rank += (qj < qi || (qj == qi && j < i)) ? 1u : 0u; }
        s_sq_[rank] = (uint8_t)qi; s_sc_[rank] = s_c_[i];
    }
    if (lane < 32) s_w_[lane] = 0;
    if (lane < 16) { s_cn_[lane] = 0; s_fs_[lane] = 0.0; s_bs_[lane] = 0.0; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane != 0) return;      // (the callers re-converge after the call)
    // ---- from the best base down (:153-166)
    for (int i = (int)n - 1; i >= 0; --i) {
        uint32_t q = s_sq_[i];
        q = q < 4u ? 4u : q > 63u ? 63u : q;
        const uint32_t bws = s_sc_[i], b = bws >> 1;
        const double f = a.fk[s_w_[bws]];
        s_fs_[b] += f;
        s_bs_[b] += f * a.beta[(q << 16) | (n << 8) | s_cn_[b]];
        s_cn_[b] += 1; s_w_[bws] += 1;
    }
    // ---- the ten genotype scores (:170-228); Base codes of A, C, G, T are 1, 2, 4, 8; Base5 codes 0..3
    const double Cc = 10.0 / 2.30258509299404568401799145468436421;                                   // 10 / LN10
    float sc[4][4];                                                                                   // sc[j][i], j >= i: genotype code j * 5 + i
    for (int i = 0; i < 4; ++i) {
        float tmp1 = 0.f; int tmp2 = 0;
        for (int k = 0; k < 4; ++k) if (k != i) { tmp1 = (float)((double)tmp1 + s_bs_[1 << k]); tmp2 += (int)s_cn_[1 << k]; }
        sc[i][i] = tmp2 > 0 ? tmp1 : 0.f;
        for (int j = i + 1; j < 4; ++j) {
            const uint32_t cij = s_cn_[1 << i] + s_cn_[1 << j];
            tmp1 = 0.f; tmp2 = 0;
            for (int k = 0; k < 4; ++k) if (k != i && k != j) { tmp1 = (float)((double)tmp1 + s_bs_[1 << k]); tmp2 += (int)s_cn_[1 << k]; }
            const double lh = Cc * a.lhet[(cij << 8) | s_cn_[1 << j]];
            sc[j][i] = tmp2 > 0 ? (float)((double)tmp1 - lh) : (float)(-lh);
        }
    }
    for (int j = 0; j < 4; ++j) for (int i = 0; i <= j; ++i) if (sc[j][i] < 0.f) sc[j][i] = 0.f;      // :221-226 (every stored genotype is visited once its first allele's turn comes)
    // ---- best and second best in rising score order, ties in genotype-code order (:251-268)
    float s0 = 0.f, s1 = 0.f; int g0 = -1, g1 = -1;
    for (int j = 0; j < 4; ++j) for (int i = 0; i <= j; ++i) {                                        // rising code j * 5 + i
        const float v = sc[j][i]; const int code = j * 5 + i;
        if (g0 < 0 || v < s0) { s1 = s0; g1 = g0; s0 = v; g0 = code; }
        else if (g1 < 0 || v < s1) { s1 = v; g1 = code; }
    }
    bgp_maq_call o;
    o.quality = s1 - s0;                                                                              // gts[1].score - gts[0].score, :470
    o.genotype = (uint8_t)g0;
    o.reference_base5 = (uint8_t)ref5;
    o.flags = (uint8_t)(BGP_MAQ_CALLED | (g0 != (int)(ref5 * 5 + ref5) ? BGP_MAQ_VARIANT : 0) | (n_valid > (uint32_t)MAQ_MAXB ? BGP_MAQ_SUBSAMPLED : 0));   // is_variant: call.d:85-87
    o.n_bases = (uint8_t)n;
    a.out[p] = o;
}

// K8 as it runs: one wave per 64 consecutive columns.  Phase A takes the block's columns one after the other with
// lane = entry: the valid bases are picked (one coalesced pass: a column with more than 64 entries is left to
// maq_column_coop), ranked by (quality, arrival) with one ballot per bit of the quality -- a lane keeps the set of lanes
// whose key agrees with its own so far and counts those that fell below -- and stored in rank order.  Phase B is
// lane = COLUMN: every lane walks its own column's bases from the best one down, the per-base state (four sums, four
// counts, eight strand counters: only A, C, G, T enter a genotype score; the other codes' counters never meet theirs) in
// registers.  The sums are the same operations in the same order as before; what changed is that 64 columns share the
// serial walk -- with a wave per column 63 lanes waited for one, a table load at a time (93 ms for 5e7 columns).
__global__ __launch_bounds__(MAQ_T, 4) void k8_maq64(MaqArgs a) {      // (four workgroups a CU: 128 registers, under 40 KB of LDS)
    __shared__ uint16_t s_srt[4][64][65];                 // [rank][column] (+1: the columns of a rank and the ranks of a column both fall into different banks); quality | code << 8
    __shared__ MaqCoopLds s_coop[4];
    __shared__ uint16_t s_tmp[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t col0 = ((uint64_t)blockIdx.x * 4 + wv) * 64;
    if (col0 >= a.np) return;
    uint16_t (*srt)[65] = s_srt[wv];
    uint32_t n = 0;                                        // valid bases of this lane's column
    bool deep = false;
    const int n_cols = (int)min<uint64_t>(64, a.np - col0);
    // the block's 65 offsets at once (lane l holds column l's)
    const uint64_t offl = a.entry_off[col0 + (uint64_t)min(lane, n_cols)];
    auto off_of = [&](int c) -> uint64_t {
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(offl >> 32), c) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)offl, c);
    };
    const uint64_t off_last = a.entry_off[col0 + (uint64_t)n_cols];
    for (int c = 0; c < n_cols; ++c) {
        const uint64_t e0 = off_of(c), e1 = c + 1 >= n_cols ? off_last : off_of(c + 1);
        if (e1 - e0 > 128) { if (lane == c) deep = true; continue; }
        // the entries that count (maq.d:400-412), compacted in file order: up to 128 entries a column ('-' entries of N skips
        // and low qualities are a good part of a column), of which up to 64 may count
        uint32_t nv = 0;
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && e1 - e0 <= 64) break;
            const uint64_t i = e0 + 64u * (uint32_t)h + lane;
            bool ok = false; uint32_t el = 0;
            if (i < e1) {
                const bgp_entry en = a.entries[i];
                ok = en.qual >= a.min_bq && en.base != '-';
                if (ok) {
                    uint2 rw; __builtin_memcpy(&rw, reinterpret_cast<const uint8_t*>(a.rec + en.read_ordinal) + 12, 8);      // bin_mq_nl, flag_nc
                    const uint32_t qq = min((uint32_t)en.qual, (rw.x >> 8) & 0xFFu);                     // min(current_base_quality, mapping_quality), :406
                    el = qq | ((base16_code(en.base) * 2u + ((rw.y >> 20) & 1u)) << 8);                // BaseWithStrand(base, is_reverse_strand), :32-34 (flag 0x10)
                }
            }
            const unsigned long long mk = __ballot(ok);
            const uint32_t idx = nv + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
            if (ok && idx < 64u) s_tmp[wv][idx] = (uint16_t)el;
            nv += (uint32_t)__popcll(mk);
        }
        if (nv > 64u) { if (lane == c) deep = true; continue; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const bool valid = (uint32_t)lane < nv;
        const uint32_t el = valid ? s_tmp[wv][lane] : 0u;
        const uint32_t q = el & 0xFFu, code = el >> 8;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const unsigned long long m = __ballot(valid);
        if (lane == c) n = (uint32_t)__popcll(m);
        if (!m) continue;
        // stable rising rank by quality (:146): most significant bit first
        unsigned long long same = m; uint32_t less = 0;
        const int top = __ballot(valid && q >= 64u) ? 7 : 5;
        for (int b = top; b >= 0; --b) {
            const unsigned long long B = __ballot(valid && ((q >> b) & 1u));
            if ((q >> b) & 1u) { less += (uint32_t)__popcll(same & ~B); same &= B; } else same &= ~B;
        }
        const uint32_t rank = less + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        if (valid) srt[rank][c] = (uint16_t)(q | (code << 8));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const uint64_t p = col0 + lane;
    const bool mine = lane < n_cols && !deep;
    // ---- from the best base down (:153-166), a column a lane
    double bs[4] = {0.0, 0.0, 0.0, 0.0}; uint32_t cn[4] = {0, 0, 0, 0}, w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t nmax = mine ? n : 0u;
    for (int d = 32; d >= 1; d >>= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, d, 64));
    for (uint32_t t = 0; t < nmax; ++t) {
        if (!mine || t >= n) continue;
        const uint32_t e = srt[n - 1 - t][lane];
        uint32_t q = e & 0xFFu;
        q = q < 4u ? 4u : q > 63u ? 63u : q;
        const uint32_t bws = e >> 8, b = bws >> 1, strand = bws & 1u;
        if (b != 1u && b != 2u && b != 4u && b != 8u) continue;                                        // (an ambiguity code: its sums and counters are its own and enter no score)
        const int bi = b == 1u ? 0 : b == 2u ? 1 : b == 4u ? 2 : 3;
        const int wi = bi * 2 + (int)strand;
        uint32_t wcur = 0, ccur = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) wcur = wi == k ? w[k] : wcur;
#pragma unroll
        for (int k = 0; k < 4; ++k) ccur = bi == k ? cn[k] : ccur;
        const double f = a.fk[wcur];
        const double add = f * a.beta[(q << 16) | (n << 8) | ccur];
#pragma unroll
        for (int k = 0; k < 4; ++k) { if (bi == k) { bs[k] += add; cn[k] += 1; } }
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] += wi == k ? 1u : 0u;
    }
    if (mine) {
        const uint32_t ref5 = a.refbase ? base5_code(a.refbase[p]) : 4u;                              // Base5(column.reference_base), :455-459 ('N' without reference bases)
        bgp_maq_call o;
        if (n == 0) { o.quality = 0.f; o.genotype = 0xFF; o.reference_base5 = (uint8_t)ref5; o.flags = 0; o.n_bases = 0; }   // no genotype likelihoods: makeCall returns null, :436-439,451
        else {
            // ---- the ten genotype scores (:170-228)
            const double Cc = 10.0 / 2.30258509299404568401799145468436421;                           // 10 / LN10
            float sc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float tmp1 = 0.f; int tmp2 = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) if (k != i) { tmp1 = (float)((double)tmp1 + bs[k]); tmp2 += (int)cn[k]; }
                sc[i][i] = tmp2 > 0 ? tmp1 : 0.f;
#pragma unroll
                for (int j = i + 1; j < 4; ++j) {
                    const uint32_t cij = cn[i] + cn[j];
                    tmp1 = 0.f; tmp2 = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (k != i && k != j) { tmp1 = (float)((double)tmp1 + bs[k]); tmp2 += (int)cn[k]; }
                    const double lh = Cc * a.lhet[(cij << 8) | cn[j]];
                    sc[j][i] = tmp2 > 0 ? (float)((double)tmp1 - lh) : (float)(-lh);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i <= j; ++i) if (sc[j][i] < 0.f) sc[j][i] = 0.f;                       // :221-226
            float s0 = 0.f, s1 = 0.f; int g0 = -1, g1 = -1;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i <= j; ++i) {                                                        // rising code j * 5 + i (:251-268)
                    const float v = sc[j][i]; const int code = j * 5 + i;
                    if (g0 < 0 || v < s0) { s1 = s0; g1 = g0; s0 = v; g0 = code; }
                    else if (g1 < 0 || v < s1) { s1 = v; g1 = code; }
                }
            o.quality = s1 - s0;                                                                      // gts[1].score - gts[0].score, :470
            o.genotype = (uint8_t)g0;
            o.reference_base5 = (uint8_t)ref5;
            o.flags = (uint8_t)(BGP_MAQ_CALLED | (g0 != (int)(ref5 * 5 + ref5) ? BGP_MAQ_VARIANT : 0));    // is_variant: call.d:85-87
            o.n_bases = (uint8_t)n;
        }
        a.out[p] = o;
    }
    // ---- the deep columns of the block, one after the other, the whole wave on each
    for (unsigned long long dm = __ballot(deep); dm; dm &= dm - 1) {
        const int c = __ffsll((long long)dm) - 1;
        maq_column_coop(a, col0 + (uint64_t)c, lane, s_coop[wv]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

}  // namespace

// ErrorModelCoefficients(depcorr, eta), maq.d:73-118, in the reference's precisions: double where it says double,
// long double (x87 80-bit, D's `real`) where it says real, the C library's pow / lgamma / logl / expl as it calls them.
static void maq_build_tables(double depcorr, double eta, std::vector<double>& fk, std::vector<double>& beta, std::vector<double>& lhet) {
    fk.assign(256, 0.0); beta.assign((size_t)256 * 256 * 64, 0.0); lhet.assign(256 * 256, 0.0);
    for (int n = 0; n < 256; ++n) fk[n] = pow(1.0 - depcorr, (double)n) * (1.0 - eta) + eta;
    std::vector<double> lC(256 * 256, 0.0);
    double lG[256];
    const long double LN2 = 0x1.62e42fefa39ef358p-1L, LN10 = 0x1.26bb1bbb5551582ep+1L;               // std.math.LN2, LN10
    for (int n = 0; n <= 255; ++n) {
        lG[n] = lgamma((double)(n + 1));
        for (int k = 0; k <= n / 2; ++k) {
            lC[n * 256 + (n - k)] = lC[n * 256 + k] = lG[n] - lG[k] - lG[n - k];
            lhet[(n << 8) | (n - k)] = lhet[(n << 8) | k] = lC[n * 256 + k] - n * (double)LN2;
        }
    }
    for (int q = 1; q < 64; ++q) {
        const long double e = powl(10.0L, -((long double)q) / 10.0L);
        const long double le = logl(e), le1 = logl(1.0L - e);
        for (int n = 1; n <= 255; ++n) {
            long double sum = 0.0L, sum1 = 0.0L;
            for (int k = n; k >= 0; --k) {
                sum = sum1 + expl(lC[n * 256 + k] + k * le + (n - k) * le1);
                beta[((size_t)q << 16) | ((size_t)n << 8) | (size_t)k] = (double)(-10.0L / LN10 * logl(sum1 / sum));
                sum1 = sum;
            }
        }
    }
}

// d_maq[0 .. np) for the columns [col_index, col_index + np) of the window whose entries K5 has just made
int k8_maq_calls(bgp_ctx* c, uint64_t col_index, uint64_t ref_index, uint64_t np, int use_md) {
    if (!c->maq_tables_ready || c->maq_tab_depcorr != c->it.maq_depcorr || c->maq_tab_eta != c->it.maq_eta) {
        std::vector<double> fk, beta, lhet;
        maq_build_tables((double)c->it.maq_depcorr, (double)c->it.maq_eta, fk, beta, lhet);         // (the caller's floats widened: MaqSnpCaller keeps float _depcorr, _eta, :329-330,320)
        BGP_TRY(c->d_maq_tab.reserve((fk.size() + lhet.size() + beta.size()) * 8));
        double* d = c->d_maq_tab.as<double>();
        BGP_HIP(hipMemcpyAsync(d, fk.data(), fk.size() * 8, hipMemcpyHostToDevice, c->stream));
        BGP_HIP(hipMemcpyAsync(d + 256, lhet.data(), lhet.size() * 8, hipMemcpyHostToDevice, c->stream));
        BGP_HIP(hipMemcpyAsync(d + 256 + 65536, beta.data(), beta.size() * 8, hipMemcpyHostToDevice, c->stream));
        BGP_HIP(hipStreamSynchronize(c->stream));
        c->maq_tables_ready = true; c->maq_tab_depcorr = c->it.maq_depcorr; c->maq_tab_eta = c->it.maq_eta;
    }
    BGP_TRY(c->d_maq.reserve((np + 1) * sizeof(bgp_maq_call)));
    if (!np) return BGP_OK;
    MaqArgs a;
    a.entries = c->d_entries.as<bgp_entry>(); a.entry_off = c->d_entry_off.as<uint64_t>(); a.rec = c->d_rec.as<bgp_record>();
    a.refbase = use_md ? c->d_refbase.as<uint8_t>() + ref_index : nullptr; a.np = np;
    const double* d = c->d_maq_tab.as<double>();
    a.fk = d; a.lhet = d + 256; a.beta = d + 256 + 65536; a.min_bq = c->it.maq_min_bq;
    a.out = c->d_maq.as<bgp_maq_call>();
    TimedRegion t(c, "k8_maq", np * sizeof(bgp_maq_call));
    hipLaunchKernelGGL(k8_maq64, dim3((unsigned)((np + 255) / 256)), dim3(MAQ_T), 0, c->stream, a);
    BGP_HIP(hipGetLastError());
    return BGP_OK;
}

// ErrorModelCoefficients(depcorr, eta) as K8 uses them, for callers and tests that want the tables themselves (host
// computation: fk[256], lhet[256 * 256] indexed n << 8 | k, beta[64 * 256 * 256] indexed q << 16 | n << 8 | k).
extern "C" int bgp_maq_error_model(float depcorr, float eta, double* fk, double* lhet, double* beta) {
    try {
        if (!fk || !lhet || !beta || !(depcorr >= 0.f && depcorr <= 1.f) || !(eta >= 0.f && eta <= 1.f)) return BGP_EINVAL;
        std::vector<double> f, b, l;
        maq_build_tables((double)depcorr, (double)eta, f, b, l);
        memcpy(fk, f.data(), f.size() * 8); memcpy(lhet, l.data(), l.size() * 8); memcpy(beta, b.data(), b.size() * 8);
        return BGP_OK;
    } catch (...) { return BGP_ENOMEM; }
}
