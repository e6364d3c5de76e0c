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

__global__ __launch_bounds__(MAQ_T) void k8_maq(MaqArgs a) {
    __shared__ uint8_t s_q[4][256], s_c[4][256], s_sq[4][256], s_sc[4][256];
    __shared__ double s_fs[4][16], s_bs[4][16];
    __shared__ uint32_t s_w[4][32], s_cn[4][16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t p = (uint64_t)blockIdx.x * 4 + wv;
    if (p >= a.np) return;
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
        if (valid && idx < (uint32_t)MAQ_MAXB) { s_q[wv][idx] = (uint8_t)q; s_c[wv][idx] = (uint8_t)code; }
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
        const uint32_t qi = s_q[wv][i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; ++j) { const uint32_t qj = s_q[wv][j]; rank += (qj < qi || (qj == qi && j < i)) ? 1u : 0u; }
        s_sq[wv][rank] = (uint8_t)qi; s_sc[wv][rank] = s_c[wv][i];
    }
    if (lane < 32) s_w[wv][lane] = 0;
    if (lane < 16) { s_cn[wv][lane] = 0; s_fs[wv][lane] = 0.0; s_bs[wv][lane] = 0.0; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane != 0) return;
    // ---- from the best base down (:153-166)
    for (int i = (int)n - 1; i >= 0; --i) {
        uint32_t q = s_sq[wv][i];
        q = q < 4u ? 4u : q > 63u ? 63u : q;
        const uint32_t bws = s_sc[wv][i], b = bws >> 1;
        const double f = a.fk[s_w[wv][bws]];
        s_fs[wv][b] += f;
        s_bs[wv][b] += f * a.beta[(q << 16) | (n << 8) | s_cn[wv][b]];
        s_cn[wv][b] += 1; s_w[wv][bws] += 1;
    }
    // ---- the ten genotype scores (:170-228); Base codes of A, C, G, T are 1, 2, 4, 8; Base5 codes 0..3
    const double Cc = 10.0 / 2.30258509299404568401799145468436421;                                   // 10 / LN10
    float sc[4][4];                                                                                   // sc[j][i], j >= i: genotype code j * 5 + i
    for (int i = 0; i < 4; ++i) {
        float tmp1 = 0.f; int tmp2 = 0;
        for (int k = 0; k < 4; ++k) if (k != i) { tmp1 = (float)((double)tmp1 + s_bs[wv][1 << k]); tmp2 += (int)s_cn[wv][1 << k]; }
        sc[i][i] = tmp2 > 0 ? tmp1 : 0.f;
        for (int j = i + 1; j < 4; ++j) {
            const uint32_t cij = s_cn[wv][1 << i] + s_cn[wv][1 << j];
            tmp1 = 0.f; tmp2 = 0;
            for (int k = 0; k < 4; ++k) if (k != i && k != j) { tmp1 = (float)((double)tmp1 + s_bs[wv][1 << k]); tmp2 += (int)s_cn[wv][1 << k]; }
            const double lh = Cc * a.lhet[(cij << 8) | s_cn[wv][1 << j]];
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
int k8_maq_calls(bgp_ctx* c, uint64_t col_index, uint64_t np, int use_md) {
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
    a.refbase = use_md ? c->d_refbase.as<uint8_t>() + col_index : nullptr; a.np = np;
    const double* d = c->d_maq_tab.as<double>();
    a.fk = d; a.lhet = d + 256; a.beta = d + 256 + 65536; a.min_bq = c->it.maq_min_bq;
    a.out = c->d_maq.as<bgp_maq_call>();
    TimedRegion t(c, "k8_maq", np * sizeof(bgp_maq_call));
    hipLaunchKernelGGL(k8_maq, dim3((unsigned)((np + 3) / 4)), dim3(MAQ_T), 0, c->stream, a);
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
