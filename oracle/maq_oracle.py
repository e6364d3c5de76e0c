"""TEST INFRASTRUCTURE -- a CPU restatement of BioD's MAQ consensus model, bio/std/hts/snpcallers/maq.d, used only by
tests/ as the checker of the device kernel (biod_amd/csrc/maq.hip).  Nothing under biod_amd/ imports it.

PARITY UNPINNED: the reference holds no unittest, fixture or expected output for this module (snpcallers/*.d: none), and
three steps of it are not functions of their input -- an unstable sort whose order among equal qualities changes the
sums (maq.d:146-166), a random subsample above 255 bases (:141-143), druntime's 80-bit `real` in the tables (:106-118).
This restatement fixes them the way include/biodgpu.h states: stable sort (file order), the first 255 bases, numpy's
longdouble (x87 80-bit on this platform, the C library's powl / logl / expl as D's core.stdc.math calls them).

Statement by statement after:
  ErrorModelCoefficients.this      maq.d:73-118
  computeLikelihoods               maq.d:133-240
  GenotypeLikelihoodInfo.this      maq.d:245-279
  MaqSnpCaller.genotypeLikelihoodInfo / makeCall   maq.d:388-486
  Base / Base5 codes               bio/core/base.d:41-63,188-213;  DiploidGenotype code  bio/core/genotype.d:33-37
"""
import ctypes
import ctypes.util
import math

import numpy as np

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.lgamma.restype = ctypes.c_double; _libm.lgamma.argtypes = [ctypes.c_double]


def c_lgamma(x: float) -> float:
    """core.stdc.math.lgamma IS the C library's (maq.d:89); CPython's math.lgamma is an implementation of its own and
    differs from it in the last place."""
    return float(_libm.lgamma(x))


LD = np.longdouble
LN10 = LD("2.30258509299404568401799145468436421")       # std.math.LN10 (real)
LN2_D = math.log(2.0)                                     # cast(double) LN2
C_REAL = LD(10.0) / LN10                                  # private immutable C = 10.0 / LN10  (maq.d:131: a real)

_CHAR2CODE16 = {"=": 0, "A": 1, "C": 2, "M": 3, "G": 4, "R": 5, "S": 6, "V": 7, "T": 8, "W": 9, "Y": 10, "H": 11, "K": 12, "D": 13, "B": 14, "N": 15,
                "0": 1, "1": 2, "2": 4, "3": 8}
NUCLEOTIDES16 = (1, 2, 4, 8)                              # Base('A'), 'C', 'G', 'T' (maq.d:71)


def base16(ch: int) -> int:                               # Base(char), base.d:41-63
    return _CHAR2CODE16.get(chr(ch).upper(), 15)


def base5(ch: int) -> int:                                # Base5(char), base.d:188-213
    return {"A": 0, "C": 1, "G": 2, "T": 3}.get(chr(ch).upper(), 4)


_TABLES = {}


def tables(depcorr: float, eta: float):
    """ErrorModelCoefficients(depcorr, eta), maq.d:73-118.  MaqSnpCaller keeps its parameters as floats (:329-330) and
    widens them: the keys here are the float32 values."""
    depcorr, eta = float(np.float32(depcorr)), float(np.float32(eta))
    if (depcorr, eta) in _TABLES:
        return _TABLES[(depcorr, eta)]
    fk = np.array([math.pow(1.0 - depcorr, float(n)) * (1.0 - eta) + eta for n in range(256)])                # :78-80
    lG = [c_lgamma(float(n + 1)) for n in range(256)]                                                       # :89
    lC = np.zeros((256, 256))
    lhet = np.zeros(256 * 256)
    for n in range(256):
        for k in range(n // 2 + 1):                                                                            # :90-95
            lC[n][n - k] = lC[n][k] = lG[n] - lG[k] - lG[n - k]
            lhet[n << 8 | (n - k)] = lhet[n << 8 | k] = lC[n][k] - n * LN2_D
    beta = np.zeros(64 * 256 * 256)
    q = np.arange(64)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        e = LD(10.0) ** (-(q.astype(LD)) / LD(10.0))                                                           # :99  (q = 0 is computed and never stored)
        le, le1 = np.log(e), np.log(LD(1.0) - e)                                                               # :100-101
        ssum = np.zeros((64, 256), LD); sum1 = np.zeros((64, 256), LD)
        n = np.arange(256)
        lCl = lC.astype(LD)
        for k in range(255, -1, -1):                                                                           # :106: for k = n .. 0, all (q, n) at once
            act = n >= k
            expo = lCl[n, np.minimum(k, n)][None, :] + LD(k) * le[:, None] + (n - k).astype(LD)[None, :] * le1[:, None]     # lC[n][k] + k * le + (n-k) * le1
            new = sum1 + np.exp(expo)
            val = (LD(-10.0) / LN10 * np.log(sum1 / new)).astype(np.float64)                                   # :108
            for qq in range(1, 64):
                idx = (qq << 16) | (n[act] << 8) | k
                beta[idx[n[act] >= 1]] = val[qq][act][n[act] >= 1]
            ssum = np.where(act[None, :], new, ssum); sum1 = np.where(act[None, :], new, sum1)
    _TABLES[(depcorr, eta)] = (fk, beta, lhet)
    return _TABLES[(depcorr, eta)]


def compute_likelihoods(read_bases, tab):
    """computeLikelihoods, maq.d:133-240.  read_bases: (code16, is_reverse, quality) in file order.
    Returns {genotype code: float32 score} for the genotypes the reference's Dict holds."""
    fk, beta, lhet = tab
    bases = list(read_bases)[:255]                                        # :139-145 (the reference: a RANDOM sample of 255)
    bases = sorted(bases, key=lambda rb: rb[2])                           # :146 (the reference: unstable)
    n = len(bases)
    w = [0] * 32; c = [0] * 16; fsum = [0.0] * 16; bsum = [0.0] * 16
    for code16, rev, quality in reversed(bases):                          # :153-166
        quality = min(max(quality, 4), 63)
        bws = code16 * 2 + (1 if rev else 0)
        fsum[code16] += fk[w[bws]]
        bsum[code16] += fk[w[bws]] * beta[quality << 16 | n << 8 | c[code16]]
        c[code16] += 1
        w[bws] += 1
    qd = {}
    for i, b1 in enumerate(NUCLEOTIDES16):                                # :172-228
        tmp1 = np.float32(0.0); tmp2 = 0
        for k, b2 in enumerate(NUCLEOTIDES16):
            if k != i:
                tmp1 = np.float32(np.float64(tmp1) + bsum[b2]); tmp2 += c[b2]
        qd[i * 5 + i] = tmp1 if tmp2 > 0 else np.float32(0.0)            # :187-191
        for j in range(i + 1, 4):
            b2 = NUCLEOTIDES16[j]
            cij = c[b1] + c[b2]
            tmp1 = np.float32(0.0); tmp2 = 0
            for k, b3 in enumerate(NUCLEOTIDES16):
                if k != i and k != j:
                    tmp1 = np.float32(np.float64(tmp1) + bsum[b3]); tmp2 += c[b3]
            lh = C_REAL * LD(lhet[cij << 8 | c[b2]])
            qd[j * 5 + i] = np.float32(LD(tmp1) - lh) if tmp2 > 0 else np.float32(-lh)      # :203-207: dG(b2, b1), real arithmetic
        for k in range(4):                                                # :221-226
            g = i * 5 + k
            if g in qd and qd[g] < 0.0:
                qd[g] = np.float32(0.0)
    return qd


def make_call(read_bases, reference_base_char: int, tab):
    """genotypeLikelihoodInfo + makeCall, maq.d:388-486, for the VALID bases of one column.
    Returns None (no valid base) or (genotype code, quality float32, reference base5, is_variant, n_used, subsampled)."""
    read_bases = list(read_bases)
    if not read_bases:
        return None                                                       # :436-439, :451
    qd = compute_likelihoods(read_bases, tab)
    order = sorted(sorted(qd), key=lambda g: qd[g])                       # :251-268: insertion sort over the genotypes in code order, stable
    g0, g1 = order[0], order[1]
    quality = np.float32(qd[g1] - qd[g0])                                 # :470
    r5 = base5(reference_base_char)
    return g0, quality, r5, g0 != r5 * 5 + r5, min(len(read_bases), 255), len(read_bases) > 255


def column_read_bases(entry_bases, entry_quals, mapqs, reverse, minimum_base_quality):
    """The reads of a column that count (maq.d:400-412) as (code16, is_reverse, min(base quality, mapping quality))."""
    out = []
    for b, q, mq, rv in zip(entry_bases, entry_quals, mapqs, reverse):
        if q < minimum_base_quality or b == ord("-"):
            continue
        out.append((base16(int(b)), bool(rv), min(int(q), int(mq))))
    return out
