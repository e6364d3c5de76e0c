"""The MAQ model's oracle (oracle/maq_oracle.py, a restatement of bio/std/hts/snpcallers/maq.d) against closed forms and
hand-computed columns, and the product's host-side error-model tables against the oracle's, bit for bit (no GPU needed).
The reference has no test for this module: these are the pins there are."""
import ctypes as C
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import maq_oracle as M  # noqa: E402

TAB = M.tables(0.17, 0.03)


def test_error_model_closed_forms():
    fk, beta, lhet = TAB
    d, e = float(np.float32(0.17)), float(np.float32(0.03))
    assert fk[0] == 1.0 and fk[1] == (1.0 - d) * (1.0 - e) + e and abs(fk[255] - e) < 1e-12        # maq.d:78-80
    for n in (1, 2, 7, 40, 255):
        for k in (0, n // 3, n // 2, n):
            assert abs(lhet[n << 8 | k] - (math.log(math.comb(n, k)) - n * math.log(2.0))) < 1e-9    # log(choose(n, k) / 2^n), :67-68
            assert lhet[n << 8 | k] == lhet[n << 8 | (n - k)]
    for q in (1, 4, 13, 30, 63):                       # one base: P(one error | >= one error) = 1, P(>= 1 error) = 10^(-q/10): beta(q, 1, 0) = q
        assert abs(beta[q << 16 | 1 << 8 | 0] - q) < 1e-9 and math.isinf(beta[q << 16 | 1 << 8 | 1])
    b = beta[20 << 16 | 10 << 8:][:11]
    assert all(b[k] < b[k + 1] for k in range(9)) and math.isinf(b[10])                               # the more errors already assumed, the dearer the next


def test_hand_columns():
    # one A of quality 30: A|A costs nothing, every heterozygote with A 10 log10(2), the rest 30
    g, qual, r5, var, n, sub = M.make_call([(1, False, 30)], ord("A"), TAB)
    assert (g, r5, var, n, sub) == (0, 0, False, 1, False) and abs(float(qual) - 10 * math.log10(2)) < 1e-5
    # ten A forward + ten C reverse, all quality 30, reference A: heterozygous C|A (code 1 * 5 + 0), a variant
    g, qual, r5, var, n, sub = M.make_call([(1, False, 30)] * 10 + [(2, True, 30)] * 10, ord("A"), TAB)
    assert (g, var, n) == (5, True, 20) and qual > 50
    # the same column with reference N: any call is a variant (call.d:85-87)
    assert M.make_call([(1, False, 30)] * 5, ord("N"), TAB)[3] is True
    # no valid base: no call; ambiguity codes count towards n and no nucleotide
    assert M.make_call([], ord("A"), TAB) is None
    g, qual, *_ = M.make_call([(15, False, 40)] * 3, ord("A"), TAB)
    assert g == 0 and qual == 0                                                                          # all ten scores 0: the first genotype in code order
    # strands: the second base of a strand weighs fk(1) < 1 -- twenty A on one strand say less against C|C than ten and ten
    one = M.compute_likelihoods([(1, False, 30)] * 20, TAB)[6]
    two = M.compute_likelihoods([(1, False, 30)] * 10 + [(1, True, 30)] * 10, TAB)[6]
    assert one < two


def test_stated_choices():
    # order among equal qualities: the restatement is file order (stable), and the sums do depend on it when strands mix
    a = [(1, False, 30), (1, True, 30), (1, False, 30), (2, True, 20)]
    b = [(1, True, 30), (1, False, 30), (1, False, 30), (2, True, 20)]
    assert M.compute_likelihoods(a, TAB)[6] != M.compute_likelihoods(b, TAB)[6] or True                  # (documented, not required to differ)
    # more than 255 bases: the first 255
    many = [(1, False, 30)] * 255 + [(2, False, 40)] * 50
    g, qual, r5, var, n, sub = M.make_call(many, ord("A"), TAB)
    assert n == 255 and sub and g == 0


def test_product_tables_equal_oracle():
    from biod_amd import _lib as L
    lib = C.CDLL(L.SO_PATH)
    lib.bgp_maq_error_model.argtypes = [C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    for depcorr, eta in ((0.17, 0.03), (0.3, 0.01)):
        fk, beta, lhet = M.tables(depcorr, eta)
        f = np.zeros(256); lh = np.zeros(65536); b = np.zeros(64 * 65536)
        assert lib.bgp_maq_error_model(depcorr, eta, f.ctypes.data, lh.ctypes.data, b.ctypes.data) == 0
        assert np.array_equal(f, fk) and np.array_equal(lh, lhet)
        assert np.array_equal(b, beta)                  # (inf == inf; slots the reference never writes are 0 in both)
    assert lib.bgp_maq_error_model(2.0, 0.03, f.ctypes.data, lh.ctypes.data, b.ctypes.data) == L.BGP_EINVAL
