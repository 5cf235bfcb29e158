"""Shared test helpers (CPU-side checkers; the product never imports these)."""
import os
from fractions import Fraction

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

FPE_VARIANTS_SUM = [(0, False), (2, False), (3, False), (4, False), (8, False), (4, True), (6, True), (8, True)]
FPE_VARIANTS_DOT = [(0, False), (3, False), (4, False), (8, False), (4, True), (6, True), (8, True)]


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def golden_cases(g, *keys):
    offs = g["offsets"]
    for i, name in enumerate(g["names"]):
        yield (str(name), i, *[g[k][offs[i]:offs[i + 1]] for k in keys])


def same_double(x, y):
    """Bitwise equality of two doubles (NaN == NaN, +0 == -0 treated as equal only if bits equal or both zero)."""
    a, b = np.float64(x), np.float64(y)
    if np.isnan(a) and np.isnan(b):
        return True
    if a == 0 and b == 0:
        return True
    return a.view(np.int64) == b.view(np.int64)


def exact_int_from_canon(limbs):
    """canonical 41 x 52-bit limbs -> Python integer in units of 2^-1092"""
    v = 0
    for i, l in enumerate(limbs):
        v += int(l) << (52 * i)
    return v


def exact_int_from_digits(digits):
    """68 x 32-bit-spaced int64 limbs (any carry state) -> Python integer in units of 2^-1074"""
    v = 0
    for i, d in enumerate(digits):
        v += int(d) << (32 * i)
    return v


def digits_from_int(v, n=68):
    """normalised digit vector (as the HIP finalize produces) of an integer in units of 2^-1074"""
    out = []
    for _ in range(n - 1):
        out.append(v & 0xffffffff)
        v >>= 32
    out.append(v)
    return np.array(out, dtype=np.int64)


def exact_sum_int(a):
    """exact sum of doubles as an integer in units of 2^-1074"""
    tot = Fraction(0)
    for x in np.asarray(a, dtype=np.float64):
        tot += Fraction(float(x))
    tot *= 2**1074
    assert tot.denominator == 1
    return tot.numerator
