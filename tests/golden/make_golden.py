"""Generates tests/golden/*.npz from the compiled reference core (oracle/_ref) and MPFR.

Run in the build container (needs /root/reference for `make -C oracle ref`):
    python tests/golden/make_golden.py
Every case stores DATA only: the raw little-endian float64 input, the 41 normalised limbs, the
double under both rounding modes and the MPFR double.  Inputs come from the reference's own
generators (src/common/common.cpp via oracle/_ref, glibc rand() seeded per case), from our
counter-based generators, and from hand-picked edge values.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
DBL_MAX = np.finfo(np.float64).max
TINY = np.finfo(np.float64).tiny


def exsum_cases():
    cases = []
    edge = {
        "one_pos_1p5": [1.5], "one_neg_1p5": [-1.5], "pos_2p53m2": [2.0**53 - 2], "neg_2p53m2": [-(2.0**53 - 2)],
        "dbl_max": [DBL_MAX], "neg_dbl_max": [-DBL_MAX], "max_max_negmax": [DBL_MAX, DBL_MAX, -DBL_MAX],
        "min_normal": [TINY], "denorm_min": [5e-324], "denorm_mix": [-5e-324, 2.5e-323, TINY, -TINY / 2],
        "zeros": [0.0, -0.0, 0.0], "cancel_exact": [1e308, 1.0, -1e308], "half_ulp_tie": [1.0, 2.0**-53],
        "half_ulp_tie_odd": [1.0 + 2.0**-52, 2.0**-53], "just_above_tie": [1.0, 2.0**-53, 2.0**-1074],
        "neg_just_above_tie": [-1.0, -2.0**-53, -2.0**-1074], "big_small": [2.0**1023, 2.0**-1074],
        "empty": [],
    }
    for name, v in edge.items():
        cases.append(("edge/" + name, np.array(v, dtype=np.float64)))
    for n in (1, 7, 8, 9, 1000, 4096):
        for seed in (1, 2):
            cases.append((f"ref_rand/naive/n{n}/s{seed}", O.ref_gen("naive", n, seed)))
            for rng, emax in ((10, 0), (50, 0), (600, 300)):
                cases.append((f"ref_rand/fpuniform_r{rng}_e{emax}/n{n}/s{seed}", O.ref_gen("fpuniform", n, seed, rng, emax)))
            for c in (1.0, 1e16, 1e32, 1e50):
                a = O.ref_gen("ill_cond", n, seed, c) if n > 1 else np.array([0.5])
                cases.append((f"ref_rand/ill_cond_c{c:g}/n{n}/s{seed}", a))
                cases.append((f"ref_rand/ill_cond_c{c:g}_negated/n{n}/s{seed}", -a))
    for n in (9, 1000, 4096):
        for kind, p0, p1 in (("lognormal", 0.0, 2.0), ("lognormal", 0.0, 50.0), ("cancel", 50.0, 0.0),
                             ("ill_cond", 1e32, 0.0), ("fpuniform_signed", 100.0, 50.0)):
            cases.append((f"ctr/{kind}_{p0:g}_{p1:g}/n{n}/s3", O.gen(kind, n, 3, p0, p1)))
    return cases


def main():
    assert O.ref() is not None and O.mpfr() is not None, "needs oracle/_ref and libmpfr_oracle.so"
    names, offs, data, limbs, d_exact, d_ref, d_mpfr = [], [0], [], [], [], [], []
    for name, a in exsum_cases():
        r_ref, l_ref = O.ref_exsum(a, 0, limbs=True)       # the reference's compiled core
        r_exact, l_or = O.exsum(a, 0, limbs=True)          # our restatement
        # subnormal inputs: the reference mis-decodes them -> neither its limbs nor its
        # double are the truth there; MPFR is
        sub = a.size and bool(np.any((np.abs(a) < TINY) & (a != 0)))
        assert sub or (l_ref == l_or).all(), name
        m = O.mpfr_exsum(a) if a.size else 0.0
        if not sub:
            assert O.exsum(a, 0, mode=O.ROUND_REFERENCE) == r_ref or (np.isnan(r_ref)), name
        assert r_exact == m or (np.isinf(m) and np.isinf(r_exact)), (name, r_exact, m)
        names.append(name)
        data.append(a)
        offs.append(offs[-1] + a.size)
        limbs.append(l_or)
        d_exact.append(r_exact)
        d_ref.append(O.exsum(a, 0, mode=O.ROUND_REFERENCE))
        d_mpfr.append(m)
    np.savez_compressed(os.path.join(HERE, "exsum_golden.npz"), names=np.array(names), offsets=np.array(offs),
                        data=np.concatenate(data) if data else np.zeros(0), limbs=np.array(limbs, dtype=np.int64),
                        exact=np.array(d_exact), refmode=np.array(d_ref), mpfr=np.array(d_mpfr))
    # ExDOT: no CPU reference implementation exists; golden = MPFR (tests/test.exdot.gpu.cpp:24-46) + our limbs
    names, offs, da, db, limbs, d_exact, d_mpfr = [], [0], [], [], [], [], []
    for n in (1, 7, 8, 9, 1000, 4096):
        for kind, p0, p1 in (("naive", 0, 0), ("fpuniform", 10, 0), ("fpuniform", 300, 150), ("lognormal", 0.0, 2.0),
                             ("lognormal", 0.0, 50.0), ("ill_cond", 1e32, 0.0), ("fpuniform_signed", 100.0, 50.0)):
            a, b = O.gen(kind, n, 11, p0, p1), O.gen(kind, n, 12, p0, p1)
            r, l = O.exdot(a, b, 0, limbs=True)
            m = O.mpfr_exdot(a, b)
            assert r == m, (kind, n)
            names.append(f"ctr/{kind}_{p0:g}_{p1:g}/n{n}")
            da.append(a); db.append(b); offs.append(offs[-1] + n); limbs.append(l); d_exact.append(r); d_mpfr.append(m)
    np.savez_compressed(os.path.join(HERE, "exdot_golden.npz"), names=np.array(names), offsets=np.array(offs),
                        a=np.concatenate(da), b=np.concatenate(db), limbs=np.array(limbs, dtype=np.int64),
                        exact=np.array(d_exact), mpfr=np.array(d_mpfr))
    print("wrote", len(names), "exdot cases;", "exsum cases:", len(exsum_cases()))


if __name__ == "__main__":
    main()
