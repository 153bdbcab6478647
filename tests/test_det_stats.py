"""CPU test of the exact statistics accumulator (misonet_amd/csrc/det_stats.hpp): the limb arithmetic is compiled for the
host with g++ (tests/helpers/det_stats_host.cpp) and checked against exact rational arithmetic -- the properties the GPU
kernels rely on for bit-reproducible instance-norm / gLN statistics (reference model.py:413,430,445,609-632 are
deterministic): every float32 partial is represented exactly, the accumulated value does not depend on the order of the
additions, and non-finite partials poison the statistic."""
import ctypes as C
import math
import os
import subprocess
from fractions import Fraction

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ds(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("ds") / "libds.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", out,
                    os.path.join(ROOT, "tests", "helpers", "det_stats_host.cpp")], check=True)
    L = C.CDLL(out)
    L.ds_accumulate.argtypes = [C.POINTER(C.c_longlong), C.c_double]
    L.ds_split.argtypes = [C.c_double, C.POINTER(C.c_longlong)]
    L.ds_value.argtypes = [C.POINTER(C.c_longlong)]
    L.ds_value.restype = C.c_double
    assert L.ds_nl() == 5
    return L


def _limbs_value(q):
    return sum(Fraction(int(v)) * Fraction(2) ** (40 * i - 80) for i, v in enumerate(q))


def test_split_is_exact_for_float32_partials(ds):
    r = np.random.default_rng(0)
    vals = np.concatenate([
        (r.standard_normal(2000) * np.exp(r.uniform(-40, 40, 2000))).astype(np.float32),      # 35 decades of scale
        np.float32([0.0, -0.0, 1.0, -1.0, 2.0 ** -70, -(2.0 ** -79), 2.0 ** 100, -3.0e35, 1e-30]),
    ])
    for v in vals:
        q = (C.c_longlong * 5)()
        ds.ds_split(float(v), q)
        assert all(abs(int(x)) < 2 ** 40 for x in q)
        exact = Fraction(float(v))
        got = _limbs_value(q)
        # exact down to the resolution 2^-80; below it the value is truncated toward zero
        assert abs(exact - got) < Fraction(2) ** -80 and (exact - got == 0 or abs(exact) < Fraction(2) ** -56)
        if abs(float(v)) >= 2.0 ** -56:                        # 24 significant bits above 2^-80: nothing is lost
            assert got == exact


@pytest.mark.parametrize("scale", [1e-6, 1.0, 1e6])
def test_accumulation_is_order_independent_and_exact(ds, scale):
    r = np.random.default_rng(1)
    # the partial sums of sum(x^2) of 4096 tiles of a layer, at three input scales
    parts = (scale * scale * np.abs(r.standard_normal(4096)) * 4096).astype(np.float32)
    orders = [np.arange(4096), np.arange(4096)[::-1], r.permutation(4096), r.permutation(4096)]
    results = []
    for o in orders:
        acc = (C.c_longlong * 5)()
        for i in o:
            assert ds.ds_accumulate(acc, float(parts[i])) == 1
        results.append((tuple(int(a) for a in acc), ds.ds_value(acc)))
    assert all(x == results[0] for x in results[1:]), "limbs and value must not depend on the order of the additions"
    exact = sum(Fraction(float(p)) for p in parts)
    got = Fraction(results[0][1])
    assert abs(got - exact) <= abs(exact) * Fraction(2) ** -52      # one float64 rounding of the exact sum


def test_signed_sums_cancel_exactly(ds):
    r = np.random.default_rng(2)
    x = (r.standard_normal(1000) * 1e3).astype(np.float32)
    acc = (C.c_longlong * 5)()
    for v in np.concatenate([x, -x])[r.permutation(2000)]:
        ds.ds_accumulate(acc, float(v))
    assert ds.ds_value(acc) == 0.0


def test_non_finite_partials_poison(ds):
    for bad in (float("inf"), float("-inf"), float("nan"), 2.0 ** 120):
        acc = (C.c_longlong * 5)()
        ds.ds_accumulate(acc, 1.5)
        assert ds.ds_accumulate(acc, bad) == 0
        ds.ds_accumulate(acc, -2.0)
        assert math.isnan(ds.ds_value(acc))


@pytest.mark.parametrize("n_bad", [1, 2, 3, 4, 5, 8, 64, 1024, 4096])
def test_poison_is_idempotent(ds, n_bad):
    """a NaN activation poisons MANY tiles of a (sample, channel) row.  With an additive poison two or three bad
    partials read as a finite -2^143 and four wrapped to exactly zero; the poison is now a MAX, so any count, in any
    interleaving with finite partials of either sign (incl. the largest the range allows), reads as NaN."""
    r = np.random.default_rng(n_bad)
    good = (r.standard_normal(4096) * np.float32(1e30)).astype(np.float32)        # top-limb sized partials, both signs
    ops = [("g", float(v)) for v in good] + [("b", b) for b in
                                             (r.choice([float("nan"), float("inf"), float("-inf"), -(2.0 ** 119)], n_bad))]
    for trial in range(3):
        acc = (C.c_longlong * 5)()
        for i in r.permutation(len(ops)):
            kind, v = ops[i]
            assert ds.ds_accumulate(acc, v) == (1 if kind == "g" else 0)
        assert math.isnan(ds.ds_value(acc)), (n_bad, trial, [int(a) for a in acc])
    # worst case by construction: 2^23 - 1 additions of the most negative legitimate top limb AFTER the poison
    acc = (C.c_longlong * 5)()
    ds.ds_accumulate(acc, float("nan"))
    acc[4] += -(2 ** 38 - 1) * (2 ** 23 - 1)
    for _ in range(n_bad):
        ds.ds_accumulate(acc, float("inf"))
        acc[4] += -(2 ** 36)
    assert math.isnan(ds.ds_value(acc))
