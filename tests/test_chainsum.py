"""pbdx_chainsum.h: a sequential float sum (every partial sum rounded to float) evaluated run by run -- integer state on the binade's
grid, elements as parity-dependent increments that compose associatively -- must equal the plain loop BIT FOR BIT on anything.
The host restatement runs the same element / composition code the device kernel does (blocks per thread, windows)."""
import ctypes as C

import numpy as np
import pytest

from positionbaseddynamics_amd import _ffi


def both(x, threads=64, per_thread=16):
    x = np.ascontiguousarray(x, dtype=np.float32)
    a, b = C.c_float(), C.c_float()
    singles = C.c_uint64()
    _ffi.check(_ffi.lib.pbdx_debug_chain_sum_host(x.ctypes.data_as(_ffi.pf), len(x), threads, per_thread, C.byref(a), C.byref(b), C.byref(singles)), "chain_sum_host")
    return np.float32(a.value), np.float32(b.value), singles.value


def same_bits(a, b):
    return np.array([a], dtype=np.float32).view(np.uint32)[0] == np.array([b], dtype=np.float32).view(np.uint32)[0]


def cases():
    rng = np.random.default_rng(7)
    n = 60000
    yield "positive uniform (monotone growth through 17 binades)", rng.random(n, dtype=np.float32) * 0.7 + 0.1
    yield "symmetric noise (sum wanders around zero)", (rng.random(n, dtype=np.float32) - 0.5)
    x = np.sort(rng.random(n, dtype=np.float32) * 2 - 1)
    yield "all negatives first, then the positives (a kd-tree's first split)", x
    yield "negatives only", -rng.random(n, dtype=np.float32)
    yield "mixed magnitudes 1e-20 .. 1e20", (rng.random(n, dtype=np.float32) - 0.5) * np.float32(10.0) ** rng.integers(-20, 20, n).astype(np.float32)
    # ties: addends that are odd multiples of half an ulp of the running sum
    base = np.full(n, 0.5, dtype=np.float32)
    base[0] = 1024.0
    base[1:] = np.float32(2.0 ** -14) * rng.integers(1, 64, n - 1).astype(np.float32)      # ulp(1024) = 2^-13: multiples of half an ulp
    yield "half-ulp addends (every second one a tie)", base
    t = np.full(n, np.float32(2.0 ** -24), dtype=np.float32)
    t[0] = 1.0
    yield "1 + n * 2^-24 (all ties, the sum never moves)", t
    t2 = np.full(n, np.float32(3 * 2.0 ** -25), dtype=np.float32)
    t2[0] = 1.0
    yield "1 + n * 3 * 2^-25 (rounds up every time)", t2
    d = (rng.integers(0, 1 << 22, n).astype(np.uint32)).view(np.float32)
    yield "denormals only", d
    e = rng.random(n, dtype=np.float32)
    e[::2] = -e[1::2]
    yield "exact cancellation to +0 every second element", e
    big = rng.random(n, dtype=np.float32)
    big[n // 2] = 3e38
    big[n // 2 + 5] = 3e38
    yield "overflow to infinity in the middle", big
    nn = rng.random(1000, dtype=np.float32)
    nn[500] = np.nan
    yield "a NaN", nn
    yield "empty", np.zeros(0, dtype=np.float32)
    yield "one element", np.array([-3.25], dtype=np.float32)
    yield "zeros", np.zeros(1000, dtype=np.float32)
    yield "negative zeros", -np.zeros(1000, dtype=np.float32)
    p2 = np.full(n, 1.0, dtype=np.float32)
    yield "ones (hits every power of two exactly)", p2
    # powers of two boundaries from above: sum decreasing through binades
    dec = np.full(n, -1.0, dtype=np.float32)
    dec[0] = 40000.0
    yield "40000 - 1 - 1 ... (exits downwards, through zero, then negative)", dec
    w = (rng.random(n, dtype=np.float32) - 0.5) * 1e-3
    w[0] = 0.25
    yield "small noise around 0.25 (crosses 0.25 back and forth)", w
    for seed in range(6):
        r = np.random.default_rng(100 + seed)
        bits = r.integers(0, 1 << 32, 20000, dtype=np.uint64).astype(np.uint32)
        bits &= np.uint32(0xBFFFFFFF)                      # exponents below 2^64: no overflow, every other bit pattern
        yield "random bit patterns %d" % seed, bits.view(np.float32)
    # mesh coordinates: a regular grid's x coordinates in an interleaved order
    g = np.linspace(-1.0, 1.0, 129, dtype=np.float32)
    xs = np.repeat(g, 33 * 33)[rng.permutation(129 * 33 * 33)]
    yield "grid coordinates, shuffled", xs
    yield "grid coordinates, sorted", np.sort(xs)


@pytest.mark.parametrize("name,x", list(cases()), ids=[c[0] for c in cases()])
def test_run_by_run_sum_equals_the_plain_loop(name, x):
    for threads, per_thread in ((64, 16), (1024, 16), (7, 3), (1, 1)):
        a, b, singles = both(x, threads, per_thread)
        assert same_bits(a, b) or (np.isnan(a) and np.isnan(b)), "%s (%d x %d): %r vs %r" % (name, threads, per_thread, a, b)


def test_regular_data_is_almost_all_runs():
    """the point of the exercise: on data like vertex coordinates nearly every element is handled inside a run"""
    rng = np.random.default_rng(3)
    x = rng.random(300000, dtype=np.float32) * 0.5 + 0.25
    a, b, singles = both(x, 1024, 16)
    assert same_bits(a, b)
    assert singles < 60, singles          # the first element and ~18 binade crossings


@pytest.mark.gpu
def test_device_run_by_run_sum_equals_the_plain_loop():
    """the kernel code of the long bounding-sphere sums (one workgroup: staging, scan of the functions, replay, fall-back bursts) on the
    same adversarial data, plus long inputs"""
    import positionbaseddynamics_amd as pbd
    sol = pbd.Solver()
    rng = np.random.default_rng(11)
    extra = [("long positive", rng.random(700000, dtype=np.float32) * 0.5 + 0.25),
             ("long symmetric", rng.random(300000, dtype=np.float32) - 0.5),
             ("long sorted", np.sort(rng.random(500000, dtype=np.float32) * 2 - 1)),
             ("window edge 16384", rng.random(16384 + 2048, dtype=np.float32) + 1.0),
             ("window edge 16385", rng.random(16385 + 2048, dtype=np.float32) + 1.0)]
    bad = []
    for name, x in list(cases()) + extra:
        x = np.ascontiguousarray(x, dtype=np.float32)
        _, plain, _ = both(x, 64, 16)
        out = C.c_float()
        _ffi.check(_ffi.lib.pbdx_debug_chain_sum(sol._h, x.ctypes.data_as(_ffi.pf), len(x), C.byref(out)), "debug_chain_sum")
        dev = np.float32(out.value)
        if not (same_bits(dev, plain) or (np.isnan(dev) and np.isnan(plain))):
            bad.append((name, dev, plain))
    assert not bad, bad
