"""Rounding margin of the one-limb external product, modelled in numpy.

The wide-launch blind rotation multiplies 7-bit digit polynomials with the 32-bit bootstrapping-key polynomials through
ONE double-precision transform (as libtfhe does) and rounds the sum of 2l = 6 products to integers.  This is the same
computation in numpy (pocketfft instead of the kernel's 8x8x8 radix-8 schedule: same precision class), compared with
exact integer arithmetic: the rounded result must be the exact product and the distance to the nearest integer must
stay far below the kernel's guard limit (1/16) -- for random data, for every operand at its extreme magnitude, and for
the worst alignment (all terms of one sign, sum 2^49.6)."""
import numpy as np

N = 1024


def _fft_product_sum(digs, bks):
    j = np.arange(N // 2)
    tw = np.exp(1j * np.pi * j / N)
    acc = np.zeros(N // 2, dtype=np.complex128)
    for d, b in zip(digs, bks):
        fd = np.fft.fft((d[: N // 2] + 1j * d[N // 2:]) * tw)
        fb = np.fft.fft((b[: N // 2].astype(np.float64) + 1j * b[N // 2:].astype(np.float64)) * tw)
        acc += fd * fb
    y = np.fft.ifft(acc) * np.conj(tw)
    return np.concatenate([y.real, y.imag])


def _exact_product_sum(digs, bks):
    out = np.zeros(N, dtype=object)
    for d, b in zip(digs, bks):
        full = np.convolve(np.array([int(v) for v in d], dtype=object), np.array([int(v) for v in b], dtype=object))
        res = full[:N].copy()
        res[: N - 1] -= full[N:]
        out += res
    return out


def _check(digs, bks, limit):
    y = _fft_product_sum(digs, bks)
    exact = _exact_product_sum(digs, bks)
    rounded = np.rint(y)
    assert all(int(rounded[i]) == exact[i] for i in range(N))
    dev = float(np.abs(y - rounded).max())
    assert dev < limit, dev
    return dev


def test_random_operands_round_to_the_exact_product():
    rng = np.random.default_rng(1)
    worst = 0.0
    for _ in range(4):
        worst = max(worst, _check(rng.integers(-64, 64, size=(6, N)), rng.integers(-2**31, 2**31, size=(6, N)), 1 / 64))
    assert worst > 0  # the transform is approximate: what is being relied on is the margin, not exactness of the FFT


def test_extreme_magnitudes_and_worst_alignment():
    rng = np.random.default_rng(2)
    _check(rng.choice([-64, 63], size=(6, N)), rng.choice([-2**31, 2**31 - 1], size=(6, N)), 1 / 16)
    # every term of the same sign -- the largest sum the parameters allow (6 x 1024 x 64 x 2^31 = 2^49.6, where a double's
    # grid is 1/8 wide): no margin is left, the computed values sit up to half a step from an integer.  This is the case the
    # guard exists for: it sees a distance far above its limit and the call is repeated on the two-limb kernels.
    digs, bks = np.full((6, N), -64), np.full((6, N), -2**31)
    y = _fft_product_sum(digs, bks)
    exact = _exact_product_sum(digs, bks)
    assert max(abs(float(y[i]) - float(exact[i])) for i in range(N)) <= 1.0
    assert np.abs(y - np.rint(y)).max() > 1 / 16


def test_old_parameter_set_has_less_headroom():
    """l=2, Bgbit=10 (libtfhe 1.0): 10-bit digits, sums up to 2^52 -- why the evaluator keeps that set on the two-limb kernels."""
    rng = np.random.default_rng(3)
    y3 = _fft_product_sum(rng.integers(-64, 64, size=(6, N)), rng.integers(-2**31, 2**31, size=(6, N)))
    y2 = _fft_product_sum(rng.integers(-512, 512, size=(4, N)), rng.integers(-2**31, 2**31, size=(4, N)))
    d3, d2 = np.abs(y3 - np.rint(y3)).max(), np.abs(y2 - np.rint(y2)).max()
    assert d2 > 3 * d3
