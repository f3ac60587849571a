"""Array model of the 12-wave narrow-launch kernel's transforms (k_blind_rotate_wide12): every step written as the kernel
does it -- 4 registers x 64 lanes per wave, radix-4 passes, register <-> lane-bit-pair transposes -- and checked against the
defining negacyclic product.  Host only, numpy; documents the index maps the HIP code relies on."""
import numpy as np

N = 1024
lane = np.arange(64)
W256 = lambda e: np.exp(2j * np.pi * (np.asarray(e) % 256) / 256.0)


def transpose(v, hi):
    """register index (2 bits) <-> lane bits (hi, hi-1)"""
    out = np.empty_like(v)
    sh = hi - 1
    for r in range(4):
        for l in range(64):
            f = (l >> sh) & 3
            l2 = (l & ~(3 << sh)) | (r << sh)
            out[f][l2] = v[r][l]
    return out


def bfly(v, sign):
    """radix-4 over the register index: u[k] = sum_a (sign*i)^(a k) v[a]"""
    w = 1j * sign
    return np.array([sum(v[a] * w ** (a * k) for a in range(4)) for k in range(4)])


TW1 = np.array([W256(lane * k) for k in range(4)])                 # lane = (b, c, d)
TW2 = np.array([W256(4 * k * (lane & 15)) for k in range(4)])      # lane & 15 = 4c + d
TW3 = np.array([W256(16 * k * (lane & 3)) for k in range(4)])      # lane & 3 = d
TWIST = np.array([np.exp(1j * np.pi * (64 * a + lane) / 512.0) for a in range(4)])  # theta^n1, n1 = 64 a + lane


def fwd256(v):
    """in: v[a][lane] = x[64 a + lane] (untwisted); out: v[kd][lane = 16 ka + 4 kb + kc] = F[ka + 4 kb + 16 kc + 64 kd],
    F[k] = sum_n x[n] theta^n W256^(n k)"""
    v = v * TWIST
    v = bfly(v, +1) * TW1
    v = transpose(v, 5)
    v = bfly(v, +1) * TW2
    v = transpose(v, 3)
    v = bfly(v, +1) * TW3
    v = transpose(v, 1)
    return bfly(v, +1)


def inv256(v):
    """exact mirror: out[a][lane] = 256 * x[64 a + lane]"""
    v = bfly(v, -1)
    v = transpose(v, 1)
    v = bfly(v * np.conj(TW3), -1)
    v = transpose(v, 3)
    v = bfly(v * np.conj(TW2), -1)
    v = transpose(v, 5)
    v = bfly(v * np.conj(TW1), -1)
    return v * np.conj(TWIST)


KMAP = np.array([[(l >> 4) + 4 * ((l >> 2) & 3) + 16 * (l & 3) + 64 * g for l in range(64)] for g in range(4)])
YK = np.exp(1j * np.pi / 512.0) * W256(KMAP)  # Y_k = theta * W256^k: the point the half transforms evaluate at


def halves(p):
    """real polynomial of 1024 coefficients -> (even, odd) register images: c[n] = p[n] + i p[n + 512], n = 2 n1 + h"""
    c = p[:512] + 1j * p[512:]
    return [np.array([[c[2 * (64 * a + l) + h] for l in range(64)] for a in range(4)]) for h in range(2)]


def negacyclic(a, b):
    full = np.convolve(a, b)
    out = full[:N].copy()
    out[: N - 1] -= full[N:]
    return out


def main():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((4, 64)) + 1j * rng.standard_normal((4, 64))
    n1 = (64 * np.arange(4)[:, None] + lane[None, :])
    flat = np.zeros(256, complex)
    flat[n1] = x
    th = np.exp(1j * np.pi * np.arange(256) / 512.0)
    F = np.array([np.sum(flat * th * W256(np.arange(256) * k)) for k in range(256)])
    out = fwd256(x)
    assert np.allclose(out, F[KMAP]), "forward map"
    assert np.allclose(inv256(out), 256 * x), "inverse"
    # the product: digits (small) x key row (32-bit), both real polynomials
    d = rng.integers(-64, 64, N).astype(float)
    k = rng.integers(-2 ** 31, 2 ** 31, N).astype(float)
    A, B = (fwd256(h) for h in halves(d))
    KA, KB = (fwd256(h) for h in halves(k))
    V0 = A * KA + B * (YK * KB)      # even coefficients of the product (in Y = X^2: Y^256 = i)
    V1 = A * KB + B * KA             # odd coefficients
    got = np.zeros(N)
    for n0, V in enumerate((V0, V1)):
        g = inv256(V) / 256.0
        for a in range(4):
            for l in range(64):
                n = 2 * (64 * a + l) + n0
                got[n] = g[a][l].real
                got[n + 512] = g[a][l].imag
    want = negacyclic(d, k)
    err = np.max(np.abs(got - want))
    assert err < 0.05, err  # sums near 2^47: double spacing 2^-5 .. 2^-6 at the top, rounding recovers the integers
    print("wide12 model: forward map, inverse and even/odd product agree with the negacyclic definition (max err %.2e)" % err)


if __name__ == "__main__":
    main()
