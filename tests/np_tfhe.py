"""Independent numpy key generation / encryption / decryption for tests.

Test support only.  Mirrors what the reference's Keygen/keygen.c,
Client1/alice.c:116-149 and Output/verif.c:41-76 obtain from libtfhe
(SURVEY.md App. A), written independently of both the product's C++ tools and
the oracle so the three can be checked against each other.
"""
import numpy as np

MU = 1 << 29  # 1/8 on the 32-bit torus


def _wrap32(x):
    return (np.asarray(x, dtype=np.int64) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)


def gaussian32(rng, sigma, size=None):
    d = rng.normal(0.0, sigma, size=size)
    return _wrap32(np.rint((d - np.rint(d)) * 4294967296.0).astype(np.int64))


def uniform32(rng, size):
    return rng.integers(-(1 << 31), 1 << 31, size=size, dtype=np.int64).astype(np.int32)


def negacyclic_mul_binary(a, s):
    """a (int32 torus poly) * s (0/1 poly) mod X^N+1, exact, wrapped to int32."""
    N = a.shape[0]
    full = np.convolve(a.astype(np.int64), s.astype(np.int64))
    res = full[:N].copy()
    res[: N - 1] -= full[N:]
    return _wrap32(res)


class ToyKeys:
    """Secret + cloud key material as raw arrays."""

    def __init__(self, n=16, N=1024, k=1, l=3, Bgbit=7, ks_t=8, ks_basebit=2,
                 lwe_sigma=2.0 ** -15, bk_sigma=2.0 ** -25, seed=1):
        assert k == 1
        rng = np.random.default_rng(seed)
        self.rng = rng
        self.n, self.N, self.k, self.l, self.Bgbit = n, N, k, l, Bgbit
        self.ks_t, self.ks_basebit = ks_t, ks_basebit
        self.lwe_sigma = lwe_sigma
        self.lwe_key = rng.integers(0, 2, size=n).astype(np.int32)
        self.tlwe_key = rng.integers(0, 2, size=N).astype(np.int32)
        kpl = (k + 1) * l
        # BK_i = TGSW(s_i): rows of TLWE(0) plus s_i * 2^(32-(p+1)Bgbit) on the
        # constant coefficient of poly `bloc` in row bloc*l+p
        bk = np.zeros((n, kpl, k + 1, N), dtype=np.int32)
        for i in range(n):
            for row in range(kpl):
                a = uniform32(rng, N)
                e = gaussian32(rng, bk_sigma, N)
                b = _wrap32(negacyclic_mul_binary(a, self.tlwe_key).astype(np.int64) + e)
                bk[i, row, 0] = a
                bk[i, row, 1] = b
            for bloc in range(k + 1):
                for p in range(l):
                    h = 1 << (32 - (p + 1) * Bgbit)
                    row = bloc * l + p
                    bk[i, row, bloc, 0] = _wrap32(int(bk[i, row, bloc, 0]) + int(self.lwe_key[i]) * h)
        self.bk = bk
        base = 1 << ks_basebit
        ksk = np.zeros((k * N, ks_t, base, n + 1), dtype=np.int32)
        a = uniform32(rng, (k * N, ks_t, base, n))
        e = gaussian32(rng, lwe_sigma, (k * N, ks_t, base))
        for i in range(k * N):
            for j in range(ks_t):
                for d in range(1, base):
                    msg = int(self.tlwe_key[i]) * d * (1 << (32 - (j + 1) * ks_basebit))
                    dot = int(np.dot(a[i, j, d].astype(np.int64), self.lwe_key.astype(np.int64)))
                    ksk[i, j, d, :n] = a[i, j, d]
                    ksk[i, j, d, n] = _wrap32(dot + msg + int(e[i, j, d]))
        self.ksk = ksk

    def encrypt_bits(self, bits, key=None, sigma=None):
        key = self.lwe_key if key is None else key
        sigma = self.lwe_sigma if sigma is None else sigma
        bits = np.asarray(bits).astype(np.int64)
        n = key.shape[0]
        out = np.zeros(bits.shape + (n + 1,), dtype=np.int32)
        a = uniform32(self.rng, bits.shape + (n,))
        e = gaussian32(self.rng, sigma, bits.shape).astype(np.int64)
        dot = (a.astype(np.int64) * key.astype(np.int64)).sum(-1)
        out[..., :n] = a
        out[..., n] = _wrap32(dot + np.where(bits > 0, MU, -MU) + e)
        return out

    def phase(self, samples, key=None):
        key = self.lwe_key if key is None else key
        n = key.shape[0]
        s = np.asarray(samples, dtype=np.int32)
        dot = (s[..., :n].astype(np.int64) * key.astype(np.int64)).sum(-1)
        return _wrap32(s[..., n].astype(np.int64) - dot)

    def decrypt_bits(self, samples, key=None):
        return (self.phase(samples, key) > 0).astype(np.int64)

    def encrypt_word(self, value, nbits=32):
        return self.encrypt_bits([(int(value) >> i) & 1 for i in range(nbits)])

    def decrypt_word(self, samples):
        bits = self.decrypt_bits(samples)
        return sum(int(b) << i for i, b in enumerate(bits))
