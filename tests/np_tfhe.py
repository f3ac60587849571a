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


# ---- an independent restatement of the gate bootstrap itself (written from the spec alone) ----
# numpy int64 / Python integers throughout, exact negacyclic products by np.convolve; shares no code with oracle/ or the
# product.  Slow (toy rings only): it exists so that the C oracle's BITS -- not just its decryptions -- have a second witness.

def _u32(x):
    return np.asarray(x, dtype=np.int64) & 0xFFFFFFFF


def _negacyclic(small, big):
    """small (int64, |.| < 2^8) * big (int32 torus) mod X^N+1 -> int32 with wraparound."""
    N = small.shape[0]
    full = np.convolve(small.astype(np.int64), big.astype(np.int64))  # |digit| <= 2^9, |key| < 2^31, N <= 2^10 terms: < 2^51, exact in int64
    res = full[:N].copy()
    res[:N - 1] -= full[N:]
    return _wrap32(res)


def _mul_by_xai(p, a):
    """X^a * p mod X^N+1 for a in [0, 2N)."""
    N = p.shape[0]
    p = p.astype(np.int64)
    out = np.zeros(N, dtype=np.int64)
    if a < N:
        out[:a] = -p[N - a:] if a else out[:0]
        out[a:] = p[:N - a]
    else:
        aa = a - N
        out[:aa] = p[N - aa:] if aa else out[:0]
        out[aa:] = -p[:N - aa]
    return _wrap32(out)


def np_modswitch(phase, log2_2N):
    return int(((int(phase) & 0xFFFFFFFF) + (1 << (31 - log2_2N))) & 0xFFFFFFFF) >> (32 - log2_2N)


def np_bootstrap(K, x):
    """tfhe_bootstrap_FFT(MU) of the LWE sample x under the ToyKeys-style material K (exact arithmetic) -> LWE sample."""
    n, N, l, Bgbit = K.n, K.N, K.l, K.Bgbit
    log2_2N = (2 * N).bit_length() - 1
    barb = np_modswitch(x[n], log2_2N)
    bara = [np_modswitch(x[i], log2_2N) for i in range(n)]
    acc = [np.zeros(N, dtype=np.int32), _mul_by_xai(np.full(N, MU, dtype=np.int32), (2 * N - barb) % (2 * N))]
    Bg, half = 1 << Bgbit, 1 << (Bgbit - 1)
    offset = sum(half << (32 - p * Bgbit) for p in range(1, l + 1)) & 0xFFFFFFFF
    for i in range(n):
        if bara[i] == 0:
            continue
        tmp = [_wrap32(_mul_by_xai(acc[u], bara[i]).astype(np.int64) - acc[u].astype(np.int64)) for u in range(2)]
        rows = []
        for u in range(2):
            w = (_u32(tmp[u]) + offset) & 0xFFFFFFFF
            for p in range(1, l + 1):
                rows.append(((w >> (32 - p * Bgbit)) & (Bg - 1)) - half)
        prod = [np.zeros(N, dtype=np.int64), np.zeros(N, dtype=np.int64)]
        for row, dec in enumerate(rows):
            for c in range(2):
                prod[c] += _negacyclic(dec.astype(np.int64), K.bk[i, row, c]).astype(np.int64)
        acc = [_wrap32(acc[c].astype(np.int64) + prod[c]) for c in range(2)]
    u = np.zeros(N + 1, dtype=np.int32)
    u[0] = acc[0][0]
    u[1:N] = _wrap32(-acc[0][N - 1:0:-1].astype(np.int64))
    u[N] = acc[1][0]
    # lweKeySwitch
    t, basebit = K.ks_t, K.ks_basebit
    base, prec = 1 << basebit, 1 << (32 - (1 + basebit * t))
    r = np.zeros(n + 1, dtype=np.int64)
    r[n] = int(u[N])
    for i in range(N):
        abar = (int(u[i]) + prec) & 0xFFFFFFFF
        for j in range(t):
            d = (abar >> (32 - (j + 1) * basebit)) & (base - 1)
            if d:
                r -= K.ksk[i, j, d].astype(np.int64)
    return _wrap32(r)


def np_gate(K, name, ca, cb):
    ca, cb = ca.astype(np.int64), cb.astype(np.int64)
    if name == "and":
        t = ca + cb
        t[K.n] -= 1 << 29
    elif name == "xor":
        t = 2 * (ca + cb)
        t[K.n] += 1 << 30
    elif name == "or":
        t = ca + cb
        t[K.n] += 1 << 29
    else:  # nand
        t = -ca - cb
        t[K.n] += 1 << 29
    return np_bootstrap(K, _wrap32(t))


def np_add(K, x, y, c, nb_bits):
    """Cloud/cloud.c:18-51 `add`, gate by gate on np_gate: per bit axc = x ^ carry, bxc = y ^ carry, sum = x ^ bxc,
    axc = axc & bxc, carry = carry ^ axc; the final carry is returned beside the sum.  (A second reading of those lines,
    independent of oracle/cloud_oracle.c.)"""
    carry = c.copy()
    out = np.zeros((nb_bits, K.n + 1), dtype=np.int32)
    for i in range(nb_bits):
        axc = np_gate(K, "xor", x[i], carry)
        bxc = np_gate(K, "xor", y[i], carry)
        out[i] = np_gate(K, "xor", x[i], bxc)
        axc = np_gate(K, "and", axc, bxc)
        carry = np_gate(K, "xor", carry, axc)
    return out, carry


def np_mul32(K, a, b, carry):
    """Cloud/cloud.c:115-218 `mul32` (nb_bits = 32), gate by gate: for every bit i of b the row a AND b_i, shifted left by i
    across two 32-bit words (zeros in front are bootsCONSTANT(0) = the noiseless (0, -1/8)), accumulated with `add` -- the low
    word with carry-in `carry`, the high word with the low word's carry-out.  Returns (high word, low word) like result /
    result2.  An independent reading, sample for sample, of what oracle/cloud_oracle.c restates."""
    n = K.n
    zero = np.zeros(n + 1, dtype=np.int32)
    zero[n] = -MU
    lo = np.tile(zero, (32, 1))
    hi = np.tile(zero, (32, 1))
    t2 = np.tile(zero, (32, 1))          # tmp3c2 keeps what earlier rounds left beyond `round`: the initial constants
    for i in range(32):
        tmp = np.stack([np_gate(K, "and", a[k], b[i]) for k in range(32)])
        t1 = np.tile(zero, (32, 1))
        t1[i:] = tmp[:32 - i]
        t2[:i] = tmp[32 - i:]
        lo, c1 = np_add(K, lo, t1, carry[0], 32)
        hi, _ = np_add(K, hi, t2, c1, 32)
    return hi, lo


def _const0(K):
    z = np.zeros(K.n + 1, dtype=np.int32)
    z[K.n] = -MU
    return z


def np_mul64(K, a, b, c, carry):
    """Cloud/cloud.c:220-385 `mul64`: the 64-bit number (b : a) times the 32-bit word c -> (top, mid, low) words.  Per bit i of
    c the rows a AND c_i and b AND c_i, shifted left by i across three words, accumulated with three chained `add`s."""
    zero = _const0(K)
    s1, s2, s3 = (np.tile(zero, (32, 1)) for _ in range(3))
    t3 = np.tile(zero, (32, 1))           # tmp3c3: entries beyond `round` keep their initial constants
    for i in range(32):
        tmp = np.stack([np_gate(K, "and", a[k], c[i]) for k in range(32)])
        tmp2 = np.stack([np_gate(K, "and", b[k], c[i]) for k in range(32)])
        c1, c2 = 32 - i, i
        t1 = np.tile(zero, (32, 1))
        t1[i:] = tmp[:c1]
        t2 = np.concatenate([tmp[c1:], tmp2[:c1]])   # the rest of tmp, then what fits of tmp2
        t3[:c2] = tmp2[c1:]
        s1, k1 = np_add(K, s1, t1, carry[0], 32)
        s2, k2 = np_add(K, s2, t2, k1, 32)
        s3, _ = np_add(K, s3, t3, k2, 32)
    return s3, s2, s1


def np_cloud_mul64(K, w1lo, w1hi, w2lo, w2hi, carry):
    """main()'s 64-bit MUL branch (cloud.c:2589-2612): two mul64 and `split` (cloud.c:65-113) -> four words, LSW first.
    split's third add takes the ARRAY carryover2 as an operand: its element 0 is the second add's carry-out, the other 31
    are still the constants they were initialised to."""
    r1, r2, r3 = np_mul64(K, w1lo, w1hi, w2lo, carry)
    r4, r5, r6 = np_mul64(K, w1lo, w1hi, w2hi, carry)
    s, co = np_add(K, r6, r2, carry[0], 32)
    s2, co2 = np_add(K, r5, r1, co, 32)
    carry_array = np.tile(_const0(K), (32, 1))
    carry_array[0] = co2
    s3, _ = np_add(K, r4, carry_array, carry[0], 32)
    return r3, s, s2, s3


def np_sub32(K, a, b, carry):
    """main()'s 32-bit SUB branch (cloud.c:1204-1236): NOT of the second operand (bootsNOT: the sample negated), + 1 through
    `add` against [bootsCONSTANT(1), 31 x bootsCONSTANT(0)] with a constant-0 carry-in, then a + that with operand 1's carry word."""
    inverse = _wrap32(-b.astype(np.int64))
    one = np.tile(_const0(K), (32, 1))
    one[0, K.n] = MU
    twos, _ = np_add(K, inverse, one, _const0(K), 32)
    res, _ = np_add(K, a, twos, carry[0], 32)
    return res


def np_mul128(K, a, b, c, d, e, carry):
    """Cloud/cloud.c:387-647 `mul128`: the 128-bit number (d : c : b : a) times the 32-bit word e -> five words, TOP first
    (result .. result5).  Per bit i of e four AND rows, shifted left by i across five words, five chained `add`s."""
    zero = _const0(K)
    sums = [np.tile(zero, (32, 1)) for _ in range(5)]
    t5 = np.tile(zero, (32, 1))           # tmp3c5: entries beyond `round` keep their initial constants
    for i in range(32):
        rows = [np.stack([np_gate(K, "and", w[k], e[i]) for k in range(32)]) for w in (a, b, c, d)]
        c1 = 32 - i
        t = [np.tile(zero, (32, 1))]
        t[0][i:] = rows[0][:c1]
        for q in range(1, 4):
            t.append(np.concatenate([rows[q - 1][c1:], rows[q][:c1]]))
        t5[:i] = rows[3][c1:]
        t.append(t5)
        cin = carry[0]
        for q in range(5):
            sums[q], cin = np_add(K, sums[q], t[q], cin, 32)
    return sums[4], sums[3], sums[2], sums[1], sums[0]


def np_cloud_mul128(K, w1, w2, carry):
    """main()'s 128-bit MUL branch (cloud.c:2434-2492): four mul128 (operand 1 times each word of operand 2) and fifteen chained
    adds -- the second and third chains take their carry-in from the TOP carry-out of the chain before (carryover5 -> sum6,
    carryover10 -> sum11), and the chains' last adds take operand 1's carry word as an operand -> eight words, LSW first."""
    r = {}
    for q in range(4):
        top = np_mul128(K, w1[0], w1[1], w1[2], w1[3], w2[q], carry)
        for j in range(5):
            r[5 * q + j + 1] = top[j]      # result1..5, result6..10, ...
    s, co = {}, {}
    s[1], co[1] = np_add(K, r[10], r[4], carry[0], 32)
    s[2], co[2] = np_add(K, r[9], r[3], co[1], 32)
    s[3], co[3] = np_add(K, r[8], r[2], co[2], 32)
    s[4], co[4] = np_add(K, r[7], r[1], co[3], 32)
    s[5], co[5] = np_add(K, r[6], carry, co[4], 32)
    s[6], co[6] = np_add(K, s[2], r[15], co[5], 32)
    s[7], co[7] = np_add(K, s[3], r[14], co[6], 32)
    s[8], co[8] = np_add(K, s[4], r[13], co[7], 32)
    s[9], co[9] = np_add(K, s[5], r[12], co[8], 32)
    s[10], co[10] = np_add(K, r[11], carry, co[9], 32)
    s[11], co[11] = np_add(K, s[7], r[20], co[10], 32)
    s[12], co[12] = np_add(K, s[8], r[19], co[11], 32)
    s[13], co[13] = np_add(K, s[9], r[18], co[12], 32)
    s[14], co[14] = np_add(K, s[10], r[17], co[13], 32)
    s[15], co[15] = np_add(K, r[16], carry, co[14], 32)
    return r[5], s[1], s[6], s[11], s[12], s[13], s[14], s[15]
