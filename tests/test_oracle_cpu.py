"""The CPU oracle against everything that pins it without libtfhe:
its two exact polynomial back-ends against each other, libtfhe's published
constants, gate truth tables under independently generated (numpy) keys, and
the reference's plaintext semantics (process.c operands, verif.c rules,
cloud.c main() dispatch table -- SURVEY.md section 8a)."""
import numpy as np
import pytest

from np_tfhe import ToyKeys, MU


def test_exact_polymul_backends_agree(O):
    rng = np.random.default_rng(0)
    for N in (16, 64, 1024):
        small = rng.integers(-64, 64, size=N).astype(np.int32)
        big = rng.integers(-2 ** 31, 2 ** 31, size=N, dtype=np.int64).astype(np.int32)
        ntt = O.negacyclic_mul(small, big, O.POLYMUL_NTT)
        school = O.negacyclic_mul(small, big, O.POLYMUL_SCHOOLBOOK)
        assert np.array_equal(ntt, school)
        # independent numpy check of the definition
        full = np.convolve(small.astype(object), big.astype(object))
        ref = [(int(full[i]) - (int(full[i + N]) if i + N < len(full) else 0)) & 0xFFFFFFFF for i in range(N)]
        assert [int(v) & 0xFFFFFFFF for v in ntt] == ref
    # extreme magnitudes: all digits -64, all coefficients INT32_MIN
    N = 1024
    small = np.full(N, -64, dtype=np.int32)
    big = np.full(N, -2 ** 31, dtype=np.int32)
    assert np.array_equal(O.negacyclic_mul(small, big, 0), O.negacyclic_mul(small, big, 1))


def test_modswitch_constants(O):
    # libtfhe modSwitchToTorus32: 1/8, -1/8, 1/4
    assert O.modswitch_to_torus32(1, 8) == 0x20000000
    assert O.modswitch_to_torus32(-1, 8) & 0xFFFFFFFF == 0xE0000000
    assert O.modswitch_to_torus32(1, 4) == 0x40000000
    # modSwitchFromTorus32(p, 2N=2048) == (uint32)(p + 2^20) >> 21, wrapping
    rng = np.random.default_rng(1)
    for p in list(rng.integers(-2 ** 31, 2 ** 31, size=200)) + [-1, 0, (1 << 20) - 1, 1 << 20, 2 ** 31 - 1, -2 ** 31]:
        assert O.modswitch_from_torus32(int(p), 2048) == ((int(p) + (1 << 20)) & 0xFFFFFFFF) >> 21
    assert O.modswitch_from_torus32(-1, 2048) == 0  # wraps to 0, not 2048


@pytest.fixture(scope="module")
def toy(O):
    K = ToyKeys(n=6, N=64, seed=3)
    ck = O.CloudKey(K.n, K.N, K.k, K.l, K.Bgbit, K.ks_t, K.ks_basebit, K.bk, K.ksk)
    return K, ck


def test_gate_truth_tables_numpy_keys(O, toy):
    K, ck = toy
    tables = {"and": lambda a, b: a & b, "xor": lambda a, b: a ^ b, "or": lambda a, b: a | b,
              "nand": lambda a, b: 1 - (a & b)}
    for rep in range(3):
        for a in (0, 1):
            for b in (0, 1):
                ca, cb = K.encrypt_bits(a), K.encrypt_bits(b)
                for name, f in tables.items():
                    out = ck.gate(name, ca, cb)
                    ph = int(K.phase(out))
                    assert int(ph > 0) == f(a, b)
                    assert abs(abs(ph) - MU) < MU // 4  # bootstrapped output sits near +-1/8
                assert int(K.phase(ck.gate("not", ca)) > 0) == 1 - a
                assert np.array_equal(ck.gate("copy", ca), ca)
    assert np.array_equal(ck.constant(1), np.r_[np.zeros(K.n, np.int32), np.int32(MU)])
    assert np.array_equal(ck.constant(0), np.r_[np.zeros(K.n, np.int32), np.int32(-MU)])


def test_bootstrap_bits_equal_an_independent_numpy_restatement(O):
    """The C oracle's gate bootstrap against tests/np_tfhe.py's np_bootstrap -- the same spec (mod-switch,
    test-vector rotation, n CMux steps with the gadget decomposition and exact negacyclic products, sample extraction,
    key switch) written independently in numpy / Python integers: every coefficient of every output sample identical,
    for all four gate types and all input combinations, on two toy rings."""
    import np_tfhe
    for (n, N, seed) in ((8, 64, 11), (5, 128, 12)):
        K = np_tfhe.ToyKeys(n=n, N=N, seed=seed)
        ck = O.CloudKey(K.n, K.N, K.k, K.l, K.Bgbit, K.ks_t, K.ks_basebit, K.bk, K.ksk)
        a = K.encrypt_bits([0, 0, 1, 1])
        b = K.encrypt_bits([0, 1, 0, 1])
        for name, f in (("and", lambda x, y: x & y), ("xor", lambda x, y: x ^ y), ("or", lambda x, y: x | y), ("nand", lambda x, y: 1 - (x & y))):
            for i in range(4):
                ref = np_tfhe.np_gate(K, name, a[i], b[i])
                out = ck.gate(name, a[i], b[i])
                assert np.array_equal(ref, out), (n, N, name, i)
                assert K.decrypt_bits(out) == f(i >> 1, i & 1)


def test_add_circuit_bits_equal_the_independent_restatement(O):
    """orc_add (oracle/cloud_oracle.c) against np_add, a second reading of Cloud/cloud.c:18-51 on top of the independent
    numpy bootstrap: all sum samples and the carry-out identical, and they decrypt to the integer sum."""
    import np_tfhe
    K = np_tfhe.ToyKeys(n=6, N=64, seed=21)
    ck = O.CloudKey(K.n, K.N, K.k, K.l, K.Bgbit, K.ks_t, K.ks_basebit, K.bk, K.ksk)
    for (xv, yv, cv) in ((0b1011, 0b0110, 1), (0b1111, 0b1111, 0), (0, 0, 0)):
        x, y = K.encrypt_word(xv, 4), K.encrypt_word(yv, 4)
        c = K.encrypt_bits([cv])
        ref_sum, ref_carry = np_tfhe.np_add(K, x, y, c[0], 4)
        s, co = ck.add(x, y, c, 4)
        assert np.array_equal(ref_sum, s) and np.array_equal(ref_carry, co.reshape(-1))
        total = xv + yv + cv
        assert K.decrypt_word(s) == total % 16 and int(K.decrypt_bits(co.reshape(1, -1))[0]) == total >> 4


def test_mul32_circuit_bits_equal_the_independent_restatement(O):
    """orc_mul32 against np_mul32, a second reading of Cloud/cloud.c:115-218 on the independent numpy bootstrap: all 64 output
    samples identical (11 264 bootstraps on a tiny ring, ~30 s), and they decrypt to the product."""
    import np_tfhe
    K = np_tfhe.ToyKeys(n=2, N=32, seed=31)
    ck = O.CloudKey(K.n, K.N, K.k, K.l, K.Bgbit, K.ks_t, K.ks_basebit, K.bk, K.ksk)
    av, bv = 0xDEADBEEF, 0x9ABCDEF1
    a, b = K.encrypt_word(av, 32), K.encrypt_word(bv, 32)
    carry = K.encrypt_word(0, 32)
    hi, lo = np_tfhe.np_mul32(K, a, b, carry)
    r_hi, r_lo = ck.mul32(a, b, carry)
    assert np.array_equal(hi, r_hi) and np.array_equal(lo, r_lo)
    assert K.decrypt_word(lo) | (K.decrypt_word(hi) << 32) == av * bv


def test_sub32_branch_bits_equal_the_independent_restatement(O):
    """The oracle's 32-bit SUB branch against np_sub32 (cloud.c:1204-1236: bootsNOT, + 1, then the addition), a > b and a < b."""
    import np_tfhe
    K = np_tfhe.ToyKeys(n=3, N=64, seed=51)
    ck = O.CloudKey(K.n, K.N, K.k, K.l, K.Bgbit, K.ks_t, K.ks_basebit, K.bk, K.ksk)
    S = K.n + 1
    for av, bv in ((0x1234ABCD, 0x0FEDCBA9), (0x0FEDCBA9, 0x1234ABCD)):
        o1 = np.zeros((8, 32, S), np.int32)
        o2 = np.zeros((8, 32, S), np.int32)
        o1[0], o2[0] = K.encrypt_word(av, 32), K.encrypt_word(bv, 32)
        carry = K.encrypt_word(0, 32)
        ref = np_tfhe.np_sub32(K, o1[0], o2[0], carry)
        rc, out = ck.cloud_values(2, 0, 32, o1, o2, carry)
        assert rc == 0 and np.array_equal(ref, out[0])
        assert K.decrypt_word(ref) == (av - bv) % (1 << 32)


def test_mul64_branch_bits_equal_the_independent_restatement(O):
    """The oracle's 64-bit MUL branch (orc_cloud_values: two mul64 + split, 35 296 bootstraps) against np_cloud_mul64, a second
    reading of Cloud/cloud.c:65-113, 220-385 and 2589-2612: the four result words identical, decrypting to the product."""
    import np_tfhe
    K = np_tfhe.ToyKeys(n=2, N=32, seed=41)
    ck = O.CloudKey(K.n, K.N, K.k, K.l, K.Bgbit, K.ks_t, K.ks_basebit, K.bk, K.ksk)
    av, bv = 0xFEDCBA9876543210, 0x8F1E2D3C4B5A6978
    S = K.n + 1
    o1 = np.zeros((8, 32, S), np.int32)
    o2 = np.zeros((8, 32, S), np.int32)
    o1[0], o1[1] = K.encrypt_word(av & 0xFFFFFFFF, 32), K.encrypt_word(av >> 32, 32)
    o2[0], o2[1] = K.encrypt_word(bv & 0xFFFFFFFF, 32), K.encrypt_word(bv >> 32, 32)
    carry = K.encrypt_word(0, 32)
    words = np_tfhe.np_cloud_mul64(K, o1[0], o1[1], o2[0], o2[1], carry)
    rc, out = ck.cloud_values(4, 0, 64, o1, o2, carry)
    assert rc == 0
    for w in range(4):
        assert np.array_equal(words[w], out[w]), w
    assert sum(K.decrypt_word(words[w]) << (32 * w) for w in range(4)) == av * bv


def test_mul128_branch_bits_equal_the_independent_restatement(O):
    """The oracle's 128-bit MUL branch (four mul128 + fifteen chained adds, 121 184 bootstraps, the false carry dependency
    between the chains included) against np_cloud_mul128, a second reading of Cloud/cloud.c:387-647 and 2434-2492: all eight
    result words identical, decrypting to the product.  ~1 min on a tiny ring."""
    import np_tfhe
    K = np_tfhe.ToyKeys(n=2, N=32, seed=61)
    ck = O.CloudKey(K.n, K.N, K.k, K.l, K.Bgbit, K.ks_t, K.ks_basebit, K.bk, K.ksk)
    av = (1 << 127) | 0xFEDCBA98765432100123456789ABCDEF
    bv = (1 << 126) | 0x0F1E2D3C4B5A69788796A5B4C3D2E1F0
    S = K.n + 1
    o1 = np.zeros((8, 32, S), np.int32)
    o2 = np.zeros((8, 32, S), np.int32)
    for w in range(4):
        o1[w] = K.encrypt_word((av >> (32 * w)) & 0xFFFFFFFF, 32)
        o2[w] = K.encrypt_word((bv >> (32 * w)) & 0xFFFFFFFF, 32)
    carry = K.encrypt_word(0, 32)
    words = np_tfhe.np_cloud_mul128(K, o1[:4], o2[:4], carry)
    rc, out = ck.cloud_values(4, 0, 128, o1, o2, carry, threads=0)
    assert rc == 0
    for w in range(8):
        assert np.array_equal(words[w], out[w]), w
    assert sum(K.decrypt_word(words[w]) << (32 * w) for w in range(8)) == av * bv


def test_schoolbook_and_ntt_bootstrap_identical(O, toy):
    K, ck = toy
    x = K.encrypt_bits(1)
    a = ck.bootstrap(x)
    ck.set_polymul(O.POLYMUL_SCHOOLBOOK)
    b = ck.bootstrap(x)
    ck.set_polymul(O.POLYMUL_FFT)
    c = ck.bootstrap(x)
    ck.set_polymul(O.POLYMUL_NTT)
    assert np.array_equal(a, b)
    assert int(K.phase(c) > 0) == 1  # the libtfhe-style FFT path agrees at decrypt level


def test_add_matches_integers(O, toy):
    K, ck = toy
    rng = np.random.default_rng(2)
    before = ck.bootstrap_count
    for nb in (1, 4, 7):
        for _ in range(3):
            a, b, cin = int(rng.integers(0, 1 << nb)), int(rng.integers(0, 1 << nb)), int(rng.integers(0, 2))
            s, co = ck.add(K.encrypt_word(a, nb), K.encrypt_word(b, nb), K.encrypt_bits([cin]), nb)
            tot = a + b + cin
            assert K.decrypt_word(s) == tot % (1 << nb)
            assert int(K.decrypt_bits(co)[0]) == tot >> nb
    assert ck.bootstrap_count - before == 3 * 5 * (1 + 4 + 7)  # 5 bootstraps per bit (cloud.c:30-43)


def _operand(K, value):
    return np.stack([K.encrypt_word((value >> (32 * w)) & 0xFFFFFFFF) for w in range(8)])


def _value(K, out, nwords):
    return sum(K.decrypt_word(out[w]) << (32 * w) for w in range(nwords))


def test_cloud_values_32bit_all_branches(O, toy):
    """main() value circuits at 32 bit, incl. the process.c operand 2^30."""
    K, ck = toy
    carry = K.encrypt_word(0)
    A, B = 0x40000000, 0x40000000  # process.c:94-99
    for (a, b) in [(A, B), (0xDEADBEEF, 0x12345678), (5, 9)]:
        o1, o2 = _operand(K, a), _operand(K, b)
        rc, out = ck.cloud_values(1, 0, 32, o1, o2, carry)  # A+B
        assert rc == 0 and _value(K, out, 1) == (a + b) & 0xFFFFFFFF
        rc, out = ck.cloud_values(2, 0, 32, o1, o2, carry)  # A-B
        assert rc == 0 and _value(K, out, 1) == (a - b) & 0xFFFFFFFF
        rc, out = ck.cloud_values(1, 1, 32, o1, o2, carry)  # (-A)+B
        assert rc == 0 and _value(K, out, 1) == (b - a) & 0xFFFFFFFF
        rc, out = ck.cloud_values(2, 1, 32, o1, o2, carry)  # (-A)-B -> magnitude A+B
        assert rc == 0 and _value(K, out, 1) == (a + b) & 0xFFFFFFFF
        # unused answer words are operand 1's carry word (cloud.c:901-916)
        assert np.array_equal(out[1], carry) and np.array_equal(out[8], carry)
    rc, out = ck.cloud_values(4, 0, 32, _operand(K, A), _operand(K, B), carry)
    assert rc == 0 and _value(K, out, 2) == 1 << 60  # words {0, 0x10000000}
    assert K.decrypt_word(out[0]) == 0 and K.decrypt_word(out[1]) == 0x10000000
    assert np.array_equal(out[2], carry)


def test_cloud_values_64bit(O, toy):
    K, ck = toy
    carry = K.encrypt_word(0)
    a, b = 0xFEDCBA9876543210, 0x0123456789ABCDEF
    o1, o2 = _operand(K, a), _operand(K, b)
    rc, out = ck.cloud_values(1, 0, 64, o1, o2, carry)
    assert rc == 0 and _value(K, out, 2) == (a + b) & (2 ** 64 - 1)
    rc, out = ck.cloud_values(2, 0, 64, o1, o2, carry)
    assert rc == 0 and _value(K, out, 2) == (a - b) & (2 ** 64 - 1)
    before = ck.bootstrap_count
    A = 1 << 62  # process.c:122-129
    rc, out = ck.cloud_values(4, 0, 64, _operand(K, A), _operand(K, A), carry)
    assert rc == 0 and _value(K, out, 4) == 1 << 124
    assert ck.bootstrap_count - before == 35296
    assert ck.cloud_values(4, 0, 256, o1, o2, carry)[0] == 126  # cloud.c:860-864
    assert ck.cloud_values(1, 0, 48, o1, o2, carry)[0] == -1  # no branch of main() matches


def test_deferred_level_parallel_mode_is_bit_identical(O, toy):
    """orc_defer_*: the oracle's own sequential gate stream, recorded and evaluated level by level on
    several threads, leaves exactly the bits of the immediate run (it is how the n=630 goldens are made)."""
    K, ck = toy
    carry = K.encrypt_word(0)
    o1, o2 = _operand(K, 0xFEDCBA9876543210), _operand(K, 0x0F1E2D3C4B5A6978)
    for op, neg, bits in ((1, 0, 64), (2, 0, 32), (1, 1, 64), (1, 2, 32), (4, 0, 32)):
        before = ck.bootstrap_count
        rc1, seq = ck.cloud_values(op, neg, bits, o1, o2, carry)
        n_seq = ck.bootstrap_count - before
        before = ck.bootstrap_count
        rc2, par = ck.cloud_values(op, neg, bits, o1, o2, carry, threads=3)
        assert rc1 == rc2 == 0 and np.array_equal(seq, par), (op, neg, bits)
        assert ck.bootstrap_count - before == n_seq
    # independent gates in one batch
    a = np.stack([K.encrypt_bits(v) for v in (0, 0, 1, 1, 1)])
    b = np.stack([K.encrypt_bits(v) for v in (0, 1, 0, 1, 1)])
    for name in ("and", "xor", "or", "nand"):
        assert np.array_equal(ck.gates_batch(name, a, b, threads=2), np.stack([ck.gate(name, a[i], b[i]) for i in range(5)]))
    assert O.max_threads() >= 1


def test_mux_gate(O, toy):
    """bootsMUX(a,b,c) = a ? b : c: two bootstraps without key switch, one key switch (boot-gates.cpp)."""
    K, ck = toy
    for a in (0, 1):
        for b in (0, 1):
            for c in (0, 1):
                ca, cb, cc = K.encrypt_bits(a), K.encrypt_bits(b), K.encrypt_bits(c)
                out = ck.mux(ca, cb, cc)
                ph = int(K.phase(out))
                assert (ph > 0) == bool(b if a else c), (a, b, c)
                assert abs(abs(ph) - MU) < MU // 2
                # = keyswitch((0,1/8) + woKS(AND(a,b)) + woKS(AND(NOT a, c)))
                t1 = (ca.astype(np.int64) + cb).astype(np.int32)
                t1[-1] = np.int32((int(t1[-1]) - MU + 2 ** 31) % 2 ** 32 - 2 ** 31)
                t2 = (cc.astype(np.int64) - ca).astype(np.int32)
                t2[-1] = np.int32((int(t2[-1]) - MU + 2 ** 31) % 2 ** 32 - 2 ** 31)
                u = (ck.bootstrap_woks(t1).astype(np.int64) + ck.bootstrap_woks(t2)).astype(np.int32)
                u[-1] = np.int32((int(u[-1]) + MU + 2 ** 31) % 2 ** 32 - 2 ** 31)
                assert np.array_equal(ck.keyswitch(u), out)


def test_metadata_dispatch_table(O):
    """SURVEY section 8a truth table (cloud.c:787-864)."""
    # (op, neg1, neg2) -> (code written, routing neg)
    for neg1, neg2, code, neg in [(0, 0, 0, 0), (2, 0, 1, 1), (0, 2, 2, 2), (2, 2, 4, 3), (1, 2, 4, 3)]:
        rc, c, bit, r, ib = O.cloud_metadata(1, neg1, 32, neg2, 64)
        assert (rc, c, r, bit, ib) == (0, code, neg, 64, 64)
    rc, c, bit, r, ib = O.cloud_metadata(4, 0, 64, 0, 32)
    assert (rc, bit, ib) == (0, 128, 64)  # MUL advertises 2*max(bit)
    assert O.cloud_metadata(4, 0, 256, 0, 32)[0] == 126
    assert O.cloud_metadata(2, 0, 256, 0, 32)[0] == 0
    # latent reference quirk: a stage-1 code 4 is NOT remapped in stage 2
    assert O.cloud_metadata(1, 4, 32, 0, 32)[1:4:2] == (0, 4)
