"""Oracle vs the committed golden vectors (tests/golden/, made by make_golden.py)."""
import hashlib
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def toy_golden(O):
    z = np.load(os.path.join(G, "toy_vectors.npz"))
    n, N, k, l, Bgbit, t, bb = (int(v) for v in z["params"])
    return z, O.CloudKey(n, N, k, l, Bgbit, t, bb, z["bk"], z["ksk"])


def test_oracle_reproduces_toy_gates(O, toy_golden):
    z, ck = toy_golden
    for g in ("and", "xor", "or", "nand"):
        for i in range(4):
            assert np.array_equal(ck.gate(g, z["ca"][i], z["cb"][i]), z["gate_" + g][i]), (g, i)
    # and they decrypt to the truth tables
    key = z["lwe_key"].astype(np.int64)
    def dec(s):
        ph = (s[..., -1].astype(np.int64) - (s[..., :-1].astype(np.int64) * key).sum(-1)) & 0xFFFFFFFF
        return (np.where(ph >= 2 ** 31, ph - 2 ** 32, ph) > 0).astype(np.uint8)
    a, b = z["a_bits"], z["b_bits"]
    assert np.array_equal(dec(z["gate_and"]), a & b) and np.array_equal(dec(z["gate_xor"]), a ^ b)
    assert np.array_equal(dec(z["gate_or"]), a | b) and np.array_equal(dec(z["gate_nand"]), 1 - (a & b))


def test_oracle_reproduces_toy_stages(O, toy_golden):
    z, ck = toy_golden
    bara, barb = ck.modswitch(z["ca"][3])
    assert np.array_equal(bara, z["bara"]) and barb == int(z["barb"])
    acc0 = ck.blind_rotate_init(barb)
    assert np.array_equal(acc0, z["acc0"])
    assert np.array_equal(ck.blind_rotate_step(acc0, 0, bara[0]), z["acc1"])
    accn = ck.blind_rotate(acc0, bara)
    assert np.array_equal(accn, z["accn"])
    u = ck.sample_extract(accn)
    assert np.array_equal(u, z["extracted"])
    assert np.array_equal(ck.keyswitch(u), z["keyswitched"])
    # schoolbook back-end gives the same bits
    ck.set_polymul(O.POLYMUL_SCHOOLBOOK)
    assert np.array_equal(ck.blind_rotate(acc0, bara), z["accn"])
    ck.set_polymul(O.POLYMUL_NTT)


def test_oracle_reproduces_toy_add(O, toy_golden):
    z, ck = toy_golden
    s, co = ck.add(z["add_x"], z["add_y"], z["add_c"], 4)
    assert np.array_equal(s, z["add_sum"]) and np.array_equal(co, z["add_carry"])


def test_oracle_reproduces_full_size_kat(ia, O):
    """n=630, N=1024 key regenerated from the keygen.c seed words; 4 gates."""
    from ieache_amd import tools
    z = np.load(os.path.join(G, "full_gate_kat.npz"))
    p = ia.default_params()
    k = tools.keygen_raw(p, tuple(int(v) for v in z["seed"]))
    h = hashlib.sha256()
    for a in (k["lwe_key"], k["bk"], k["ksk"]):
        h.update(np.ascontiguousarray(a).tobytes())
    assert h.hexdigest() == str(z["key_sha256"]), "product keygen is no longer deterministic for this seed"
    ck = O.CloudKey(p.n, p.N, p.k, p.l, p.Bgbit, p.ks_t, p.ks_basebit, k["bk"], k["ksk"])
    for i in range(2):
        assert np.array_equal(ck.gate("and", z["ca"][i], z["cb"][i]), z["gate_and"][i])
        assert np.array_equal(ck.gate("xor", z["ca"][i], z["cb"][i]), z["gate_xor"][i])
    assert list(tools.decrypt_bits(p, k["lwe_key"], z["gate_and"])) == [1, 0]
    assert list(tools.decrypt_bits(p, k["lwe_key"], z["gate_xor"])) == [0, 1]


def predicted_gate_output_noise(p, tlwe_key_ones=None):
    """(variance of a bootstrapped gate's output phase about its mean over many gates under ONE key, standard deviation of that
    mean over keys), as the published analysis of the algorithm gives them (CGGI: Chillotti, Gama, Georgieva, Izabachene,
    "TFHE: fast fully homomorphic encryption over the torus": external product / CMux noise, key-switch noise), average case,
    for libtfhe's conventions:
      blind rotation   n CMux steps, each adds over the (k+1) l digit rows (digit polynomial x the row's noise polynomial):
                       n (k+1) l N E[d^2] sigma_bk^2, digits uniform on [-Bg/2, Bg/2): E[d^2] = (Bg^2 + 2) / 12
      decomposition    libtfhe's gadget decomposition TRUNCATES (its offset has no half unit of the last digit): every coefficient
                       of (X^a - 1) acc is short by eps in [0, Bg^-l), mean m = Bg^-l / 2, and through the h ones of the ring key
                       those means add up coherently: a step with s_i = 1 shifts the phase by m (2 c(j) - h - 1) at coefficient j
                       (c(j) = ones among the key's first j + 1 coefficients), i.e. by something uniform on [-h m, h m] at the
                       coefficient that ends up extracted: (n / 2) (h m)^2 / 3.  (A decomposition that rounded would not have this
                       term -- 15 % of the total at the reference's parameters, so the samples tell the two apart.)
      key switch       for each of the N t (coefficient, digit position) pairs one of `base` key-switch-key samples, the one for
                       digit 0 being exactly zero: the noise values are part of the KEY, so their average over the digit is a
                       constant offset of every output under that key (variance N t (base-1)/base^2 sigma_ks^2 over keys) and only
                       the rest varies from gate to gate: N t ((base-1)/base)^2 sigma_ks^2
    The output noise does not depend on the inputs' noise -- that is what bootstrapping is for."""
    Bg, base = 1 << p.Bgbit, 1 << p.ks_basebit
    h = p.N // 2 if tlwe_key_ones is None else int(tlwe_key_ones)
    br = p.n * (p.k + 1) * p.l * p.N * (Bg * Bg + 2) / 12.0 * p.tlwe_alpha_min ** 2
    m = 0.5 * 2.0 ** (-p.l * p.Bgbit)
    trunc = 0.5 * p.n * (h * m) ** 2 / 3.0
    ks = p.N * p.k * p.ks_t * ((base - 1.0) / base) ** 2 * p.lwe_alpha_min ** 2
    offset_sd = (p.N * p.k * p.ks_t * (base - 1.0) / base ** 2) ** 0.5 * p.lwe_alpha_min
    return br + trunc + ks, offset_sd


def phase_errors(p, lwe_key, samples, bits):
    """phase - (+-1/8) of each sample, as a fraction of the torus"""
    s = np.asarray(lwe_key[: p.n], dtype=np.int64)
    a = samples[:, : p.n].astype(np.int64)
    ph = (samples[:, p.n].astype(np.int64) - a @ s) & 0xFFFFFFFF
    mu = np.where(np.asarray(bits) != 0, 1 << 29, (1 << 32) - (1 << 29))
    e = (ph - mu) & 0xFFFFFFFF
    e = np.where(e >= 1 << 31, e - (1 << 32), e)
    return e / 2.0 ** 32


def test_oracle_output_noise_matches_the_published_variance(ia, O):
    """A pin on the ALGORITHM that needs no libtfhe binary: the oracle's bootstrapped outputs at the reference's parameters carry
    the noise the published analysis predicts for them (a decomposition off by a digit, a missing row, a wrong gadget or
    key-switch base would all move it).  64 gates here -- the sample variance of 64 values is within +-45 % of the truth at
    3 sigma --; the GPU suite does the same on 16 384 gates to +-5 % (tests/test_gpu_parity.py)."""
    from ieache_amd import tools
    p = ia.default_params()
    k = tools.keygen_raw(p, (27, 18, 28))
    ck = O.CloudKey(p.n, p.N, p.k, p.l, p.Bgbit, p.ks_t, p.ks_basebit, k["bk"], k["ksk"])
    rng = np.random.default_rng(7)
    bits = rng.integers(0, 2, size=(2, 64)).astype(np.uint8)
    a, b = tools.encrypt_bits(p, k["lwe_key"], bits[0], 5), tools.encrypt_bits(p, k["lwe_key"], bits[1], 6)
    out = ck.gates_batch("xor", a, b, threads=0)
    e = phase_errors(p, k["lwe_key"], out, bits[0] ^ bits[1])
    var, offset_sd = predicted_gate_output_noise(p, np.sum(k["tlwe_key"]))
    assert abs(var - 1.05e-5) < 0.03e-5           # 4.69e-6 blind rotation + 1.5e-6 truncation + 4.29e-6 key switch
    assert abs(offset_sd - 1.2e-3) < 0.05e-3
    assert np.all(np.abs(e) < 1.0 / 16)           # every output decrypts with margin
    assert 0.55 * var < np.var(e) < 1.6 * var, (np.var(e), var)
    assert abs(np.mean(e)) < 4 * offset_sd + 4 * np.sqrt(var / 64)
