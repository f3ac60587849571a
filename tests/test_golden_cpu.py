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
