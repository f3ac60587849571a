"""Regenerates the committed golden vectors under tests/golden/.

The reference holds no ciphertext-level vectors (no tests directory, libtfhe
not vendored), so these are produced HERE by the exact-integer CPU oracle
(oracle/) on keys from the product's keygen tool; they pin both the oracle and
the HIP path against regressions.  Plaintext-level vectors come from the
reference's own canned operands (Client1/process.c:94-99,122-129,152-163,
185-204: value = 2^(bits-2), sign code 0 or 2).

Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import ieache_amd as ia  # noqa: E402
from ieache_amd import tools  # noqa: E402
from oracle import oracle as O  # noqa: E402


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def toy_vectors():
    p = ia.default_params().copy(n=6, N=64)
    seed = (2024, 10, 3)
    k = tools.keygen_raw(p, seed)
    ck = O.CloudKey(p.n, p.N, p.k, p.l, p.Bgbit, p.ks_t, p.ks_basebit, k["bk"], k["ksk"])
    a_bits = np.array([0, 0, 1, 1], dtype=np.uint8)
    b_bits = np.array([0, 1, 0, 1], dtype=np.uint8)
    ca = tools.encrypt_bits(p, k["lwe_key"], a_bits, 1001)
    cb = tools.encrypt_bits(p, k["lwe_key"], b_bits, 1002)
    out = {"params": np.array([p.n, p.N, p.k, p.l, p.Bgbit, p.ks_t, p.ks_basebit], dtype=np.int32),
           "seed": np.array(seed, dtype=np.uint32), "lwe_key": k["lwe_key"], "bk": k["bk"], "ksk": k["ksk"],
           "ca": ca, "cb": cb, "a_bits": a_bits, "b_bits": b_bits}
    for g in ("and", "xor", "or", "nand"):
        out["gate_" + g] = np.stack([ck.gate(g, ca[i], cb[i]) for i in range(4)])
    # stages of one bootstrap
    bara, barb = ck.modswitch(ca[3])
    acc0 = ck.blind_rotate_init(barb)
    acc1 = ck.blind_rotate_step(acc0, 0, bara[0])
    accn = ck.blind_rotate(acc0, bara)
    u = ck.sample_extract(accn)
    out.update(bara=bara, barb=np.int32(barb), acc0=acc0, acc1=acc1, accn=accn, extracted=u,
               keyswitched=ck.keyswitch(u))
    # add(nb_bits=4): x=0b1011, y=0b0110, carry-in 1
    x = tools.encrypt_bits(p, k["lwe_key"], tools.int_to_bits(0b1011, 4), 1003)
    y = tools.encrypt_bits(p, k["lwe_key"], tools.int_to_bits(0b0110, 4), 1004)
    c = tools.encrypt_bits(p, k["lwe_key"], np.array([1], dtype=np.uint8), 1005)
    s, co = ck.add(x, y, c, 4)
    out.update(add_x=x, add_y=y, add_c=c, add_sum=s, add_carry=co)
    np.savez_compressed(os.path.join(HERE, "toy_vectors.npz"), **out)
    print("toy_vectors.npz written")


def full_size_kat():
    p = ia.default_params()
    seed = (314, 1592, 657)  # Keygen/keygen.c:30
    k = tools.keygen_raw(p, seed)
    ck = O.CloudKey(p.n, p.N, p.k, p.l, p.Bgbit, p.ks_t, p.ks_basebit, k["bk"], k["ksk"])
    ca = tools.encrypt_bits(p, k["lwe_key"], np.array([1, 0], dtype=np.uint8), 2001)
    cb = tools.encrypt_bits(p, k["lwe_key"], np.array([1, 1], dtype=np.uint8), 2002)
    np.savez_compressed(os.path.join(HERE, "full_gate_kat.npz"),
                        seed=np.array(seed, dtype=np.uint32), key_sha256=np.array(digest(k["lwe_key"], k["bk"], k["ksk"])),
                        lwe_key=k["lwe_key"], ca=ca, cb=cb,
                        gate_and=np.stack([ck.gate("and", ca[i], cb[i]) for i in range(2)]),
                        gate_xor=np.stack([ck.gate("xor", ca[i], cb[i]) for i in range(2)]))
    print("full_gate_kat.npz written")


def mul32_n630():
    """BASELINE configs[2]'s circuit at the product parameter set: the oracle's own sequential
    mul32 (oracle/cloud_oracle.c, cloud.c:115-218, 2655-2718) on ONE expression, 11 264 exact
    bootstraps at n=630.  The gate stream is recorded and its independent gates evaluated on
    all host cores (oracle/tfhe_oracle.c orc_defer_*; bit-identical to the sequential run, which
    takes ~45 min).  Committed: operands, seeds, sha256 of the 64 output samples, and the first
    and last sample in full; the key is regenerated from its seed (114 MB)."""
    import time
    p = ia.default_params()
    seed = (314, 1592, 657)
    k = tools.keygen_raw(p, seed)
    ck = O.CloudKey(p.n, p.N, p.k, p.l, p.Bgbit, p.ks_t, p.ks_basebit, k["bk"], k["ksk"])
    a, b, enc_seed = 0xDEADBEEF, 0x9ABCDEF1, 3001
    S = p.n + 1
    inb = np.zeros(96, dtype=np.uint8)
    inb[:32] = tools.int_to_bits(a, 32)
    inb[32:64] = tools.int_to_bits(b, 32)
    inp = tools.encrypt_bits(p, k["lwe_key"], inb, enc_seed)  # A bits, B bits, carry word (zeros)
    o1 = np.zeros((8, 32, S), np.int32)
    o2 = np.zeros((8, 32, S), np.int32)
    o1[0], o2[0] = inp[:32], inp[32:64]
    t0 = time.time()
    rc, out = ck.cloud_values(4, 0, 32, o1, o2, inp[64:96], threads=0)
    assert rc == 0
    res = np.ascontiguousarray(out[:2].reshape(64, S))  # low word, high word (cloud.c:2683-2686)
    dec = tools.decrypt_bits(p, k["lwe_key"], res)
    assert tools.bits_to_int(dec) == a * b
    with open(os.path.join(HERE, "mul32_n630.json"), "w") as f:
        json.dump({"params": "n=630 N=1024 k=1 l=3 Bgbit=7 ks_t=8 ks_basebit=2", "key_seed": list(seed),
                   "a": a, "b": b, "encrypt_seed": enc_seed, "input_sha256": digest(inp),
                   "output_sha256": digest(res), "first_sample": res[0].tolist(), "last_sample": res[-1].tolist(),
                   "bootstraps": int(ck.bootstrap_count), "oracle_seconds": round(time.time() - t0, 1),
                   "made_by": "tests/golden/make_golden.py mul32_n630 (oracle exact NTT back-end, deferred level-parallel mode)"}, f, indent=0)
    print("mul32_n630.json written in %.0f s" % (time.time() - t0))


def muladd64_n630():
    """BASELINE configs[3]'s circuit at the product parameter set: 64-bit a*b+c as the reference evaluates it
    -- compute() = MUL at 64 bits, then compute_final() = ADD at 128 bits on [answer | c]
    (dragonfly_cipher_cloud.py:1219-1327) -- by the oracle's two sequential cloud.c runs: 35 296 + 640 exact
    bootstraps at n=630 (deferred level-parallel mode; ~40 min on 8 cores).  Committed: operands, seeds,
    sha256 of the 128 output samples, first and last sample."""
    import time
    p = ia.default_params()
    seed = (314, 1592, 657)
    k = tools.keygen_raw(p, seed)
    ck = O.CloudKey(p.n, p.N, p.k, p.l, p.Bgbit, p.ks_t, p.ks_basebit, k["bk"], k["ksk"])
    a, b, c, enc_seed = 0xFEDCBA9876543210, 0x0F1E2D3C4B5A6978, (1 << 127) | 0x1234567890ABCDEF, 3002
    S = p.n + 1
    inb = np.zeros(2 * 64 + 32 + 128, dtype=np.uint8)  # IEACHE_CIRC_MULADD inputs: A, B, carry word, C (128 bits)
    inb[:64], inb[64:128], inb[160:] = tools.int_to_bits(a, 64), tools.int_to_bits(b, 64), tools.int_to_bits(c, 128)
    inp = tools.encrypt_bits(p, k["lwe_key"], inb, enc_seed)
    o1 = np.zeros((8, 32, S), np.int32)
    o2 = np.zeros((8, 32, S), np.int32)
    o1[:2], o2[:2] = inp[:64].reshape(2, 32, S), inp[64:128].reshape(2, 32, S)
    t0 = time.time()
    rc, st1 = ck.cloud_values(4, 0, 64, o1, o2, inp[128:160], threads=0)
    assert rc == 0
    C = np.zeros((8, 32, S), np.int32)
    C[:4] = inp[160:].reshape(4, 32, S)
    rc, st2 = ck.cloud_values(1, 0, 128, np.ascontiguousarray(st1[:8]), C, inp[128:160], threads=0)
    assert rc == 0
    res = np.ascontiguousarray(st2[:4].reshape(128, S))
    assert tools.bits_to_int(tools.decrypt_bits(p, k["lwe_key"], res)) == (a * b + c) % (1 << 128)
    with open(os.path.join(HERE, "muladd64_n630.json"), "w") as f:
        json.dump({"params": "n=630 N=1024 k=1 l=3 Bgbit=7 ks_t=8 ks_basebit=2", "key_seed": list(seed),
                   "a": a, "b": b, "c": c, "encrypt_seed": enc_seed, "input_sha256": digest(inp),
                   "output_sha256": digest(res), "first_sample": res[0].tolist(), "last_sample": res[-1].tolist(),
                   "bootstraps": int(ck.bootstrap_count), "oracle_seconds": round(time.time() - t0, 1),
                   "made_by": "tests/golden/make_golden.py muladd64_n630 (oracle exact NTT back-end, two cloud.c runs, deferred level-parallel mode)"}, f, indent=0)
    print("muladd64_n630.json written in %.0f s" % (time.time() - t0))


def mul128_n630():
    """BASELINE configs[4]'s circuit at the product parameter set: one 128-bit multiplication as cloud.c does it
    (4 x mul128 + 15 chained adds, cloud.c:2371-2567): 121 184 exact bootstraps at n=630 by the oracle's sequential
    gate stream in deferred level-parallel mode (~2.5 h on 8 cores)."""
    import time
    p = ia.default_params()
    seed = (314, 1592, 657)
    k = tools.keygen_raw(p, seed)
    ck = O.CloudKey(p.n, p.N, p.k, p.l, p.Bgbit, p.ks_t, p.ks_basebit, k["bk"], k["ksk"])
    a, b, enc_seed = (1 << 126) | 0xFFFFFFFFFFFFFFFFFFFFFFFF, (1 << 127) | 0x123456789ABCDEF0FEDCBA9, 3003
    S = p.n + 1
    inb = np.zeros(2 * 128 + 32, dtype=np.uint8)
    inb[:128], inb[128:256] = tools.int_to_bits(a, 128), tools.int_to_bits(b, 128)
    inp = tools.encrypt_bits(p, k["lwe_key"], inb, enc_seed)
    o1 = np.zeros((8, 32, S), np.int32)
    o2 = np.zeros((8, 32, S), np.int32)
    o1[:4], o2[:4] = inp[:128].reshape(4, 32, S), inp[128:256].reshape(4, 32, S)
    t0 = time.time()
    rc, out = ck.cloud_values(4, 0, 128, o1, o2, inp[256:288], threads=0)
    assert rc == 0
    res = np.ascontiguousarray(out[:8].reshape(256, S))
    assert tools.bits_to_int(tools.decrypt_bits(p, k["lwe_key"], res)) == a * b
    with open(os.path.join(HERE, "mul128_n630.json"), "w") as f:
        json.dump({"params": "n=630 N=1024 k=1 l=3 Bgbit=7 ks_t=8 ks_basebit=2", "key_seed": list(seed),
                   "a": a, "b": b, "encrypt_seed": enc_seed, "input_sha256": digest(inp),
                   "output_sha256": digest(res), "first_sample": res[0].tolist(), "last_sample": res[-1].tolist(),
                   "bootstraps": int(ck.bootstrap_count), "oracle_seconds": round(time.time() - t0, 1),
                   "made_by": "tests/golden/make_golden.py mul128_n630 (oracle exact NTT back-end, deferred level-parallel mode)"}, f, indent=0)
    print("mul128_n630.json written in %.0f s" % (time.time() - t0))


def misc_n630():
    """Smaller product-parameter vectors (about 1 500 exact bootstraps, ~6 min on 8 cores): the chained a+b-c at 32 bits
    (compute() + compute_final(), ADD then SUB), (-A)+B at 64 bits, and bootsOR / bootsNAND / bootsMUX gates."""
    import time
    p = ia.default_params()
    seed = (314, 1592, 657)
    k = tools.keygen_raw(p, seed)
    ck = O.CloudKey(p.n, p.N, p.k, p.l, p.Bgbit, p.ks_t, p.ks_basebit, k["bk"], k["ksk"])
    S = p.n + 1
    t0 = time.time()
    out = {"params": "n=630 N=1024 k=1 l=3 Bgbit=7 ks_t=8 ks_basebit=2", "key_seed": list(seed)}
    # a + b - c, 32 bits: IEACHE_CIRC_CHAIN(ADD, SUB, flip) inputs = A, B, carry word, C
    a, b, c = 0xF0E1D2C3, 0x1F2E3D4C, 0x89ABCDEF
    inb = np.zeros(32 * 4, dtype=np.uint8)
    inb[:32], inb[32:64], inb[96:128] = tools.int_to_bits(a, 32), tools.int_to_bits(b, 32), tools.int_to_bits(c, 32)
    inp = tools.encrypt_bits(p, k["lwe_key"], inb, 3004)
    o1 = np.zeros((8, 32, S), np.int32)
    o2 = np.zeros((8, 32, S), np.int32)
    o1[0], o2[0] = inp[:32], inp[32:64]
    rc, st1 = ck.cloud_values(1, 0, 32, o1, o2, inp[64:96], threads=0)
    assert rc == 0
    C = np.zeros((8, 32, S), np.int32)
    C[0] = inp[96:128]
    rc, st2 = ck.cloud_values(2, 0, 32, np.ascontiguousarray(st1[:8]), C, inp[64:96], threads=0)
    assert rc == 0 and tools.bits_to_int(tools.decrypt_bits(p, k["lwe_key"], st2[0])) == (a + b - c) % (1 << 32)
    out["addsub32"] = {"a": a, "b": b, "c": c, "encrypt_seed": 3004, "input_sha256": digest(inp),
                       "output_sha256": digest(np.ascontiguousarray(st2[0])), "first_sample": st2[0][0].tolist()}
    # (-A) + B at 64 bits = circuit RSUB (operator 1, first operand negative)
    a, b = 0x0123456789ABCDEF, 0xFEDCBA9876543210
    inb = np.zeros(64 * 2 + 32, dtype=np.uint8)
    inb[:64], inb[64:128] = tools.int_to_bits(a, 64), tools.int_to_bits(b, 64)
    inp = tools.encrypt_bits(p, k["lwe_key"], inb, 3005)
    o1[:] = 0
    o2[:] = 0
    o1[:2], o2[:2] = inp[:64].reshape(2, 32, S), inp[64:128].reshape(2, 32, S)
    rc, r = ck.cloud_values(1, 1, 64, o1, o2, inp[128:160], threads=0)
    res = np.ascontiguousarray(r[:2].reshape(64, S))
    assert rc == 0 and tools.bits_to_int(tools.decrypt_bits(p, k["lwe_key"], res)) == (b - a) % (1 << 64)
    out["rsub64"] = {"a": a, "b": b, "encrypt_seed": 3005, "input_sha256": digest(inp), "output_sha256": digest(res),
                     "last_sample": res[-1].tolist()}
    # gates: OR, NAND, MUX on the four / eight input combinations
    bits = np.array([[0, 0, 0, 0, 1, 1, 1, 1], [0, 0, 1, 1, 0, 0, 1, 1], [0, 1, 0, 1, 0, 1, 0, 1]], dtype=np.uint8)
    ga, gb, gc = (tools.encrypt_bits(p, k["lwe_key"], bits[i], 3006 + i) for i in range(3))
    g_or = ck.gates_batch("or", ga, gb)
    g_nand = ck.gates_batch("nand", ga, gb)
    g_mux = np.stack([ck.mux(ga[i], gb[i], gc[i]) for i in range(8)])
    assert np.array_equal(tools.decrypt_bits(p, k["lwe_key"], g_mux), np.where(bits[0] == 1, bits[1], bits[2]))
    out["gates"] = {"encrypt_seeds": [3006, 3007, 3008], "or_sha256": digest(g_or), "nand_sha256": digest(g_nand),
                    "mux_sha256": digest(g_mux), "mux_first_sample": g_mux[0].tolist()}
    out["oracle_seconds"] = round(time.time() - t0, 1)
    out["made_by"] = "tests/golden/make_golden.py misc_n630 (oracle exact NTT back-end)"
    with open(os.path.join(HERE, "misc_n630.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("misc_n630.json written in %.0f s" % (time.time() - t0))


CLOUD_N630_CASES = (
    # name, operator.txt code, bit size, (sign code, value, alice seed) x 2
    ("add16_in_32bit_word", 1, 32, (0, 0xBEEF, 4101), (0, 0x1234, 4102)),   # BASELINE configs[0]: 16-bit a+b, zero-extended in the 32-bit word
    ("sub32", 2, 32, (0, 0x1234ABCD, 4103), (0, 0x0FEDCBA9, 4104)),
    ("mul32", 4, 32, (0, 0xC0FFEE11, 4105), (0, 0x89ABCDEF, 4106)),
    ("add64", 1, 64, (0, 0xFEDCBA9876543210, 4107), (0, 0x0123456789ABCDEF, 4108)),                    # sum = 2^64 - 1: every sum bit set, no carry out
    ("sub32_borrow", 2, 32, (0, 0x0FEDCBA9, 4109), (0, 0x1234ABCD, 4110)),                               # a < b: two's complement result
    # the sign branches of main() (cloud.c:812-821, 870, 1194-1196, 1809): process.c's sign code 2 = negative
    ("add32_first_negative", 1, 32, (2, 0x0BADF00D, 4111), (0, 0x1234ABCD, 4112)),                       # (-A)+B runs B-A
    ("sub64_second_negative", 2, 64, (0, 0x0123456789ABCDEF, 4113), (2, 0x0FEDCBA987654321, 4114)),      # A-(-B) runs A+B
    ("add128_both_negative", 1, 128, (2, 0x0123456789ABCDEF0FEDCBA987654321, 4115), (2, 0x01111111222222223333333344444444, 4116)),  # -(A+B): ADD, total negatives 4
    ("sub32_both_negative", 2, 32, (2, 0x0FEDCBA9, 4117), (2, 0x1234ABCD, 4118)),                        # (-A)-(-B) runs B-A
    # the widest operands the contract carries: all eight words of each operand, eight result words
    ("add256", 1, 256, (0, 0x0123456789ABCDEF0FEDCBA98765432100112233445566778899AABBCCDDEEFF, 4119),
     (0, 0x0EDCBA9876543210F0123456789ABCDEFFEEDDCCBBAA99887766554433221100, 4120)),
    ("sub128", 2, 128, (0, 0x80000000000000000000000000000001, 4121), (0, 0x00000000FFFFFFFFFFFFFFFF00000002, 4122)),   # borrows across word boundaries
    ("sub256", 2, 256, (0, 0x8000000000000000000000000000000000000000000000000000000000000000, 4125),
     (0, 0x0000000100000000FFFFFFFF00000001FFFFFFFE00000000FFFFFFFFFFFFFFFF, 4126)),                      # the widest SUB: borrows through all eight words
    ("mul64_second_negative", 4, 64, (0, 0xFEDCBA9876543210, 4123), (2, 0x0123456789ABCDEF, 4124)),   # split / mul32 x4 / recombination (cloud.c:220-385); ~45 min of oracle on 8 cores
    ("mul128_both_negative", 4, 128, (2, 0xF0E1D2C3B4A5968778695A4B3C2D1E0F, 4127), (2, 0x0123456789ABCDEFFEDCBA9876543210, 4128)),  # cloud.c:387-647; ~2.5 h of oracle on 8 cores
)
CLOUD_N630_KEY_SEED, CLOUD_N630_NBIT_SEED = (314, 1592, 657), (2718, 2818)


def cloud_n630(only=None):
    """The ./cloud FILE contract at the product parameter set (cloud.c:650-917): keygen from the documented seeds,
    `alice` twice per case, then the oracle's orc_cloud_values on that cloud.data.  Committed per case: sha256 of
    cloud.data's 704 samples, sha256 of the 288 value samples of answer.data (words r1..r8 + carry filler; the two
    metadata words are fresh encryptions and compared by decryption), first and last value sample.
    ~15 min on 8 cores (the 32-bit MUL is 11 264 exact bootstraps)."""
    import tempfile
    import time
    p = ia.default_params()
    S = p.n + 1
    path = os.path.join(HERE, "cloud_n630.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    out.update({"params": "n=630 N=1024 k=1 l=3 Bgbit=7 ks_t=8 ks_basebit=2", "key_seed": list(CLOUD_N630_KEY_SEED),
                "nbit_seed": list(CLOUD_N630_NBIT_SEED), "sample_bytes": 4 * p.n + 16,
                "made_by": "tests/golden/make_golden.py cloud_n630 (keygen_files + alice x2 per case; oracle exact NTT back-end, "
                           "orc_cloud_values in deferred level-parallel mode)"})
    out.setdefault("cases", {})
    with tempfile.TemporaryDirectory() as d:
        tools.keygen_files(d, p, seed=CLOUD_N630_KEY_SEED, nbit_seed=CLOUD_N630_NBIT_SEED)
        _, bk, ksk = tools.read_cloud_key(os.path.join(d, "cloud.key"))
        _, lwe_key, _ = tools.read_secret_key(os.path.join(d, "secret.key"))
        out["cloud_key_sha256"] = digest(bk, ksk)
        ck = O.CloudKey(p.n, p.N, p.k, p.l, p.Bgbit, p.ks_t, p.ks_basebit, bk, ksk)
        for name, op, bits, (sa, a, seed_a), (sb, b, seed_b) in CLOUD_N630_CASES:
            if only and name not in only:
                continue
            t0 = time.time()
            tools.alice(d, sa, bits, a, seed=seed_a)
            tools.alice(d, sb, bits, b, seed=seed_b, append=True)
            data = tools.read_samples(os.path.join(d, "cloud.data"), p.n)
            assert data.shape == (704, S)
            w = data.reshape(22, 32, S)
            neg = {0: 0, 1: 1, 2: 1}[sa] + sb  # cloud.c:787-789: code 2 of operand 1 is remapped to 1
            rc, ref = ck.cloud_values(op, neg, bits, np.ascontiguousarray(w[2:10]), np.ascontiguousarray(w[13:21]),
                                      np.ascontiguousarray(w[10]), threads=0)
            assert rc == 0
            ref = np.ascontiguousarray(ref.reshape(288, S))
            nw = (2 * bits if op == 4 else bits) // 32
            val = tools.bits_to_int(tools.decrypt_bits(p, lwe_key, ref[:32 * nw]))
            # which circuit main() runs on the magnitudes (cloud.c:870, 1194-1196, 1809): ADD, A-B or B-A
            run = 4 if op == 4 else {(1, 0): 1, (1, 3): 1, (1, 2): 2, (1, 1): 3, (2, 1): 1, (2, 2): 1, (2, 0): 2, (2, 3): 3}[(op, neg)]
            exp = {1: (a + b) % (1 << bits), 2: (a - b) % (1 << bits), 3: (b - a) % (1 << bits), 4: a * b}[run]
            assert val == exp, (name, hex(val), hex(exp))
            if op == 4 and neg in (1, 2):
                exp = -exp  # verif.c:1409-1435
            if op != 4:  # what verif.c reads back (verif.c:120-179, 733-789): the signed result of the signed operands
                sgn_a, sgn_b = (-a if sa else a), (-b if sb else b)
                exp = sgn_a + sgn_b if op == 1 else sgn_a - sgn_b
            out["cases"][name] = {"operator": op, "bits": bits, "a": a, "sign_a": sa, "seed_a": seed_a, "b": b, "sign_b": sb,
                                  "seed_b": seed_b, "cloud_data_sha256": digest(data), "value_samples_sha256": digest(ref),
                                  "first_value_sample": ref[0].tolist(), "last_value_sample": ref[-1].tolist(),
                                  "expect": exp, "bootstraps": int(ck.bootstrap_count), "oracle_seconds": round(time.time() - t0, 1)}
            print("cloud_n630 %s done in %.0f s" % (name, time.time() - t0), flush=True)
            with open(path, "w") as f:
                json.dump(out, f, indent=0)
    print("cloud_n630.json written")


CLOUD_N630_CHAIN = {"first": "add64", "operator": 2, "c": 0xFF00FF00FF00FF00, "seed_c": 4131}  # (2^64 - 1) - c stays below 2^63: verif.c reads SUB results as two's complement


def cloud_n630_chain():
    """compute() followed by compute_final() through the FILE boundary at the product parameter set
    (Cloud/dragonfly_cipher_cloud.py:1219-1327): stage 1 = the `add64` case above, then the third operand alone in
    cloud.data, answer.data spliced in front of it (flip) and ./cloud run again with operator SUB: (a + b) - c.
    The second run's operand words are the first run's value samples and the carry-word filler, so its 288 value
    samples are fixed too; committed: their sha256, first / last sample, the integer verif reads.  ~5 min on 8 cores."""
    import tempfile
    import time
    p = ia.default_params()
    S = p.n + 1
    path = os.path.join(HERE, "cloud_n630.json")
    out = json.load(open(path))
    first = next(c for c in CLOUD_N630_CASES if c[0] == CLOUD_N630_CHAIN["first"])
    name, op, bits, (sa, a, seed_a), (sb, b, seed_b) = first
    with tempfile.TemporaryDirectory() as d:
        t0 = time.time()
        tools.keygen_files(d, p, seed=CLOUD_N630_KEY_SEED, nbit_seed=CLOUD_N630_NBIT_SEED)
        _, bk, ksk = tools.read_cloud_key(os.path.join(d, "cloud.key"))
        _, lwe_key, _ = tools.read_secret_key(os.path.join(d, "secret.key"))
        ck = O.CloudKey(p.n, p.N, p.k, p.l, p.Bgbit, p.ks_t, p.ks_basebit, bk, ksk)
        tools.alice(d, sa, bits, a, seed=seed_a)
        tools.alice(d, sb, bits, b, seed=seed_b, append=True)
        w = tools.read_samples(os.path.join(d, "cloud.data"), p.n).reshape(22, 32, S)
        rc, ref1 = ck.cloud_values(op, 0, bits, np.ascontiguousarray(w[2:10]), np.ascontiguousarray(w[13:21]), np.ascontiguousarray(w[10]), threads=0)
        ref1 = np.ascontiguousarray(ref1.reshape(288, S))
        assert rc == 0 and digest(ref1) == out["cases"][name]["value_samples_sha256"]
        # stage 2: answer.data = [neg, bit, r1..r8, filler] in front of the third operand's 11 words
        c, seed_c, op2 = CLOUD_N630_CHAIN["c"], CLOUD_N630_CHAIN["seed_c"], CLOUD_N630_CHAIN["operator"]
        tools.alice(d, 0, bits, c, seed=seed_c)
        wc = tools.read_samples(os.path.join(d, "cloud.data"), p.n).reshape(11, 32, S)
        r = ref1.reshape(9, 32, S)
        rc, ref2 = ck.cloud_values(op2, 0, bits, np.ascontiguousarray(r[:8]), np.ascontiguousarray(wc[2:10]), np.ascontiguousarray(r[8]), threads=0)
        ref2 = np.ascontiguousarray(ref2.reshape(288, S))
        assert rc == 0
        val = tools.bits_to_int(tools.decrypt_bits(p, lwe_key, ref2[:bits]))
        assert val == (a + b - c) % (1 << bits), hex(val)
        out["chain"] = dict(CLOUD_N630_CHAIN, bits=bits, value_samples_sha256=digest(ref2), first_value_sample=ref2[0].tolist(),
                            last_value_sample=ref2[-1].tolist(), expect=a + b - c, oracle_seconds=round(time.time() - t0, 1),
                            note="compute(ADD) on the `add64` case, then compute_final(SUB, flip=True) with the third operand c")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("cloud_n630.json: chain written in %.0f s" % (time.time() - t0))


def plaintext_kats():
    kats = []
    for bits in (32, 64, 128, 256):
        v = 1 << (bits - 2)  # process.c
        for op, name in ((1, "+"), (2, "-"), (4, "*")):
            if op == 4 and bits == 256:
                kats.append({"op": op, "bits": bits, "a": str(v), "sa": 0, "b": str(v), "sb": 0, "exit": 126})
                continue
            for sa in (0, 2):       # process.c:80,86 sign codes: 0 positive, 2 negative
                for sb in (0, 2):
                    A, B = (-v if sa else v), (-v if sb else v)
                    r = A + B if op == 1 else (A - B if op == 2 else A * B)
                    kats.append({"op": op, "bits": bits, "a": str(v), "sa": sa, "b": str(v), "sb": sb,
                                 "exit": 0, "expect": str(r)})
    with open(os.path.join(HERE, "plaintext_kats.json"), "w") as f:
        json.dump(kats, f, indent=0)
    print("plaintext_kats.json written,", len(kats), "cases")


if __name__ == "__main__":
    if sys.argv[1:] == ["mul32_n630"]:  # ~12 min on 8 cores; not part of the default regeneration
        mul32_n630()
    elif sys.argv[1:] == ["muladd64_n630"]:  # ~40 min on 8 cores
        muladd64_n630()
    elif sys.argv[1:] == ["mul128_n630"]:    # ~2.5 h on 8 cores
        mul128_n630()
    elif sys.argv[1:] == ["misc_n630"]:      # ~6 min
        misc_n630()
    elif sys.argv[1:] == ["cloud_n630_chain"]:  # ~5 min
        cloud_n630_chain()
    elif sys.argv[1:2] == ["cloud_n630"]:    # ~15 min on 8 cores; optional case names after it
        cloud_n630(sys.argv[2:] or None)
    else:
        toy_vectors()
        full_size_kat()
        plaintext_kats()
