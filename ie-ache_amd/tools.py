"""CPU tools around the Cloud path: key generation, operand encryption and
answer decryption -- what the reference's Keygen/keygen.c:22-51,
Client1/alice.c:116-191 and Output/verif.c:41-179 do through libtfhe.  All of
it runs inside libieache.so's host code (no GPU, no oracle)."""
import ctypes as C
import os

import numpy as np

from .evaluator import Params, check, lib, _i32


def _u32(seq):
    if seq is None:
        return None, 0
    arr = (C.c_uint32 * len(seq))(*seq)
    return arr, len(seq)


def keygen_raw(params, seed=(314, 1592, 657), with_cloud=True):
    """-> dict(lwe_key [n], tlwe_key [kN], bk, ksk) as numpy int32 (bk/ksk None if with_cloud=False)."""
    s, ns = _u32(seed)
    lwe = np.zeros(params.n, dtype=np.int32)
    tlwe = np.zeros(params.k * params.N, dtype=np.int32)
    bk = np.zeros(params.bk_count, dtype=np.int32) if with_cloud else None
    ksk = np.zeros(params.ksk_count, dtype=np.int32) if with_cloud else None
    check(lib().ieache_keygen_raw(C.byref(params), s, ns, _i32(lwe), _i32(tlwe),
                                  _i32(bk) if with_cloud else None, _i32(ksk) if with_cloud else None))
    if with_cloud:
        kpl = (params.k + 1) * params.l
        bk = bk.reshape(params.n, kpl, params.k + 1, params.N)
        ksk = ksk.reshape(params.k * params.N, params.ks_t, 1 << params.ks_basebit, params.n + 1)
    return {"lwe_key": lwe, "tlwe_key": tlwe, "bk": bk, "ksk": ksk}


def keygen_files(directory, params=None, seed=None, nbit_seed=None):
    """keygen.c: writes secret.key, cloud.key, nbit.key into `directory`."""
    s, ns = _u32(seed)
    b, nb = _u32(nbit_seed)
    check(lib().ieache_keygen_files(os.fsencode(directory), C.byref(params) if params is not None else None,
                                    s, ns, b, nb))


def encrypt_bits(params, lwe_key, bits, seed):
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    lwe_key = np.ascontiguousarray(lwe_key, dtype=np.int32)
    out = np.zeros(bits.shape + (params.n + 1,), dtype=np.int32)
    check(lib().ieache_encrypt_bits(C.byref(params), _i32(lwe_key), bits.ctypes.data_as(C.POINTER(C.c_uint8)),
                                    bits.size, seed, _i32(out)))
    return out


def decrypt_bits(params, lwe_key, samples):
    samples = np.ascontiguousarray(samples, dtype=np.int32)
    lwe_key = np.ascontiguousarray(lwe_key, dtype=np.int32)
    shape = samples.shape[:-1]
    bits = np.zeros(shape, dtype=np.uint8)
    check(lib().ieache_decrypt_bits(C.byref(params), _i32(lwe_key), _i32(samples), int(np.prod(shape, dtype=np.int64)),
                                    bits.ctypes.data_as(C.POINTER(C.c_uint8))))
    return bits


def int_to_bits(value, nbits):
    return np.array([(int(value) >> i) & 1 for i in range(nbits)], dtype=np.uint8)


def bits_to_int(bits):
    return sum(int(b) << i for i, b in enumerate(np.asarray(bits).reshape(-1)))


def read_secret_key(path):
    p = Params()
    check(lib().ieache_read_secret_key(os.fsencode(path), C.byref(p), None, None))
    lwe = np.zeros(p.n, dtype=np.int32)
    tlwe = np.zeros(p.k * p.N, dtype=np.int32)
    check(lib().ieache_read_secret_key(os.fsencode(path), C.byref(p), _i32(lwe), _i32(tlwe)))
    return p, lwe, tlwe


def read_cloud_key(path):
    p = Params()
    check(lib().ieache_read_cloud_key(os.fsencode(path), C.byref(p), None, None))
    bk = np.zeros(p.bk_count, dtype=np.int32)
    ksk = np.zeros(p.ksk_count, dtype=np.int32)
    check(lib().ieache_read_cloud_key(os.fsencode(path), C.byref(p), _i32(bk), _i32(ksk)))
    return p, bk, ksk


def write_cloud_key(path, params, bk, ksk):
    bk = np.ascontiguousarray(bk, dtype=np.int32)
    ksk = np.ascontiguousarray(ksk, dtype=np.int32)
    check(lib().ieache_write_cloud_key(os.fsencode(path), C.byref(params), _i32(bk), _i32(ksk)))


def write_secret_key(path, params, lwe_key, tlwe_key, bk, ksk):
    arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in (lwe_key, tlwe_key, bk, ksk)]
    check(lib().ieache_write_secret_key(os.fsencode(path), C.byref(params), *[_i32(a) for a in arrs]))


def read_samples(path, n, first=0, count=None):
    if count is None:
        count = os.path.getsize(path) // (4 * n + 16) - first
    out = np.zeros((count, n + 1), dtype=np.int32)
    check(lib().ieache_read_samples(os.fsencode(path), n, first, count, _i32(out)))
    return out


def write_samples(path, rows, append=False):
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    n = rows.shape[-1] - 1
    check(lib().ieache_write_samples(os.fsencode(path), n, rows.size // (n + 1), _i32(rows), int(append)))


def alice(directory, sign_code, bit_size, value, seed, append=False, cloud_data="cloud.data"):
    """alice.c: encrypt one operand (magnitude `value`, sign code 0/1/2) into cloud.data."""
    words = [(int(value) >> (32 * w)) & 0xFFFFFFFF for w in range(8)]
    arr = (C.c_uint32 * 8)(*words)
    d = os.fspath(directory)
    check(lib().ieache_alice(os.fsencode(os.path.join(d, "secret.key")), os.fsencode(os.path.join(d, "nbit.key")),
                             os.fsencode(os.path.join(d, cloud_data)), int(append), sign_code, bit_size, arr, seed))


def verif(directory, answer_data="answer.data"):
    """verif.c decrypt step -> (sign_code, bit_size, [9 words])."""
    d = os.fspath(directory)
    code, bits = C.c_uint32(), C.c_uint32()
    words = (C.c_uint32 * 9)()
    check(lib().ieache_verif(os.fsencode(os.path.join(d, "secret.key")), os.fsencode(os.path.join(d, "nbit.key")),
                             os.fsencode(os.path.join(d, answer_data)), C.byref(code), C.byref(bits), words))
    return code.value, bits.value, list(words)


def verif_interpret(op, sign_code, bit_size, words):
    """Output/verif.c's reconstruction rules -> Python int.

    op: operator.txt code (1 add, 2 sub, 4 mul).  Words are LSW first
    (verif.c:229).  ADD (verif.c:120-179): codes 0/4 read the magnitude
    unsigned (4 negates); codes 1/2 read two's complement.  SUB
    (verif.c:733-789): code 2 unsigned, otherwise two's complement, code 1
    negates.  MUL (verif.c:1409-1435): unsigned, codes 1/2 negate."""
    nwords = bit_size // 32
    total = 0
    for w in reversed(range(nwords)):
        total = (total << 32) | (words[w] & 0xFFFFFFFF)
    top = 1 << (bit_size - 1)

    def twos(v):
        return v - (1 << bit_size) if v & top else v

    if op == 1:
        if sign_code in (0, 4):
            return -total if sign_code == 4 else total
        return twos(total)
    if op == 2:
        v = total if sign_code == 2 else twos(total)
        return -v if sign_code == 1 else v
    if op == 4:
        return -total if sign_code in (1, 2) else total
    raise ValueError("unknown operator code %r" % (op,))
