"""ctypes front end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.

Samples are numpy int32 arrays whose last axis is n+1 (a[0..n-1], b).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

POLYMUL_NTT, POLYMUL_SCHOOLBOOK, POLYMUL_FFT = 0, 1, 2


class Params(C.Structure):
    _fields_ = [(f, C.c_int32) for f in ("n", "N", "k", "l", "Bgbit", "ks_t", "ks_basebit")]


def build():
    """Compile liboracle.so with gcc (a few seconds)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        # idle OpenMP threads sleep instead of spinning: on a box whose CPU quota is below its thread count
        # spinning waiters starve the workers at every level barrier (read by libgomp when it is loaded)
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        os.environ.setdefault("GOMP_SPINCOUNT", "0")
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        i32p, vp = C.POINTER(C.c_int32), C.c_void_p
        L.orc_cloudkey_new.restype = vp
        L.orc_cloudkey_new.argtypes = [C.POINTER(Params), i32p, i32p]
        L.orc_cloudkey_free.argtypes = [vp]
        L.orc_cloudkey_set_polymul.argtypes = [vp, C.c_int]
        L.orc_cloudkey_bootstrap_count.restype = C.c_uint64
        L.orc_cloudkey_bootstrap_count.argtypes = [vp]
        L.orc_modswitch_to_torus32.restype = C.c_int32
        L.orc_modswitch_to_torus32.argtypes = [C.c_int32, C.c_int32]
        L.orc_modswitch_from_torus32.restype = C.c_int32
        L.orc_modswitch_from_torus32.argtypes = [C.c_int32, C.c_int32]
        L.orc_modswitch_sample.restype = C.c_int32
        L.orc_modswitch_sample.argtypes = [vp, i32p, i32p]
        L.orc_blind_rotate_init.argtypes = [vp, i32p, C.c_int32, C.c_int32]
        L.orc_blind_rotate_step.argtypes = [vp, i32p, C.c_int32, C.c_int32]
        L.orc_blind_rotate.argtypes = [vp, i32p, i32p]
        L.orc_sample_extract.argtypes = [vp, i32p, i32p]
        L.orc_keyswitch.argtypes = [vp, i32p, i32p]
        L.orc_bootstrap.argtypes = [vp, i32p, i32p]
        L.orc_gate_constant.argtypes = [vp, i32p, C.c_int32]
        for g in ("not", "copy"):
            getattr(L, "orc_gate_" + g).argtypes = [vp, i32p, i32p]
        for g in ("and", "xor", "or", "nand"):
            getattr(L, "orc_gate_" + g).argtypes = [vp, i32p, i32p, i32p]
        L.orc_negacyclic_mul.argtypes = [C.c_int, C.c_int32, i32p, i32p, i32p]
        L.orc_lwe_phase.restype = C.c_int32
        L.orc_lwe_phase.argtypes = [i32p, i32p, C.c_int32]
        L.orc_add.argtypes = [vp, i32p, i32p, i32p, i32p, i32p, C.c_int32]
        L.orc_mul32.argtypes = [vp] + [i32p] * 5 + [C.c_int32]
        L.orc_cloud_values.restype = C.c_int
        L.orc_cloud_values.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, i32p, i32p, i32p, i32p]
        L.orc_gate_mux.argtypes = [vp, i32p, i32p, i32p, i32p]
        L.orc_bootstrap_woks.argtypes = [vp, i32p, i32p]
        L.orc_defer_begin.argtypes = [vp]
        L.orc_defer_run.restype = C.c_int64
        L.orc_defer_run.argtypes = [vp, C.c_int]
        L.orc_gates_batch.argtypes = [vp, C.c_int32, C.c_size_t, i32p, i32p, i32p, C.c_int]
        L.orc_max_threads.restype = C.c_int
        L.orc_cloud_metadata.restype = C.c_int
        L.orc_cloud_metadata.argtypes = [C.c_int32] * 5 + [i32p] * 4
        _LIB = L
    return _LIB


def _p(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def negacyclic_mul(small, big, mode=POLYMUL_NTT):
    small = np.ascontiguousarray(small, dtype=np.int32)
    big = np.ascontiguousarray(big, dtype=np.int32)
    out = np.empty_like(small)
    lib().orc_negacyclic_mul(mode, small.shape[0], _p(out), _p(small), _p(big))
    return out


def modswitch_to_torus32(mu, msize):
    return lib().orc_modswitch_to_torus32(mu, msize)


def modswitch_from_torus32(phase, msize):
    return lib().orc_modswitch_from_torus32(int(np.int32(phase)), msize)


class CloudKey:
    """Holds BK [n][(k+1)l][k+1][N] and KSK [kN][t][base][n+1] as raw int32."""

    def __init__(self, n, N, k, l, Bgbit, ks_t, ks_basebit, bk, ksk, polymul=POLYMUL_NTT):
        self.p = Params(n, N, k, l, Bgbit, ks_t, ks_basebit)
        bk = np.ascontiguousarray(bk, dtype=np.int32)
        ksk = np.ascontiguousarray(ksk, dtype=np.int32)
        assert bk.size == n * (k + 1) * l * (k + 1) * N, bk.shape
        assert ksk.size == k * N * ks_t * (1 << ks_basebit) * (n + 1), ksk.shape
        self.h = lib().orc_cloudkey_new(C.byref(self.p), _p(bk), _p(ksk))
        assert self.h, "unsupported parameters"
        self.n, self.N, self.k = n, N, k
        if polymul != POLYMUL_NTT:
            self.set_polymul(polymul)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_cloudkey_free(self.h)
            self.h = None

    def set_polymul(self, mode):
        lib().orc_cloudkey_set_polymul(self.h, mode)

    @property
    def bootstrap_count(self):
        return lib().orc_cloudkey_bootstrap_count(self.h)

    def _new(self, count=None):
        shape = (self.n + 1,) if count is None else (count, self.n + 1)
        return np.zeros(shape, dtype=np.int32)

    # --- stages ---
    def modswitch(self, x):
        x = np.ascontiguousarray(x, dtype=np.int32)
        bara = np.zeros(self.n, dtype=np.int32)
        barb = lib().orc_modswitch_sample(self.h, _p(x), _p(bara))
        return bara, barb

    def blind_rotate_init(self, barb, mu=1 << 29):
        acc = np.zeros((self.k + 1, self.N), dtype=np.int32)
        lib().orc_blind_rotate_init(self.h, _p(acc), barb, mu)
        return acc

    def blind_rotate_step(self, acc, i, barai):
        acc = np.ascontiguousarray(acc, dtype=np.int32).copy()
        lib().orc_blind_rotate_step(self.h, _p(acc), i, int(barai))
        return acc

    def blind_rotate(self, acc, bara):
        acc = np.ascontiguousarray(acc, dtype=np.int32).copy()
        bara = np.ascontiguousarray(bara, dtype=np.int32)
        lib().orc_blind_rotate(self.h, _p(acc), _p(bara))
        return acc

    def sample_extract(self, acc):
        acc = np.ascontiguousarray(acc, dtype=np.int32)
        u = np.zeros(self.k * self.N + 1, dtype=np.int32)
        lib().orc_sample_extract(self.h, _p(u), _p(acc))
        return u

    def keyswitch(self, u):
        u = np.ascontiguousarray(u, dtype=np.int32)
        out = self._new()
        lib().orc_keyswitch(self.h, _p(out), _p(u))
        return out

    def bootstrap(self, x):
        x = np.ascontiguousarray(x, dtype=np.int32)
        out = self._new()
        lib().orc_bootstrap(self.h, _p(out), _p(x))
        return out

    # --- gates ---
    def constant(self, v):
        out = self._new()
        lib().orc_gate_constant(self.h, _p(out), int(v))
        return out

    def gate(self, name, ca, cb=None):
        ca = np.ascontiguousarray(ca, dtype=np.int32)
        out = self._new()
        f = getattr(lib(), "orc_gate_" + name)
        if cb is None:
            f(self.h, _p(out), _p(ca))
        else:
            cb = np.ascontiguousarray(cb, dtype=np.int32)
            f(self.h, _p(out), _p(ca), _p(cb))
        return out

    def mux(self, a, b, c):
        """bootsMUX(a, b, c) = a ? b : c."""
        a, b, c = (np.ascontiguousarray(v, dtype=np.int32) for v in (a, b, c))
        out = self._new()
        lib().orc_gate_mux(self.h, _p(out), _p(a), _p(b), _p(c))
        return out

    def bootstrap_woks(self, x):
        x = np.ascontiguousarray(x, dtype=np.int32)
        u = np.zeros(self.k * self.N + 1, dtype=np.int32)
        lib().orc_bootstrap_woks(self.h, _p(u), _p(x))
        return u

    def gates_batch(self, name, a, b, threads=0):
        """`count` independent gates on `threads` host threads (0 = all): out[i] = gate(a[i], b[i])."""
        a = np.ascontiguousarray(a, dtype=np.int32)
        b = np.ascontiguousarray(b, dtype=np.int32)
        out = np.zeros_like(a)
        lib().orc_gates_batch(self.h, {"and": 0, "xor": 1, "or": 2, "nand": 3}[name], a.shape[0], _p(out), _p(a), _p(b), threads)
        return out

    # --- circuits ---
    def add(self, x, y, c, nb_bits):
        """cloud.c add(): returns (sum[nb_bits], carryover[1])."""
        x = np.ascontiguousarray(x, dtype=np.int32)
        y = np.ascontiguousarray(y, dtype=np.int32)
        c = np.ascontiguousarray(c, dtype=np.int32).reshape(-1, self.n + 1)
        s = self._new(nb_bits)
        co = self._new(1)
        lib().orc_add(self.h, _p(s), _p(co), _p(x), _p(y), _p(c), nb_bits)
        return s, co

    def mul32(self, a, b, carry):
        """cloud.c mul32(): returns (high[32], low[32])."""
        a = np.ascontiguousarray(a, dtype=np.int32)
        b = np.ascontiguousarray(b, dtype=np.int32)
        carry = np.ascontiguousarray(carry, dtype=np.int32)
        hi, lo = self._new(32), self._new(32)
        lib().orc_mul32(self.h, _p(hi), _p(lo), _p(a), _p(b), _p(carry), 32)
        return hi, lo

    def cloud_values(self, op, neg, int_bit, opnd1, opnd2, carry1, threads=1):
        """Value part of cloud.c main(): operands [8][32][n+1], carry1 [32][n+1]
        -> (rc, out [9][32][n+1]).  threads != 1: the same sequential gate stream is recorded and
        its independent gates evaluated on that many host threads (0 = all); identical bits."""
        opnd1 = np.ascontiguousarray(opnd1, dtype=np.int32)
        opnd2 = np.ascontiguousarray(opnd2, dtype=np.int32)
        carry1 = np.ascontiguousarray(carry1, dtype=np.int32)
        assert opnd1.shape == (8, 32, self.n + 1) and carry1.shape == (32, self.n + 1)
        out = np.zeros((9, 32, self.n + 1), dtype=np.int32)
        if threads != 1:
            lib().orc_defer_begin(self.h)
        rc = lib().orc_cloud_values(self.h, op, neg, int_bit, _p(opnd1), _p(opnd2), _p(carry1), _p(out))
        if threads != 1:
            lib().orc_defer_run(self.h, threads)
        return rc, out


def cloud_metadata(op, neg1, bit1, neg2, bit2):
    """-> (rc, sign_code, bit_word, neg_routing, int_bit), cloud.c:775-864."""
    vals = [C.c_int32() for _ in range(4)]
    rc = lib().orc_cloud_metadata(op, neg1, bit1, neg2, bit2, *[C.byref(v) for v in vals])
    return (rc,) + tuple(v.value for v in vals)


def max_threads():
    return lib().orc_max_threads()


def lwe_phase(sample, key):
    sample = np.ascontiguousarray(sample, dtype=np.int32)
    key = np.ascontiguousarray(key, dtype=np.int32)
    return lib().orc_lwe_phase(_p(sample), _p(key), key.shape[0])
