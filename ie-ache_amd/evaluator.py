"""ctypes binding of libieache.so (include/ieache.h).

There is no CPU fallback: if the HIP library is missing this module raises on
first use, and every evaluation entry point needs a GPU.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB = None

CIRC_ADD, CIRC_SUB, CIRC_RSUB, CIRC_MUL, CIRC_MULADD = 1, 2, 3, 4, 5
CIRC_ADD_KS, CIRC_SUB_KS, CIRC_RSUB_KS = 6, 7, 8  # Kogge-Stone variants (decrypt-identical, not bit-identical)
CIRC_MUL_WALLACE = 9  # carry-save multiplier (decrypt-identical, not bit-identical)
GATE_AND, GATE_XOR, GATE_OR, GATE_NAND, GATE_MUX = 0, 1, 2, 3, 4


def circ_chain(k1, k2, flip=True):
    """IEACHE_CIRC_CHAIN: stage 1 = k1(A, B), stage 2 = k2(answer, C) (flip) or k2(C, answer) --
    compute() followed by compute_final() (Cloud/dragonfly_cipher_cloud.py:1219-1327) as one DAG."""
    assert 1 <= k1 <= 4 and 1 <= k2 <= 4
    return 32 + (k1 - 1) + 4 * (k2 - 1) + (0 if flip else 16)


class IeacheError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ieache error %d: %s" % (code, msg))
        self.code = code


class Params(C.Structure):
    _fields_ = [(f, C.c_int32) for f in ("n", "N", "k", "l", "Bgbit", "ks_t", "ks_basebit")] + \
               [(f, C.c_double) for f in ("lwe_alpha_min", "lwe_alpha_max", "tlwe_alpha_min", "tlwe_alpha_max")]

    @property
    def bk_count(self):
        return self.n * (self.k + 1) * self.l * (self.k + 1) * self.N

    @property
    def ksk_count(self):
        return self.k * self.N * self.ks_t * (1 << self.ks_basebit) * (self.n + 1)

    @property
    def lwe_stride(self):
        return (self.n + 1 + 3) & ~3

    def copy(self, **kw):
        p = Params.from_buffer_copy(bytes(self))
        for k, v in kw.items():
            setattr(p, k, v)
        return p


class Stats(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("blind_rotate_ms", C.c_double), ("keyswitch_ms", C.c_double),
                ("blind_rotate_launches", C.c_int64), ("keyswitch_launches", C.c_int64),
                ("bootstraps", C.c_int64), ("levels", C.c_int64), ("chunks", C.c_int64)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


class CircuitInfo(C.Structure):
    _fields_ = [("n_inputs", C.c_int32), ("n_outputs", C.c_int32), ("n_slots", C.c_int32), ("depth", C.c_int32),
                ("max_width", C.c_int32), ("bootstraps", C.c_int64), ("n_and", C.c_int64), ("n_xor", C.c_int64),
                ("sched_max_width", C.c_int32), ("folded", C.c_int32), ("reference_bootstraps", C.c_int64),
                ("sched_levels", C.c_int32), ("level_cap", C.c_int32)]


def library_path():
    # IEACHE_LIBRARY: another build of the same library (A/B of compiler flags: scripts/build_alt.sh)
    return os.environ.get("IEACHE_LIBRARY") or os.path.join(_PKG, "libieache.so")


def build_library(jobs=4):
    """Compile libieache.so and the `cloud` shim for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-j%d" % jobs, "-C", os.path.join(_PKG, "csrc")])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise IeacheError(-19, "HIP extension %s is missing: run __graft_entry__.build() "
                               "(make -C ie-ache_amd/csrc); there is no CPU fallback" % path)
    # PyTorch bundles its own HIP runtime under the same SONAME (libamdhip64.so.7) as
    # /opt/rocm's.  Two HIP runtimes in one process cannot both own the GPU, so when
    # torch is installed let it load first: the dynamic loader then binds libieache.so
    # to the runtime already in the process.  (The standalone `cloud` executable has
    # no torch in-process and uses /opt/rocm's.)
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(path)
    i32p, u32p, u8p, vp = C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8), C.c_void_p
    pp, sp = C.POINTER(Params), C.POINTER(Stats)
    L.ieache_version.restype = C.c_char_p
    L.ieache_last_error.restype = C.c_char_p
    L.ieache_last_key_layout.restype = C.c_char_p
    L.ieache_default_params.argtypes = [pp]
    L.ieache_cloud_run.argtypes = [C.c_char_p]
    L.ieache_ctx_create.restype = vp
    L.ieache_ctx_create.argtypes = [C.c_char_p, C.c_int]
    L.ieache_ctx_create_raw.restype = vp
    L.ieache_ctx_create_raw.argtypes = [pp, i32p, i32p, C.c_int]
    L.ieache_ctx_create_device.restype = vp
    L.ieache_ctx_create_device.argtypes = [pp, vp, vp, C.c_int]
    L.ieache_ctx_destroy.argtypes = [vp]
    L.ieache_ctx_params.argtypes = [vp, pp]
    L.ieache_lwe_stride.argtypes = [vp]
    L.ieache_ctx_stream.restype = vp
    L.ieache_ctx_stream.argtypes = [vp]
    L.ieache_ctx_cloud_run.argtypes = [vp, C.c_char_p]
    L.ieache_ctx_set_chunk.argtypes = [vp, C.c_int64]
    L.ieache_ctx_force_generic.argtypes = [vp, C.c_int]
    L.ieache_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.ieache_ctx_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64)]
    L.ieache_ctx_fft_guard.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.ieache_ctx_fft_audit.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.ieache_ctx_kernel_variant.restype = C.c_char_p
    L.ieache_ctx_kernel_variant.argtypes = [vp]
    L.ieache_ctx_kernel_for_launch.restype = C.c_char_p
    L.ieache_ctx_kernel_for_launch.argtypes = [vp, C.c_int64]
    L.ieache_circuit_info_get.argtypes = [C.c_int, C.c_int, C.POINTER(CircuitInfo)]
    L.ieache_circuit_info_get_ex.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(CircuitInfo)]
    L.ieache_circuit_level_cap.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int]
    L.ieache_ctx_circuit_level_cap.argtypes = [vp, C.c_int, C.c_int, C.c_int64]
    L.ieache_circuit_info_get_cap.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(CircuitInfo)]
    L.ieache_circuit_simulate_cap.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, u8p, u8p]
    L.ieache_circuit_simulate.argtypes = [C.c_int, C.c_int, u8p, u8p]
    L.ieache_circuit_simulate_ex.argtypes = [C.c_int, C.c_int, C.c_int, u8p, u8p]
    L.ieache_ctx_wait_stream.argtypes = [vp, vp]
    L.ieache_mux_device.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, sp]
    L.ieache_mux.argtypes = [vp, C.c_size_t, i32p, i32p, i32p, i32p, sp]
    L.ieache_eval_batch.argtypes = [vp, C.c_int, C.c_int, C.c_size_t, i32p, i32p, sp]
    L.ieache_eval_batch_device.argtypes = [vp, C.c_int, C.c_int, C.c_size_t, vp, vp, sp]
    L.ieache_prepare_batch.argtypes = [vp, C.c_int, C.c_int, C.c_size_t]
    L.ieache_gates_device.argtypes = [vp, C.c_int, C.c_size_t, vp, vp, vp, sp]
    L.ieache_gates.argtypes = [vp, C.c_int, C.c_size_t, i32p, i32p, i32p, sp]
    L.ieache_debug_blind_rotate.argtypes = [vp, C.c_size_t, i32p, i32p, C.c_int32]
    L.ieache_debug_keyswitch.argtypes = [vp, C.c_size_t, i32p, i32p]
    L.ieache_keygen_raw.argtypes = [pp, u32p, C.c_int, i32p, i32p, i32p, i32p]
    L.ieache_keygen_files.argtypes = [C.c_char_p, pp, u32p, C.c_int, u32p, C.c_int]
    L.ieache_encrypt_bits.argtypes = [pp, i32p, u8p, C.c_size_t, C.c_uint64, i32p]
    L.ieache_decrypt_bits.argtypes = [pp, i32p, i32p, C.c_size_t, u8p]
    L.ieache_read_secret_key.argtypes = [C.c_char_p, pp, i32p, i32p]
    L.ieache_read_cloud_key.argtypes = [C.c_char_p, pp, i32p, i32p]
    L.ieache_write_cloud_key.argtypes = [C.c_char_p, pp, i32p, i32p]
    L.ieache_write_secret_key.argtypes = [C.c_char_p, pp, i32p, i32p, i32p, i32p]
    L.ieache_read_samples.argtypes = [C.c_char_p, C.c_int32, C.c_size_t, C.c_size_t, i32p]
    L.ieache_write_samples.argtypes = [C.c_char_p, C.c_int32, C.c_size_t, i32p, C.c_int]
    L.ieache_alice.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_uint32, C.c_uint32, u32p, C.c_uint64]
    L.ieache_verif.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, u32p, u32p, u32p]
    L.ieache_serve.restype = C.c_int64
    L.ieache_serve.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int64]
    L.ieache_serve_devices.restype = C.c_int64
    L.ieache_serve_devices.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.c_int, C.c_int64]
    L.ieache_debug_mix_plan.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.ieache_shard_slice.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.ieache_client_ping.argtypes = [C.c_char_p]
    L.ieache_client_run_dir.argtypes = [C.c_char_p, C.c_char_p]
    L.ieache_client_run_data.argtypes = [C.c_char_p, C.c_int, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.ieache_client_shutdown.argtypes = [C.c_char_p]
    _LIB = L
    return L


def check(rc):
    if rc < 0:
        raise IeacheError(rc, lib().ieache_last_error().decode())
    return rc


def _i32(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def default_params():
    p = Params()
    lib().ieache_default_params(C.byref(p))
    return p


def device_count():
    return lib().ieache_device_count()


def circuit_info(kind, bits, fold=False, level_cap=0):
    info = CircuitInfo()
    check(lib().ieache_circuit_info_get_cap(kind, bits, int(fold), int(level_cap), C.byref(info)))
    return info


def circuit_level_cap(kind, bits, batch, resident_workgroups=1024, fold=False):
    """Level width for `batch` expressions on a GPU that holds `resident_workgroups` blind rotations at once ("level_quantum");
    0 = the default schedule.  What a given context picks (its device's residency, both kernels): Context.circuit_level_cap."""
    return check(lib().ieache_circuit_level_cap(kind, bits, int(fold), int(batch), int(resident_workgroups)))


def circuit_simulate(kind, bits, in_bits, fold=False, level_cap=0):
    """Plaintext run of the levelised, slot-allocated circuit (host only)."""
    info = circuit_info(kind, bits, fold)
    in_bits = np.ascontiguousarray(in_bits, dtype=np.uint8)
    assert in_bits.shape == (info.n_inputs,)
    out = np.zeros(info.n_outputs, dtype=np.uint8)
    check(lib().ieache_circuit_simulate_cap(kind, bits, int(fold), int(level_cap), in_bits.ctypes.data_as(C.POINTER(C.c_uint8)),
                                            out.ctypes.data_as(C.POINTER(C.c_uint8))))
    return out


class Context:
    """Cloud key resident on one GPU (ieache_ctx)."""

    def __init__(self, handle):
        if not handle:
            raise IeacheError(-19, lib().ieache_last_error().decode())
        self.h = handle
        self.params = Params()
        check(lib().ieache_ctx_params(self.h, C.byref(self.params)))

    @classmethod
    def from_file(cls, cloud_key_path, device=0):
        return cls(lib().ieache_ctx_create(os.fsencode(cloud_key_path), device))

    @classmethod
    def from_arrays(cls, params, bk, ksk, device=0):
        bk = np.ascontiguousarray(bk, dtype=np.int32)
        ksk = np.ascontiguousarray(ksk, dtype=np.int32)
        assert bk.size == params.bk_count and ksk.size == params.ksk_count
        return cls(lib().ieache_ctx_create_raw(C.byref(params), _i32(bk), _i32(ksk), device))

    @classmethod
    def from_device_pointers(cls, params, d_bk, d_ksk, device=0):
        """d_bk / d_ksk: integer device addresses (e.g. torch tensor .data_ptr())."""
        return cls(lib().ieache_ctx_create_device(C.byref(params), C.c_void_p(d_bk), C.c_void_p(d_ksk), device))

    def close(self):
        if getattr(self, "h", None):
            lib().ieache_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def lwe_stride(self):
        return lib().ieache_lwe_stride(self.h)

    @property
    def stream(self):
        return lib().ieache_ctx_stream(self.h)

    @property
    def kernel_variant(self):
        return lib().ieache_ctx_kernel_variant(self.h).decode()

    def kernel_for_launch(self, gates):
        """Name of the blind-rotation kernel a launch of `gates` gate instances takes."""
        return lib().ieache_ctx_kernel_for_launch(self.h, int(gates)).decode()

    def fft_guard(self):
        """(largest distance to an integer the one-limb blind rotation rounded away -- 0.5 would be a wrong bit --,
        calls repeated on the two-limb kernel because a launch exceeded 1/16)."""
        m, r = C.c_double(0), C.c_int64(0)
        check(lib().ieache_ctx_fft_guard(self.h, C.byref(m), C.byref(r)))
        return m.value, r.value

    def circuit_level_cap(self, kind, bits, batch):
        """Level width this context's eval_batch* uses for `batch` expressions; 0 = the default schedule."""
        return check(lib().ieache_ctx_circuit_level_cap(self.h, kind, bits, int(batch)))

    def fft_audit(self):
        """The sampled bit-for-bit audit of the one-limb kernel against the two-limb one (option "fft_audit" = K):
        {"audits": launches audited, "gates_compared": gate instances re-run and compared, "mismatches": rows that differed}."""
        a, g, m = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(lib().ieache_ctx_fft_audit(self.h, C.byref(a), C.byref(g), C.byref(m)))
        return {"audits": a.value, "gates_compared": g.value, "mismatches": m.value}

    def set_chunk(self, items):
        check(lib().ieache_ctx_set_chunk(self.h, items))

    def set_option(self, name, value):
        """Tuning knobs: chunk, force_generic, ks_batch_min, br_slice, ..., fold_constants."""
        check(lib().ieache_ctx_set_option(self.h, name.encode(), int(value)))
        if name == "fold_constants":
            self._fold = bool(value)

    def set_option_ok(self, name, value):
        """set_option without raising: False when the name is unknown or the value out of range (nothing is changed then)."""
        return lib().ieache_ctx_set_option(self.h, name.encode(), int(value)) == 0

    def get_option(self, name):
        """Current value of an option, or of a read-only figure ("cus", "resident_gates", "overlapped_levels")."""
        v = C.c_int64(0)
        check(lib().ieache_ctx_get_option(self.h, name.encode(), C.byref(v)))
        return v.value

    def wait_stream(self, hip_stream=None):
        """Order the context's stream after the work queued on `hip_stream` (int handle; None = default stream)."""
        check(lib().ieache_ctx_wait_stream(self.h, C.c_void_p(hip_stream)))

    def force_generic(self, on=True):
        check(lib().ieache_ctx_force_generic(self.h, int(on)))

    def cloud_run(self, workdir):
        return check(lib().ieache_ctx_cloud_run(self.h, os.fsencode(workdir)))

    def eval_batch(self, kind, bits, in_lwe, stats=None):
        """in_lwe [batch][n_inputs][n+1] int32 on the host -> [batch][n_outputs][n+1]."""
        info = circuit_info(kind, bits, getattr(self, "_fold", False))
        in_lwe = np.ascontiguousarray(in_lwe, dtype=np.int32)
        batch = in_lwe.shape[0]
        assert in_lwe.shape == (batch, info.n_inputs, self.params.n + 1), in_lwe.shape
        out = np.zeros((batch, info.n_outputs, self.params.n + 1), dtype=np.int32)
        check(lib().ieache_eval_batch(self.h, kind, bits, batch, _i32(in_lwe), _i32(out),
                                      C.byref(stats) if stats is not None else None))
        return out

    def prepare(self, kind, bits, batch):
        """Allocates what eval_batch* of this circuit and batch needs (circuit tables, wire store, scratch), so that a timed
        first evaluation measures the steady state."""
        check(lib().ieache_prepare_batch(self.h, kind, bits, batch))

    def eval_batch_device(self, kind, bits, batch, d_in, d_out, stats=None):
        """Device pointers (ints); rows of lwe_stride int32."""
        check(lib().ieache_eval_batch_device(self.h, kind, bits, batch, C.c_void_p(d_in), C.c_void_p(d_out),
                                             C.byref(stats) if stats is not None else None))

    def gates(self, gate_type, a, b, stats=None):
        a = np.ascontiguousarray(a, dtype=np.int32)
        b = np.ascontiguousarray(b, dtype=np.int32)
        assert a.shape == b.shape and a.shape[-1] == self.params.n + 1
        out = np.zeros_like(a)
        count = a.size // (self.params.n + 1)
        check(lib().ieache_gates(self.h, gate_type, count, _i32(a), _i32(b), _i32(out),
                                 C.byref(stats) if stats is not None else None))
        return out

    def gates_device(self, gate_type, count, d_a, d_b, d_out, stats=None):
        check(lib().ieache_gates_device(self.h, gate_type, count, C.c_void_p(d_a), C.c_void_p(d_b), C.c_void_p(d_out),
                                        C.byref(stats) if stats is not None else None))

    def mux(self, a, b, c, stats=None):
        """bootsMUX on host rows: out[i] = a[i] ? b[i] : c[i]."""
        a, b, c = (np.ascontiguousarray(v, dtype=np.int32) for v in (a, b, c))
        assert a.shape == b.shape == c.shape and a.shape[-1] == self.params.n + 1
        out = np.zeros_like(a)
        check(lib().ieache_mux(self.h, a.size // (self.params.n + 1), _i32(a), _i32(b), _i32(c), _i32(out),
                               C.byref(stats) if stats is not None else None))
        return out

    def mux_device(self, count, d_a, d_b, d_c, d_out, stats=None):
        check(lib().ieache_mux_device(self.h, count, C.c_void_p(d_a), C.c_void_p(d_b), C.c_void_p(d_c), C.c_void_p(d_out),
                                      C.byref(stats) if stats is not None else None))

    def debug_blind_rotate(self, x, steps=-1):
        x = np.ascontiguousarray(x, dtype=np.int32).reshape(-1, self.params.n + 1)
        acc = np.zeros((x.shape[0], 2, self.params.N), dtype=np.int32)
        check(lib().ieache_debug_blind_rotate(self.h, x.shape[0], _i32(x), _i32(acc), steps))
        return acc

    def debug_keyswitch(self, u):
        u = np.ascontiguousarray(u, dtype=np.int32).reshape(-1, self.params.N + 1)
        out = np.zeros((u.shape[0], self.params.n + 1), dtype=np.int32)
        check(lib().ieache_debug_keyswitch(self.h, u.shape[0], _i32(u), _i32(out)))
        return out
