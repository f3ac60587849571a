"""MI355X-native evaluator for the IE-ACHE Cloud path (Cloud/cloud.c + libtfhe
gate bootstrapping).  The product is the HIP shared library `libieache.so`
(C ABI: include/ieache.h); this package is the thin ctypes host side mirroring
the reference's Python caller (Cloud/dragonfly_cipher_cloud.py:1219-1327).
"""
from .evaluator import (  # noqa: F401
    CIRC_ADD, CIRC_SUB, CIRC_RSUB, CIRC_MUL, CIRC_MULADD, CIRC_ADD_KS, CIRC_SUB_KS, CIRC_RSUB_KS, CIRC_MUL_WALLACE,
    GATE_AND, GATE_XOR, GATE_OR, GATE_NAND, GATE_MUX, circ_chain,
    CircuitInfo, Context, IeacheError, Params, Stats, build_library, circuit_info, circuit_level_cap, circuit_simulate,
    default_params, device_count, lib, library_path,
)
from . import tools  # noqa: F401
from .compute import compute, compute_final, FAILURE_SIZE  # noqa: F401
