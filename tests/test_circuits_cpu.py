"""Host logic: the gate DAG of cloud.c's circuits (no GPU, no ciphertexts).

Statistics are SURVEY.md App. C (replayed from Cloud/cloud.c); plaintext
semantics are pinned by integer arithmetic and the reference's canned operands
(Client1/process.c:94-99,122-129,152-163,185-204: value = 2^(bits-2))."""
import os

import numpy as np
import pytest

APP_C = {  # (kind, bits): (bootstraps, and, xor, depth, max_width)
    (1, 16): (80, 16, 64, 48, 2),
    (1, 32): (160, 32, 128, 96, 2),
    (1, 64): (320, 64, 256, 192, 2),
    (1, 128): (640, 128, 512, 384, 2),
    (1, 256): (1280, 256, 1024, 768, 2),
    (2, 32): (320, 64, 256, 98, 4),
    (2, 64): (640, 128, 512, 194, 4),
    (2, 128): (1280, 256, 1024, 386, 4),
    (2, 256): (2560, 512, 2048, 770, 4),
    (3, 32): (320, 64, 256, 98, 4),
    (4, 32): (11264, 3072, 8192, 255, 1056),
    (4, 64): (35296, 10336, 24960, 449, 4160),
    (4, 128): (121184, 37344, 83840, 1601, 16512),
    (5, 64): (35936, 10464, 25472, 451, 4161),
    # fused chains: stage 1 + stage 2 of SURVEY App. C, e.g. BASELINE.md's A*B*C =
    # 32-bit MUL (11 264) then 64-bit MUL (35 296) = 46 560; A+B*C at 32 bits = 11 264 + 64-bit ADD (320)
    (5, 32): (11584, 3136, 8448, 257, 1057),
    (5, 128): (122464, 37600, 84864, 1603, 16513),
}
CHAIN_COUNTS = {  # (k1, k2, bits): bootstraps = stage 1 + stage 2 at stage 2's width
    (1, 1, 32): 160 + 160, (1, 2, 32): 160 + 320, (2, 2, 32): 320 + 320, (2, 1, 64): 640 + 320,
    (4, 1, 32): 11264 + 320, (4, 2, 32): 11264 + 640, (4, 4, 32): 11264 + 35296, (4, 4, 64): 35296 + 121184,
    (1, 4, 32): 160 + 11264,
}


@pytest.mark.parametrize("key", sorted(APP_C))
def test_circuit_statistics_match_reference(ia, key):
    info = ia.circuit_info(*key)
    assert (info.bootstraps, info.n_and, info.n_xor, info.depth, info.max_width) == APP_C[key]
    kind, bits = key
    assert info.n_outputs == (2 * bits if kind in (4, 5) else bits)


def _run(ia, kind, bits, a, b, c=0, carry_bits=0):
    from ieache_amd.tools import int_to_bits, bits_to_int
    info = ia.circuit_info(kind, bits)
    x = np.zeros(info.n_inputs, dtype=np.uint8)
    x[:bits] = int_to_bits(a, bits)
    x[bits:2 * bits] = int_to_bits(b, bits)
    x[2 * bits:2 * bits + 32] = int_to_bits(carry_bits, 32)
    if kind == 5:
        x[2 * bits + 32:] = int_to_bits(c, 2 * bits)
    return bits_to_int(ia.circuit_simulate(kind, bits, x))


@pytest.mark.parametrize("bits", [16, 32, 64, 128, 256])
def test_add_sub_plaintext(ia, bits):
    rng = np.random.default_rng(bits)
    m = 1 << bits
    cases = [(0, 0), (1, m - 1), (m - 1, m - 1), (1 << (bits - 2), 1 << (bits - 2))]  # last: process.c operands
    cases += [(int.from_bytes(rng.bytes(bits // 8), "little"), int.from_bytes(rng.bytes(bits // 8), "little")) for _ in range(6)]
    for a, b in cases:
        assert _run(ia, 1, bits, a, b) == (a + b) % m          # final carry dropped (cloud.c:891-916)
        assert _run(ia, 2, bits, a, b) == (a - b) % m          # A + (~B + 1)
        assert _run(ia, 3, bits, a, b) == (b - a) % m          # B + (~A + 1)
    # the carry word's bit 0 is the carry-in (cloud.c:24); a client always sends 0 there
    assert _run(ia, 1, bits, 5, 6, carry_bits=1) == 12


@pytest.mark.parametrize("bits", [32, 64, 128])
def test_mul_plaintext(ia, bits):
    rng = np.random.default_rng(100 + bits)
    m = 1 << bits
    cases = [(0, 0), (1, m - 1), (m - 1, m - 1), (1 << (bits - 2), 1 << (bits - 2))]
    n_random = {32: 6, 64: 3, 128: 1}[bits]
    cases += [(int.from_bytes(rng.bytes(bits // 8), "little"), int.from_bytes(rng.bytes(bits // 8), "little")) for _ in range(n_random)]
    for a, b in cases:
        assert _run(ia, 4, bits, a, b) == a * b
    # process.c known answers: 2^30*2^30 = 2^60 -> words {0, 0x10000000}; 2^62^2 = 2^124; 2^126^2 = 2^252
    assert _run(ia, 4, bits, 1 << (bits - 2), 1 << (bits - 2)) == 1 << (2 * bits - 4)


@pytest.mark.parametrize("bits", [32, 64, 128])
def test_muladd_plaintext(ia, bits):
    rng = np.random.default_rng(5 + bits)
    for _ in range({32: 3, 64: 3, 128: 1}[bits]):
        a, b = (int.from_bytes(rng.bytes(bits // 8), "little") for _ in range(2))
        c = int.from_bytes(rng.bytes(bits // 4), "little")
        assert _run(ia, 5, bits, a, b, c) == (a * b + c) % (1 << (2 * bits))
    assert ia.circ_chain(ia.CIRC_MUL, ia.CIRC_ADD) != 5  # MULADD keeps its own code, same DAG
    i5, ic = ia.circuit_info(5, bits), ia.circuit_info(ia.circ_chain(ia.CIRC_MUL, ia.CIRC_ADD), bits)
    assert (i5.bootstraps, i5.depth, i5.n_inputs) == (ic.bootstraps, ic.depth, ic.n_inputs)


def _run_chain(ia, k1, k2, flip, bits, a, b, c, fold=False):
    """plaintext run of IEACHE_CIRC_CHAIN(k1, k2, flip): inputs A, B, carry word, C [, C's carry word]"""
    from ieache_amd.tools import int_to_bits, bits_to_int
    kind = ia.circ_chain(k1, k2, flip)
    info = ia.circuit_info(kind, bits, fold)
    w2 = 2 * bits if k1 == 4 else bits
    assert info.n_inputs == 2 * bits + 32 + w2 + (0 if flip else 32)
    assert info.n_outputs == (2 * w2 if k2 == 4 else w2)
    x = np.zeros(info.n_inputs, dtype=np.uint8)
    x[:bits] = int_to_bits(a, bits)
    x[bits:2 * bits] = int_to_bits(b, bits)
    x[2 * bits + 32:2 * bits + 32 + w2] = int_to_bits(c, w2)
    return bits_to_int(ia.circuit_simulate(kind, bits, x, fold)), info


def _stage(k, x, y, m):
    return {1: (x + y) % m, 2: (x - y) % m, 3: (y - x) % m, 4: x * y}[k]


@pytest.mark.parametrize("fold", [False, True])
def test_chained_operators_plaintext(ia, fold):
    """compute() + compute_final() (dragonfly_cipher_cloud.py:1219-1327) fused: every operator pair the paper
    times (AC058.pdf Fig. 7: A+B+C, A+B-C, A+B*C, A-B*C, A-B-C, A*B*C) and both operand orders."""
    rng = np.random.default_rng(77)
    for bits in (32, 64):
        m = 1 << bits
        a, b = (int.from_bytes(rng.bytes(bits // 8), "little") for _ in range(2))
        for k1 in (1, 2, 3, 4):
            w2 = 2 * bits if k1 == 4 else bits
            c = int.from_bytes(rng.bytes(w2 // 8), "little")
            for k2 in (1, 2, 3, 4):
                if k2 == 4 and (w2 > 128 or (bits == 64 and k1 != 4)):
                    continue  # keep the CPU suite short: one 128-bit stage-2 multiplier is enough
                for flip in (True, False):
                    s1 = _stage(k1, a, b, m)
                    exp = _stage(k2, s1, c, 1 << w2) if flip else _stage(k2, c, s1, 1 << w2)
                    got, info = _run_chain(ia, k1, k2, flip, bits, a, b, c, fold)
                    assert got == exp, (bits, k1, k2, flip)
                    if not fold and (k1, k2, bits) in CHAIN_COUNTS:
                        assert info.bootstraps == info.reference_bootstraps == CHAIN_COUNTS[(k1, k2, bits)]


def test_constant_folded_circuits(ia):
    """Opt-in folding (SURVEY App. C note): same plaintext function, fewer bootstraps, reference count kept."""
    from ieache_amd.tools import int_to_bits, bits_to_int
    rng = np.random.default_rng(31)
    for kind, bits, min_gain in ((1, 32, 1.0), (2, 32, 1.2), (3, 64, 1.2), (4, 32, 1.4), (4, 64, 1.2), (5, 32, 1.4), (6, 32, 1.0)):
        plain, folded = ia.circuit_info(kind, bits), ia.circuit_info(kind, bits, fold=True)
        assert plain.folded == 0 and folded.folded == 1
        assert plain.reference_bootstraps == plain.bootstraps == folded.reference_bootstraps
        assert folded.bootstraps * min_gain <= plain.bootstraps and folded.depth <= plain.depth
        assert folded.n_and + folded.n_xor == folded.bootstraps
        for _ in range(4):
            x = rng.integers(0, 2, size=plain.n_inputs, dtype=np.uint8)
            x[2 * bits:2 * bits + 32] = 0
            assert np.array_equal(ia.circuit_simulate(kind, bits, x), ia.circuit_simulate(kind, bits, x, fold=True))
    # the carry word's bit 0 is still a live carry-in after folding
    x = np.zeros(96, dtype=np.uint8)
    x[:32], x[32:64], x[64] = int_to_bits(5, 32), int_to_bits(6, 32), 1
    assert bits_to_int(ia.circuit_simulate(1, 32, x, fold=True)) == 12
    assert ia.circuit_info(4, 32, fold=True).bootstraps == 7568  # regression pin: 11 264 in the reference


@pytest.mark.parametrize("bits", [32, 64, 128])
def test_carry_save_multiplier_plaintext(ia, bits):
    """Opt-in CIRC_MUL_WALLACE: the same product as cloud.c's shift-add multipliers from a Dadda carry-save tree and
    one Kogge-Stone addition -- XOR/AND only, a fraction of the depth, fewer bootstraps."""
    from ieache_amd.tools import int_to_bits, bits_to_int
    rng = np.random.default_rng(2000 + bits)
    info, ref = ia.circuit_info(ia.CIRC_MUL_WALLACE, bits), ia.circuit_info(ia.CIRC_MUL, bits)
    assert info.n_inputs == ref.n_inputs and info.n_outputs == ref.n_outputs == 2 * bits
    assert info.reference_bootstraps == ref.bootstraps and info.bootstraps < 0.85 * ref.bootstraps
    assert info.depth <= {32: 40, 64: 46, 128: 54}[bits] and info.n_and + info.n_xor == info.bootstraps
    m = 1 << bits
    cases = [(0, 0), (1, m - 1), (m - 1, m - 1), (1 << (bits - 2), 1 << (bits - 2)), (m - 1, 0)]
    cases += [(int.from_bytes(rng.bytes(bits // 8), "little"), int.from_bytes(rng.bytes(bits // 8), "little")) for _ in range(5)]
    for a, b in cases:
        x = np.zeros(info.n_inputs, dtype=np.uint8)
        x[:bits], x[bits:2 * bits] = int_to_bits(a, bits), int_to_bits(b, bits)
        x[2 * bits:] = rng.integers(0, 2, size=32)  # the carry word plays no part in this circuit
        assert bits_to_int(ia.circuit_simulate(ia.CIRC_MUL_WALLACE, bits, x)) == a * b


def test_unsupported_circuits_rejected(ia):
    bad = [(4, 256), (4, 16), (5, 16), (5, 256), (9, 16), (9, 256), (10, 32), (1, 0), (1, 257), (64, 32),
           (ia.circ_chain(4, 4), 128),   # stage 2 would be a 256-bit MUL: cloud.c:860-864 exits 126
           (ia.circ_chain(1, 4), 256), (ia.circ_chain(4, 1), 48)]
    for kind, bits in bad:
        with pytest.raises(ia.IeacheError):
            ia.circuit_info(kind, bits)


def test_slack_balanced_schedule(ia):
    """The executor's schedule keeps the ASAP depth but spreads gates with slack over the levels."""
    for (kind, bits), (maxw, slots) in {(4, 64): (160, 2000), (4, 128): (160, 4000), (5, 64): (160, 2000)}.items():
        i = ia.circuit_info(kind, bits)
        assert i.depth == APP_C[(kind, bits)][3]                 # same number of levels as ASAP
        assert i.sched_max_width <= maxw < i.max_width           # but no 1000+-gate level 1 any more
        assert i.sched_max_width >= -(-i.bootstraps // i.depth)  # at least the mean width
        assert i.n_slots <= slots
    i = ia.circuit_info(1, 32)
    assert i.sched_max_width == 2  # a ripple adder has no slack: schedule == ASAP
    i = ia.circuit_info(4, 32)
    assert i.sched_max_width == i.max_width == 1056  # mul32 keeps ASAP levels (measured faster, store is small)


def test_batch_aware_level_width(ia):
    """"level_quantum": for the slack-balanced circuits a context picks the level width that makes level x batch a
    whole number of resident-workgroup rounds (1 024 on an MI355X: 4 per CU).  Same DAG, same outputs; more levels,
    each exactly full, when the batch is small."""
    rng = np.random.default_rng(8)
    assert ia.circuit_level_cap(4, 128, 16) == 64        # 64 x 16 = one round; the mean width 76 would cost two
    assert ia.circuit_level_cap(5, 64, 128) == 80        # 80 x 128 = ten rounds: what the default schedule already does
    assert ia.circuit_level_cap(4, 128, 1024) == 0       # a whole round per gate of a level: nothing to quantise
    assert ia.circuit_level_cap(4, 128, 2) == 0          # under one round per level: narrow levels are the cheap ones
    # the ASAP-scheduled 32-bit multiplier family at small batches: floor(m x resident / batch) gates per expression, m = the
    # whole number of rounds nearest its mean level (44 gates per expression); nothing for adders, the shallow carry-save
    # trees, or batches whose mean level is already three rounds or more
    assert ia.circuit_level_cap(4, 32, 58, 2048) == 35 and ia.circuit_level_cap(4, 32, 64, 2048) == 32
    assert ia.circuit_level_cap(5, 32, 100, 2048) == 40  # two rounds per level
    assert ia.circuit_level_cap(4, 32, 16, 2048) == 0 and ia.circuit_level_cap(4, 32, 256, 2048) == 0
    assert ia.circuit_level_cap(4, 32, 1024, 2048) == 0  # BASELINE configs[2] keeps cloud.c's ASAP levels
    assert ia.circuit_level_cap(1, 32, 58, 2048) == 0 and ia.circuit_level_cap(ia.CIRC_MUL_WALLACE, 32, 8, 2048) == 0
    c35 = ia.circuit_info(4, 32, level_cap=35)
    assert c35.sched_levels == 334 and c35.sched_max_width <= 36 and c35.bootstraps == 11264
    x = rng.integers(0, 2, size=ia.circuit_info(4, 32).n_inputs, dtype=np.uint8)
    assert np.array_equal(ia.circuit_simulate(4, 32, x), ia.circuit_simulate(4, 32, x, level_cap=35))
    assert ia.circuit_level_cap(ia.CIRC_MUL_WALLACE, 128, 16) == 0  # no slack to flatten
    assert ia.circuit_level_cap(4, 64, 100) == 0         # quantum 256 is beyond what a level offers
    base, capped = ia.circuit_info(4, 64), ia.circuit_info(4, 64, level_cap=64)
    assert (base.sched_levels, base.level_cap) == (449, 0) and capped.level_cap == 64
    assert capped.bootstraps == base.bootstraps and capped.depth == base.depth == 449
    assert -(-base.bootstraps // 64) <= capped.sched_levels <= 1.1 * base.bootstraps / 64
    assert capped.sched_max_width <= 70 and capped.n_slots <= base.n_slots
    for _ in range(3):
        x = rng.integers(0, 2, size=base.n_inputs, dtype=np.uint8)
        assert np.array_equal(ia.circuit_simulate(4, 64, x), ia.circuit_simulate(4, 64, x, level_cap=64))
    x = rng.integers(0, 2, size=ia.circuit_info(5, 64).n_inputs, dtype=np.uint8)
    assert np.array_equal(ia.circuit_simulate(5, 64, x), ia.circuit_simulate(5, 64, x, level_cap=64))
    assert ia.circuit_info(1, 32, level_cap=64).sched_levels == 96   # ignored where there is nothing to balance


@pytest.mark.parametrize("bits", [1, 5, 16, 32, 64, 256])
def test_kogge_stone_adders_plaintext(ia, bits):
    """XOR/AND-only parallel-prefix adders decrypt like the ripple ones."""
    rng = np.random.default_rng(900 + bits)
    m = 1 << bits
    nbytes = (bits + 7) // 8
    cases = [(0, 0), (m - 1, 1), (m - 1, m - 1), (1, m - 1)]
    cases += [(int.from_bytes(rng.bytes(nbytes), "little") % m, int.from_bytes(rng.bytes(nbytes), "little") % m) for _ in range(8)]
    for a, b in cases:
        assert _run(ia, ia.CIRC_ADD_KS, bits, a, b) == (a + b) % m
        assert _run(ia, ia.CIRC_SUB_KS, bits, a, b) == (a - b) % m
        assert _run(ia, ia.CIRC_RSUB_KS, bits, a, b) == (b - a) % m
    assert _run(ia, ia.CIRC_ADD_KS, bits, 0, 0, carry_bits=1) == 1 % m  # carry-in honoured like cloud.c:24
    if bits >= 16:
        ks, rc = ia.circuit_info(ia.CIRC_ADD_KS, bits), ia.circuit_info(ia.CIRC_ADD, bits)
        assert ks.depth < rc.depth // 3 and ks.n_and + ks.n_xor == ks.bootstraps  # only the reference's gate types


def test_twisted_radix8_passes_match_their_defining_sums(tmp_path):
    """The radix-8 passes that carry the register part of the negacyclic twist (csrc/dft8_twist.h, compiled into every
    blind-rotation kernel through fft512.h) are plain arithmetic: on the host they must equal the sums they stand for --
    X_k = sum_r y_r e^{i pi r/16} W8^{rk} forward, its conjugate transpose with the pending real gains inverse."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "dft8_twist_test"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(root, "ie-ache_amd", "csrc"),
                           os.path.join(root, "tests", "native", "dft8_twist_test.cpp"), "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "forward" in r.stdout and "inverse" in r.stdout


def test_rotation_of_roles_plan(ia):
    """csrc/mix_plan.h through ieache_debug_mix_plan: which mid-size launches run as a rotation of roles between the two-waves- and
    the one-wave-per-gate kernel (4 .. 7 gates per CU, and 8 .. 10.5: a full round plus a small remainder), with three subsets of
    which two are on two waves at a time, and that the rounds never reach the end of the rotation (the ordinary loop needs at
    least one step: it extracts).  256 CUs."""
    import ctypes as C
    L = ia.lib()

    def plan(gates, cus=256, n=630, s1=16, ratio=200):
        out = (C.c_int * 9)()
        used = L.ieache_debug_mix_plan(cus, n, gates, s1, ratio, out)
        return used, list(out)

    for gates in (1, 1024, 1793, 2048, 2689, 4096, 16384):  # single kernels outside the two ranges
        assert plan(gates)[0] == 0, gates
    for gates in list(range(1025, 1793)) + list(range(2049, 2689)):
        used, (k, tw, s1, s2, cycles, t1, t2, covered, per) = plan(gates)
        assert used == 1 and (k, tw) == (3, 2) and (s1, s2) == (16, 32), gates
        assert per % 12 == 0 and per * k >= gates > per * (k - 1)                                   # whole workgroups, no empty subset
        rnd = tw * s2 + (k - tw) * s1
        assert cycles >= 1 and covered == cycles * rnd + (tw * t2 + (k - tw) * t1) and covered < 630
        assert (t1 == 0 and t2 == 0) or (t1 >= 4 and t2 >= 4)
        assert 630 - covered <= 3 * k + 8, (gates, covered)                                          # the ordinary loop is left with little
    # other devices, other rotations: the thresholds are per CU; a rotation too short for one round is left to the single kernels
    assert plan(4 * 304 + 1, cus=304)[0] == 1 and plan(4 * 304, cus=304)[0] == 0 and plan(7 * 304 + 1, cus=304)[0] == 0
    assert plan(8 * 304 + 1, cus=304)[0] == 1 and plan(1300, n=40)[0] == 0
    used, out = plan(1300, n=500, s1=64, ratio=188)
    assert used == 1 and out[:5] == [3, 2, 64, 120, 1] and out[7] < 500
