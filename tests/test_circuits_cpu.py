"""Host logic: the gate DAG of cloud.c's circuits (no GPU, no ciphertexts).

Statistics are SURVEY.md App. C (replayed from Cloud/cloud.c); plaintext
semantics are pinned by integer arithmetic and the reference's canned operands
(Client1/process.c:94-99,122-129,152-163,185-204: value = 2^(bits-2))."""
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


def test_muladd_plaintext(ia):
    rng = np.random.default_rng(5)
    for _ in range(3):
        a, b = (int.from_bytes(rng.bytes(8), "little") for _ in range(2))
        c = int.from_bytes(rng.bytes(16), "little")
        assert _run(ia, 5, 64, a, b, c) == (a * b + c) % (1 << 128)


def test_unsupported_circuits_rejected(ia):
    for kind, bits in [(4, 256), (4, 16), (5, 32), (9, 32), (1, 0), (1, 257)]:
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
