"""Blind rotation alone by launch size and gates per workgroup of the one-wave-per-gate kernels ("wg_gates" 4 / 3 / 2 / 1), against
the two-waves-per-gate kernel where the launch size allows it: ms per rotation (best of REPS), bit identity with the two-limb kernel.
usage: wg_sweep.py [count ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ieache_amd as ia
from ieache_amd import tools
counts = [int(c) for c in sys.argv[1:]] or [1100, 1216, 1280, 1400, 1536, 1792, 2048]
p = ia.default_params()
k = tools.keygen_raw(p, (1, 2, 3))
ctx = ia.Context.from_arrays(p, k["bk"], k["ksk"])
rng = np.random.default_rng(5)
cmax = max(counts)
bits = rng.integers(0, 2, size=(2, cmax)).astype(np.uint8)
a = tools.encrypt_bits(p, k["lwe_key"], bits[0], 11)
b = tools.encrypt_bits(p, k["lwe_key"], bits[1], 12)
ctx.set_option("exact_fft", 1)
ref = ctx.gates(ia.GATE_XOR, a, b)
ctx.set_option("exact_fft", 0)
ctx.set_option("overlap", 0)
reps = int(os.environ.get("REPS", "5"))
cus = ctx.get_option("cus")
exact = int(os.environ.get("EXACT_FFT", "0"))
ctx.set_option("exact_fft", exact)


def run(c, label):
    best = None
    for _ in range(reps):
        st = ia.Stats()
        out = ctx.gates(ia.GATE_XOR, a[:c], b[:c], st)
        best = st.blind_rotate_ms if best is None else min(best, st.blind_rotate_ms)
    print("%5d gates %-34s BR %.3f ms in %d launches (%.0f gates/s) identical %s" % (c, label + " " + ctx.kernel_for_launch(c).split("<")[0], best, st.blind_rotate_launches, c / best * 1e3, np.array_equal(ref[:c], out)), flush=True)


for c in counts:
    if not exact:
        ctx.set_option("two_wave_max", 1 << 30)   # two waves per gate whatever the size (several rounds of 4 per CU)
        run(c, "two waves/gate")
        ctx.set_option("two_wave_max", 0)
    for wg in (4, 3, 2, 1):
        ctx.set_option("wg_gates", wg)
        run(c, "one wave/gate, %d per workgroup" % wg)
    ctx.set_option("wg_gates", 0)
    if not exact:
        ctx.set_option("two_wave_max", 5 * cus)
    run(c, "default")
