"""Rotation of roles for mid-size launches ("br_mix"): blind rotation alone by launch size, the default kernel choice against the
mix at several one-wave turn lengths and step ratios; bit identity with the two-limb kernel.  usage: mix_sweep.py [count ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ieache_amd as ia
from ieache_amd import tools
counts = [int(c) for c in sys.argv[1:]] or [1100, 1216, 1280, 1365, 1400, 1536, 1600, 1638]
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
reps = int(os.environ.get("REPS", "4"))
ratios = [int(x) for x in os.environ.get("RATIOS", "160,188,210").split(",")]
s1s = [int(x) for x in os.environ.get("S1", "32,64").split(",")]


def run(c, label):
    best = None
    for _ in range(reps):
        st = ia.Stats()
        out = ctx.gates(ia.GATE_XOR, a[:c], b[:c], st)
        best = st.blind_rotate_ms if best is None else min(best, st.blind_rotate_ms)
    print("%5d gates %-44s BR %.3f ms in %3d launches (%.0f gates/s) identical %s" % (c, label, best, st.blind_rotate_launches, c / best * 1e3, np.array_equal(ref[:c], out)), flush=True)
    return best


for c in counts:
    ctx.set_option("br_mix", 0)
    base = run(c, "default (" + ctx.kernel_for_launch(c).split("<")[0] + ")")
    ctx.set_option("br_mix", 1)
    if os.environ.get("FORCE"):  # FORCE=k,tw: that geometry whatever the launch size
        fk, ftw = (int(x) for x in os.environ["FORCE"].split(","))
        ctx.set_option("mix_k", fk)
        ctx.set_option("mix_tw", ftw)
    for wg in [int(x) for x in os.environ.get("WG", "4").split(",")]:
        ctx.set_option("mix_wg", wg)
        for s1 in s1s:
            ctx.set_option("mix_s1", s1)
            for r in ratios:
                ctx.set_option("mix_ratio", r)
                before = ctx.get_option("mixed_launches")
                t = run(c, "mix s1=%d ratio=%.2f wg=%d" % (s1, r / 100, wg))
                assert ctx.get_option("mixed_launches") > before
