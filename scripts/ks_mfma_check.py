"""Key switch as an int8 MFMA product against the hand-scheduled walk at full size: bit identity and time (development aid).
usage: ks_mfma_check.py [counts...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ieache_amd as ia
from ieache_amd import tools
counts = [int(c) for c in sys.argv[1:]] or [1, 37, 511, 512, 513, 1024, 2304, 4096, 8192, 16384]
p = ia.default_params()
k = tools.keygen_raw(p, (1, 2, 3))
ctx = ia.Context.from_arrays(p, k["bk"], k["ksk"])
rng = np.random.default_rng(7)
mx = max(counts)
bits = rng.integers(0, 2, size=(2, mx)).astype(np.uint8)
a = tools.encrypt_bits(p, k["lwe_key"], bits[0], 21)
b = tools.encrypt_bits(p, k["lwe_key"], bits[1], 22)
for c in counts:
    res = {}
    for name, mn in (("walk", 1 << 40), ("mfma", 1)):
        ctx.set_option("ks_mfma_min", mn)
        best = None
        for rep in range(3):
            st = ia.Stats()
            out = ctx.gates(ia.GATE_XOR, a[:c], b[:c], st)
            best = st.keyswitch_ms if best is None else min(best, st.keyswitch_ms)
        res[name] = (out, best)
    for split in (1, 2, 4, 8):
        ctx.set_option("ks_mfma_split", split)
        st = ia.Stats()
        out = ctx.gates(ia.GATE_XOR, a[:c], b[:c], st)
        st = ia.Stats()
        out = ctx.gates(ia.GATE_XOR, a[:c], b[:c], st)
        res["mfma/%d" % split] = (out, st.keyswitch_ms)
    ctx.set_option("ks_mfma_split", 0)
    ok = np.array_equal(tools.decrypt_bits(p, k["lwe_key"], res["walk"][0]), bits[0][:c] ^ bits[1][:c])
    print("count", c, "decrypts", ok, " ".join("%s %.3f ms%s" % (n, t, "" if np.array_equal(o, res["walk"][0]) else " DIFFERS") for n, (o, t) in res.items()), flush=True)
