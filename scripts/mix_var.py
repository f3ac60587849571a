"""Run-to-run spread of the rotation of roles: REPS timings per size (min / median / max), several contexts in turn.  usage: mix_var.py [count ...]"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ieache_amd as ia
from ieache_amd import tools
counts = [int(c) for c in sys.argv[1:]] or [1100, 1536, 2304]
p = ia.default_params()
k = tools.keygen_raw(p, (1, 2, 3))
rng = np.random.default_rng(5)
cmax = max(counts)
bits = rng.integers(0, 2, size=(2, cmax)).astype(np.uint8)
a = tools.encrypt_bits(p, k["lwe_key"], bits[0], 11)
b = tools.encrypt_bits(p, k["lwe_key"], bits[1], 12)
reps = int(os.environ.get("REPS", "12"))
for trial in range(int(os.environ.get("CONTEXTS", "3"))):
    ctx = ia.Context.from_arrays(p, k["bk"], k["ksk"])
    for c in counts:
        for mix in (0, 1):
            ctx.set_option("br_mix", mix)
            ts = []
            for _ in range(reps):
                st = ia.Stats()
                ctx.gates(ia.GATE_XOR, a[:c], b[:c], st)
                ts.append(st.blind_rotate_ms)
            print("context %d, %5d gates, br_mix %d: min %.3f median %.3f max %.3f ms (%s)" % (trial, c, mix, min(ts), statistics.median(ts), max(ts), " ".join("%.2f" % t for t in ts)), flush=True)
    ctx.close()
