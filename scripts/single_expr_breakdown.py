"""Where one single-expression evaluation's time goes (the reference's own mode): wall time of the C-ABI call against the
stream's blind-rotation / key-switch time from HIP events (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ieache_amd as ia
from ieache_amd import tools
p = ia.default_params()
k = tools.keygen_raw(p, (1, 2, 3))
ctx = ia.Context.from_arrays(p, k["bk"], k["ksk"])
for name, kind, bits in (("32-bit A+B", ia.CIRC_ADD, 32), ("32-bit A*B", ia.CIRC_MUL, 32), ("64-bit A*B", ia.CIRC_MUL, 64)):
    info = ia.circuit_info(kind, bits)
    inb = np.random.default_rng(3).integers(0, 2, size=(1, info.n_inputs), dtype=np.uint8)
    inb[:, 2 * bits:2 * bits + 32] = 0
    inp = tools.encrypt_bits(p, k["lwe_key"], inb, 9)
    ctx.eval_batch(kind, bits, inp)  # warm: allocations, circuit build
    for stats in (True, False):
        st = ia.Stats() if stats else None
        t0 = time.perf_counter()
        ctx.eval_batch(kind, bits, inp, st)
        wall = (time.perf_counter() - t0) * 1e3
        if stats:
            print("%-11s %5d levels %6d bootstraps: wall %.1f ms with timers; stream total %.1f = blind rotation %.1f (%.3f per level) + key switch %.1f (%.3f) + other %.1f"
                  % (name, st.levels, st.bootstraps, wall, st.total_ms, st.blind_rotate_ms, st.blind_rotate_ms / st.levels, st.keyswitch_ms,
                     st.keyswitch_ms / st.levels, st.total_ms - st.blind_rotate_ms - st.keyswitch_ms), flush=True)
        else:
            print("%-11s wall %.1f ms without timers" % (name, wall), flush=True)
