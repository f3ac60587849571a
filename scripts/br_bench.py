"""Blind-rotation / key-switch micro-benchmark at full size (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ieache_amd as ia
from ieache_amd import tools
p = ia.default_params()
k = tools.keygen_raw(p, (1, 2, 3))
ctx = ia.Context.from_arrays(p, k["bk"], k["ksk"])
for opt in ("br_variant", "br_slice", "exact_fft", "one_limb_min", "br_wide_max", "ks_split_max", "two_wave_max", "ks_batch_min", "ks_sliced_min", "ks_slice", "ks_gates"):  # e.g. BR_VARIANT=7 BR_SLICE=630
    if os.environ.get(opt.upper()):
        ctx.set_option(opt, int(os.environ[opt.upper()]))
rng = np.random.default_rng(0)
counts = [int(c) for c in (sys.argv[1:] or ["1024", "4096", "8192"])]
mx = max(counts)
bits = rng.integers(0, 2, size=(2, mx)).astype(np.uint8)
a = tools.encrypt_bits(p, k["lwe_key"], bits[0], 1)
b = tools.encrypt_bits(p, k["lwe_key"], bits[1], 2)
for count in counts:
    best = None
    for rep in range(3):
        st = ia.Stats()
        out = ctx.gates(ia.GATE_AND, a[:count], b[:count], st)
        if best is None or st.blind_rotate_ms < best.blind_rotate_ms:
            best = st
    ok = np.array_equal(tools.decrypt_bits(p, k["lwe_key"], out), bits[0][:count] & bits[1][:count])
    print("variant", os.environ.get("BR_VARIANT", os.environ.get("IEACHE_BR_VARIANT", "0")), "slice", os.environ.get("BR_SLICE", "auto"), ctx.kernel_variant, "count", count, "ok", ok,
          "BR ms %.2f in %d launches of %s (%.0f gates/s)  KS ms %.2f (%.0f gates/s)" % (best.blind_rotate_ms, best.blind_rotate_launches,
          ctx.kernel_for_launch(count).split("<")[0], count / best.blind_rotate_ms * 1e3,
          best.keyswitch_ms, count / best.keyswitch_ms * 1e3), "guard", ctx.fft_guard(), flush=True)
