"""Rounding-guard statistics of the one-limb blind rotation over many gates (development aid / evidence for DESIGN.md section 2).

Runs full 32-bit multiplier circuits (reference, folded and carry-save forms) on 1 024 random operand pairs per pass at
n=630, checks every product by decryption and prints the guard record after each pass: the largest distance to an
integer any rounded coefficient had, and whether any call had to be repeated on the two-limb kernels."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ieache_amd as ia
from ieache_amd import tools

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
p = ia.default_params()
k = tools.keygen_raw(p, (314, 1592, 657))
ctx = ia.Context.from_arrays(p, k["bk"], k["ksk"])
if os.environ.get("FFT_AUDIT"):  # e.g. FFT_AUDIT=1: every one-limb launch re-runs 64 of its gates on the two-limb kernel and compares bits
    ctx.set_option("fft_audit", int(os.environ["FFT_AUDIT"]))
rng = np.random.default_rng(int(os.environ.get("SEED", "2026")))
total = 0
for it in range(passes):
    kind = (ia.CIRC_MUL_WALLACE, ia.CIRC_MUL)[it % 2]
    info = ia.circuit_info(kind, 32)
    a = rng.integers(0, 2**32, size=batch, dtype=np.uint64)
    b = rng.integers(0, 2**32, size=batch, dtype=np.uint64)
    inb = np.zeros((batch, info.n_inputs), dtype=np.uint8)
    for e in range(batch):
        inb[e, :32] = tools.int_to_bits(int(a[e]), 32)
        inb[e, 32:64] = tools.int_to_bits(int(b[e]), 32)
    inp = tools.encrypt_bits(p, k["lwe_key"], inb, 100 + it)
    st = ia.Stats()
    t0 = time.time()
    out = ctx.eval_batch(kind, 32, inp, st)
    dec = tools.decrypt_bits(p, k["lwe_key"], out)
    ok = all(tools.bits_to_int(dec[e][:64]) == int(a[e]) * int(b[e]) for e in range(batch))
    total += st.bootstraps
    dev, reruns = ctx.fft_guard()
    print("pass %d %s: %d bootstraps in %.1f s, all %d products correct: %s; so far %.3g rounded coefficients, "
          "largest distance to an integer among the watched quarter %.6f (limit 0.0625, wrong bit at 0.5), calls repeated on the "
          "two-limb kernels: %d; audit %s"
          % (it, "carry-save" if kind == ia.CIRC_MUL_WALLACE else "cloud.c mul32", st.bootstraps, time.time() - t0, batch, ok,
             total * 630.0 * 2048, dev, reruns, ctx.fft_audit()), flush=True)
