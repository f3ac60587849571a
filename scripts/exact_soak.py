"""Whole circuits on the guarded one-limb kernels against the same circuits on the provably exact two-limb kernels (exact_fft = 1):
every output sample of every expression compared word for word (not a sample of gates: a wrong intermediate bit anywhere in
an expression's 11 264 bootstraps changes its output ciphertexts).  Evidence for DESIGN.md section 3.
usage: exact_soak.py [passes=4] [batch=1024]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ieache_amd as ia
from ieache_amd import tools

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
p = ia.default_params()
k = tools.keygen_raw(p, (2718, 2818, 2845))
ctx = ia.Context.from_arrays(p, k["bk"], k["ksk"])
rng = np.random.default_rng(int(os.environ.get("SEED", "4")))
total = words = 0
for it in range(passes):
    kind, bits, name = ((ia.CIRC_MUL, 32, "cloud.c mul32"), (ia.CIRC_ADD, 32, "add32"), (ia.CIRC_MUL, 32, "cloud.c mul32"), (ia.CIRC_SUB, 32, "sub32"))[it % 4]
    info = ia.circuit_info(kind, bits)
    inb = rng.integers(0, 2, size=(batch, info.n_inputs)).astype(np.uint8)
    inp = tools.encrypt_bits(p, k["lwe_key"], inb, 300 + it)
    outs = {}
    for exact in (0, 1):
        ctx.set_option("exact_fft", exact)
        # round 5: the one-limb pass runs the product's default stream mode (two pipelines for these batches); CROSS=1 puts the
        # two-limb pass on ONE stream, so the comparison also spans the stream modes
        ctx.set_option("overlap", 0 if (exact and os.environ.get("CROSS") == "1") else 1)
        st = ia.Stats()
        t0 = time.time()
        outs[exact] = ctx.eval_batch(kind, bits, inp, st)
        dt = time.time() - t0
        print("pass %d %s x %d on the %s kernels (%s): %d bootstraps in %.1f s" % (it, name, batch, "two-limb" if exact else "one-limb",
              ctx.kernel_variant, st.bootstraps, dt), flush=True)
    ctx.set_option("exact_fft", 0)
    ctx.set_option("overlap", 1)
    same = np.array_equal(outs[0], outs[1])
    total += st.bootstraps
    words += outs[0].size
    dev, reruns = ctx.fft_guard()
    print("pass %d: outputs identical word for word: %s (%d x %d samples of %d words); so far %d bootstraps per mode, %d output words compared, "
          "guard maximum %.6f, repeats %d; pipelined evaluations so far %d, launches run as a rotation of roles %d" % (it, same, outs[0].shape[0], outs[0].shape[1], outs[0].shape[2] if outs[0].ndim == 3 else 1, total, words, dev, reruns, ctx.get_option("pipelined_evals"), ctx.get_option("mixed_launches")), flush=True)
    assert same
