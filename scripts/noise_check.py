"""Output noise of 65 536 bootstrapped gates under three keys against the published analysis (tests/test_golden_cpu.py:
predicted_gate_output_noise); evidence for DESIGN.md section 7 (viii).  usage: noise_check.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, ieache_amd as ia
from ieache_amd import tools
from test_golden_cpu import predicted_gate_output_noise, phase_errors
p = ia.default_params()
for seed in ((1, 2, 3), (27, 18, 28), (5, 6, 7)):
    k = tools.keygen_raw(p, seed)
    ctx = ia.Context.from_arrays(p, k["bk"], k["ksk"])
    var, osd = predicted_gate_output_noise(p, np.sum(k["tlwe_key"]))
    rng = np.random.default_rng(seed[0])
    bits = rng.integers(0, 2, size=(2, 65536)).astype(np.uint8)
    a = tools.encrypt_bits(p, k["lwe_key"], bits[0], 11); b = tools.encrypt_bits(p, k["lwe_key"], bits[1], 12)
    out = ctx.gates(ia.GATE_XOR, a, b)
    e = phase_errors(p, k["lwe_key"], out, bits[0] ^ bits[1])
    print("key seed", seed, "ring key ones", int(np.sum(k["tlwe_key"])), ": 65 536 gates, variance %.4e predicted %.4e ratio %.3f; mean %.3e (predicted sd of the per-key offset %.2e); max |e| %.4f"
          % (np.var(e), var, np.var(e) / var, np.mean(e), osd, np.max(np.abs(e))), flush=True)
    ctx.close()
