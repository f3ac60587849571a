"""Quick GPU parity + timing probe (development aid; the real tests live in tests/)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ieache_amd as ia
from ieache_amd import tools
from oracle import oracle as O

def mk(n, N=1024, seed=(1, 2, 3)):
    p = ia.default_params().copy(n=n, N=N)
    k = tools.keygen_raw(p, seed)
    return p, k

def main():
    p, k = mk(16)
    ck = O.CloudKey(p.n, p.N, p.k, p.l, p.Bgbit, p.ks_t, p.ks_basebit, k["bk"], k["ksk"])
    ctx = ia.Context.from_arrays(p, k["bk"], k["ksk"])
    print("variant", ctx.kernel_variant)
    rng = np.random.default_rng(0)
    bits = rng.integers(0, 2, size=8).astype(np.uint8)
    x = tools.encrypt_bits(p, k["lwe_key"], bits, 7)
    # blind-rotate stage parity
    for steps in (0, 1, 2, -1):
        acc = ctx.debug_blind_rotate(x, steps)
        ok = True
        for i in range(x.shape[0]):
            bara, barb = ck.modswitch(x[i])
            ref = ck.blind_rotate_init(barb)
            nst = p.n if steps < 0 else steps
            for s in range(nst):
                ref = ck.blind_rotate_step(ref, s, bara[s])
            if not np.array_equal(ref, acc[i]):
                ok = False
                d = np.abs(ref.astype(np.int64) - acc[i].astype(np.int64))
                print("  mismatch item", i, "steps", steps, "max diff", d.max(), "count", (d > 0).sum())
        print("blind_rotate steps", steps, "OK" if ok else "FAIL")
    # keyswitch parity
    u = rng.integers(-2**31, 2**31, size=(4, p.N + 1), dtype=np.int64).astype(np.int32)
    ks = ctx.debug_keyswitch(u)
    print("keyswitch", all(np.array_equal(ck.keyswitch(u[i]), ks[i]) for i in range(4)))
    # gates
    a = tools.encrypt_bits(p, k["lwe_key"], np.array([0, 0, 1, 1], dtype=np.uint8), 11)
    b = tools.encrypt_bits(p, k["lwe_key"], np.array([0, 1, 0, 1], dtype=np.uint8), 12)
    for name, gt in (("and", ia.GATE_AND), ("xor", ia.GATE_XOR), ("or", ia.GATE_OR), ("nand", ia.GATE_NAND)):
        out = ctx.gates(gt, a, b)
        ref = np.stack([ck.gate(name, a[i], b[i]) for i in range(4)])
        print("gate", name, "bit-exact", np.array_equal(out, ref), "decrypt", tools.decrypt_bits(p, k["lwe_key"], out))
    # circuit: add 16
    for kind, bits_, nm in ((ia.CIRC_ADD, 16, "add16"), (ia.CIRC_SUB, 32, "sub32")):
        info = ia.circuit_info(kind, bits_)
        B = 3
        va = [int(v) for v in rng.integers(0, 2**bits_, size=B)]
        vb = [int(v) for v in rng.integers(0, 2**bits_, size=B)]
        inb = np.zeros((B, info.n_inputs), dtype=np.uint8)
        for e in range(B):
            inb[e, :bits_] = tools.int_to_bits(va[e], bits_)
            inb[e, bits_:2 * bits_] = tools.int_to_bits(vb[e], bits_)
        inp = tools.encrypt_bits(p, k["lwe_key"], inb, 99)
        st = ia.Stats()
        out = ctx.eval_batch(kind, bits_, inp, st)
        dec = tools.decrypt_bits(p, k["lwe_key"], out)
        got = [tools.bits_to_int(dec[e]) for e in range(B)]
        exp = [((va[e] + vb[e]) if kind == ia.CIRC_ADD else (va[e] - vb[e])) % 2**bits_ for e in range(B)]
        # oracle bit-exact
        S = p.n + 1
        if kind == ia.CIRC_ADD:
            s, _ = ck.add(inp[0, :bits_], inp[0, bits_:2 * bits_], inp[0, 2 * bits_:2 * bits_ + 1], bits_)
            exact = np.array_equal(s, out[0])
        else:
            o1 = np.zeros((8, 32, S), np.int32); o2 = np.zeros((8, 32, S), np.int32)
            o1[0] = inp[0, :32]; o2[0] = inp[0, 32:64]
            rc, ov = ck.cloud_values(2, 0, 32, o1, o2, inp[0, 64:96])
            exact = np.array_equal(ov[0], out[0])
        print(nm, "decrypt ok", got == exp, "bit-exact vs oracle", exact, st.as_dict())
    ctx.close()
    # full-size timing
    p, k = mk(630)
    ctx = ia.Context.from_arrays(p, k["bk"], k["ksk"])
    for count in (256, 2048):
        bits = rng.integers(0, 2, size=(2, count)).astype(np.uint8)
        a = tools.encrypt_bits(p, k["lwe_key"], bits[0], 1)
        b = tools.encrypt_bits(p, k["lwe_key"], bits[1], 2)
        st = ia.Stats()
        out = ctx.gates(ia.GATE_AND, a, b, st)
        dec = tools.decrypt_bits(p, k["lwe_key"], out)
        print("n=630 AND x", count, "correct", np.array_equal(dec, bits[0] & bits[1]), st.as_dict(),
              "gates/s %.0f" % (count / (st.total_ms / 1e3)))
    # one oracle check at full size
    ck = O.CloudKey(p.n, p.N, p.k, p.l, p.Bgbit, p.ks_t, p.ks_basebit, k["bk"], k["ksk"])
    t = time.time(); ref = ck.gate("and", a[0], b[0]); print("oracle gate sec", time.time() - t)
    print("n=630 bit-exact vs oracle", np.array_equal(ref, out[0]))

main()
