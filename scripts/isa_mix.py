"""Instruction mix of one CMux step of a blind-rotation kernel, by class, from hipcc's -S output (development aid).

usage: isa_mix.py file.s <substring of the mangled kernel name> label:weight[,label:weight...]
Basic blocks are cut at labels AND after conditional branches; a block is named by its label, the fall-through part
after its n-th conditional branch "<label>+n".  The weights say how often each block runs per CMux step (loop trip counts);
blocks not named are ignored (prologue, epilogue, skipped-step path)."""
import re
import sys
import collections

s = open(sys.argv[1]).read()
pat = sys.argv[2]
weights = {kv.split(":")[0]: float(kv.split(":")[1]) for kv in sys.argv[3].split(",")}
name = [l.split(":")[0] for l in s.split("\n") if pat in l and re.match(r"^_Z\w+:", l)][0]
body = s.split("\n" + name + ":")[1].split(".Lfunc_end")[0]
blocks, cur, nbr = collections.OrderedDict(), "entry", 0
blocks[cur] = []
for line in body.split("\n"):
    t = line.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        cur, nbr = m.group(1), 0
        blocks[cur] = []
        continue
    if not t or t.startswith((".", ";", "//")):
        continue
    blocks[cur].append(t)
    if t.startswith("s_cbranch"):
        nbr += 1
        cur = cur.split("+")[0] + "+%d" % nbr
        blocks[cur] = []


def klass(i):
    op = i.split()[0]
    if re.match(r"v_(add|mul|fma|fmac|max|min|ldexp)_f64", op):
        return "FP64 arithmetic"
    if "cvt" in op:
        return "int <-> f64 conversion"
    if "dpp" in op or "permlane" in op:
        return "cross-lane move (DPP / permlane swap)"
    if op.startswith("v_bfe"):
        return "bit-field extract (digits, signs)"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "lane <-> scalar"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "register move"
    if op.startswith("v_"):
        return "integer / logic (addresses, rotation, decomposition)"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vector memory"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    return "scalar"


tot = collections.Counter()
fp64 = collections.Counter()  # FP64 opcodes, weighted: how many of the slots are fused multiply-adds (two flops) and how many one
print(name)
for b, ins in blocks.items():
    if b in weights and ins:
        c = collections.Counter(klass(i) for i in ins)
        print("  block %-12s x%-3g %5d instructions: %s" % (b, weights[b], len(ins), ", ".join("%s %d" % kv for kv in c.most_common())))
        for k, v in c.items():
            tot[k] += v * weights[b]
        for i in ins:
            if klass(i) == "FP64 arithmetic":
                fp64[re.sub(r"_e(32|64)$", "", i.split()[0])] += weights[b]
valu = sum(v for k, v in tot.items() if k not in ("LDS", "vector memory", "s_waitcnt", "scalar"))
print("per CMux step:")
for k, v in tot.most_common():
    print("  %-55s %7.0f" % (k, v))
print("  %-55s %7.0f" % ("= vector ALU instructions", valu))
fused = sum(v for k, v in fp64.items() if "fma" in k)
print("  FP64 by opcode: %s" % ", ".join("%s %.0f" % kv for kv in fp64.most_common()))
print("  => %.0f fused multiply-adds (2 flop) + %.0f one-flop slots = %.0f flop per lane, %.0f per wave and CMux step"
      % (fused, sum(fp64.values()) - fused, 2 * fused + sum(fp64.values()) - fused, 64 * (2 * fused + sum(fp64.values()) - fused)))
