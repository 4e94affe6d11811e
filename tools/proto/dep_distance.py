"""Dependency distance of the vector instructions of one kernel in a hipcc .s file: for each VALU instruction, how many
instructions earlier its nearest source operand was written (1 = the instruction right before it)."""
import re, sys, collections
asm = open(sys.argv[1]).read()
pat = sys.argv[2]
i = asm.find("\n_Z" + asm.split("\n_Z", 1)[1].split(":")[0]) if False else None
m = re.search(r"\n(_Z\S*%s\S*):" % pat, asm)
start = m.end(); end = asm.find("s_endpgm", start)
def regs(tok):
    out = []
    for mm in re.finditer(r"\b([vs])(\d+)\b|\b([vs])\[(\d+):(\d+)\]", tok):
        if mm.group(1): out.append((mm.group(1), int(mm.group(2))))
        else: out += [(mm.group(3), r) for r in range(int(mm.group(4)), int(mm.group(5)) + 1)]
    if "vcc" in tok: out.append(("vcc", 0))
    return out
last = {}
hist = collections.Counter(); n = 0; idx = 0
for line in asm[start:end].split("\n"):
    t = line.split(";")[0].strip()
    if not t or t.endswith(":") or t.startswith("."): continue
    op, _, rest = t.partition(" ")
    ops = [o.strip() for o in rest.split(",")]
    idx += 1
    if op.startswith("v_") and not op.startswith("v_cmp"):
        dst, srcs = ops[0], ops[1:]
        d = min([idx - last[r] for s in srcs for r in regs(s) if r in last] or [99])
        hist[min(d, 12)] += 1; n += 1
        for r in regs(dst): last[r] = idx
        if op.endswith("_co_u32") or "vcc" in dst: last[("vcc", 0)] = idx
    elif op.startswith("v_cmp"):
        srcs = ops[1:] if not ops[0].startswith(("v", "s[")) else ops[1:]
        d = min([idx - last[r] for s in ops for r in regs(s) if r in last] or [99])
        hist[min(d, 12)] += 1; n += 1
        last[("vcc", 0)] = idx
        for r in regs(ops[0]): last[r] = idx
    else:
        for r in regs(ops[0] if ops else ""): last[r] = idx
print(pat, "VALU", n, " distance histogram (12 = 12 or more):", " ".join("%d:%d%%" % (k, round(100 * v / n)) for k, v in sorted(hist.items())))
