import re, collections, sys
asm = open(sys.argv[1]).read()
for name in ("tree_split_stepILi0E", "tree_split_stepILi1E", "tree_lane_stepILi0E", "tree_split_env_stepILi0E", "tree_lane_env_stepILi0E"):
    m = re.search(r"\n(_Z\S*%s\S*):" % name, asm)
    if not m: print(name, "not found"); continue
    end = asm.find("s_endpgm", m.end())
    body = asm[m.end():end]
    c = collections.Counter()
    for line in body.split("\n"):
        t = line.split(";")[0].strip().split()
        if not t: continue
        op = t[0]
        if op.startswith("v_pk_"): c[op] += 1; c["pk"] += 1
        elif op.startswith("v_accvgpr"): c["agpr_move"] += 1
        elif op.startswith("scratch_"): c["scratch"] += 1
        elif op.startswith("v_mov"): c["v_mov"] += 1
        if op.startswith("v_"): c["valu"] += 1
    k = asm.find(".amdhsa_kernel " + m.group(1))
    desc = asm[k:k + 3000]
    g = lambda key: re.search(r"\.amdhsa_%s\s+(\S+)" % key, desc).group(1)
    print(name, "valu", c["valu"], "pk", c["pk"], {k2: v for k2, v in c.items() if k2.startswith("v_pk")}, "agpr_move", c["agpr_move"], "v_mov", c["v_mov"], "scratch ops", c["scratch"], "vgpr", g("next_free_vgpr"), "accum_offset", g("accum_offset"), "scratch", g("private_segment_fixed_size"))
