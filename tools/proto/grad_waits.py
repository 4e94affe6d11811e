import re, sys
asm = open(sys.argv[1]).read()
for k in ("ILi0ELi1ELi8ELb1E", "ILi1ELi1ELi8ELb1E"):
    i = asm.find("\n_ZN12_GLOBAL__N_115mlp_grad_kernel%sEEvNS_9TrainArgsE:" % k)
    j = asm.find("s_endpgm", i)
    lines = asm[i:j].split("\n")
    print(k, len(lines))
    for n, l in enumerate(lines):
        if n > 600 and re.search(r"vmcnt|global_load|Loop Header: Depth=1|scratch", l): print(n, l.strip()[:110])
