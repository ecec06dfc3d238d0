"""Dev aid: in hipcc -S output, flag any vector instruction that reads or writes the destination registers of an inline-asm
`global_load_dwordx4` between that load and the next `s_waitcnt vmcnt` (hipcc counts an asm load's destination as written at
;;#ASMEND and may copy it -- for a phi, a tied operand -- before the data has landed; attnw2_fwd_kernel's first version did).
usage: asm_inflight_check.py file.s mangled_kernel_name_prefix"""
import re, sys
s = open(sys.argv[1]).read()
pat = re.compile(r"\bv(\d+)\b|v\[(\d+):(\d+)\]")


def regs(t):
    r = set()
    for m in pat.finditer(t):
        if m.group(1):
            r.add(int(m.group(1)))
        else:
            r.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return r


bad = 0
for m0 in re.finditer(r"^(%s\w*):" % re.escape(sys.argv[2]), s, re.M):
    i = m0.start()
    body = s[i:s.index(".end_amdhsa_kernel", i)].split("\n")
    loaded = set()
    for k, l in enumerate(body):
        t = l.strip()
        if t.startswith("global_load_dwordx4"):
            m = re.match(r"global_load_dwordx4 v\[(\d+):(\d+)\]", t)
            loaded |= set(range(int(m.group(1)), int(m.group(2)) + 1))
            continue
        if "s_waitcnt vmcnt" in t:
            loaded = set()
            continue
        if loaded and not t.startswith(";") and t.split(" ")[0].startswith(("v_", "ds_", "global_", "buffer_")) and regs(t) & loaded:
            bad += 1
            print(f"{m0.group(1)[:50]} line {k}: {t[:100]}")
print("in-flight register touches:", bad)
sys.exit(1 if bad else 0)
