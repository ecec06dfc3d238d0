"""Dev aid: find registers that an in-flight LDS read targets and that another instruction touches before a wait covers it.
Inline-asm ds_reads are complete at once as far as the compiler knows: an asm read whose result nothing consumes has its
destination reused immediately, and the late LDS return then lands on the new contents (seen in tailpair_kernel<7, 14>).
LDS operations return in order, so `s_waitcnt lgkmcnt(N)` retires all but the last N of them (scalar loads, which share
the counter and return out of order, are ignored: the check is conservative towards silence there).
usage: asm_lds_hazard.py kernel.s   (hipcc -S --cuda-device-only output)"""
import re, sys


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


inflight = []   # (line, destination registers) of LDS operations, oldest first
func = None
for n, line in enumerate(open(sys.argv[1]), 1):
    t = line.split(";")[0].strip()
    if t.endswith(":") and not t.startswith("."):
        func, inflight = t[:-1], []
        continue
    if not t or t.startswith(".") or t.endswith(":"):
        continue
    op, _, rest = t.partition(" ")
    if op == "s_waitcnt":
        m = re.search(r"lgkmcnt\((\d+)\)", rest)
        if m:
            k = int(m.group(1))
            inflight = inflight[len(inflight) - k:] if k else []
        continue
    if op in ("s_barrier", "s_endpgm"):
        continue
    toks = [x.strip() for x in re.split(r",\s*(?![^\[]*\])", rest)] if rest else []
    touched = set()
    for x in toks:
        touched |= regs(x.split(" ")[0])
    for ln, dst in inflight:
        hit = touched & dst
        if hit:
            print(f"{func}\n  line {n}: {t}   <- register(s) {sorted(hit)} still awaited from the LDS read at line {ln}")
    if op.startswith("ds_") or op.startswith("buffer_load") and " lds" in rest:
        inflight.append((n, regs(toks[0]) if op.startswith("ds_read") or op.startswith("ds_bpermute") or op.startswith("ds_swizzle") else set()))
