"""Dev aid: print the instruction-class sequence of one kernel from hipcc -S output.
usage: python tools/asm_seq.py file.s mangled_kernel_name [max_chars]"""
import re, sys
s = open(sys.argv[1]).read()
i = s.index(sys.argv[2] + ':')
j = s.index('.end_amdhsa_kernel', i)
seq = []
for l in s[i:j].split('\n'):
    t = l.strip()
    if t.startswith('.LBB') or t.startswith('s_cbranch') or t.startswith('s_branch'):
        w = t.split()
        seq.append('\n' + w[0] + (' ' + w[1] if len(w) > 1 and not w[1].startswith(';') else '') + '\n')
    elif 'ds_read' in t: seq.append('R')
    elif 'ds_write' in t: seq.append('W')
    elif 'v_mfma' in t: seq.append('M')
    elif '_load' in t and ' lds' in t: seq.append('G')
    elif 'global_store' in t or 'buffer_store' in t: seq.append('S')
    elif 'scratch_' in t: seq.append('x')
    elif 'global_load' in t or 'buffer_load' in t: seq.append('L')
    elif 's_barrier' in t: seq.append('|BAR|')
    elif 's_waitcnt' in t: seq.append('[' + t.split('s_waitcnt ')[1] + ']')
    elif t.startswith('v_'): seq.append('.')
out = re.sub(r'\n+', '\n', ''.join(seq))
out = re.sub(r'\.{8,}', lambda m: f'.{{{len(m.group(0))}}}', out)
print(out[:int(sys.argv[3]) if len(sys.argv) > 3 else 8000])
