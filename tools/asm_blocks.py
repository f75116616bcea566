"""dev: per-basic-block instruction counts of one kernel in a hipcc -S listing.  usage: asm_blocks.py file.s <kernel name substring> [min]"""
import re, sys
s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]; mn = int(sys.argv[3]) if len(sys.argv) > 3 else 8
st = next(i for i, l in enumerate(s) if key in l and l.startswith('_Z') and ':' in l)
en = next(i for i in range(st, len(s)) if s[i].startswith('.Lfunc_end'))
counts = [['entry', 0, 0, 0, 0, st]]
for i in range(st + 1, en):
    t = s[i].strip()
    if re.match(r'^\.LBB\d+_\d+:', t):
        counts.append([t.split(':')[0], 0, 0, 0, 0, i]); continue
    if t.startswith('v_'): counts[-1][1] += 1
    elif t.startswith('s_'): counts[-1][2] += 1
    elif t.startswith(('global_', 'buffer_', 'scratch_', 'flat_')): counts[-1][3] += 1
    elif t.startswith('ds_'): counts[-1][4] += 1
print('block valu salu vmem lds line')
for c in counts:
    if c[1] + c[3] + c[4] >= mn: print(*c)
print('total valu', sum(c[1] for c in counts), 'salu', sum(c[2] for c in counts), 'vmem', sum(c[3] for c in counts))
