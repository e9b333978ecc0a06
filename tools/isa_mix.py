"""Instruction mix of one function in a gfx950 assembly listing: tools/isa_mix.py <file.s> <name substring> [top]"""
import re, sys
from collections import Counter
T = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]; top = int(sys.argv[3]) if len(sys.argv) > 3 else 16
for i, l in enumerate(T):
    if re.match(r'^_Z\w+:', l) and key in l:
        end = next(j for j in range(i, len(T)) if T[j].startswith('.Lfunc_end'))
        L = [x.strip() for x in T[i:end] if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
        c = Counter(x.split()[0] for x in L)
        print(l.split(':')[0][:70], len(L), 'instructions; scratch', sum(x.startswith('scratch_') for x in L))
        for a, b in c.most_common(top): print('  ', b, a)
