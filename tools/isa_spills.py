"""Scratch stores / loads per basic block of one function in a gfx950 listing: tools/isa_spills.py <file.s> <name substring>"""
import re, sys
T = open(sys.argv[1]).read().split('\n')
i = next(i for i, l in enumerate(T) if re.match(r'^_Z\w+:', l) and sys.argv[2] in l)
e = next(j for j in range(i, len(T)) if T[j].startswith('.Lfunc_end'))
lab = None; cnt = {}
for n, l in enumerate(T[i:e]):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: lab = m.group(1)
    if 'scratch_' in l:
        cnt.setdefault(lab, [0, 0, n]); cnt[lab][0 if 'store' in l else 1] += 1
print(e - i, 'lines;', {k: v for k, v in cnt.items() if v[0]}, 'loads', sum(v[1] for v in cnt.values()))
