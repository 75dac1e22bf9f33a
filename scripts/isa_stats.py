"""Instruction statistics of one kernel from a hipcc -S listing: python scripts/isa_stats.py lk.s <symbol substring> ..."""
import re
import sys
from collections import Counter

text = open(sys.argv[1]).read()
for key in sys.argv[2:]:
    m = re.search(r'^(\S*' + re.escape(key) + r'\S*):.*?\n(.*?)\.Lfunc_end', text, re.S | re.M)
    if not m:
        print(key, "not found")
        continue
    ins = [l.strip() for l in m.group(2).split('\n')
           if l.strip() and not l.strip().startswith(('.', ';', '//')) and not l.strip().split(';')[0].strip().endswith(':')]
    c = Counter(i.split()[0] for i in ins)
    print(key, len(ins), dict(c.most_common(16)))
    tail = text[m.end():m.end() + 6000]
    for k in ('NumVgprs', 'NumAgprs', 'NumSgprs', 'ScratchSize', 'Occupancy', 'LDSByteSize'):
        mm = re.search(r';\s*' + k + r':\s*\S+', tail)
        if mm:
            print('   ', mm.group(0).strip('; '))
