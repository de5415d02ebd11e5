"""Developer aid: per-loop instruction / scratch statistics of one kernel from hipcc's assembly output.
usage: asm_report.py <file.s> <mangled-name-substring>"""
import re, sys
from collections import Counter
txt = open(sys.argv[1]).read()
name = sys.argv[2]
i = txt.index(name); i = txt.index('\n', txt.index(':', i)); j = txt.index('.Lfunc_end', i)
body = txt[i:j].split('\n')
# blocks: label lines
blocks = []
cur = ('entry', 0, '')
for k, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):\s*;?\s*(.*)', l)
    if m:
        blocks.append((cur[0], cur[1], k, cur[2]))
        cur = (m.group(1), k, m.group(2))
blocks.append((cur[0], cur[1], len(body), cur[2]))
print(f"{len(body)} lines, {len(blocks)} blocks")
tot = Counter()
for lab, a, b, info in blocks:
    seg = [l.strip() for l in body[a:b] if l.strip() and not l.strip().startswith((';', '.'))]
    c = Counter(x.split()[0] for x in seg)
    sc = sum(v for k, v in c.items() if k.startswith('scratch'))
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    tot.update(c)
    if sc or valu > 150:
        print(f"{lab:12s} lines {a:5d}-{b:5d} instrs {len(seg):4d} VALU {valu:4d} scratch {sc:3d} pk_fma {c.get('v_pk_fma_f32', 0):3d} fma64 {c.get('v_fma_f64', 0):3d} ds {sum(v for k, v in c.items() if k.startswith('ds_')):3d}  {info[:60]}")
print('total scratch', sum(v for k, v in tot.items() if k.startswith('scratch')))
