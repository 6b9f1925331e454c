"""Instruction mix of the loops of one kernel in a hipcc -S dump.  usage: asm_loops.py file.s <mangled-substring>"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and True)
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
for i, l in enumerate(body):
    m = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l) or re.search(r's_branch (\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        a = labels[m.group(1)]
        seg = [x.strip().split()[0] for x in body[a:i] if x.strip() and not x.strip().startswith(('.', ';')) and not x.strip().endswith(':')]
        c = Counter(seg)
        f = lambda p: sum(v for k, v in c.items() if p(k))
        print('loop %s lines %d-%d insts %d | mfma %d valu %d ds %d vmem %d salu %d waitcnt %d barrier %d' % (m.group(1), a, i, len(seg),
              f(lambda k: 'mfma' in k), f(lambda k: k.startswith('v_') and 'mfma' not in k), f(lambda k: k.startswith('ds_')),
              f(lambda k: k.startswith(('global_', 'buffer_', 'flat_', 'scratch_'))), f(lambda k: k.startswith('s_') and 'waitcnt' not in k and 'barrier' not in k),
              c.get('s_waitcnt', 0), c.get('s_barrier', 0)))
        print('   ', sorted(((k, v) for k, v in c.items() if k.startswith('v_') and 'mfma' not in k), key=lambda kv: -kv[1]))
for l in body:
    if any(t in l for t in ('.vgpr_count', 'NumVgprs', 'ScratchSize', 'Occupancy', 'NumAgprs')): print(l.strip())
