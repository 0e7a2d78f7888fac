"""Instruction census of the coupling loop of one stack_kernel instance.
usage: isa_census.py stack.s MB NB MODE LISTS NJL   (assembly from hipcc -S --cuda-device-only)"""
import re, sys, collections
path, MB, NB, MODE, LISTS, NJL = sys.argv[1], *sys.argv[2:7]
sym = f"stack_kernelILi{MB}ELi{NB}ELi{MODE}ELb{LISTS}ELi{NJL}E"
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if sym in l and l.rstrip().split(';')[0].strip().endswith(':'))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start:end]
labels = {l.split(':')[0]: i for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)}
edges = []
for i, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        edges.append((i - labels[m.group(1)], labels[m.group(1)], i))
edges.sort(reverse=True)
print('back edges (len, from, to):', edges[:4])
_, a, b = edges[0]
ops = collections.Counter()
for l in body[a:b]:
    l = l.strip()
    if not l or l.startswith(('.', ';')) or l.endswith(':'):
        continue
    ops[l.split()[0]] += 1
tot = sum(ops.values())
valu = sum(v for k, v in ops.items() if k.startswith('v_') and 'mfma' not in k)
print('total', tot, 'valu', valu, 'mfma', sum(v for k, v in ops.items() if 'mfma' in k),
      'ds', sum(v for k, v in ops.items() if k.startswith('ds_')), 'salu', sum(v for k, v in ops.items() if k.startswith('s_')))
for k, v in ops.most_common(45):
    print(f'{v:5d} {k}')
