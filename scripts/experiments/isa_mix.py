"""Instruction mix of a kernel's loops from hipcc's -S output: python isa_mix.py netsq.s <symbol substring> [min loop size]."""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 100
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.rstrip().split(':')[0].endswith(key) or (l.startswith('_Z') and key in l.split(':')[0]))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
body = lines[start:end + 1]
labels = {}
ins = []
for l in body:
    t = l.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', t)
    if m:
        labels[m.group(1)] = len(ins); continue
    if not t or t.startswith((';', '.', '_Z')):
        continue
    ins.append(t.split(';')[0].strip())
def cls(i):
    op = i.split()[0]
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem' if not op.startswith('scratch_') else 'scratch'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_barrier'): return 'barrier'
    if op.startswith('s_load') or op.startswith('s_buffer'): return 'smem'
    if op.startswith('s_'): return 'salu'
    return 'other'
print('kernel %s: %d instructions' % (body[0].split(':')[0][:90], len(ins)), dict(collections.Counter(cls(i) for i in ins)))
loops = []
for n, i in enumerate(ins):
    m = re.match(r'^s_cbranch_\w+\s+(\.LBB\d+_\d+)|^s_branch\s+(\.LBB\d+_\d+)', i)
    if m:
        lab = m.group(1) or m.group(2)
        if lab in labels and labels[lab] <= n and n - labels[lab] >= minsz:
            loops.append((labels[lab], n))
for a, b in loops:
    c = collections.Counter(cls(i) for i in ins[a:b + 1])
    ops = collections.Counter(i.split()[0] for i in ins[a:b + 1] if cls(i) == 'valu')
    print('loop [%d, %d] %d instr' % (a, b, b - a + 1), dict(c))
    print('   valu:', ', '.join('%s %d' % kv for kv in ops.most_common(24)))
