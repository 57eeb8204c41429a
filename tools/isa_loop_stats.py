import re, sys
src = open(sys.argv[1]).read().split('\n')
# split into functions
funcs = {}
cur = None
for i, l in enumerate(src):
    m = re.match(r'^(_Z\w+):', l)
    if m: cur = m.group(1); funcs[cur] = []
    elif cur is not None:
        if l.startswith('.Lfunc_end'): cur = None
        else: funcs[cur].append(l)
def cls(op):
    if op.startswith('v_mfma'): return 'MFMA'
    if op.startswith('v_'): return 'VALU'
    if op.startswith('s_waitcnt') or op.startswith('s_nop') or op.startswith('s_barrier'): return op.split()[0]
    if op.startswith('s_'): return 'SALU'
    if op.startswith('ds_'): return 'LDS'
    if op.startswith('buffer_') or op.startswith('global_'): return 'VMEM'
    return 'other'
for name, body in funcs.items():
    if not any('v_mfma' in l for l in body): continue
    # find innermost loops containing mfma: header labels with "Loop Header" comments
    # approach: for each back-edge branch to a label earlier in the text, take [label, branch]; choose smallest range containing >= 8 mfma
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m: labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.match(r'^\s+s_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            a = labels[m.group(1)]
            n = sum('v_mfma' in x for x in body[a:i + 1])
            if n >= 4: loops.append((i - a, a, i, n))
    if not loops: continue
    loops.sort()
    seen = []
    for ln, a, b, n in loops[:3]:
        if any(a >= x and b <= y for x, y in seen): continue
        seen.append((a, b))
        c = {}
        for l in body[a:b + 1]:
            m = re.match(r'^\s+([a-z]\S+)', l)
            if m and not l.strip().startswith(';'):
                k = cls(m.group(1)); c[k] = c.get(k, 0) + 1
        mf = c.get('MFMA', 1)
        print(f"{name[:70]:70s} loop {b-a:5d} lines MFMA {mf:3d} | per MFMA: VALU {c.get('VALU',0)/mf:5.2f} SALU {c.get('SALU',0)/mf:5.2f} "
              f"LDS {c.get('LDS',0)/mf:4.2f} VMEM {c.get('VMEM',0)/mf:4.2f} wait {c.get('s_waitcnt',0)/mf:4.2f} nop {c.get('s_nop',0)/mf:4.2f} bar {c.get('s_barrier',0)}")
