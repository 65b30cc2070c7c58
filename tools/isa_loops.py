#!/usr/bin/env python3
"""Per-loop instruction mix of one kernel in a hipcc -S listing: tools/isa_loops.py file.s <mangled-name-substring>"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = None
for i, l in enumerate(lines):
    if re.match(r'^[_A-Za-z0-9]+:', l) and key in l and not l.startswith('.'):
        start = i; break
assert start is not None, "kernel not found"
body = []
for l in lines[start + 1:]:
    body.append(l)
    if 's_endpgm' in l: break
labels = {}
ins = []
for l in body:
    m = re.match(r'^(\.LBB\S+):', l)
    if m: labels[m.group(1)] = len(ins); continue
    t = l.strip()
    if not t or t.startswith(('.', ';')): continue
    ins.append(t)
def cat(op):
    if op.startswith('v_pk_'): return 'valu_pk'
    if op.startswith(('v_mov', 'v_cndmask', 'v_pk_mov', 'v_accvgpr')): return 'valu_mov/sel'
    if op.startswith(('v_rcp', 'v_sqrt', 'v_rsq', 'v_exp', 'v_log')): return 'valu_trans'
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_'): return 'valu'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('s_'): return 'salu'
    return 'other'
tot = collections.Counter(cat(i.split()[0]) for i in ins)
print("whole kernel:", dict(tot), "total", len(ins))
loops = []
for idx, t in enumerate(ins):
    m = re.match(r's_cbranch_\S+\s+(\.LBB\S+)|s_branch\s+(\.LBB\S+)', t)
    if m:
        lab = m.group(1) or m.group(2)
        if lab in labels and labels[lab] <= idx:
            loops.append((labels[lab], idx, lab))
seen = set()
shown = 0
for a, b, lab in sorted(loops, key=lambda x: x[0] - x[1]):
    if lab in seen: continue
    seen.add(lab)
    c = collections.Counter(cat(i.split()[0]) for i in ins[a:b + 1])
    ops = collections.Counter(i.split()[0] for i in ins[a:b + 1])
    nval = sum(v for k, v in c.items() if k.startswith('valu') or k == 'mfma')
    print(f"loop {lab}: {b - a + 1} instrs, VALU {nval} (readlane {ops['v_readlane_b32']}, cndmask {ops['v_cndmask_b32_e64'] + ops['v_cndmask_b32_e32']}, "
          f"mov {ops['v_mov_b32_e32'] + ops['v_mov_b64_e32'] + ops['v_mov_b32_dpp']}) salu {c['salu']} vmem {c['vmem']} lds {c['lds']}")
    if len(sys.argv) > 3:
        print("   ", [(k, v) for k, v in ops.most_common(60) if k.startswith('v_')])
    shown += 1
    if shown >= 8: break
